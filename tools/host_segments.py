"""Host time of a training step by segment (forward, backward, optimiser), device not waited for: which part of the step the
CPU spends its issue time on.  usage: host_segments.py cfg1 [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gemm_tune as G  # noqa: E402
from bayeslms_amd import model as M, ops  # noqa: E402


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    step, tokens = G.build(name, dev)
    m = G._LAST_MODEL[0]
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # the same step taken apart (engine.Trainer.step without the data-parallel parts)
    from bayeslms_amd import engine
    from bayeslms_amd.data import batchify, get_batch, synthetic_corpus
    T, B = (35, 20) if name == "cfg1" else ((35, 64) if name == "cfg2" else (100, 32))
    V = m.decoder.weight.shape[0]
    data = batchify(synthetic_corpus(V, B * (8 * T + 1) + 17, seed=1111), B, dev)
    tr = engine.Trainer(m, lr=1.0, clip=1.0, kl_scale=0.0, seed=1111)
    hidden = m.init_hidden(B)
    seg = [0.0, 0.0, 0.0, 0.0]
    tot = 0.0
    for i in range(n + 5):
        if i == 5:
            torch.cuda.synchronize()
            seg = [0.0, 0.0, 0.0, 0.0]
            t_all = time.perf_counter()
        d, t = get_batch(data, (i % 8) * T, T)
        hidden = M.repackage_hidden(hidden)
        a = time.perf_counter()
        m.train(); m.set_step(i); m.set_columns(0, B); tr.flat.zero_grad()
        b = time.perf_counter()
        out, hidden = m(d, hidden)
        mle, _ = ops.cross_entropy(out.view(-1, V), t, unit_grad=True)
        c = time.perf_counter()
        mle.backward()
        e = time.perf_counter()
        ops.clip_sgd(tr.table, tr.clip, tr.lr, tr.momentum, tr.first, 1.0, tr.weight_decay)
        tr.first = False
        f = time.perf_counter()
        seg[0] += b - a; seg[1] += c - b; seg[2] += e - c; seg[3] += f - e
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    tot = time.perf_counter() - t_all
    print("%s: host per step: prologue %.0f us, forward + CE %.0f us, backward %.0f us, clip + SGD %.0f us; host total %.0f us, device done after %.0f us"
          % (name, *(1e6 * s / n for s in seg), 1e6 * host / n, 1e6 * tot / n))


if __name__ == "__main__":
    main()
