#!/usr/bin/env python3
"""BASELINE.json configs[1]: Bayesian LSTM LM (--uncertainty Bayesian --L_bayes_pos 3), 2x1024,
batch 64, seq_len 35, V = 33,000, dropout 0.2, tied: training tokens/s on one MI355X (not the
headline bench; a parity-test configuration timed for DESIGN.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import engine, model as M, ops  # noqa: E402
from bayeslms_amd.data import batchify, get_batch, synthetic_corpus  # noqa: E402


def main():
    V, H, T, B, steps, warm = 33000, 1024, 35, 64, int(os.environ.get("STEPS", "30")), int(os.environ.get("WARM", "30"))
    dev = torch.device("cuda:0")
    stream = synthetic_corpus(V, B * ((steps + warm) * T + 1) + 5, seed=1111)
    train = batchify(stream, B, dev)
    torch.manual_seed(1111)
    m = M.BayesRNNModel("LSTM", V, H, H, 2, 0.2, True, 3).to(dev)
    tr = engine.Trainer(m, lr=1.0, clip=1.0, kl_scale=float(T) / train.size(0), seed=1111)
    kl_fn = lambda mm: mm.rnn.kl_divergence()  # noqa: E731
    kl_fn.fusable = False
    hidden = m.init_hidden(B)
    t0 = None
    if os.environ.get("WAVEFRONT", "0") == "1":
        ops.set_lstm_wavefront(True)
    timer = ops.KernelTimer()
    for i in range(steps + warm):
        if i == warm:
            torch.cuda.synchronize()
            ops.set_kernel_timer(timer)
            t0 = time.perf_counter()
        data, tgt = get_batch(train, i * T, T)
        hidden = M.repackage_hidden(hidden)
        loss, kl, hidden = tr.step(data, tgt, hidden, kl_fn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.set_kernel_timer(None)
    print("cfg2 Bayes LSTM pos3 2x1024 B64 T35 V33000: %.2f ms/step, %.0f tokens/s, loss %.4f" % (1e3 * dt, T * B / dt, float(loss)))
    for k, v in timer.summary().items():
        if k.startswith("lstm"):
            print("   %-24s %.1f us per launch bracket (n=%d)" % (k, 1e3 * v["avg_ms"], v["n"]))


if __name__ == "__main__":
    main()
