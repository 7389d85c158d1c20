"""Per-GEMM breakdown of one cfg3 training step: every blm_gemm launch bracketed with events
(layout, shape, epilogue), average duration and TFLOP/s.  Bracketing serialises nothing (same stream),
but the step runs a little slower than untimed."""
import sys, re
import torch
sys.path.insert(0, ".")
from bayeslms_amd import engine, model as M, ops
from bayeslms_amd.data import batchify, get_batch, synthetic_corpus

def main():
    V, D, H, FF, NL, T, B = 33000, 512, 8, 4096, 6, 128, 64
    dev = torch.device("cuda:0")
    steps = 4
    stream = synthetic_corpus(V, B * ((steps + 2) * T + 1) + 17, seed=1111)
    train = batchify(stream, B, dev)
    torch.manual_seed(1111)
    gauss = len(sys.argv) > 1 and sys.argv[1] == "gauss"
    lstm = len(sys.argv) > 1 and sys.argv[1] == "lstm"  # BASELINE configs[1]: Bayesian LSTM pos 3, T 35
    if lstm:
        T = 35
        train = batchify(synthetic_corpus(V, B * ((steps + 12) * T + 1) + 17, seed=1111), B, dev)
        model = M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3).to(dev)
        kl_fn = lambda mm: mm.rnn.kl_divergence()
        kl_fn.fusable = False
    else:
        model = (M.GaussTransformerModel(V, D, H, FF, NL, 0.2, True, 3) if gauss else M.BayesTransformerModel(V, D, H, FF, NL, 0.2, True, "FFN")).to(dev)
        kl_fn = (lambda mm: mm.transformerlayers[0].gpnn.kl_divergence()) if gauss else (lambda mm: mm.transformerlayers[0].linear2.kl_divergence())
        kl_fn.fusable = not gauss
    tr = engine.Trainer(model, lr=0.1, clip=0.25, kl_scale=float(T) / train.size(0), seed=1111)
    timer = ops.KernelTimer(all_gemms=True)
    warm = 12 if lstm else 2
    hidden = model.init_hidden(B) if lstm else None
    for i in range(steps + warm):
        data, targets = get_batch(train, i * T, T)
        ops.set_kernel_timer(timer if i >= warm else None)
        if lstm:
            hidden = M.repackage_hidden(hidden)
        _, _, hidden = tr.step(data, targets, hidden=hidden, kl_fn=kl_fn)
    ops.set_kernel_timer(None)
    rows = []
    for tag, r in timer.summary().items():
        m = re.match(r"(\w+) (\d+)x(\d+)x(\d+)", tag)
        if m is None:  # non-GEMM brackets (LSTM step sequences)
            continue
        fl = 2.0 * int(m.group(2)) * int(m.group(3)) * int(m.group(4))
        rows.append((r["avg_ms"] * r["n"] / steps, tag, r["n"] / steps, r["avg_ms"], fl / (r["avg_ms"] * 1e-3) / 1e12))
    rows.sort(reverse=True)
    tot = 0.0
    for ms_step, tag, n, avg, tf in rows:
        tot += ms_step
        print(f"{tag:58s} x{n:4.1f}/step  {avg * 1000:8.1f} us  {tf:6.1f} TF  {ms_step:6.2f} ms/step")
    print(f"GEMM total {tot:.2f} ms/step")

if __name__ == "__main__":
    main()
