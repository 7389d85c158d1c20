#!/usr/bin/env python3
"""Training trajectories of the cfg3 model under the three GEMM arithmetic modes (same initial weights, same Philox
noise / dropout streams, same batches): per-step loss of the opt-in split-bf16 modes against the fp32 MFMA mode.
Two fp32 runs give the run-to-run spread (split-K float atomics make the summation order differ between runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd import engine, model as M, ops  # noqa: E402
from bayeslms_amd.data import batchify, get_batch, synthetic_corpus  # noqa: E402

V, D, NH, FF, L, T, B = 33000, 512, 8, 4096, 6, 128, 64
STEPS = int(os.environ.get("STEPS", "60"))
dev = torch.device("cuda:0")
stream = synthetic_corpus(V, B * (STEPS * T + 1) + 17, seed=1111)
train = batchify(stream, B, dev)


def kl_fn(mm):
    return mm.transformerlayers[0].linear2.kl_divergence()


kl_fn.fusable = True


def run(mode):
    ops.set_gemm_mode(mode)
    try:
        torch.manual_seed(1111)
        m = M.BayesTransformerModel(V, D, NH, FF, L, 0.2, True, "FFN").to(dev)
        tr = engine.Trainer(m, lr=0.1, clip=1.0, kl_scale=float(T) / float(train.size(0)), seed=1111)
        out = []
        for i in range(STEPS):
            d, t = get_batch(train, i * T, T)
            loss, _, _ = tr.step(d, t, kl_fn=kl_fn)
            out.append(loss)
        out = torch.stack(out).double().cpu()
        ops.set_grad_ready_hook(None)
        return out
    finally:
        ops.set_gemm_mode("f32")


ref = run("f32")
again = run("f32")
print("steps %d, loss %.4f -> %.4f" % (STEPS, float(ref[0]), float(ref[-1])))
IDX = [0, 1, 2, 5, 10, 20, 40, STEPS - 1]


def show(name, cur):
    rel = (cur - ref).abs() / ref
    print("%-22s |dloss|/loss at steps %s: %s" % (name, IDX, " ".join("%.1e" % float(rel[i]) for i in IDX)))


show("f32 (second run)", again)
for mode in ("bf16x6", "bf16x3"):
    cur = run(mode)
    show(mode, cur)
