#!/usr/bin/env python3
"""Race check of the layer wavefront's stream schedule: the same N training steps of the recipe-shaped LSTM (T 100, B 32: the
three-stream form) and of configs[0] (T 35, B 20: the two-stream form) run twice as a wavefront and once with the layers in
sequence (set_lstm_wavefront(False)); the loss trajectories must agree (a missing wait shows as a run-to-run difference long before
it shows as a NaN).  usage: wavefront_soak.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import engine, model as M, ops  # noqa: E402
from bayeslms_amd.data import batchify, get_batch, synthetic_corpus  # noqa: E402


def run(T, B, V, steps, wavefront, dev):
    ops.set_lstm_wavefront(wavefront)
    torch.manual_seed(1111)
    m = M.RNNModel("LSTM", V, 1024, 1024, 2, 0.2, True).to(dev)
    data = batchify(synthetic_corpus(V, B * (steps * T + 1) + 17, seed=1111), B, dev)
    tr = engine.Trainer(m, lr=1.0, clip=1.0, kl_scale=0.0, seed=1111)
    hidden = m.init_hidden(B)
    losses = []
    for i in range(steps):
        d, t = get_batch(data, i * T, T)
        hidden = M.repackage_hidden(hidden)
        loss, _, hidden = tr.step(d, t, hidden=hidden)
        losses.append(loss)
    out = torch.stack(losses).double().cpu()
    ops.set_lstm_wavefront(None)
    return out, bool(torch.isfinite(tr.flat.flat_param).all())


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    for T, B, V in ((100, 32, 10000), (35, 20, 10000)):
        a, fa = run(T, B, V, steps, True, dev)
        b, fb = run(T, B, V, steps, True, dev)
        c, fc = run(T, B, V, steps, False, dev)
        d_ab = float((a - b).abs().max() / a.abs().max())
        d_ac = float((a - c).abs().max() / a.abs().max())
        print("T %3d B %2d, %d steps: wavefront run 1 vs run 2 max rel loss diff %.2e; wavefront vs sequential layers %.2e; finite: %s; "
              "loss %.4f -> %.4f" % (T, B, steps, d_ab, d_ac, fa and fb and fc, float(a[0]), float(a[-1])), flush=True)


if __name__ == "__main__":
    main()
