#!/usr/bin/env python3
"""engine.evaluate() (train.py:441-458) tokens/s at eval batch 20 for the headline Transformer (T 128) and the configs[1] LSTM
(T 35) on a synthetic stream.  BLM_EVAL_FUSED_NLL=0|1 switches the decoder between logits + CE kernel and ops.linear_nll."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import engine, model as M  # noqa: E402
from bayeslms_amd.data import batchify, synthetic_corpus  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    V = 33000
    torch.manual_seed(1111)
    for name, m, T in (("Transformer", M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "FFN").to(dev), 128),
                       ("LSTM", M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3).to(dev), 35)):
        src = batchify(synthetic_corpus(V, 20 * (40 * T + 1), seed=2222), 20, dev)
        engine.evaluate(m, src[:4 * T + 1], T)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            loss = engine.evaluate(m, src, T)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%s evaluate(): %.1f k tokens/s, loss %.6f" % (name, 20 * 40 * T / best / 1e3, loss), flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
