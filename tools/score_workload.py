#!/usr/bin/env python3
"""The n-best rescoring workload of bench.py's extra_configs (1000 utterances x 20-best) on its own: for rocprofv3, and host
against device time.  usage: score_workload.py tlm|lstm [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bayeslms_amd import compute_sentence_scores as css, model as M  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "tlm"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(1111)
    V = int(os.environ.get("V", bench.V))  # e.g. V=33278 (wikitext-2): a vocabulary that is not a multiple of 4 words
    if kind == "tlm":
        m, mtype = M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "FFN").to(dev), "Transformer"
    else:
        m, mtype = M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3).to(dev), "LSTM"
    nbest, vocab, ntok = bench.synthetic_nbest(1000, 20, V)
    from collections import OrderedDict
    css.compute_scores_batched(OrderedDict(list(nbest.items())[:150]), m, vocab, mtype, dev)
    torch.cuda.synchronize()
    bt = int(os.environ.get("BATCH_TOKENS", "0")) or None  # padded tokens per batch (None: the scorer's default)
    for _ in range(reps):
        t0 = time.perf_counter()
        css.compute_scores_batched(nbest, m, vocab, mtype, dev, batch_tokens=bt)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("%s: %.1f ms for 20000 hypotheses (%.0f hyp/s, %.0f tokens/s); the call returned after %.1f ms" % (kind, 1e3 * el, 20000 / el, ntok / el, 1e3 * host))


if __name__ == "__main__":
    main()
