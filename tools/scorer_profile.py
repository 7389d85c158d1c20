#!/usr/bin/env python3
"""The n-best rescoring leg of bench.py (SURVEY 8(d) workload: 1000 utterances x 20-best) for one model, to put behind
rocprofv3 --kernel-trace --stats: scorer_profile.py lstm|tlm [n_utt]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from bayeslms_amd import compute_sentence_scores as css, model as M  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
    n_utt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = torch.device("cuda:0")
    V = 33000
    torch.manual_seed(1111)
    if kind == "lstm":
        model, mtype = M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3).to(dev), "LSTM"
    else:
        model, mtype = M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "FFN").to(dev), "Transformer"
    model.eval()
    if os.environ.get("TIME_CHAIN"):  # wall time of the carry chain alone (synchronised: changes the total)
        orig = css._carry_chain

        def timed_chain(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = orig(*a, **k)
            torch.cuda.synchronize()
            print("   carry chain: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
            return r
        css._carry_chain = timed_chain
    nbest, vocab, ntok = bench.synthetic_nbest(n_utt, 20, V)
    sub = dict(list(nbest.items())[:int(os.environ.get("WARM_UTT", "50"))])
    css.compute_scores_batched(sub, model, vocab, mtype, dev)
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    css.compute_scores_batched(nbest, model, vocab, mtype, dev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    print("allocator during the timed call: device mallocs %d, device frees %d, retries %d, reserved %.2f -> %.2f GB" % (
        ms1["num_device_alloc"] - ms0["num_device_alloc"], ms1["num_device_free"] - ms0["num_device_free"],
        ms1["num_alloc_retries"] - ms0["num_alloc_retries"], ms0["reserved_bytes.all.current"] / 2**30, ms1["reserved_bytes.all.current"] / 2**30))
    print("%s: %d hypotheses in %.1f ms = %.1f hypotheses/s, %.0f tokens/s" % (kind, 20 * n_utt, 1e3 * el, 20 * n_utt / el, ntok / el))
    if os.environ.get("NOGC"):
        import gc
        gc.disable()
    for _ in range(int(os.environ.get("REPEAT", "0"))):
        t0 = time.perf_counter()
        css.compute_scores_batched(nbest, model, vocab, mtype, dev)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("   again: %.1f ms = %.1f hypotheses/s (the call returned after %.1f ms)" % (1e3 * el, 20 * n_utt / el, 1e3 * host))


if __name__ == "__main__":
    main()
