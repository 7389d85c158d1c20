"""N-best rescoring throughput (hypotheses/s) of bayeslms_amd.compute_sentence_scores on a synthetic
AMI-shaped n-best list: the reference's one-hypothesis-per-forward loop vs the padded per-utterance
batch, for the cfg3 Bayesian Transformer and the cfg2 Bayesian LSTM (mean weights), plus 8 Monte-Carlo
weight samples for the Transformer (BASELINE.json configs[4] inference shape)."""
import random
import sys
import time
from collections import OrderedDict

import torch

sys.path.insert(0, ".")
from bayeslms_amd import compute_sentence_scores as css, model as M  # noqa: E402


BT = int(__import__('os').environ.get('BATCH_TOKENS', '8192'))  # padded tokens per batch (compute_scores_batched default)


def main():
    dev = torch.device("cuda:0")
    V = 33000
    rnd = random.Random(7)
    words = ["w%d" % i for i in range(V - 2)]
    vocab = {w: i + 2 for i, w in enumerate(words)}
    vocab["<s>"], vocab["<unk>"] = 0, 1
    nbest = OrderedDict()
    import os
    n_utt, n_hyp = int(os.environ.get("N_UTT", "40")), int(os.environ.get("N_HYP", "100"))
    for u in range(n_utt):
        base = [rnd.choice(words) for _ in range(rnd.randint(6, 30))]
        hyps = []
        for _ in range(n_hyp):
            h = list(base)
            for _ in range(rnd.randint(0, 3)):
                h[rnd.randrange(len(h))] = rnd.choice(words)
            hyps.append(" ".join(h))
        nbest["utt%03d" % u] = hyps
    total = n_utt * n_hyp
    torch.manual_seed(1)
    models = [("Transformer", M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "FFN").to(dev)),
              ("LSTM", M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3).to(dev))]
    for mtype, model in models:
        sub = OrderedDict(list(nbest.items())[:4])
        css.compute_scores(sub, model, vocab, mtype, dev)
        css.compute_scores_batched(sub, model, vocab, mtype, dev, batch_tokens=BT)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); css.compute_scores(sub, model, vocab, mtype, dev); torch.cuda.synchronize()
        t_loop = (time.perf_counter() - t0) / (4 * n_hyp)
        t0 = time.perf_counter(); sb = css.compute_scores_batched(nbest, model, vocab, mtype, dev, batch_tokens=BT); torch.cuda.synchronize()
        t_b = (time.perf_counter() - t0) / total
        line = f"{mtype:12s} loop {1 / t_loop:8.0f} hyp/s   batched {1 / t_b:8.0f} hyp/s"
        if mtype == "Transformer":
            t0 = time.perf_counter(); css.compute_scores_batched(nbest, model, vocab, mtype, dev, mc_samples=8, batch_tokens=BT); torch.cuda.synchronize()
            line += f"   batched, 8 MC weight samples {total / (time.perf_counter() - t0):8.0f} hyp/s"
        print(line)


if __name__ == "__main__":
    main()
