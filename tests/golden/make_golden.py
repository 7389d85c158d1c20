#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REFERENCE.

Run only in the build container (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

It imports /root/reference/steps/pytorchnn/{model,data}.py read-only, drives the
reference classes on tiny seeded inputs and stores inputs + state_dict +
outputs as .npz (data only; no reference source is copied).  eps recovery
follows SURVEY.md Appendix D: re-seed, replay the same ``zeros().normal_()``
draws the layer makes, re-seed again, run the layer.

Recipes (first argument; none = the layer / model / scorer fixtures of round 1): rnnv, search, search_btlm, scorer_gp,
gp_sample, vt11, rnn_gpnn2, pos5, gpnn2, interp, late, traj (six train.py trajectories from a saved initial state),
init (initial state_dict digests of 60 model builds under one seed), traj_seed (train.py runs started from --seed 1111 alone:
nothing sampled / weight noise / weight noise and dropout; TRAJ_ONLY=tag,tag limits the list), headline_seed [names]
(BASELINE configurations at full size from the seed alone, three steps + evaluate), scorer_full [names] (the reference scorer on
the full-size models the seed gives).
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference/steps/pytorchnn"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import model as ref  # noqa: E402
    import data as refdata  # noqa: E402

torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy()


def pack_sd(m):
    # pos_encoder.pe is a deterministic (5000,1,d) table: keep 64 rows so fixtures stay small
    return {"sd/" + k: (npy(v)[:64] if k.endswith("pos_encoder.pe") else npy(v)) for k, v in m.state_dict().items()}


def grads(m):
    return {"grad/" + k: npy(p.grad) for k, p in m.named_parameters() if p.grad is not None}


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


# ---------------------------------------------------------------- F1 BayesLinear
def f1_bayes_linear():
    torch.manual_seed(11)
    lin = ref.BayesLinear(24, 10)
    x = torch.randn(7, 3, 24, requires_grad=True)
    g = torch.randn(7, 3, 10)
    kl_scale = 0.37
    lin.train()
    torch.manual_seed(5)
    eps = torch.zeros(10, 24).normal_(0, 1)
    torch.manual_seed(5)
    y = lin(x)
    kl = lin.kl_divergence()
    ((y * g).sum() + kl * kl_scale).backward()
    lin.eval()
    y_eval = lin(x)
    save("bayes_linear", x=npy(x), g=npy(g), eps=npy(eps), kl_scale=np.float32(kl_scale),
         mu=npy(lin.weight_mean), lgstd=npy(lin.weight_lgstd), y_train=npy(y), y_eval=npy(y_eval), kl=npy(kl),
         dx=npy(x.grad), dmu=npy(lin.weight_mean.grad), dlgstd=npy(lin.weight_lgstd.grad))


# ---------------------------------------------------------------- F3 Transformer LMs
def f3_transformer(bayes_pos):
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    torch.manual_seed(21)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.BayesTransformerModel(V, d, h, ff, L, 0.2, True, bayes_pos)
    zero_dropout(m)  # layer 0 is built with a hard-coded 0.2 (model.py:1202)
    src = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    kl_scale = float(T) / 123.0
    shape = {"FFN": (d, ff), "MHA": (d, d), "EMB": (d, d)}.get(bayes_pos)
    m.train()
    torch.manual_seed(9)
    eps = torch.zeros(*shape).normal_() if shape else None
    torch.manual_seed(9)
    logits = m(src)
    mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
    if bayes_pos == "FFN":
        kl = m.transformerlayers[0].linear2.kl_divergence()
    elif bayes_pos == "MHA":
        kl = m.transformerlayers[0].self_attn.o_net.kl_divergence()
    elif bayes_pos == "EMB":
        kl = m.embed_kl_divergence()
    else:
        kl = torch.zeros(())
    loss = mle + kl * kl_scale
    loss.backward()
    m.eval()
    with torch.no_grad():
        logits_eval = m(src)
        nll_eval = torch.nn.functional.cross_entropy(logits_eval.view(-1, V), tgt, reduction="none")
    kw = dict(src=npy(src), tgt=npy(tgt), nhead=np.int64(h), kl_scale=np.float32(kl_scale),
              logits_train=npy(logits), mle=npy(mle), kl=npy(kl), loss=npy(loss),
              logits_eval=npy(logits_eval), nll_eval=npy(nll_eval))
    if eps is not None:
        kw["eps"] = npy(eps)
    kw.update(pack_sd(m))
    kw.update(grads(m))
    save("bayes_tlm_" + bayes_pos, **kw)


def f3_transformer_baseline():
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    torch.manual_seed(22)
    m = ref.TransformerModel(V, d, h, ff, L, 0.2, "gelu", True)
    src = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    m.eval()
    with torch.no_grad():
        try:
            torch.backends.mha.set_fastpath_enabled(False)
        except Exception:
            pass
        logits = m(src)
        nll = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt, reduction="none")
    save("transformer_baseline", src=npy(src), tgt=npy(tgt), nhead=np.int64(h),
         logits_eval=npy(logits), nll_eval=npy(nll), **pack_sd(m))


def f4_gauss_transformer(gp):
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    torch.manual_seed(23 + gp)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.GaussTransformerModel(V, d, h, ff, L, 0.0, True, gp)
    src = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    m.train()  # GPNN.sample is False under train.py -> deterministic (model.py:1799)
    logits = m(src)
    mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
    kl = m.transformerlayers[0].gpnn.kl_divergence() if 1 <= gp <= 3 else torch.zeros(())
    if not torch.is_tensor(kl):
        kl = torch.tensor(float(kl))
    (mle + kl * 0.05).backward()
    m.eval()
    with torch.no_grad():
        logits_eval = m(src)
    save("gauss_tlm_%d" % gp, src=npy(src), tgt=npy(tgt), nhead=np.int64(h), kl_scale=np.float32(0.05),
         logits_train=npy(logits), logits_eval=npy(logits_eval), mle=npy(mle), kl=npy(kl),
         **pack_sd(m), **grads(m))


def f4_gauss_transformer4():
    """gauss_pos 4: layer 0 uses GPNN2 (random-feature GP, model.py:2036-2102), whose frequencies are
    sampled in train mode with one N(0,1) draw of shape (d_model, 150) -- the first draw of the forward
    (dropout 0), recovered by replaying the seed.  train.py adds no KL for this position."""
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    torch.manual_seed(27)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.GaussTransformerModel(V, d, h, ff, L, 0.0, True, 4)
    src = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    m.train()
    g2 = m.transformerlayers[0].gpnn
    torch.manual_seed(301)
    eps = torch.zeros(g2.input_dim, g2.n_MC_terms).normal_()
    torch.manual_seed(301)
    logits = m(src)
    mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
    mle.backward()
    m.eval()
    with torch.no_grad():
        logits_eval = m(src)
    save("gauss_tlm_4", src=npy(src), tgt=npy(tgt), nhead=np.int64(h), eps=npy(eps), logits_train=npy(logits),
         logits_eval=npy(logits_eval), mle=npy(mle), **pack_sd(m), **grads(m))


# ---------------------------------------------------------------- F2 Bayes LSTM
EPS_ORDER = ("weight_hh_lgstd_1", "weight_ih_lgstd_1", "bias_hh_lgstd_1", "bias_ih_lgstd_1",
             "weight_hh_lgstd_2", "weight_ih_lgstd_2", "bias_hh_lgstd_2", "bias_ih_lgstd_2")


def f2_bayes_rnn(pos):
    V, H, T, B = 40, 12, 5, 3
    torch.manual_seed(31 + pos)
    m = ref.BayesRNNModel("LSTM", V, H, H, 2, 0.0, True, pos)
    x1 = torch.randint(0, V, (T, B))
    x2 = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    kl_scale = float(T) / 77.0
    m.train()
    hidden = m.init_hidden(B)
    kw = {}
    eps_all = []
    for w, x in enumerate((x1, x2)):
        torch.manual_seed(100 + w)
        eps8 = []
        if 1 <= pos <= 4:
            for k in EPS_ORDER:
                eps8.append(torch.zeros_like(getattr(m.rnn, k)).normal_())
        eps_all.append(eps8)
        torch.manual_seed(100 + w)
        hidden = tuple(h.detach() for h in hidden)
        logits, hidden = m(x, hidden)
        kw["logits_train_%d" % w] = npy(logits)
        for j, e in enumerate(eps8):
            kw["eps_%d_%d" % (w, j)] = npy(e)
    mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
    # train.py:337 only asks for the KL when 1 <= pos <= 5 (pos 0 would raise in the reference)
    kl = m.rnn.kl_divergence() if 1 <= pos <= 5 else torch.zeros(())  # train.py:337 adds it for 1..5
    (mle + kl * kl_scale).backward()
    kw.update(h_train=npy(hidden[0]), c_train=npy(hidden[1]), mle=npy(mle), kl=npy(kl))
    m.eval()
    with torch.no_grad():
        hidden = m.init_hidden(B)
        l1, hidden = m(x1, hidden)
        l2, hidden = m(x2, hidden)
    kw.update(logits_eval_0=npy(l1), logits_eval_1=npy(l2), h_eval=npy(hidden[0]), c_eval=npy(hidden[1]),
              x1=npy(x1), x2=npy(x2), tgt=npy(tgt), pos=np.int64(pos), kl_scale=np.float32(kl_scale))
    kw.update(pack_sd(m))
    kw.update(grads(m))
    save("bayes_rnn_pos%d" % pos, **kw)


def f2_rnn_baseline():
    V, H, T, B = 40, 12, 5, 3
    torch.manual_seed(41)
    m = ref.RNNModel("LSTM", V, H, H, 2, 0.2, True)
    x1 = torch.randint(0, V, (T, B))
    x2 = torch.randint(0, V, (T, B))
    m.eval()
    with torch.no_grad():
        hidden = m.init_hidden(B)
        l1, hidden = m(x1, hidden)
        l2, hidden = m(x2, hidden)
    save("rnn_baseline", x1=npy(x1), x2=npy(x2), logits_eval_0=npy(l1), logits_eval_1=npy(l2),
         h_eval=npy(hidden[0]), c_eval=npy(hidden[1]), **pack_sd(m))


# ---------------------------------------------------------------- F8 data layout
def f8_data():
    words = ["<s>", "<unk>"] + ["w%03d" % i for i in range(2, 30)]
    rng = np.random.RandomState(3)
    with tempfile.TemporaryDirectory() as dtmp:
        with open(os.path.join(dtmp, "words.txt"), "w") as f:
            for i, w in enumerate(words):
                f.write("%s %d\n" % (w, i))
        texts = {}
        for split, n in (("train", 40), ("valid", 9), ("test", 7)):
            lines = []
            for _ in range(n):
                ln = 1 + rng.poisson(4)
                toks = [words[rng.randint(2, 30)] if rng.rand() > 0.1 else "oov%d" % rng.randint(9) for _ in range(ln)]
                lines.append(" ".join(toks))
            texts[split] = "\n".join(lines) + "\n"
            with open(os.path.join(dtmp, split + ".txt"), "w") as f:
                f.write(texts[split])
        c = refdata.Corpus(dtmp)
    # batchify / get_batch restated from the call sites (train.py:167-179, 299-303) on the reference's ids
    ids = c.train
    bsz, seq_len = 4, 5
    nbatch = ids.size(0) // bsz
    b = ids.narrow(0, 0, nbatch * bsz).view(bsz, -1).t().contiguous()
    i = 5
    sl = min(seq_len, len(b) - 1 - i)
    save("data_layout", words=np.array(words), train_txt=np.array(texts["train"]), valid_txt=np.array(texts["valid"]),
         test_txt=np.array(texts["test"]), train_ids=npy(c.train), valid_ids=npy(c.valid), test_ids=npy(c.test),
         bsz=np.int64(bsz), seq_len=np.int64(seq_len), batchified=npy(b), get_batch_i=np.int64(i),
         get_batch_data=npy(b[i:i + sl]), get_batch_target=npy(b[i + 1:i + 1 + sl].view(-1)))


# ---------------------------------------------------------------- F7 scorer, F6 train.py checkpoints
def _tiny_corpus(dtmp, nwords=30, seed=5, sizes=(("train", 260), ("valid", 40), ("test", 36))):
    rng = np.random.RandomState(seed)
    words = ["<s>", "<unk>"] + ["w%03d" % i for i in range(2, nwords)]
    with open(os.path.join(dtmp, "words.txt"), "w") as f:
        for i, w in enumerate(words):
            f.write("%s %d\n" % (w, i))
    texts = {}
    for split, n in sizes:
        lines = [" ".join(words[rng.randint(2, nwords)] for _ in range(1 + rng.poisson(5))) for _ in range(n)]
        texts[split] = "\n".join(lines) + "\n"
        with open(os.path.join(dtmp, split + ".txt"), "w") as f:
            f.write(texts[split])
    return words, texts


SCORER_CASES = (
    ("lstm_bayes3", ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Bayesian",
                     "--L_bayes_pos", "3"],
     lambda V: ref.BayesRNNModel("LSTM", V, 12, 12, 2, 0.5, True, 3)),
    ("tlm_ffn", ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                 "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN"],
     lambda V: ref.BayesTransformerModel(V, 16, 4, 32, 2, 0.5, True, "FFN")),
)
# round 4: the Gaussian / Variational branches of the reference scorer (:391-447; GaussRNNModel is built UNTIED there, :428-429)
SCORER_CASES_GP = (
    ("tlm_gauss3", ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                    "--uncertainty", "Gaussian", "--T_gauss_pos", "3"],
     lambda V: ref.GaussTransformerModel(V, 16, 4, 32, 2, 0.5, True, 3)),
    ("lstm_gauss33", ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Gaussian",
                      "--L_gauss_pos", "33"],
     lambda V: ref.GaussRNNModel("LSTM", V, 12, 12, 2, 0.5, False, "33")),
    ("lstm_var11", ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Variational",
                    "--L_v_pos", "11"],
     lambda V: ref.VariationalRNNModel("LSTM", V, 12, 12, 2, 0.5, True, "11")),
)


def f7_scorer(cases=None):
    """Drives the reference scorer's main() on CPU (its hard-coded .cuda() calls are patched to
    identity in this process) and keeps its output file as the vector."""
    import importlib
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    scorer = importlib.import_module("compute_sentence_scores_bayes_jianwei")
    for tag, margs, build in (cases or SCORER_CASES):
        with tempfile.TemporaryDirectory() as dtmp:
            words, _ = _tiny_corpus(dtmp)
            rng = np.random.RandomState(11)
            nb = []
            for u in range(4):
                for n in range(1, 4):
                    ln = rng.randint(0, 7)
                    toks = [words[rng.randint(2, 30)] if rng.rand() > 0.15 else "zzz" for _ in range(ln)]
                    nb.append("utt%d-A-%d %s" % (u, n, " ".join(toks)))
            nbest_txt = "\n".join(nb) + "\n"
            with open(os.path.join(dtmp, "nbest.txt"), "w") as f:
                f.write(nbest_txt)
            torch.manual_seed(77)
            with contextlib.redirect_stdout(io.StringIO()):
                m = build(len(words))
            torch.save(m.state_dict(), os.path.join(dtmp, "model.pt"))
            argv = ["scorer", "--nbest-list", os.path.join(dtmp, "nbest.txt"), "--outfile", os.path.join(dtmp, "out.txt"),
                    "--vocabulary", os.path.join(dtmp, "words.txt"), "--model-path", os.path.join(dtmp, "model.pt")] + margs
            old = sys.argv
            sys.argv = argv
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    scorer.main()
            finally:
                sys.argv = old
            out_txt = open(os.path.join(dtmp, "out.txt")).read()
            save("scorer_" + tag, words=np.array(words), nbest_txt=np.array(nbest_txt), scores_txt=np.array(out_txt),
                 argv=np.array(margs), **pack_sd(m))


SCORER_FULL_SIZE = {
    # fixture -> (scorer flags, reference constructor under torch.manual_seed(1111)): BASELINE.json configs[1] / [2] / [4] at their real
    # sizes, 33,000 words; the parameters are NOT stored -- the model is what the seed gives (tests/test_init_state_cpu.py)
    "scorer_cfg1_from_seed": (["--model", "LSTM", "--emsize", "1024", "--nhid", "1024", "--nlayers", "2", "--uncertainty", "Bayesian",
                               "--L_bayes_pos", "3"], lambda V: ref.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.5, True, 3)),
    "scorer_cfg2_from_seed": (["--model", "Transformer", "--emsize", "512", "--nhid", "4096", "--nlayers", "6", "--nhead", "8",
                               "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN"],
                              lambda V: ref.BayesTransformerModel(V, 512, 8, 4096, 6, 0.5, True, "FFN")),
    "scorer_cfg4_from_seed": (["--model", "Transformer", "--emsize", "512", "--nhid", "4096", "--nlayers", "6", "--nhead", "8",
                               "--uncertainty", "Gaussian", "--T_gauss_pos", "3"],
                              lambda V: ref.GaussTransformerModel(V, 512, 8, 4096, 6, 0.5, True, 3)),
}


def f7_scorer_full_size(names=None):
    """The reference scorer's main() (mean weights, one hypothesis per forward, compute_sentence_scores_bayes_jianwei.py:123-173,
    :237-274) on the BASELINE models at their real sizes: 8 utterances x 5-best over a 33,000-word vocabulary, out-of-vocabulary
    words and an empty hypothesis among them.  Kept: the n-best text, the score file, the flags."""
    import importlib
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    scorer = importlib.import_module("compute_sentence_scores_bayes_jianwei")
    V = 33000
    words = ["<s>", "<unk>"] + ["w%d" % i for i in range(V - 2)]
    for name in (names or list(SCORER_FULL_SIZE)):
        margs, build = SCORER_FULL_SIZE[name]
        with tempfile.TemporaryDirectory() as dtmp:
            with open(os.path.join(dtmp, "words.txt"), "w") as f:
                for i, w in enumerate(words):
                    f.write("%s %d\n" % (w, i))
            rng = np.random.RandomState(13)
            nb = []
            for u in range(8):
                for n in range(1, 6):
                    ln = 0 if (u, n) == (3, 2) else int(rng.randint(1, 31))
                    toks = [words[2 + min(int(rng.pareto(1.1) * 40), V - 3)] if rng.rand() > 0.1 else "zzz" for _ in range(ln)]
                    nb.append(("utt%d-C-%d %s" % (u, n, " ".join(toks))).rstrip())
            nbest_txt = "\n".join(nb) + "\n"
            with open(os.path.join(dtmp, "nbest.txt"), "w") as f:
                f.write(nbest_txt)
            torch.manual_seed(1111)
            with contextlib.redirect_stdout(io.StringIO()):
                m = build(V)
            torch.save(m.state_dict(), os.path.join(dtmp, "model.pt"))
            old = sys.argv
            sys.argv = ["scorer", "--nbest-list", os.path.join(dtmp, "nbest.txt"), "--outfile", os.path.join(dtmp, "out.txt"),
                        "--vocabulary", os.path.join(dtmp, "words.txt"), "--model-path", os.path.join(dtmp, "model.pt")] + margs
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    scorer.main()
            finally:
                sys.argv = old
            out_txt = open(os.path.join(dtmp, "out.txt")).read()
            print(name, out_txt.splitlines()[:3])
            save(name, words_n=np.int64(V), nbest_txt=np.array(nbest_txt), scores_txt=np.array(out_txt), argv=np.array(margs),
                 seed=np.int64(1111))


INTERP_CASES = (
    ("lstm_bayes3_interp", ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty",
                            "Bayesian", "--L_bayes_pos", "3"],
     lambda V: ref.BayesRNNModel("LSTM", V, 12, 12, 2, 0.5, True, 3),
     lambda V: ref.BayesRNNModel("LSTM", V, 12, 12, 2, 0.5, False, 0)),
    ("tlm_ffn_interp", ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                        "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN"],
     lambda V: ref.BayesTransformerModel(V, 16, 4, 32, 2, 0.5, True, "FFN"),
     lambda V: ref.BayesTransformerModel(V, 16, 4, 32, 2, 0.5, True, "none")),
)
# round 4: the Gaussian branch of the reference scorer with --interpolation_flag 1 (:391-398, :426-436)
INTERP_CASES_GP = (
    ("tlm_gauss3_interp", ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                           "--uncertainty", "Gaussian", "--T_gauss_pos", "3"],
     lambda V: ref.GaussTransformerModel(V, 16, 4, 32, 2, 0.5, True, 3),
     lambda V: ref.BayesTransformerModel(V, 16, 4, 32, 2, 0.5, True, "none")),
    ("lstm_gauss33_interp", ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty",
                             "Gaussian", "--L_gauss_pos", "33"],
     lambda V: ref.GaussRNNModel("LSTM", V, 12, 12, 2, 0.5, False, "33"),
     lambda V: ref.BayesRNNModel("LSTM", V, 12, 12, 2, 0.5, False, 0)),
)


def f7_scorer_interp(cases=None):
    """Reference scorer with interpolation: logits of the Bayesian model and of a second, standard
    model (built as the reference builds it, :385-436) mixed with alpha = 0.7 before the log-softmax
    (:157-168).  The second state dict travels under the sd2/ prefix."""
    import importlib
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    scorer = importlib.import_module("compute_sentence_scores_bayes_jianwei")
    for tag, margs, build, build2 in (cases or INTERP_CASES):
        with tempfile.TemporaryDirectory() as dtmp:
            words, _ = _tiny_corpus(dtmp)
            rng = np.random.RandomState(12)
            nb = []
            for u in range(3):
                for n in range(1, 4):
                    ln = rng.randint(1, 7)
                    toks = [words[rng.randint(2, 30)] if rng.rand() > 0.15 else "zzz" for _ in range(ln)]
                    nb.append("utt%d-B-%d %s" % (u, n, " ".join(toks)))
            nbest_txt = "\n".join(nb) + "\n"
            with open(os.path.join(dtmp, "nbest.txt"), "w") as f:
                f.write(nbest_txt)
            torch.manual_seed(78)
            with contextlib.redirect_stdout(io.StringIO()):
                m, m2 = build(len(words)), build2(len(words))
            torch.save(m.state_dict(), os.path.join(dtmp, "model.pt"))
            torch.save(m2.state_dict(), os.path.join(dtmp, "model2.pt"))
            extra = ["--interpolation_flag", "1", "--inter_alpha", "0.7"]
            # the reference's main() overwrites --inter_path with an absolute path of its authors' cluster
            # (:451-454), so the interpolation run calls its compute_scores()/write_scores() directly
            mtype = margs[1]
            vocab = scorer.read_vocab(os.path.join(dtmp, "words.txt"))
            nbest = scorer.load_nbest(os.path.join(dtmp, "nbest.txt"))
            with contextlib.redirect_stdout(io.StringIO()):
                res = scorer.compute_scores(nbest, m, torch.nn.CrossEntropyLoss(), len(vocab), vocab, model_type=mtype,
                                            inter_flag=1, alpha=0.7, model_2=m2)
                scorer.write_scores(res, os.path.join(dtmp, "out.txt"))
            out_txt = open(os.path.join(dtmp, "out.txt")).read()
            sd2 = {"sd2/" + k: (npy(v)[:64] if k.endswith("pos_encoder.pe") else npy(v)) for k, v in m2.state_dict().items()}
            save("scorer_" + tag, words=np.array(words), nbest_txt=np.array(nbest_txt), scores_txt=np.array(out_txt),
                 argv=np.array(margs + extra), **pack_sd(m), **sd2)


def f6_train_checkpoint():
    """Runs the reference train.py itself (subprocess, CPU, 1 epoch, tiny corpus) and keeps the
    checkpoint it saved plus the eval losses: the printed ones (2 decimals) and the same quantity
    recomputed at full precision with the reference model classes on that checkpoint."""
    import re
    import subprocess
    for tag, margs, build in (
        ("tlm_ffn", ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                     "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN"],
         lambda V: ref.BayesTransformerModel(V, 16, 4, 32, 2, 0.0, True, "FFN")),
        ("lstm_bayes3", ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Bayesian",
                         "--L_bayes_pos", "3"],
         lambda V: ref.BayesRNNModel("LSTM", V, 12, 12, 2, 0.0, True, 3)),
    ):
        with tempfile.TemporaryDirectory() as dtmp:
            words, texts = _tiny_corpus(dtmp)
            save_path = os.path.join(dtmp, "model.pt")
            cmd = [sys.executable, os.path.join(REF, "train.py"), "--data", dtmp, "--epochs", "1", "--batch-size", "4",
                   "--seq_len", "7", "--dropout", "0.0", "--lr", "0.5", "--clip", "1.0", "--tied", "--save", save_path,
                   "--log-interval", "5"] + margs
            env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=REF)
            log = subprocess.run(cmd, cwd=dtmp, env=env, capture_output=True, text=True, check=True).stdout
            printed_valid = float(re.search(r"valid loss\s+([0-9.]+)", log).group(1))
            printed_test = float(re.search(r"test loss\s+([0-9.]+)", log).group(1))
            sd = torch.load(save_path, map_location="cpu")
            with contextlib.redirect_stdout(io.StringIO()):
                m = build(len(words))
            m.load_state_dict(sd)
            m.eval()
            c = refdata.Corpus(dtmp)
            out = {}
            for split in ("valid", "test"):
                ids = getattr(c, split)
                bsz, seq_len = 20, 7
                nb = ids.size(0) // bsz
                src = ids.narrow(0, 0, nb * bsz).view(bsz, -1).t().contiguous()
                total = 0.0
                hidden = m.init_hidden(bsz) if hasattr(m, "init_hidden") else None
                with torch.no_grad():
                    for i in range(0, src.size(0) - 1, seq_len):
                        n = min(seq_len, len(src) - 1 - i)
                        data, tgt = src[i:i + n], src[i + 1:i + 1 + n].view(-1)
                        if hidden is None:
                            o = m(data)
                        else:
                            o, hidden = m(data, hidden)
                        total += len(data) * torch.nn.functional.cross_entropy(o.view(-1, len(words)), tgt).item()
                out[split] = total / (len(src) - 1)
            assert abs(out["valid"] - printed_valid) < 0.006 and abs(out["test"] - printed_test) < 0.006, (out, log[-400:])
            save("train_ckpt_" + tag, words=np.array(words), train_txt=np.array(texts["train"]),
                 valid_txt=np.array(texts["valid"]), test_txt=np.array(texts["test"]), argv=np.array(margs),
                 valid_loss=np.float64(out["valid"]), test_loss=np.float64(out["test"]),
                 printed_valid=np.float64(printed_valid), printed_test=np.float64(printed_test),
                 **{"sd/" + k: npy(v) for k, v in sd.items() if not k.endswith("pos_encoder.pe")})


# runs INSIDE the child process before the reference's train.py: instruments torch / math (not the reference) so that
# the quantities train.py only prints with two decimals are kept at full precision
_TRAIN_PROBE = r"""
import math, os, runpy, sys
import numpy as np, torch
if os.environ.get("PROBE_ZERO_DROPOUT"):   # harness switch: every nn.Dropout of the run is built with p = 0 (no mask is drawn)
    _dinit = torch.nn.Dropout.__init__
    def _zdrop(self, p=0.5, inplace=False):
        _dinit(self, 0.0, inplace)
    torch.nn.Dropout.__init__ = _zdrop
rec = {"exp": [], "bwd": [], "snaps": [], "sgd_lr": []}
_exp = math.exp
def _rexp(x):
    rec["exp"].append(float(x)); return _exp(x)
math.exp = _rexp                       # train.py prints math.exp(cur_loss / val_loss / test_loss)
_bw = torch.Tensor.backward
def _rbw(self, *a, **k):
    rec["bwd"].append(float(self.detach())); return _bw(self, *a, **k)
torch.Tensor.backward = _rbw           # loss.backward(): the total loss of every step
_ev = torch.nn.Module.eval
def _rev(self):
    if hasattr(self, "encoder") and hasattr(self, "decoder"):
        rec["snaps"].append({k: v.detach().clone().numpy() for k, v in self.state_dict().items()})
    return _ev(self)
torch.nn.Module.eval = _rev            # evaluate() starts with model.eval(): parameters at the end of every epoch
_sgd = torch.optim.SGD.__init__
def _rsgd(self, params, *a, **k):
    rec["sgd_lr"].append(float(k.get("lr", a[0] if a else 0.0))); return _sgd(self, params, *a, **k)
torch.optim.SGD.__init__ = _rsgd       # a fresh optimizer per LR halving
script, out = sys.argv[1], sys.argv[2]
sys.argv = [script] + sys.argv[3:]
g = runpy.run_path(script, run_name="__main__")
kw = {"exp": np.array(rec["exp"]), "bwd": np.array(rec["bwd"]), "sgd_lr": np.array(rec["sgd_lr"]),
      "final_lr": np.float64(g["lr"]), "counter": np.int64(g["counter"]), "best_val_loss": np.float64(g["best_val_loss"]),
      "test_loss": np.float64(g["test_loss"]), "rows": np.int64(len(g["train_data"]))}
for i, sd in enumerate(rec["snaps"]):
    for k, v in sd.items():
        if not k.endswith("pos_encoder.pe"):
            kw["snap%d/%s" % (i, k)] = v
np.savez(out, **kw)
"""


FULL_SIZE = {
    # fixture name -> (V, B, T, train.py's flags): BASELINE.json configs[0] / [1] / [2] / [4] as written
    "train_cfg0_from_seed": (10000, 20, 35, ["--model", "LSTM", "--emsize", "1024", "--nhid", "1024", "--nlayers", "2", "--uncertainty", "none",
                                             "--dropout", "0.2", "--clip", "1.0", "--lr", "0.1"]),
    "train_cfg1_from_seed": (33000, 64, 35, ["--model", "LSTM", "--emsize", "1024", "--nhid", "1024", "--nlayers", "2", "--uncertainty", "Bayesian",
                                             "--L_bayes_pos", "3", "--dropout", "0.2", "--clip", "1.0", "--lr", "0.1"]),
    "train_headline_from_seed": (33000, 64, 128, ["--model", "Transformer", "--emsize", "512", "--nhid", "4096", "--nlayers", "6", "--nhead", "8",
                                                  "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN", "--dropout", "0.2", "--clip", "1.0",
                                                  "--lr", "0.1"]),
    "train_cfg4_from_seed": (33000, 64, 128, ["--model", "Transformer", "--emsize", "512", "--nhid", "4096", "--nlayers", "6", "--nhead", "8",
                                              "--uncertainty", "Gaussian", "--T_gauss_pos", "3", "--dropout", "0.2", "--clip", "1.0", "--lr", "0.1"]),
    # the same two Bayesian configurations with the HARNESS building every nn.Dropout with p = 0 (name ends in _nodrop, fixture field
    # zero_dropout): weight noise alone -- on our side that is the PRODUCTION path (fused feed-forward with eps handed in, matrix-core
    # attention, fused LSTM steps), not the unfused parity blocks
    "train_headline_nodrop_from_seed": (33000, 64, 128, ["--model", "Transformer", "--emsize", "512", "--nhid", "4096", "--nlayers", "6", "--nhead", "8",
                                                         "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN", "--dropout", "0.0", "--clip", "1.0",
                                                         "--lr", "0.1"]),
    "train_cfg1_nodrop_from_seed": (33000, 64, 35, ["--model", "LSTM", "--emsize", "1024", "--nhid", "1024", "--nlayers", "2", "--uncertainty", "Bayesian",
                                                    "--L_bayes_pos", "3", "--dropout", "0.0", "--clip", "1.0", "--lr", "0.1"]),
}


def f6_headline_from_seed(name="train_headline_from_seed"):
    """(``name``: which of FULL_SIZE; the text below describes configs[2].)  BASELINE.json configs[2] AT ITS REAL SIZE under the reference's own train.py, the recipe's flags (Bayesian Transformer-FFN, 6
    layers, d_model 512, d_ff 4096, 8 heads, 33,000 words, tied, dropout 0.2, clip 1.0, batch 64 x seq_len 128), started from
    ``--seed 1111`` alone: three training steps of 8,192 tokens with weight noise and every dropout site on, then its
    evaluate() on the valid and test text.  Kept: the corpus, the total loss of every step, valid and test loss (no parameters:
    the model is what the seed gives).  ~50 M parameters on the CPU: a few minutes."""
    import subprocess
    (V, B, T, flags), steps = FULL_SIZE[name], 3
    rng = np.random.RandomState(20)
    words = ["<s>", "<unk>"] + ["w%d" % i for i in range(V - 2)]

    def text(ntok):
        lines, left = [], ntok
        while left > 0:
            n = min(left, int(rng.randint(3, 30)))
            ids = 2 + np.minimum((rng.pareto(1.1, n) * 40).astype(np.int64), V - 3)  # heavy tail over the vocabulary
            lines.append(" ".join(words[i] for i in ids))
            left -= n + 1  # + the <s> the tokenizer appends
        return lines
    extra = 40 if B > 41 else B // 2  # rows = steps * T + 1 exactly (text() may add one token)
    texts = {"train": text(B * (steps * T + 1) + extra), "valid": text(20 * (T + 1) + 30), "test": text(20 * (T + 1) + 30)}
    margs = flags + ["--batch-size", str(B), "--seq_len", str(T), "--epochs", "1", "--tied", "--log-interval", "1", "--seed", "1111"]
    with tempfile.TemporaryDirectory() as dtmp:
        with open(os.path.join(dtmp, "words.txt"), "w") as f:
            for i, w in enumerate(words):
                f.write("%s %d\n" % (w, i))
        for k, lines in texts.items():
            with open(os.path.join(dtmp, k + ".txt"), "w") as f:
                f.write("\n".join(lines) + "\n")
        probe = os.path.join(dtmp, "probe.py")
        open(probe, "w").write(_TRAIN_PROBE.replace('rec["snaps"].append(', 'os.environ.get("PROBE_NO_SNAPS") or rec["snaps"].append('))
        out_npz = os.path.join(dtmp, "rec.npz")
        cmd = [sys.executable, probe, os.path.join(REF, "train.py"), out_npz, "--data", dtmp, "--save", os.path.join(dtmp, "model.pt")] + margs
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=REF, OMP_NUM_THREADS="8", PROBE_NO_SNAPS="1")
        if "_nodrop" in name:
            env["PROBE_ZERO_DROPOUT"] = "1"
        run = subprocess.run(cmd, cwd=dtmp, env=env, capture_output=True, text=True)
        assert run.returncode == 0, run.stderr[-3000:]
        z = np.load(out_npz)
        ppl = [ln for ln in run.stdout.splitlines() if " ppl " in ln]
        assert len(ppl) == len(z["exp"]) and len(z["bwd"]) == steps, (len(ppl), len(z["exp"]), len(z["bwd"]))
        valid = [v for ln, v in zip(ppl, z["exp"]) if "end of epoch" in ln]
        test = [v for ln, v in zip(ppl, z["exp"]) if "End of training" in ln]
        interval = [v for ln, v in zip(ppl, z["exp"]) if "batches" in ln]
        print(name, "step losses", [round(float(v), 5) for v in z["bwd"]], "valid", valid, "test", test)
        save(name, words_n=np.int64(V), train_txt=np.array(texts["train"]), valid_txt=np.array(texts["valid"]),
             test_txt=np.array(texts["test"]), argv=np.array(margs), step_loss=z["bwd"], interval_loss=np.array(interval),
             valid_loss=np.array(valid), test_loss=np.float64(test[0]), rows=z["rows"], zero_dropout=np.int64("_nodrop" in name))


def f6_train_trajectory(seed_only=False):
    """``seed_only``: NO saved initial state -- train.py is started with ``--seed 1111`` alone, so the run begins from whatever
    its own constructors draw under that seed (incl. the second construction of ``--uncertainty none``, train.py:196-199, and the
    discarded ``sample_parameters()`` draws of the GP layers); three of the families, fixtures ``train_traj_seed_<tag>.npz``.
    SURVEY 8(c) F6: RNG-free runs of the reference's own train.py (dropout 0, nothing sampled: --uncertainty none;
    Bayesian LSTM position 5 = KL in the loss but no draw, model.py:716; the GP families, whose GPNN.sample is never
    raised by train.py) from an initial state saved here and loaded through its --prior True path (train.py:239-258).
    Kept at full precision: the total loss of every step, the interval means / valid losses / test loss train.py
    prints, the parameters at the end of every epoch, the LR halvings."""
    import re
    import subprocess
    lr_l, lr_t = os.environ.get("TRAJ_LR_LSTM", "1.5"), os.environ.get("TRAJ_LR_TLM", "0.4")
    common = ["--epochs", os.environ.get("TRAJ_EPOCHS", "6"), "--batch-size", "4", "--seq_len", "7", "--dropout", "0.0", "--clip", "1.0", "--tied",
              "--log-interval", "10"] + (["--seed", "1111"] if seed_only else ["--prior", "True"])
    for tag, lr, margs, build in (
        ("lstm_none", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "none"],
         lambda V: ref.RNNModel("LSTM", V, 12, 12, 2, 0.0, True)),
        ("tlm_none", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                      "--uncertainty", "none"],
         lambda V: ref.TransformerModel(V, 16, 4, 32, 2, 0.0, "gelu", True)),
        ("lstm_bayes5", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Bayesian",
                         "--L_bayes_pos", "5"],
         lambda V: ref.BayesRNNModel("LSTM", V, 12, 12, 2, 0.0, True, 5)),
        ("tlm_gauss3", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                        "--uncertainty", "Gaussian", "--T_gauss_pos", "3"],
         lambda V: ref.GaussTransformerModel(V, 16, 4, 32, 2, 0.0, True, 3)),
        ("lstm_gauss33", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Gaussian",
                          "--L_gauss_pos", "33"],
         lambda V: ref.GaussRNNModel("LSTM", V, 12, 12, 2, 0.0, True, "33")),
        ("lstm_var00", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Variational",
                        "--L_v_pos", "00"],
         lambda V: ref.VariationalRNNModel("LSTM", V, 12, 12, 2, 0.0, True, "00")),
        # seed_only alone -- runs WITH weight noise (dropout 0): the eps of every training step comes from torch's CPU generator,
        # which train.py seeded and the constructors advanced (bayeslms_amd.train --noise-source torch draws the same)
        # --T_bayes_pos FFN / MHA: their layer 0 is built with a hard-coded dropout of 0.2 (model.py:1202,1207) whose masks come
        # from the same generator and cannot be followed; recorded with the HARNESS building every nn.Dropout with p = 0
        # (PROBE_ZERO_DROPOUT in the probe above; the fixture says so: zero_dropout = 1), which leaves the weight noise alone
        ("noisy_tlm_bayes_ffn_nodrop", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                        "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN"], None),
        ("noisy_tlm_bayes_mha_nodrop", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                        "--uncertainty", "Bayesian", "--T_bayes_pos", "MHA"], None),
        ("noisy_tlm_bayes_emb", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                 "--uncertainty", "Bayesian", "--T_bayes_pos", "EMB"], None),
        ("noisy_lstm_bayes3", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Bayesian",
                               "--L_bayes_pos", "3"], None),
        ("noisy_lstm_var11", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Variational",
                              "--L_v_pos", "11"], None),
        # ... and WITH dropout 0.2 as well (the LSTM language models: embedding, nn.LSTM's inter-layer and output dropout all draw
        # their masks through torch's CPU dropout; a later --dropout wins over the common 0.0)
        ("noisy_drop_lstm_none", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "none",
                                  "--dropout", "0.2"], None),
        ("noisy_drop_lstm_bayes3", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Bayesian",
                                    "--L_bayes_pos", "3", "--dropout", "0.2"], None),
        ("noisy_drop_lstm_gauss33", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Gaussian",
                                     "--L_gauss_pos", "33", "--dropout", "0.2"], None),
        ("noisy_drop_lstm_var11", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Variational",
                                   "--L_v_pos", "11", "--dropout", "0.2"], None),
        # the Transformers with --dropout 0.2: positional-encoding, attention-probability, dropout1, feed-forward and dropout2 masks
        # (for FFN / MHA on top of layer 0's hard-coded 0.2), every one of them torch's CPU dropout
        ("noisy_drop_tlm_none", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                 "--uncertainty", "none", "--dropout", "0.2"], None),
        ("noisy_drop_tlm_bayes_ffn", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                      "--uncertainty", "Bayesian", "--T_bayes_pos", "FFN", "--dropout", "0.2"], None),
        ("noisy_drop_tlm_bayes_mha", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                      "--uncertainty", "Bayesian", "--T_bayes_pos", "MHA", "--dropout", "0.2"], None),
        ("noisy_drop_tlm_bayes_emb", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                      "--uncertainty", "Bayesian", "--T_bayes_pos", "EMB", "--dropout", "0.2"], None),
        ("noisy_drop_tlm_gauss3", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                   "--uncertainty", "Gaussian", "--T_gauss_pos", "3", "--dropout", "0.2"], None),
        # GPNN2 (random features: fresh frequencies drawn at EVERY call in train mode, model.py:2064-2066): once per forward in the
        # Transformer layer, once per time step inside the GP-LSTM cell
        ("noisy_drop_tlm_gauss4", lr_t, ["--model", "Transformer", "--emsize", "16", "--nhid", "32", "--nlayers", "2", "--nhead", "4",
                                   "--uncertainty", "Gaussian", "--T_gauss_pos", "4", "--dropout", "0.2"], None),
        ("noisy_drop_lstm_gauss34", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Gaussian",
                                     "--L_gauss_pos", "34", "--dropout", "0.2"], None),
        ("noisy_drop_lstm_gauss74", lr_l, ["--model", "LSTM", "--emsize", "12", "--nhid", "12", "--nlayers", "2", "--uncertainty", "Gaussian",
                                     "--L_gauss_pos", "74", "--dropout", "0.2"], None),
    ):
        if (seed_only and tag not in ("lstm_none", "tlm_gauss3", "lstm_gauss33") and build is not None) or (not seed_only and build is None):
            continue
        if os.environ.get("TRAJ_ONLY") and tag not in os.environ["TRAJ_ONLY"].split(","):
            continue
        with tempfile.TemporaryDirectory() as dtmp:
            words, texts = _tiny_corpus(dtmp)
            prior_dir = os.path.join(dtmp, "prior")
            if not seed_only:
                torch.manual_seed(61)
                with contextlib.redirect_stdout(io.StringIO()):
                    m0 = build(len(words))
                os.makedirs(prior_dir)
                init = {k: v.detach().clone() for k, v in m0.state_dict().items()}
                torch.save(init, os.path.join(prior_dir, "model.pt"))
            probe = os.path.join(dtmp, "probe.py")
            open(probe, "w").write(_TRAIN_PROBE)
            out_npz = os.path.join(dtmp, "rec.npz")
            cmd = [sys.executable, probe, os.path.join(REF, "train.py"), out_npz, "--data", dtmp, "--lr", lr,
                   "--save", os.path.join(dtmp, "model.pt")] + ([] if seed_only else ["--prior_path", prior_dir]) + common + margs
            env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=REF, OMP_NUM_THREADS="1")
            if tag.endswith("_nodrop"):
                env["PROBE_ZERO_DROPOUT"] = "1"
            run = subprocess.run(cmd, cwd=dtmp, env=env, capture_output=True, text=True)
            assert run.returncode == 0, run.stderr[-3000:]
            log = run.stdout
            z = np.load(out_npz)
            ppl_lines = [ln for ln in log.splitlines() if " ppl " in ln]
            assert len(ppl_lines) == len(z["exp"]), (len(ppl_lines), len(z["exp"]))
            interval, valid, test = [], [], None
            for ln, v in zip(ppl_lines, z["exp"]):
                if "batches" in ln:
                    interval.append(v)
                elif "end of epoch" in ln:
                    valid.append(v)
                elif "End of training" in ln:
                    test = v
            kl_printed = [float(x) for x in re.findall(r"kl_loss\s+([0-9.eE+-]+)", log)]
            nsnap = 1 + max(int(k[4:].split("/")[0]) for k in z.files if k.startswith("snap"))
            assert nsnap == len(valid) + 1 and abs(test - float(z["test_loss"])) == 0.0
            halved = [i + 1 for i in range(len(valid)) if i > 0 and not valid[i] < min(valid[:i])]
            assert len(z["sgd_lr"]) == 1 + len(halved) and (len(halved) >= 1 or seed_only), (valid, z["sgd_lr"])
            margins = [abs(valid[i] - min(valid[:i])) / min(valid[:i]) for i in range(1, len(valid))]
            print(tag, "valid", [round(float(v), 4) for v in valid], "halved at", halved, "test", round(float(test), 4),
                  "min decision margin %.1e" % min(margins))
            kw = {} if seed_only else {"init/" + k: npy(v) for k, v in init.items() if not k.endswith("pos_encoder.pe")}
            kw.update({k: z[k] for k in z.files if k.startswith("snap")})
            save("train_traj_" + ("seed_" if seed_only else "") + tag, words=np.array(words), train_txt=np.array(texts["train"]),
                 valid_txt=np.array(texts["valid"]), test_txt=np.array(texts["test"]), argv=np.array(common + margs + ["--lr", lr]),
                 step_loss=z["bwd"], interval_loss=np.array(interval), valid_loss=np.array(valid), test_loss=np.float64(test),
                 kl_printed=np.array(kl_printed), sgd_lr=z["sgd_lr"], halved_epochs=np.array(halved, dtype=np.int64),
                 final_lr=z["final_lr"], rows=z["rows"], zero_dropout=np.int64(tag.endswith("_nodrop")), **kw)


def f5_gauss_variational_rnn():
    V, H, T, B = 40, 12, 5, 3
    for gp in ("33", "31", "13", "23", "43", "330", "6360", "3333", "53", "73", "00"):
        torch.manual_seed(51 + len(gp) + int(gp[0]))
        with contextlib.redirect_stdout(io.StringIO()):
            m = ref.GaussRNNModel("LSTM", V, H, H, 2, 0.0, True, gp)
        x1, x2 = torch.randint(0, V, (T, B)), torch.randint(0, V, (T, B))
        tgt = torch.randint(0, V, (T * B,))
        m.train()
        hid = m.init_hidden(B)
        l1, hid = m(x1, hid)
        hid = tuple(h.detach() for h in hid)
        l2, hid = m(x2, hid)
        mle = torch.nn.functional.cross_entropy(l2.view(-1, V), tgt)
        kl = torch.zeros(())
        if int(gp[0]) > 0 and 0 < int(gp[1]) <= 3:  # train.py:366-376
            if len(gp) < 3:
                kl = m.rnn.rnn[0].gpnn.kl_divergence()
            elif len(gp) == 3:
                kl = m.rnn.rnn[1].gpnn.kl_divergence()
            else:
                kl = m.rnn.rnn[0].gpnn.kl_divergence() + m.rnn.rnn[1].gpnn.kl_divergence()
        if not torch.is_tensor(kl):
            kl = torch.tensor(float(kl))
        (mle + kl * 0.07).backward()
        m.eval()
        with torch.no_grad():
            hid = m.init_hidden(B)
            e1, hid = m(x1, hid)
            e2, hid = m(x2, hid)
        save("gauss_rnn_" + gp, x1=npy(x1), x2=npy(x2), tgt=npy(tgt), kl_scale=np.float32(0.07), logits_train_0=npy(l1),
             logits_train_1=npy(l2), mle=npy(mle), kl=npy(kl), logits_eval_0=npy(e1), logits_eval_1=npy(e2),
             h_eval=npy(hid[0]), c_eval=npy(hid[1]), **pack_sd(m), **grads(m))
    for vp in ("00", "01", "10", "11"):
        torch.manual_seed(61 + int(vp))
        m = ref.VariationalRNNModel("LSTM", V, H, H, 2, 0.0, True, vp)
        x1 = torch.randint(0, V, (T, B))
        tgt = torch.randint(0, V, (T * B,))
        m.train()
        torch.manual_seed(200)
        eps = {}
        for c in (0, 1):  # cell 0 draws its T rows first, then cell 1 (model.py:2503-2512)
            if int(vp[c]) == 1:
                eps[c] = torch.stack([torch.zeros(1, H).normal_(0, 0.1)[0] for _ in range(T)])
        torch.manual_seed(200)
        l1, hid = m(x1, m.init_hidden(B))
        mle = torch.nn.functional.cross_entropy(l1.view(-1, V), tgt)
        kl = torch.zeros(())
        for c in (0, 1):  # train.py:379-382
            if int(vp[c]) == 1:
                kl = kl + m.rnn.rnn[c].vnn.kl_divergence()
        (mle + kl * 0.07).backward()
        m.eval()
        with torch.no_grad():
            e1, hid = m(x1, m.init_hidden(B))
        kw = dict(x1=npy(x1), tgt=npy(tgt), kl_scale=np.float32(0.07), logits_train_0=npy(l1), mle=npy(mle), kl=npy(kl),
                  logits_eval_0=npy(e1), h_eval=npy(hid[0]), c_eval=npy(hid[1]))
        for c, e in eps.items():
            kw["eps_%d" % c] = npy(e)
        save("variational_rnn_" + vp, **kw, **pack_sd(m), **grads(m))
    for v_pos in (0, 1, 2, 3):
        torch.manual_seed(71 + v_pos)
        m = ref.VTransformerModel(50, 16, 4, 32, 4, 0.0, True, v_pos)
        src = torch.randint(0, 50, (6, 3))
        m.eval()
        with torch.no_grad():
            out = m(src)
        save("vtransformer_%d" % v_pos, src=npy(src), nhead=np.int64(4), logits_eval=npy(out), **pack_sd(m))


def f5_vtransformer_11():
    """`--T_v_pos 11`, the literal flag of BASELINE.json configs[4] (README.md:85-93 writes the option as 00/01/10/11):
    model.py:2822-2843 handles 0..3 only, so the int 11 builds ZERO encoder layers -- embedding * sqrt(d) + positional
    table straight into the tied decoder.  Eval logits, and one train-mode step (dropout 0: RNG-free) with every gradient."""
    torch.manual_seed(82)
    m = ref.VTransformerModel(50, 16, 4, 32, 4, 0.0, True, 11)
    assert len(m.transformerlayers) == 0
    src = torch.randint(0, 50, (6, 3))
    tgt = torch.randint(0, 50, (18,))
    m.eval()
    with torch.no_grad():
        out = m(src)
    m.train()
    lt = m(src)
    mle = torch.nn.functional.cross_entropy(lt.view(-1, 50), tgt)
    mle.backward()
    save("vtransformer_11", src=npy(src), tgt=npy(tgt), nhead=np.int64(4), logits_eval=npy(out), logits_train=npy(lt),
         mle=npy(mle), **pack_sd(m), **grads(m))


def f5_gauss_rnn_gpnn2():
    """--L_gauss_pos with type digit 4: GPNN2 inside the GP-LSTM cells (model.py:1698-1702, 1763-1770).
    Every GPNN2 call of the time loop draws fresh frequencies in train mode; layer 0 runs all its steps
    before layer 1, so the draws are replayed in that order (T tensors per GPNN2 cell and window)."""
    V, H, T, B = 40, 12, 5, 3
    for gp in ("34", "14", "64", "74", "54", "340", "3464"):
        torch.manual_seed(81 + len(gp) + int(gp[0]))
        with contextlib.redirect_stdout(io.StringIO()):
            m = ref.GaussRNNModel("LSTM", V, H, H, 2, 0.0, True, gp)
        x1, x2 = torch.randint(0, V, (T, B)), torch.randint(0, V, (T, B))
        tgt = torch.randint(0, V, (T * B,))
        cells = [c for c in (0, 1) if hasattr(m.rnn.rnn[c], "gpnn") and isinstance(m.rnn.rnn[c].gpnn, ref.GPNN2)]
        m.train()
        hid = m.init_hidden(B)
        kw = {}
        outs = []
        for w, x in enumerate((x1, x2)):
            torch.manual_seed(400 + w)
            for c in cells:
                g2 = m.rnn.rnn[c].gpnn
                for t in range(T):
                    kw["eps_%d_%d_%d" % (w, c, t)] = npy(torch.zeros(g2.input_dim, g2.n_MC_terms).normal_())
            torch.manual_seed(400 + w)
            hid = tuple(h.detach() for h in hid)
            l, hid = m(x, hid)
            outs.append(l)
        mle = torch.nn.functional.cross_entropy(outs[1].view(-1, V), tgt)
        mle.backward()  # train.py:367 adds no KL when the type digit is 4
        m.eval()
        with torch.no_grad():
            hid = m.init_hidden(B)
            e1, hid = m(x1, hid)
            e2, hid = m(x2, hid)
        save("gauss_rnn_" + gp, x1=npy(x1), x2=npy(x2), tgt=npy(tgt), logits_train_0=npy(outs[0]), logits_train_1=npy(outs[1]),
             mle=npy(mle), logits_eval_0=npy(e1), logits_eval_1=npy(e2), h_eval=npy(hid[0]), c_eval=npy(hid[1]),
             cells=np.array(cells, dtype=np.int64), **kw, **pack_sd(m), **grads(m))

def _gp_sample_eps(g):
    """The eps buffers the GPNN's last forward used (model.py:1855-1861: sample_parameters() overwrites them at the
    start of every forward of the enclosing layer / cell, so after the forward they ARE that forward's draw)."""
    out = {}
    for name in ("coef", "weights", "bias"):
        if hasattr(g, name + "_lgstd"):
            out[name] = npy(getattr(g, name + "_sample"))
    return out


def f4_gauss_transformer_sample(gp):
    """GPNN with ``sample`` raised (model.py:1863-1884): coef / weights / bias = mean + exp(lgstd) * eps in train mode.
    No reference entry point sets the flag; the fixture pins the branch itself (sample False AND True)."""
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    torch.manual_seed(123 + gp)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.GaussTransformerModel(V, d, h, ff, L, 0.0, True, gp)
    src = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    g = m.transformerlayers[0].gpnn
    g.sample = True
    m.train()
    logits = m(src)
    eps = {"eps_" + k: v for k, v in _gp_sample_eps(g).items()}
    mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
    kl = g.kl_divergence()
    (mle + kl * 0.05).backward()
    m.eval()
    with torch.no_grad():
        logits_eval = m(src)
    save("gauss_tlm_%d_sample" % gp, src=npy(src), tgt=npy(tgt), nhead=np.int64(h), kl_scale=np.float32(0.05),
         logits_train=npy(logits), logits_eval=npy(logits_eval), mle=npy(mle), kl=npy(kl), **eps,
         **pack_sd(m), **grads(m))


def f5_gauss_rnn_sample():
    """GP-LSTM cells with GPNN.sample raised: GPLSTMCell.forward redraws the eps buffers ONCE per call (model.py:1721-1723),
    so all T steps of a window share one draw and the second window gets a fresh one."""
    V, H, T, B = 40, 12, 5, 3
    for gp in ("33", "31", "32", "13", "23", "43", "53", "63", "73", "330", "3333"):
        torch.manual_seed(151 + len(gp) + int(gp[0]) + 7 * int(gp[1]))
        with contextlib.redirect_stdout(io.StringIO()):
            m = ref.GaussRNNModel("LSTM", V, H, H, 2, 0.0, True, gp)
        x1, x2 = torch.randint(0, V, (T, B)), torch.randint(0, V, (T, B))
        tgt = torch.randint(0, V, (T * B,))
        cells = [c for c in (0, 1) if hasattr(m.rnn.rnn[c], "gpnn")]
        for c in cells:
            m.rnn.rnn[c].gpnn.sample = True
        m.train()
        hid = m.init_hidden(B)
        kw, outs = {}, []
        for w, x in enumerate((x1, x2)):
            hid = tuple(h.detach() for h in hid)
            l, hid = m(x, hid)
            outs.append(l)
            for c in cells:
                for k, v in _gp_sample_eps(m.rnn.rnn[c].gpnn).items():
                    kw["eps_%d_%d_%s" % (w, c, k)] = v
        mle = torch.nn.functional.cross_entropy(outs[1].view(-1, V), tgt)
        kl = sum(m.rnn.rnn[c].gpnn.kl_divergence() for c in cells)  # train.py:366-376
        if not torch.is_tensor(kl):
            kl = torch.tensor(float(kl))
        (mle + kl * 0.07).backward()
        m.eval()
        with torch.no_grad():
            hid = m.init_hidden(B)
            e1, hid = m(x1, hid)
            e2, hid = m(x2, hid)
        save("gauss_rnn_%s_sample" % gp, x1=npy(x1), x2=npy(x2), tgt=npy(tgt), kl_scale=np.float32(0.07),
             logits_train_0=npy(outs[0]), logits_train_1=npy(outs[1]), mle=npy(mle), kl=npy(kl), logits_eval_0=npy(e1),
             logits_eval_1=npy(e2), h_eval=npy(hid[0]), c_eval=npy(hid[1]), cells=np.array(cells, dtype=np.int64), **kw,
             **pack_sd(m), **grads(m))


# ---------------------------------------------------------------- F9 architecture search
def _load_search():
    """model_search_bayes.py / architect.py call ``.cuda()`` on sub-modules while building
    (model_search_bayes.py:99,258); with no GPU here the harness maps ``.cuda()`` to identity."""
    torch.nn.Module.cuda = lambda self, device=None: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    with contextlib.redirect_stdout(io.StringIO()):
        import model_search_bayes as S
        import architect as A
    return S, A


def _gp_eps(layer):
    g = layer.gpnn
    return {"coef": npy(g.coef_sample), "weights": npy(g.weights_sample), "bias": npy(g.bias_sample)}


def f9_search_models():
    """One forward/backward of the two super-nets train_search_bayes.py builds (:158-163)."""
    S, _ = _load_search()
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    for sample in (False, True):
        torch.manual_seed(301 + sample)
        with contextlib.redirect_stdout(io.StringIO()):
            m = S.GaussTransModelSearch(V, d, h, ff, L, 0.0, True)
        m.weights.data.copy_(torch.randn(L, 1, 2) * 0.7)
        src = torch.randint(0, V, (T, B))
        tgt = torch.randint(0, V, (T * B,))
        m.train()
        for lyr in m.transformerlayers:
            lyr.gpnn.sample = sample
        logits = m(src)
        kw = {}
        for i, lyr in enumerate(m.transformerlayers):  # the eps of THIS forward (layer forward redraws first, :232-233)
            for k, v in _gp_eps(lyr).items():
                kw["eps_%d_%s" % (i, k)] = v
        mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
        kl = sum(lyr.gpnn.kl_divergence() for lyr in m.transformerlayers)
        (mle + kl * 0.05).backward()
        m.eval()
        with torch.no_grad():
            logits_eval = m(src)
        save("search_gauss_tlm_%d" % sample, src=npy(src), tgt=npy(tgt), nhead=np.int64(h), kl_scale=np.float32(0.05),
             arch=npy(m.weights), arch_grad=npy(m.weights.grad), logits_train=npy(logits), logits_eval=npy(logits_eval),
             mle=npy(mle), kl=npy(kl), **kw, **pack_sd(m), **grads(m))
    V, H, T, B = 40, 12, 5, 3
    for sample in (False, True):
        torch.manual_seed(311 + sample)
        with contextlib.redirect_stdout(io.StringIO()):
            m = S.BayesLSTMModelSearch("LSTM", V, H, H, 2, 0.0, True)
        m.weights.data.copy_(torch.randn(2, 4, 2) * 0.7)
        x1, x2 = torch.randint(0, V, (T, B)), torch.randint(0, V, (T, B))
        tgt = torch.randint(0, V, (T * B,))
        m.train()
        gates = ("ingate", "forgate", "cellgate", "outgate")
        for c in m.rnn.rnn:
            for g in gates:
                getattr(c, "bayes_" + g).sample = sample
        hid = m.init_hidden(B)
        l1, hid = m(x1, hid)
        hid = tuple(t.detach() for t in hid)
        l2, hid = m(x2, hid)
        kw = {}
        for ci, c in enumerate(m.rnn.rnn):  # eps of the SECOND window (cell forward redraws first, :662-665)
            for g in gates:
                b = getattr(c, "bayes_" + g)
                kw["eps_%d_%s_w" % (ci, g)] = npy(b.weights_sample)
                kw["eps_%d_%s_b" % (ci, g)] = npy(b.bias_sample)
        mle = torch.nn.functional.cross_entropy(l2.view(-1, V), tgt)
        kl = torch.zeros(())
        for c in m.rnn.rnn:  # train_search_bayes.py:312-318 evaluates the KL with sample = True
            for g in gates:
                b = getattr(c, "bayes_" + g)
                old, b.sample = b.sample, True
                kl = kl + b.kl_divergence()
                b.sample = old
        (mle + kl * 0.07).backward()
        m.eval()
        with torch.no_grad():
            hid = m.init_hidden(B)
            e1, hid = m(x1, hid)
            e2, hid = m(x2, hid)
        save("search_bayes_lstm_%d" % sample, x1=npy(x1), x2=npy(x2), tgt=npy(tgt), kl_scale=np.float32(0.07),
             arch=npy(m.weights), arch_grad=npy(m.weights.grad), logits_train_1=npy(l2), mle=npy(mle), kl=npy(kl),
             logits_eval_0=npy(e1), logits_eval_1=npy(e2), h_eval=npy(hid[0]), c_eval=npy(hid[1]), **kw, **pack_sd(m),
             **grads(m))


def f9_search_bayes_tlm():
    """BayesTransModelSearch (model_search_bayes.py:33-194): FFN output mixed from a standard and a Bayesian
    linear2 by Gumbel-softmax'd logits.  Draw order per layer: the (1,2) uniform of the Gumbel sample (:26), then
    the eps of bayes_linear2 (model.py:1087); replayed from the same seed to recover both."""
    S, _ = _load_search()
    V, d, h, ff, L, T, B = 50, 16, 4, 32, 2, 6, 3
    torch.manual_seed(331)
    with contextlib.redirect_stdout(io.StringIO()):
        m = S.BayesTransModelSearch(V, d, h, ff, L, 0.0, True)
    zero_dropout(m)  # dropout2 is a hard-coded 0.1 (:51)
    m.weights.data.copy_(torch.randn(L, 1, 2) * 0.7)
    src = torch.randint(0, V, (T, B))
    tgt = torch.randint(0, V, (T * B,))
    m.train()
    torch.manual_seed(17)
    kw = {}
    for i in range(L):
        kw["u_%d" % i] = npy(torch.zeros(1, 2).uniform_(0, 1))
        kw["eps_%d" % i] = npy(torch.zeros(d, ff).normal_(0, 1))
    torch.manual_seed(17)
    logits = m(src)
    mle = torch.nn.functional.cross_entropy(logits.view(-1, V), tgt)
    kl = sum(lyr.bayes_linear2.kl_divergence() for lyr in m.transformerlayers)
    (mle + kl * 0.05).backward()
    m.eval()
    for lyr in m.transformerlayers:
        lyr.gumble_flag = False  # eval-mode fixture without the Gumbel draw: probs = the raw logits (:61)
    with torch.no_grad():
        logits_eval = m(src)
    save("search_bayes_tlm", src=npy(src), tgt=npy(tgt), nhead=np.int64(h), kl_scale=np.float32(0.05), arch=npy(m.weights),
         arch_grad=npy(m.weights.grad), logits_train=npy(logits), logits_eval_nogumbel=npy(logits_eval), mle=npy(mle),
         kl=npy(kl), **kw, **pack_sd(m), **grads(m))


def f9_search_loop():
    """The alternating loop of train_search_bayes.py:203-290 (first-order Architect step on a validation
    window, then the SGD(momentum 0.9, weight_decay 1e-5) network step), driven with the reference's own
    Architect and model classes for a few windows; every quantity at full precision."""
    import types
    S, A = _load_search()
    nsteps, T, B, lr, clip = 6, 6, 3, 0.5, 1.0
    for kind in ("tlm", "lstm"):
        torch.manual_seed(321)
        V = 40
        with contextlib.redirect_stdout(io.StringIO()):
            if kind == "tlm":
                m = S.GaussTransModelSearch(V, 16, 4, 32, 2, 0.0, True)
            else:
                m = S.BayesLSTMModelSearch("LSTM", V, 12, 12, 2, 0.0, True)
        args = types.SimpleNamespace(wdecay=5e-7, clip=clip, arch_lr=3e-3, arch_wdecay=1e-3)
        arch = A.Architect(m, V, args)
        opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=0.9, weight_decay=1e-5)
        train = torch.randint(0, V, (nsteps * T + 1, B))
        valid = torch.randint(0, V, (nsteps * T + 1, B))
        kw = {"init/" + k: (npy(v)[:64] if k.endswith("pos_encoder.pe") else npy(v)).copy() for k, v in m.state_dict().items()}
        kw["arch_init"] = npy(m.weights).copy()
        kl_scale = 0.01
        gates = ("ingate", "forgate", "cellgate", "outgate")
        m.train()
        if kind == "lstm":
            hidden = m.init_hidden(B)
            hidden_valid = m.init_hidden(B)
        losses, kls, archs = [], [], []
        for s in range(nsteps):
            data, tg = train[s * T:(s + 1) * T], train[s * T + 1:(s + 1) * T + 1].reshape(-1)
            dv, tv = valid[s * T:(s + 1) * T], valid[s * T + 1:(s + 1) * T + 1].reshape(-1)
            opt.zero_grad()
            if kind == "tlm":
                arch.step(data, tg, dv, tv, opt, False)
            else:
                arch.step(data, tg, dv, tv, opt, False, hidden_valid)
            archs.append(npy(m.weights).copy())
            opt.zero_grad()
            if kind == "tlm":
                for lyr in m.transformerlayers:
                    lyr.gpnn.sample = True
                out = m(data)
                for i, lyr in enumerate(m.transformerlayers):
                    for k, v in _gp_eps(lyr).items():
                        kw["eps_%d_%d_%s" % (s, i, k)] = v
                kl = sum(lyr.gpnn.kl_divergence() for lyr in m.transformerlayers) * kl_scale  # --T_bayes_pos FFN
                for lyr in m.transformerlayers:
                    lyr.gpnn.sample = False
            else:
                hidden = tuple(t.detach() for t in hidden)
                out, hidden = m(data, hidden)
                kl = torch.zeros(())
                for c in m.rnn.rnn:  # --L_bayes_pos > 0
                    for g in gates:
                        b = getattr(c, "bayes_" + g)
                        b.sample = True
                        kl = kl + b.kl_divergence() * kl_scale
                        b.sample = False
            mle = torch.nn.functional.cross_entropy(out.view(-1, V), tg)
            (mle + kl).backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
            opt.step()
            losses.append(float(mle.detach()))
            kls.append(float(kl.detach()))
        m.eval()
        with torch.no_grad():
            if kind == "tlm":
                ev = m(valid[:T])
            else:
                ev, _ = m(valid[:T], m.init_hidden(B))
        save("search_loop_" + kind, train=npy(train), valid=npy(valid), nhead=np.int64(4), lr=np.float64(lr),
             clip=np.float64(clip), kl_scale=np.float64(kl_scale), T=np.int64(T), mle=np.array(losses), kl=np.array(kls),
             arch_after=np.stack(archs), logits_eval=npy(ev), **kw, **pack_sd(m))



# ---------------------------------------------------------------- F10 initial state under a torch seed
def init_state_cases():
    """name -> (module, class, constructor arguments): every constructor train.py / train_search_bayes.py can reach
    (train.py:186-221, train_search_bayes.py:158-163), tied and untied, every position / type string the parity
    fixtures use."""
    V = 50
    cases = {}
    for tied in (True, False):
        cases["rnn_none_t%d" % tied] = ("model", "RNNModel", ["LSTM", V, 12, 12 if tied else 20, 2, 0.2, tied])
        cases["tlm_none_t%d" % tied] = ("model", "TransformerModel", [V, 16, 4, 32, 3, 0.2, "gelu", tied])
    for pos in range(6):
        cases["rnn_bayes%d" % pos] = ("model", "BayesRNNModel", ["LSTM", V, 12, 12, 2, 0.2, True, pos])
    cases["rnn_bayes2_untied"] = ("model", "BayesRNNModel", ["LSTM", V, 12, 12, 2, 0.2, False, 2])
    for g in ("00", "13", "23", "33", "330", "3333", "6360", "34", "74", "53", "43", "63", "73", "14", "340", "3464", "54",
              "64", "31", "32", "10", "20"):
        cases["rnn_gauss%s" % g] = ("model", "GaussRNNModel", ["LSTM", V, 12, 12, 2, 0.2, True, g])
    for v in ("00", "01", "10", "11"):
        cases["rnn_var%s" % v] = ("model", "VariationalRNNModel", ["LSTM", V, 12, 12, 2, 0.2, True, v])
    for b in ("none", "FFN", "MHA", "EMB"):
        cases["tlm_bayes_%s" % b] = ("model", "BayesTransformerModel", [V, 16, 4, 32, 2, 0.2, True, b])
    for g in range(5):
        cases["tlm_gauss%d" % g] = ("model", "GaussTransformerModel", [V, 16, 4, 32, 3, 0.2, True, g])
    for v in (0, 1, 2, 3, 11):
        cases["tlm_var%d" % v] = ("model", "VTransformerModel", [V, 16, 4, 32, 4, 0.2, True, v])
    cases["search_bayes_tlm"] = ("model_search_bayes", "BayesTransModelSearch", [V, 16, 4, 32, 3, 0.2, True])
    cases["search_bayes_tlm_untied"] = ("model_search_bayes", "BayesTransModelSearch", [V, 16, 4, 32, 2, 0.2, False])
    cases["search_gauss_tlm"] = ("model_search_bayes", "GaussTransModelSearch", [V, 16, 4, 32, 3, 0.2, True])
    cases["search_bayes_lstm"] = ("model_search_bayes", "BayesLSTMModelSearch", ["LSTM", V, 12, 12, 2, 0.2, True])
    cases["search_bayes_lstm_untied"] = ("model_search_bayes", "BayesLSTMModelSearch", ["LSTM", V, 12, 12, 2, 0.2, False])
    return cases


def init_state_cli_cases():
    """name -> (train.py's flags, the constructor calls train.py:193-223 makes for them in order, the LAST one being the
    model it trains): ``--uncertainty none`` builds the model twice (``model_2`` first, :196-199 / :211-214)."""
    V = 50
    base = dict(emsize=16, nhead=4, nhid=32, nlayers=2, dropout=0.2, tied=True, T_bayes_pos="none", L_bayes_pos=0, T_gauss_pos=3,
                L_gauss_pos="00", L_v_pos="11", T_v_pos=0)
    tlm = ("TransformerModel", [V, 16, 4, 32, 2, 0.2, "gelu", True])
    rnn = ("RNNModel", ["LSTM", V, 16, 16, 2, 0.2, True])
    return {
        "cli_tlm_none": (dict(base, model="Transformer", uncertainty="none"), [tlm, tlm]),
        "cli_lstm_none": (dict(base, model="LSTM", uncertainty="none", nhid=16), [rnn, rnn]),
        "cli_tlm_bayes_ffn": (dict(base, model="Transformer", uncertainty="Bayesian", T_bayes_pos="FFN"),
                              [("BayesTransformerModel", [V, 16, 4, 32, 2, 0.2, True, "FFN"])]),
        "cli_lstm_bayes3": (dict(base, model="LSTM", uncertainty="Bayesian", L_bayes_pos=3, nhid=16),
                            [("BayesRNNModel", ["LSTM", V, 16, 16, 2, 0.2, True, 3])]),
    }


def tensor_digest(t):
    """[shape, sha256 of the little-endian bytes, float64 sum] -- the bit-exact identity of one tensor in ~100 bytes
    (the sum is there for the failure message: which way and how far a mismatching tensor is off)."""
    import hashlib
    a = np.ascontiguousarray(npy(t))
    return [list(a.shape), hashlib.sha256(a.tobytes()).hexdigest(), float(a.astype(np.float64).sum())]


def f10_init_state(seed=1111):
    """The state_dict (keys in order, one digest per tensor) and the architecture logits every reference constructor
    leaves behind under ``torch.manual_seed(seed)`` (train.py:122 seeds before it builds the model), plus the first
    values torch's generator yields AFTER the constructor returned: a run that starts from the same seed then also
    meets the same stream in whatever draws next.  Written as JSON: tests/golden/init_state.json."""
    import json
    S, _ = _load_search()
    mods = {"model": ref, "model_search_bayes": S}
    out = {"seed": seed, "torch": torch.__version__, "cases": {}}
    for name, (mod, cls, args) in init_state_cases().items():
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            m = getattr(mods[mod], cls)(*args)
        after = torch.rand(4, dtype=torch.float64).tolist()
        entry = {"module": mod, "cls": cls, "args": args, "state": [[k] + tensor_digest(v) for k, v in m.state_dict().items()],
                 "generator_after": after}
        if hasattr(m, "arch_parameters"):
            entry["arch"] = [tensor_digest(a) for a in m.arch_parameters()]
        out["cases"][name] = entry
    out["ntokens"] = 50
    for name, (flags, calls) in init_state_cli_cases().items():
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            for cls, args in calls:
                m = getattr(ref, cls)(*args)
        after = torch.rand(4, dtype=torch.float64).tolist()
        out["cases"][name] = {"module": "model", "cls": calls[-1][0], "args": calls[-1][1], "cli": flags,
                              "state": [[k] + tensor_digest(v) for k, v in m.state_dict().items()], "generator_after": after}
    path = os.path.join(OUT, "init_state.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "%d constructors" % len(out["cases"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "init":
        f10_init_state()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rnnv":
        f5_gauss_variational_rnn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "search_btlm":
        f9_search_bayes_tlm()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        f9_search_models()
        f9_search_loop()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "scorer_gp":
        f7_scorer(SCORER_CASES_GP)
        f7_scorer_interp(INTERP_CASES_GP)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gp_sample":
        for gp in (1, 2, 3):
            f4_gauss_transformer_sample(gp)
        f5_gauss_rnn_sample()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vt11":
        f5_vtransformer_11()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rnn_gpnn2":
        f5_gauss_rnn_gpnn2()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pos5":
        f2_bayes_rnn(5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gpnn2":
        f4_gauss_transformer4()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "interp":
        f7_scorer_interp()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "traj":
        f6_train_trajectory()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "scorer_full":
        f7_scorer_full_size(sys.argv[2:])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "headline_seed":
        for name in (sys.argv[2:] or list(FULL_SIZE)):
            f6_headline_from_seed(name)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "traj_seed":
        f6_train_trajectory(seed_only=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "late":
        f6_train_checkpoint()
        f7_scorer()
        sys.exit(0)
    f1_bayes_linear()
    for bp in ("FFN", "MHA", "EMB", "none"):
        f3_transformer(bp)
    f3_transformer_baseline()
    for gp in (0, 1, 2, 3):
        f4_gauss_transformer(gp)
    for pos in (0, 1, 2, 3, 4, 5):
        f2_bayes_rnn(pos)
    f2_rnn_baseline()
    f8_data()
    f6_train_checkpoint()
    f7_scorer()
    f7_scorer_interp()
    f4_gauss_transformer4()
    f5_gauss_rnn_gpnn2()
    f9_search_models()
    f9_search_loop()
    f9_search_bayes_tlm()
    for gp in (1, 2, 3):
        f4_gauss_transformer_sample(gp)
    f5_gauss_rnn_sample()
    f7_scorer(SCORER_CASES_GP)
    f7_scorer_interp(INTERP_CASES_GP)
    f6_train_trajectory()
    f5_vtransformer_11()
