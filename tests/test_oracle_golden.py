"""CPU: the oracle (oracle/bayes_oracle.py) against vectors produced by the
reference itself (tests/golden/make_golden.py).  Gate: <= 1e-5 abs on logits."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import bayes_oracle as O
from oracle import philox as P

TOL = dict(rtol=1e-5, atol=2e-6)


def test_philox_known_answers():
    for c, k, out in P.KAT:
        r = P.philox4x32_10(*[np.array([x], np.uint32) for x in c], k[0], k[1])
        assert tuple(int(x[0]) for x in r) == out


def test_philox_normal_moments_and_determinism():
    z = P.normal(1 << 18, 1111, P.STREAM_WEIGHT + 3, 7)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert np.isfinite(z).all()
    np.testing.assert_array_equal(z[:1001], P.normal(1001, 1111, P.STREAM_WEIGHT + 3, 7))
    assert not np.array_equal(z[:64], P.normal(64, 1111, P.STREAM_WEIGHT + 3, 8)[:64])
    keep = P.keep_mask(1 << 18, 0.2, 1111, P.STREAM_DROPOUT, 0)
    assert abs(float(keep.mean()) - 0.8) < 5e-3


def test_bayes_linear_matches_reference():
    g, _, _ = load_golden("bayes_linear")
    x = g["x"].clone().requires_grad_(True)
    mu = g["mu"].clone().requires_grad_(True)
    lg = g["lgstd"].clone().requires_grad_(True)
    y = O.bayes_linear(x, mu, lg, g["eps"])
    kl = O.kl_mean_form(mu, lg)
    torch.testing.assert_close(y, g["y_train"], **TOL)
    torch.testing.assert_close(kl, g["kl"], **TOL)
    torch.testing.assert_close(O.bayes_linear(x, mu, lg, None), g["y_eval"], **TOL)
    ((y * g["g"]).sum() + kl * g["kl_scale"]).backward()
    torch.testing.assert_close(x.grad, g["dx"], **TOL)
    torch.testing.assert_close(mu.grad, g["dmu"], **TOL)
    torch.testing.assert_close(lg.grad, g["dlgstd"], **TOL)
    # closed forms used by the fused HIP epilogue
    dW = torch.einsum("tbn,tbk->nk", g["g"], g["x"])
    n = mu.numel()
    lam = float(g["kl_scale"])
    torch.testing.assert_close(dW + lam * g["mu"] / n, g["dmu"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dW * g["eps"] * torch.exp(g["lgstd"]) + lam * (torch.exp(2 * g["lgstd"]) - 1) / n,
                               g["dlgstd"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("pos", ["FFN", "MHA", "EMB", "none"])
def test_bayes_transformer_matches_reference(pos):
    g, sd, grad = load_golden("bayes_tlm_" + pos)
    nhead = int(g["nhead"])
    logits = O.transformer_lm(g["src"], sd, nhead, None)
    torch.testing.assert_close(logits, g["logits_eval"], **TOL)
    torch.testing.assert_close(O.token_nll(logits, g["tgt"]), g["nll_eval"], **TOL)
    # train mode (dropout 0, eps injected) + every parameter gradient
    leaf = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "pos_encoder.pe") for k, v in sd.items()}
    if "decoder.weight" in leaf:  # tied (model.py:1240): one tensor under two names
        leaf["decoder.weight"] = leaf["encoder.weight"]
    loss, mle, kl = O.transformer_train_loss(g["src"], g["tgt"], leaf, nhead, pos, g.get("eps"), float(g["kl_scale"]))
    torch.testing.assert_close(mle, g["mle"], **TOL)
    torch.testing.assert_close(kl / float(g["kl_scale"]) if pos != "none" else kl, g["kl"], **TOL)
    torch.testing.assert_close(loss, g["loss"], **TOL)
    loss.backward()
    for k, gv in grad.items():
        if k == "decoder.weight":
            continue
        torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)


def test_transformer_baseline_matches_reference():
    g, sd, _ = load_golden("transformer_baseline")
    logits = O.transformer_lm(g["src"], sd, int(g["nhead"]), None)
    torch.testing.assert_close(logits, g["logits_eval"], **TOL)


@pytest.mark.parametrize("gp", [0, 1, 2, 3])
def test_gauss_transformer_matches_reference(gp):
    g, sd, grad = load_golden("gauss_tlm_%d" % gp)
    nhead = int(g["nhead"])
    torch.testing.assert_close(O.transformer_lm(g["src"], sd, nhead, None), g["logits_eval"], **TOL)
    # GPNN.sample stays False under train.py -> train forward == eval forward without dropout
    torch.testing.assert_close(g["logits_train"], g["logits_eval"], **TOL)
    pre = "transformerlayers.0.gpnn."
    kl = torch.zeros(())
    if gp in (1, 3):
        kl = kl + O.kl_mean_form_minus1(sd[pre + "coef_mean"], sd[pre + "coef_lgstd"])
    if gp in (2, 3):
        kl = kl + O.kl_mean_form_minus1(sd[pre + "weights_mean"], sd[pre + "weights_lgstd"])
        kl = kl + O.kl_mean_form_minus1(sd[pre + "bias_mean"], sd[pre + "bias_lgstd"])
    torch.testing.assert_close(kl, g["kl"], **TOL)


def _gp_eps(g, prefix="eps_"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix) and k[len(prefix):] in ("coef", "weights", "bias")}


@pytest.mark.parametrize("gp", [1, 2, 3])
def test_gauss_transformer_sample_branch_matches_reference(gp):
    """GPNN.sample raised (model.py:1863-1884): coef / weights / bias = mean + exp(lgstd) * eps with the eps buffers
    the reference's forward drew; logits, KL and every parameter gradient (the lgstd tensors get both the
    reparameterisation and the KL gradient)."""
    g, sd, grad = load_golden("gauss_tlm_%d_sample" % gp)
    nhead = int(g["nhead"])
    V = sd["encoder.weight"].shape[0]
    eps = _gp_eps(g)
    assert sorted(eps) == {1: ["coef"], 2: ["bias", "weights"], 3: ["bias", "coef", "weights"]}[gp]
    torch.testing.assert_close(O.transformer_lm(g["src"], sd, nhead, None), g["logits_eval"], **TOL)
    assert (g["logits_train"] - g["logits_eval"]).abs().max() > 1e-3  # the branch really fired
    leaf = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "pos_encoder.pe") for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    logits = O.transformer_lm(g["src"], leaf, nhead, eps)
    torch.testing.assert_close(logits, g["logits_train"], **TOL)
    pre = "transformerlayers.0.gpnn."
    kl = torch.zeros(())
    if gp in (1, 3):
        kl = kl + O.kl_mean_form_minus1(leaf[pre + "coef_mean"], leaf[pre + "coef_lgstd"])
    if gp in (2, 3):
        kl = kl + O.kl_mean_form_minus1(leaf[pre + "weights_mean"], leaf[pre + "weights_lgstd"])
        kl = kl + O.kl_mean_form_minus1(leaf[pre + "bias_mean"], leaf[pre + "bias_lgstd"])
    torch.testing.assert_close(kl, g["kl"], **TOL)
    (O.cross_entropy_mean(logits.view(-1, V), g["tgt"]) + kl * float(g["kl_scale"])).backward()
    for k, gv in grad.items():
        if k != "decoder.weight":
            torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)


def test_gauss_transformer_gpnn2_matches_reference():
    """--T_gauss_pos 4: GPNN2 random features in layer 0 (model.py:2036-2076); train mode samples the
    frequencies with the recovered draw; gradients of every parameter against the reference's."""
    g, sd, grad = load_golden("gauss_tlm_4")
    nhead = int(g["nhead"])
    V = sd["encoder.weight"].shape[0]
    torch.testing.assert_close(O.transformer_lm(g["src"], sd, nhead, None), g["logits_eval"], **TOL)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    logits = O.transformer_lm(g["src"], leaf, nhead, g["eps"])
    torch.testing.assert_close(logits, g["logits_train"], **TOL)
    O.cross_entropy_mean(logits.view(-1, V), g["tgt"]).backward()
    for k, v in grad.items():
        if k == "decoder.weight":
            continue
        torch.testing.assert_close(leaf[k].grad, v, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("pos", [0, 1, 2, 3, 4, 5])
def test_bayes_lstm_matches_reference(pos):
    g, sd, grad = load_golden("bayes_rnn_pos%d" % pos)
    B = g["x1"].shape[1]
    H = sd["rnn.weight_hh_mean_1"].shape[1]
    zeros = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    # eval: two windows with hidden carry-over
    l1, hid = O.bayes_rnn_lm(g["x1"], zeros, sd, pos, None)
    l2, hid = O.bayes_rnn_lm(g["x2"], hid, sd, pos, None)
    torch.testing.assert_close(l1, g["logits_eval_0"], **TOL)
    torch.testing.assert_close(l2, g["logits_eval_1"], **TOL)
    torch.testing.assert_close(hid[0], g["h_eval"], **TOL)
    torch.testing.assert_close(hid[1], g["c_eval"], **TOL)
    # train: eps in the reference's draw order, hidden detached between windows
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    hid = zeros
    for w, x in enumerate((g["x1"], g["x2"])):
        eps8 = [g["eps_%d_%d" % (w, j)] for j in range(8)] if 1 <= pos <= 4 else None
        hid = tuple(h.detach() for h in hid)
        logits, hid = O.bayes_rnn_lm(x, hid, leaf, pos, eps8)
        torch.testing.assert_close(logits, g["logits_train_%d" % w], **TOL)
    torch.testing.assert_close(hid[0], g["h_train"], **TOL)
    mle = O.cross_entropy_mean(logits, g["tgt"])
    kl = O.kl_bayes2lstm(leaf, "rnn.", pos)
    torch.testing.assert_close(mle, g["mle"], **TOL)
    torch.testing.assert_close(kl, g["kl"], **TOL)
    (mle + kl * float(g["kl_scale"])).backward()
    for k, gv in grad.items():
        if k == "decoder.weight":
            continue
        torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)


def test_rnn_baseline_matches_reference():
    g, sd, _ = load_golden("rnn_baseline")
    B = g["x1"].shape[1]
    H = sd["rnn.weight_hh_l0"].shape[1]
    hid = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    l1, hid = O.rnn_lm(g["x1"], hid, sd)
    l2, hid = O.rnn_lm(g["x2"], hid, sd)
    torch.testing.assert_close(l1, g["logits_eval_0"], **TOL)
    torch.testing.assert_close(l2, g["logits_eval_1"], **TOL)
    torch.testing.assert_close(hid[1], g["c_eval"], **TOL)


def _write_corpus(g, d):
    import os
    with open(os.path.join(d, "words.txt"), "w") as f:
        f.write("".join("%s %d\n" % (w, i) for i, w in enumerate(g["words"])))
    for s in ("train", "valid", "test"):
        if s + "_txt" in g:
            with open(os.path.join(d, s + ".txt"), "w") as f:
                f.write(str(g[s + "_txt"]))


def oracle_eval_loss(sd, ids, is_rnn, nhead, pos, bsz=20, seq_len=7):
    """train.py:441-458 with the oracle forward."""
    nb = ids.size(0) // bsz
    src = ids.narrow(0, 0, nb * bsz).view(bsz, -1).t().contiguous()
    total = 0.0
    H = sd["rnn.weight_hh_mean_1"].shape[1] if is_rnn else 0
    hid = (torch.zeros(2, bsz, H), torch.zeros(2, bsz, H)) if is_rnn else None
    for i in range(0, src.size(0) - 1, seq_len):
        n = min(seq_len, len(src) - 1 - i)
        data, tgt = src[i:i + n], src[i + 1:i + 1 + n].reshape(-1)
        if is_rnn:
            out, hid = O.bayes_rnn_lm(data, hid, sd, pos, None)
        else:
            out = O.transformer_lm(data, sd, nhead, None)
        total += len(data) * float(O.cross_entropy_mean(out, tgt))
    return total / (len(src) - 1)


@pytest.mark.parametrize("tag", ["tlm_ffn", "lstm_bayes3"])
def test_eval_loss_of_reference_checkpoint(tag, tmp_path):
    """A checkpoint written by the reference's own train.py: the oracle reproduces its valid/test loss."""
    from bayeslms_amd import data as D
    g, sd, _ = load_golden("train_ckpt_" + tag)
    _write_corpus(g, str(tmp_path))
    c = D.Corpus(str(tmp_path))
    is_rnn = tag.startswith("lstm")
    if not is_rnn:
        sd["pos_encoder.pe"] = O.positional_table(64, sd["encoder.weight"].shape[1])
    for split in ("valid", "test"):
        got = oracle_eval_loss(sd, getattr(c, split), is_rnn, 4, 3)
        assert abs(got - float(g[split + "_loss"])) < 1e-5 * float(g[split + "_loss"])
        assert abs(got - float(g["printed_" + split])) < 0.006  # what train.py printed (2 decimals)


@pytest.mark.parametrize("tag", ["tlm_ffn", "lstm_bayes3", "tlm_gauss3", "lstm_gauss33", "lstm_var11"])
def test_scorer_scores_of_reference(tag):
    """Per-hypothesis scores written by the reference scorer: oracle restatement of its loop
    (sum of token NLL, '<s>' wrapping, OOV -> <unk>, LSTM hidden carried from the first hypothesis) -- round 4: also its
    Gaussian (GP Transformer; GP-LSTM, built UNTIED by the scorer, :428-429) and Variational branches."""
    from bayeslms_amd import compute_sentence_scores as S
    g, sd, _ = load_golden("scorer_" + tag)
    vocab = {w: i for i, w in enumerate(g["words"])}
    want = [(ln.split()[0], float(ln.split()[1])) for ln in str(g["scores_txt"]).splitlines()]
    is_rnn = tag.startswith("lstm")
    H = sd["encoder.weight"].shape[1] if is_rnn else 0
    if "pos_encoder.pe" in sd:
        sd["pos_encoder.pe"] = O.positional_table(5000, sd["encoder.weight"].shape[1])
    rnn_fwd = {"lstm_bayes3": lambda xs, hid: O.bayes_rnn_lm(xs, hid, sd, 3, None),
               "lstm_gauss33": lambda xs, hid: O.gauss_rnn_lm(xs, hid, sd, "33"),
               "lstm_var11": lambda xs, hid: O.variational_rnn_lm(xs, hid, sd, "11")[:2]}.get(tag)
    hid = (torch.zeros(2, 1, H), torch.zeros(2, 1, H)) if is_rnn else None
    got = []
    import collections
    nbest = collections.OrderedDict()
    for line in str(g["nbest_txt"]).splitlines():
        parts = line.strip().split(' ', 1)
        key, hyp = (parts[0], parts[1]) if len(parts) == 2 else (line.strip(), ' ')
        nbest.setdefault(key.rsplit('-', 1)[0], []).append(hyp)
    for key, hyps in nbest.items():
        first = None
        for n, hyp in enumerate(hyps, 1):
            x, t = S.get_input_and_target(hyp, vocab)
            xs, ts = torch.tensor(x).view(-1, 1), torch.tensor(t)
            if is_rnn:
                out, h_new = rnn_fwd(xs, hid)
                first = h_new if first is None else first
            else:
                out = O.transformer_lm(xs, sd, 4, None)
            got.append(("%s-%d" % (key, n), float(O.sentence_score(out, ts))))
        if is_rnn:
            hid = first
    assert [k for k, _ in got] == [k for k, _ in want]
    for (_, a), (_, b) in zip(got, want):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b))


@pytest.mark.parametrize("tag", ["tlm_ffn_interp", "lstm_bayes3_interp", "tlm_gauss3_interp", "lstm_gauss33_interp"])
def test_scorer_interpolation_of_reference(tag):
    """Two-model scoring by the reference's compute_scores (:157-168): the LOGITS are mixed with
    alpha = 0.7 before the log-softmax; the second LSTM carries its own hidden state."""
    from bayeslms_amd import compute_sentence_scores as S
    import collections
    g, sd, _ = load_golden("scorer_" + tag)
    sd2 = g["sd2"]
    vocab = {w: i for i, w in enumerate(g["words"])}
    want = [(ln.split()[0], float(ln.split()[1])) for ln in str(g["scores_txt"]).splitlines()]
    is_rnn = tag.startswith("lstm")
    H = sd["encoder.weight"].shape[1] if is_rnn else 0
    hid = (torch.zeros(2, 1, H), torch.zeros(2, 1, H)) if is_rnn else None
    hid2 = (torch.zeros(2, 1, H), torch.zeros(2, 1, H)) if is_rnn else None
    for d_ in (sd, sd2):
        if "pos_encoder.pe" in d_:
            d_["pos_encoder.pe"] = O.positional_table(5000, d_["encoder.weight"].shape[1])
    first_model = (lambda xs, h: O.gauss_rnn_lm(xs, h, sd, "33")) if "gauss" in tag else (lambda xs, h: O.bayes_rnn_lm(xs, h, sd, 3, None))
    nbest = collections.OrderedDict()
    for line in str(g["nbest_txt"]).splitlines():
        parts = line.strip().split(' ', 1)
        key, hyp = (parts[0], parts[1]) if len(parts) == 2 else (line.strip(), ' ')
        nbest.setdefault(key.rsplit('-', 1)[0], []).append(hyp)
    got = []
    for key, hyps in nbest.items():
        first = first2 = None
        for n, hyp in enumerate(hyps, 1):
            x, t = S.get_input_and_target(hyp, vocab)
            xs, ts = torch.tensor(x).view(-1, 1), torch.tensor(t)
            if is_rnn:
                o1, h1 = first_model(xs, hid)
                o2, h2 = O.bayes_rnn_lm(xs, hid2, sd2, 0, None)
                first, first2 = (h1, h2) if first is None else (first, first2)
            else:
                o1, o2 = O.transformer_lm(xs, sd, 4, None), O.transformer_lm(xs, sd2, 4, None)
            got.append(("%s-%d" % (key, n), float(O.sentence_score(0.7 * o1 + 0.3 * o2, ts))))
        if is_rnn:
            hid, hid2 = first, first2
    assert [k for k, _ in got] == [k for k, _ in want]
    for (_, a), (_, b) in zip(got, want):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b))


GAUSS_RNN = ["33", "31", "13", "23", "43", "330", "6360", "3333", "53", "73", "00"]


GAUSS_RNN_GPNN2 = ["34", "14", "64", "74", "54", "340", "3464"]


def _gpnn2_eps(g, w, T):
    return {int(c): [g["eps_%d_%d_%d" % (w, int(c), t)] for t in range(T)] for c in g["cells"]}


@pytest.mark.parametrize("gp", GAUSS_RNN_GPNN2)
def test_gauss_rnn_gpnn2_matches_reference(gp):
    """Type digit 4: GPNN2 inside the GP-LSTM cells; fresh frequencies at every time step in train mode
    (the replayed draws), no KL (train.py:367)."""
    g, sd, grad = load_golden("gauss_rnn_" + gp)
    T, B = g["x1"].shape
    H, V = sd["encoder.weight"].shape[1], sd["encoder.weight"].shape[0]
    zeros = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    e1, hid = O.gauss_rnn_lm(g["x1"], zeros, sd, gp)
    e2, hid = O.gauss_rnn_lm(g["x2"], hid, sd, gp)
    torch.testing.assert_close(e1, g["logits_eval_0"], **TOL)
    torch.testing.assert_close(e2, g["logits_eval_1"], **TOL)
    torch.testing.assert_close(hid[1], g["c_eval"], **TOL)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    l1, hid = O.gauss_rnn_lm(g["x1"], zeros, leaf, gp, _gpnn2_eps(g, 0, T))
    l2, hid = O.gauss_rnn_lm(g["x2"], tuple(h.detach() for h in hid), leaf, gp, _gpnn2_eps(g, 1, T))
    torch.testing.assert_close(l1, g["logits_train_0"], **TOL)
    torch.testing.assert_close(l2, g["logits_train_1"], **TOL)
    O.cross_entropy_mean(l2.view(-1, V), g["tgt"]).backward()
    for k, v in grad.items():
        if k == "decoder.weight":
            continue
        torch.testing.assert_close(leaf[k].grad, v, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("gp", GAUSS_RNN)
def test_gauss_rnn_matches_reference(gp):
    g, sd, grad = load_golden("gauss_rnn_" + gp)
    B, H = g["x1"].shape[1], sd["encoder.weight"].shape[1]
    zeros = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    e1, hid = O.gauss_rnn_lm(g["x1"], zeros, sd, gp)
    e2, hid = O.gauss_rnn_lm(g["x2"], hid, sd, gp)
    torch.testing.assert_close(e1, g["logits_eval_0"], **TOL)
    torch.testing.assert_close(e2, g["logits_eval_1"], **TOL)
    torch.testing.assert_close(hid[1], g["c_eval"], **TOL)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    l1, hid = O.gauss_rnn_lm(g["x1"], zeros, leaf, gp)
    l2, hid = O.gauss_rnn_lm(g["x2"], tuple(h.detach() for h in hid), leaf, gp)
    torch.testing.assert_close(l2, g["logits_train_1"], **TOL)
    mle = O.cross_entropy_mean(l2, g["tgt"])
    kl = O.kl_gauss_rnn(leaf, gp)
    torch.testing.assert_close(kl, g["kl"], **TOL)
    (mle + kl * float(g["kl_scale"])).backward()
    for k, gv in grad.items():
        if k != "decoder.weight":
            torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)


GAUSS_RNN_SAMPLE = ["33", "31", "32", "13", "23", "43", "53", "63", "73", "330", "3333"]


def _gp_cell_eps(g, w):
    return {int(c): _gp_eps(g, "eps_%d_%d_" % (w, int(c))) for c in g["cells"]}


@pytest.mark.parametrize("gp", GAUSS_RNN_SAMPLE)
def test_gauss_rnn_sample_branch_matches_reference(gp):
    """GP-LSTM cells with GPNN.sample raised: one draw per cell and window shared by all its steps (model.py:1721-1723)."""
    g, sd, grad = load_golden("gauss_rnn_%s_sample" % gp)
    B, H = g["x1"].shape[1], sd["encoder.weight"].shape[1]
    zeros = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    e1, hid = O.gauss_rnn_lm(g["x1"], zeros, sd, gp)
    e2, hid = O.gauss_rnn_lm(g["x2"], hid, sd, gp)
    torch.testing.assert_close(e1, g["logits_eval_0"], **TOL)
    torch.testing.assert_close(e2, g["logits_eval_1"], **TOL)
    assert (g["logits_train_0"] - g["logits_eval_0"]).abs().max() > 1e-4  # the branch fired
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    l1, hid = O.gauss_rnn_lm(g["x1"], zeros, leaf, gp, _gp_cell_eps(g, 0))
    l2, hid = O.gauss_rnn_lm(g["x2"], tuple(h.detach() for h in hid), leaf, gp, _gp_cell_eps(g, 1))
    torch.testing.assert_close(l1, g["logits_train_0"], **TOL)
    torch.testing.assert_close(l2, g["logits_train_1"], **TOL)
    kl = O.kl_gauss_rnn(leaf, gp)
    torch.testing.assert_close(kl, g["kl"], **TOL)
    (O.cross_entropy_mean(l2, g["tgt"]) + kl * float(g["kl_scale"])).backward()
    for k, gv in grad.items():
        if k != "decoder.weight":
            torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)


@pytest.mark.parametrize("vp", ["00", "01", "10", "11"])
def test_variational_rnn_matches_reference(vp):
    g, sd, grad = load_golden("variational_rnn_" + vp)
    B, H = g["x1"].shape[1], sd["encoder.weight"].shape[1]
    zeros = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    e1, hid, _ = O.variational_rnn_lm(g["x1"], zeros, sd, vp)
    torch.testing.assert_close(e1, g["logits_eval_0"], **TOL)
    torch.testing.assert_close(hid[0], g["h_eval"], **TOL)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    eps = {c: g["eps_%d" % c] for c in (0, 1) if "eps_%d" % c in g}
    l1, hid, kl = O.variational_rnn_lm(g["x1"], zeros, leaf, vp, eps)
    torch.testing.assert_close(l1, g["logits_train_0"], **TOL)
    torch.testing.assert_close(kl, g["kl"], **TOL)
    (O.cross_entropy_mean(l1, g["tgt"]) + kl * float(g["kl_scale"])).backward()
    for k, gv in grad.items():
        if k != "decoder.weight":
            torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)


@pytest.mark.parametrize("v_pos", [0, 1, 2, 3])
def test_vtransformer_matches_reference(v_pos):
    g, sd, _ = load_golden("vtransformer_%d" % v_pos)
    torch.testing.assert_close(O.transformer_lm(g["src"], sd, int(g["nhead"]), None), g["logits_eval"], **TOL)
    nl = len({k.split(".")[1] for k in sd if k.startswith("transformerlayers.")})
    assert nl == {0: 4, 1: 4, 2: 3, 3: 3}[v_pos]  # the reference's layer-count arithmetic (model.py:2822-2843)


def test_vtransformer_11_builds_zero_layers_like_the_reference():
    """`--T_v_pos 11` (the literal flag of BASELINE configs[4]): model.py:2822-2843 builds no encoder layer at all."""
    g, sd, grad = load_golden("vtransformer_11")
    assert not any(k.startswith("transformerlayers.") for k in sd)
    torch.testing.assert_close(O.transformer_lm(g["src"], sd, int(g["nhead"]), None), g["logits_eval"], **TOL)
    leaf = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "pos_encoder.pe") for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    lt = O.transformer_lm(g["src"], leaf, int(g["nhead"]), None)
    torch.testing.assert_close(lt, g["logits_train"], **TOL)
    mle = O.cross_entropy_mean(lt, g["tgt"])
    torch.testing.assert_close(mle, g["mle"], **TOL)
    mle.backward()
    for k, gv in grad.items():
        if k != "decoder.weight":
            torch.testing.assert_close(leaf[k].grad, gv, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: k + ": " + m)
    # the engine's class and the CLI dispatch build the same thing: same state-dict keys and shapes, no layers
    from bayeslms_amd import model as M, train as T
    m = M.VTransformerModel(50, 16, 4, 32, 4, 0.0, True, 11)
    assert len(m.transformerlayers) == 0
    shapes = lambda d: {k: tuple(v.shape) for k, v in d.items() if k != "pos_encoder.pe"}  # noqa: E731 (fixtures keep 64 rows of the table)
    assert shapes(m.state_dict()) == shapes(sd) and "pos_encoder.pe" in m.state_dict()
    a = T.build_parser().parse_args("--model Transformer --uncertainty Variational --T_v_pos 11 --emsize 16 --nhid 32 --nlayers 4 "
                                    "--nhead 4 --tied".split())
    assert len(T.build_model(a, 50).transformerlayers) == 0 and T.kl_selector(a) is None


def _nbest_of(g):
    import collections
    nbest = collections.OrderedDict()
    for line in str(g["nbest_txt"]).splitlines():
        parts = line.strip().split(' ', 1)
        key, hyp = (parts[0], parts[1]) if len(parts) == 2 else (line.strip(), ' ')
        nbest.setdefault(key.rsplit('-', 1)[0], []).append(hyp)
    return nbest


def test_mc_sentence_score_is_log_mean_exp():
    import math
    assert abs(O.mc_sentence_score([3.25]) - 3.25) < 1e-12                    # S = 1: the sample's NLL
    assert abs(O.mc_sentence_score([2.0, 2.0, 2.0]) - 2.0) < 1e-12            # identical samples
    want = -math.log((math.exp(-1.0) + math.exp(-2.0) + math.exp(-4.0)) / 3)  # probabilities averaged, not NLLs
    assert abs(O.mc_sentence_score([1.0, 2.0, 4.0]) - want) < 1e-12
    assert O.mc_sentence_score([1.0, 2.0, 4.0]) < (1.0 + 2.0 + 4.0) / 3       # Jensen


@pytest.mark.parametrize("tag", ["tlm_ffn", "lstm_bayes3"])
def test_mc_scoring_oracle_degenerates_to_reference_scores(tag):
    """sigma -> 0 (log-sigma = -50): every weight sample IS the mean weight, so the Monte-Carlo oracle with any S
    must reproduce the score file the reference scorer wrote; and with the fixture's real sigmas the S-sample score
    differs from the mean-weight one but stays a proper log-mean-exp of its per-sample scores."""
    from bayeslms_amd import compute_sentence_scores as S
    g, sd, _ = load_golden("scorer_" + tag)
    vocab = {w: i for i, w in enumerate(g["words"])}
    want = [(ln.split()[0], float(ln.split()[1])) for ln in str(g["scores_txt"]).splitlines()]
    fam = "lstm_bayes" if tag.startswith("lstm") else "tlm_ffn"
    ids = list(range(8)) if fam == "lstm_bayes" else [0]
    cold = {k: (torch.full_like(v, -50.0) if "lgstd" in k else v) for k, v in sd.items()}
    got = O.mc_scores(_nbest_of(g), vocab, cold, fam, 3, 1111, ids, get_input_and_target=S.get_input_and_target)
    assert [k for k, _ in got] == [k for k, _ in want]
    for (_, a), (_, b) in zip(got, want):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b))
    one = O.mc_scores(_nbest_of(g), vocab, sd, fam, 1, 1111, ids, get_input_and_target=S.get_input_and_target)
    four = O.mc_scores(_nbest_of(g), vocab, sd, fam, 4, 1111, ids, get_input_and_target=S.get_input_and_target)
    assert any(abs(a - b) > 1e-6 for (_, a), (_, b) in zip(one, want))  # the noise does reach the scores
    assert all(v == v and v >= 0 for _, v in four)
