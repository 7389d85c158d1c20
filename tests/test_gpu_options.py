"""GPU: every kernel-selection switch that survives (include/bayeslm.h blm_set_option, the scorer / evaluate switches) runs its
NON-default form against the default one and against the CPU oracle -- an untested non-default kernel is product surface, not
documentation.  INTEGRATION.md lists the switches with these tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture
def option():
    """set(name, value) for the duration of a test; every option goes back to what it was."""
    from bayeslms_amd import ops
    saved = {}

    def setter(name, value):
        saved.setdefault(name, ops.get_option(name))
        ops.set_option(name, value)
    yield setter
    for k, v in saved.items():
        ops.set_option(k, v)


def _attn(dev, T, B, nhead, hd, p):
    from bayeslms_amd import ops
    from oracle import bayes_oracle as O
    d = nhead * hd
    g = torch.Generator().manual_seed(T * 3 + B)
    qkv = torch.randn(T, B, 3 * d, generator=g)
    go = torch.randn(T, B, d, generator=g)
    drop = ops.Drop(p, 77, 5, 2, 0, B) if p > 0 else ops.NO_DROP
    x = qkv.to(dev).requires_grad_(True)
    out = ops.attention(x, nhead, drop)
    out.backward(go.to(dev))
    ref = None
    if p == 0:
        qr = qkv.clone().requires_grad_(True)
        q, k, v = qr.chunk(3, dim=-1)
        r = O.attention_core(q, k, v, nhead, O.causal_mask(T))
        r.backward(go)
        ref = (r.detach(), qr.grad)
    return out.detach(), x.grad, ref


@pytest.mark.parametrize("T,B,nhead,p", [(128, 8, 8, 0.0), (128, 8, 8, 0.2), (97, 2, 3, 0.0), (64, 64, 8, 0.2)])
def test_attention_heads_per_workgroup_forms_agree(dev, option, T, B, nhead, p):
    """"attn_hpw" 1 / 2: one head per 4-wave workgroup against two heads per 8-wave workgroup (the default picks by head
    count), forward and the one-launch backward, with the oracle at dropout 0 and the same Philox masks otherwise."""
    res = {}
    for v in (1, 2):
        option("attn_hpw", v)
        res[v] = _attn(dev, T, B, nhead, 64, p)
    assert rel(res[1][0], res[2][0]) < 2e-6 and rel(res[1][1], res[2][1]) < 5e-6
    if p == 0:
        for v in (1, 2):
            assert rel(res[v][0], res[v][2][0]) < 1e-5 and rel(res[v][1], res[v][2][1]) < 2e-5


@pytest.mark.parametrize("T,B,nhead,p", [(32, 16, 8, 0.0), (20, 40, 8, 0.0), (7, 3, 2, 0.0), (30, 64, 8, 0.3)])
def test_attention_short_forward_equals_the_general_one(dev, option, T, B, nhead, p):
    """"attn_short" 0: the 128-row forward also for T <= 32, where the default runs one wave per head (inference: no grad)."""
    from bayeslms_amd import ops
    from oracle import bayes_oracle as O
    d = nhead * 64
    qkv = torch.randn(T, B, 3 * d, generator=torch.Generator().manual_seed(T)).to(dev)
    drop = ops.Drop(p, 5, 1, 0, 0, B) if p > 0 else ops.NO_DROP
    outs = {}
    with torch.no_grad():
        for v in (1, 0):
            option("attn_short", v)
            outs[v] = ops.attention(qkv, nhead, drop)
    assert rel(outs[1], outs[0]) < 2e-6
    if p == 0:
        q, k, v = qkv.cpu().chunk(3, dim=-1)
        assert rel(outs[0], O.attention_core(q, k, v, nhead, O.causal_mask(T))) < 1e-5


@pytest.mark.parametrize("T,B,nhead", [(128, 2, 4), (50, 3, 2), (160, 1, 2)])
def test_attention_vector_alu_kernels_at_head_dim_64(dev, option, T, B, nhead):
    """"attn_valu" 1: the untiled vector-ALU kernels (the path of every other head size) at head_dim 64 against the MFMA kernels
    and the oracle."""
    res = {}
    for v in (0, 1):
        option("attn_valu", v)
        res[v] = _attn(dev, T, B, nhead, 64, 0.0)
    assert rel(res[1][0], res[0][0]) < 1e-5 and rel(res[1][1], res[0][1]) < 2e-5
    assert rel(res[1][0], res[1][2][0]) < 1e-5 and rel(res[1][1], res[1][2][1]) < 2e-5


def _lstm(dev, T, B, H, grad):
    from bayeslms_amd import ops
    g = torch.Generator().manual_seed(H + B)
    s = 1.0 / H ** 0.5
    x = torch.randn(T, B, H, generator=g)
    w_ih, w_hh = (torch.rand(4 * H, H, generator=g) * 2 - 1) * s, (torch.rand(4 * H, H, generator=g) * 2 - 1) * s
    b_ih, b_hh = (torch.rand(4 * H, generator=g) * 2 - 1) * s, (torch.rand(4 * H, generator=g) * 2 - 1) * s
    h0, c0 = torch.randn(B, H, generator=g) * 0.1, torch.randn(B, H, generator=g) * 0.1
    t = [v.to(dev) for v in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    if not grad:
        with torch.no_grad():
            return ops.lstm_layer(*t), (x, h0, c0, w_ih, w_hh, b_ih, b_hh)
    t = [v.requires_grad_(True) for v in t]
    y, hT, cT = ops.lstm_layer(*t)
    (y.sum() * 0.3 + hT.sum() + cT.sum() * 0.5).backward()
    return (y.detach(), hT.detach(), cT.detach(), t[0].grad, t[3].grad, t[4].grad), (x, h0, c0, w_ih, w_hh, b_ih, b_hh)


@pytest.mark.parametrize("B,H", [(1, 1024), (2, 256), (4, 64)])
def test_lstm_tiny_batch_kernel_equals_the_matrix_core_step(dev, option, B, H):
    """"lstm_gemv" 0: B <= 4 on the matrix-core step kernel instead of the one-wave-per-unit form (the scorer's carry chain)."""
    from oracle import bayes_oracle as O
    res = {}
    for v in (1, 0):
        option("lstm_gemv", v)
        res[v], cpu = _lstm(dev, 9, B, H, False)
    for a, b in zip(res[1], res[0]):
        assert rel(a, b) < 5e-6
    y, h, c = O.lstm_layer(*cpu)
    assert rel(res[0][0], y) < 1e-5 and rel(res[0][2], c) < 1e-5


@pytest.mark.parametrize("opt,val", [("lstm_pipe", 0), ("lstm_tail", 1)])
@pytest.mark.parametrize("B,H", [(64, 1024), (20, 512)])
def test_lstm_step_kernel_forms_agree(dev, option, opt, val, B, H):
    """"lstm_pipe" 0 (the K loop without software pipelining) and "lstm_tail" 1 (the general K-tail form of the pipelined loop)
    against the default form: forward, carried state and the input / weight gradients of a whole layer."""
    base, _ = _lstm(dev, 6, B, H, True)
    option(opt, val)
    other, _ = _lstm(dev, 6, B, H, True)
    for a, b in zip(other, base):
        assert rel(a, b) < 5e-6


@pytest.mark.parametrize("B,H", [(64, 512), (50, 512), (96, 512), (64, 1024), (40, 1024)])
def test_search_step_two_batch_tiles_per_workgroup_equals_one(dev, option, B, H):
    """"lstm_mb2" (default 1): at B > 32 the search cell's forward step multiplies its 32 weight rows against TWO batch tiles per
    workgroup (the stacked 8H x H weight streams once, half the workgroups); 0 = one tile per workgroup.  Same K order per wave and
    the same fixed cross-wave reduction: h, c and the eight activations are bit-identical, incl. a ragged last tile (B = 50, 40) and a
    second row of workgroups (B = 96).  H = 512 takes the plain K loop of the two-tile form, H = 1024 its software-pipelined one
    (H = 256 has one chunk per lane and keeps the one-tile kernel)."""
    from bayeslms_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(B + H)
    xw = torch.randn(B, 8 * H, generator=g).to(dev)
    w8 = (0.05 * torch.randn(8 * H, H, generator=g)).to(dev)
    h0, c0 = torch.randn(B, H, generator=g).to(dev), torch.randn(B, H, generator=g).to(dev)
    probs = torch.softmax(torch.randn(4, 2, generator=g), -1).contiguous().to(dev)
    outs = []
    for mb2 in (1, 0):
        option("lstm_mb2", mb2)
        h, c, a = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, 8 * H, device=dev)
        check(lib().blm_lstm_search_step_fwd(ptr(xw), ptr(w8), ptr(h0), ptr(c0), ptr(probs), ptr(h), ptr(c), ptr(a), B, H, stream()))
        outs.append((h.cpu(), c.cpu(), a.cpu()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    # and against fp64: z = xw + h0 W8^T, eight activations, four mixes, cell
    z = xw.double().cpu() + h0.double().cpu() @ w8.double().cpu().t()
    act = torch.cat([torch.tanh(z[:, k * H:(k + 1) * H]) if k % 4 == 2 else torch.sigmoid(z[:, k * H:(k + 1) * H]) for k in range(8)], 1)
    pr = probs.double().cpu()
    gate = [act[:, k * H:(k + 1) * H] * pr[k, 0] + act[:, (k + 4) * H:(k + 5) * H] * pr[k, 1] for k in range(4)]
    cn = gate[1] * c0.double().cpu() + gate[0] * gate[2]
    hn = gate[3] * torch.tanh(cn)
    assert rel(outs[0][0].double(), hn) < 2e-5 and rel(outs[0][1].double(), cn) < 2e-5 and rel(outs[0][2].double(), act) < 2e-5


def test_options_are_validated(dev):
    from bayeslms_amd import ops
    from bayeslms_amd._lib import BayesLMError
    with pytest.raises(BayesLMError, match="unknown option"):
        ops.set_option("no_such_switch", 1)
    with pytest.raises(BayesLMError, match="takes 0..2"):
        ops.set_option("attn_hpw", 3)
    assert ops.get_option("lstm_pipe") == 1 and ops.get_option("attn_valu") == 0


def test_scorer_and_evaluate_fused_nll_switches(dev, monkeypatch):
    """BLM_SCORER_FUSED_NLL=0 / BLM_EVAL_FUSED_NLL=0: decoder logits materialised + cross-entropy kernel (two models: two logit
    matrices + blm_ce_interp_fwd) against the default (blm_linear_nll / blm_linear_nll2: no logits stored)."""
    import collections
    from bayeslms_amd import compute_sentence_scores as S, engine, model as M
    torch.manual_seed(4)
    V = 64
    vocab = {"w%d" % i: i for i in range(V)}
    vocab["<s>"] = 0
    g = torch.Generator().manual_seed(2)
    nbest = collections.OrderedDict()
    for u in range(6):
        nbest["utt%d" % u] = [" ".join("w%d" % int(t) for t in torch.randint(1, V, (int(torch.randint(1, 9, (1,), generator=g)),), generator=g))
                              for _ in range(4)]
    m1 = M.BayesTransformerModel(V, 32, 4, 64, 2, 0.1, True, "FFN").to(dev)
    m2 = M.TransformerModel(V, 16, 4, 32, 2, 0.1, "gelu", True).to(dev)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(S, "_FUSED_NLL", fused)
        one = S.compute_scores_batched(nbest, m1, vocab, "Transformer", dev)
        two = S.compute_scores_batched(nbest, m1, vocab, "Transformer", dev, m2, 0.7)
        res[fused] = ([v for hv in one.values() for _, v in hv], [v for hv in two.values() for _, v in hv])
    for a, b in zip(res[True][0] + res[True][1], res[False][0] + res[False][1]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b))
    assert any(abs(a - b) > 1e-3 for a, b in zip(res[True][0], res[True][1]))  # the second model matters
    data = torch.randint(0, V, (90, 5), generator=g).to(dev)
    monkeypatch.setenv("BLM_EVAL_FUSED_NLL", "1")
    a = engine.evaluate(m1, data, 16)
    monkeypatch.setenv("BLM_EVAL_FUSED_NLL", "0")
    b = engine.evaluate(m1, data, 16)
    assert abs(a - b) <= 1e-5 * abs(b)


def test_evaluate_batches_the_windows_of_a_stateless_model(dev, monkeypatch):
    """evaluate() (train.py:441-458) walks the stream window by window; a Transformer carries nothing from one window to the next, so
    G full windows run as ONE batch of G x columns columns (BLM_EVAL_WINDOWS=G; default: as many as fill ~16384 rows; 1: one by one, the
    reference's walk).  Same loss to rounding for every G, fused and unfused decoder, with a ragged last window and with fewer
    windows than G.  A recurrent model carries its state from window to window, so G of its windows are one window of G x seq_len steps:
    the same recurrence with G times the rows in the input / decoder products (all four LSTM families)."""
    from bayeslms_amd import engine, model as M
    torch.manual_seed(6)
    V = 64
    g = torch.Generator().manual_seed(8)
    m = M.BayesTransformerModel(V, 32, 4, 64, 2, 0.1, True, "FFN").to(dev)
    calls = []
    real = m.forward
    m.forward = lambda x, *a, **k: (calls.append(tuple(x.shape)), real(x, *a, **k))[1]
    for rows in (90, 97, 17, 12):  # 5 full windows + ragged, 6 full exactly (96 + 1), one full + ragged, no full window
        data = torch.randint(0, V, (rows, 5), generator=g).to(dev)
        for fused in ("1", "0"):
            monkeypatch.setenv("BLM_EVAL_FUSED_NLL", fused)
            monkeypatch.setenv("BLM_EVAL_WINDOWS", "1")
            del calls[:]
            ref = engine.evaluate(m, data, 16)
            assert all(c[1] == 5 for c in calls) and len(calls) == -(-(rows - 1) // 16)
            for G in ("0", "2", "3", "64"):
                monkeypatch.setenv("BLM_EVAL_WINDOWS", G)
                del calls[:]
                got = engine.evaluate(m, data, 16)
                assert abs(got - ref) <= 2e-6 * abs(ref), (rows, fused, G, got, ref)
                if G == "0" and rows >= 33:
                    assert calls[0][1] == 5 * ((rows - 1) // 16)      # every full window in one batch at this size
                if G == "3" and rows == 97:
                    assert [c[1] for c in calls] == [15, 15]
                if G == "64" and rows == 90:
                    assert [c[1] for c in calls] == [25, 5]           # five full windows at once, the ragged one alone
                if G == "2" and rows == 90:
                    assert [c[1] for c in calls] == [10, 10, 5, 5]    # 2 + 2 + the fifth full window + the ragged one
    # a recurrent model carries its state across windows: G consecutive windows are ONE window of G x seq_len steps (never a wider batch)
    for build in (lambda: M.RNNModel("LSTM", V, 32, 32, 2, 0.1, True), lambda: M.BayesRNNModel("LSTM", V, 32, 32, 2, 0.1, True, 3),
                  lambda: M.GaussRNNModel("LSTM", V, 32, 32, 2, 0.1, True, "33"), lambda: M.VariationalRNNModel("LSTM", V, 32, 32, 2, 0.1, True, "11")):
        lstm = build().to(dev)
        seen = []
        real2 = lstm.forward
        lstm.forward = lambda x, h, real2=real2, seen=seen: (seen.append(tuple(x.shape)), real2(x, h))[1]
        stream = torch.randint(0, V, (90, 5), generator=g).to(dev)
        for fused in ("1", "0"):
            monkeypatch.setenv("BLM_EVAL_FUSED_NLL", fused)
            monkeypatch.setenv("BLM_EVAL_WINDOWS", "1")
            del seen[:]
            ref = engine.evaluate(lstm, stream, 16)
            assert [sh for sh in seen] == [(16, 5)] * 5 + [(9, 5)]
            for G, shapes in (("2", [(32, 5), (32, 5), (25, 5)]), ("0", [(80, 5), (9, 5)]), ("4", [(64, 5), (25, 5)])):
                monkeypatch.setenv("BLM_EVAL_WINDOWS", G)
                del seen[:]
                got = engine.evaluate(lstm, stream, 16)
                assert seen == shapes and abs(got - ref) <= 3e-6 * abs(ref), (type(lstm).__name__, fused, G, seen, got, ref)
