"""GPU: the one tensor of the path that outgrows 32-bit indexing -- the logits.

At configs[3]'s GLOBAL batch on ONE GPU (512 columns x 128 tokens x 33,000 words = 2.16 G elements, 8.7 GB) the logits
pass 2^31 elements and every operand that touches them passes 4 GB: the decoder's forward product writes them, its two
backward products read them (as the A operand of an NN product and, transposed, of a TN product: the GEMM leaves its
32-bit-offset loaders for the general ones, gemm_api.hip), the cross-entropy kernels walk them row by row.  The
reference gets all of this from torch (model.py:1305, train.py:330-333 at whatever batch the user passes); here each
piece is checked against torch's own kernels on the same device at M x V = 70,016 x 33,000 = 2.31 G elements:

* ``ops.linear`` forward (NT), dX (NN, A > 4 GB), dW (TN, A > 4 GB) and the bias gradient (column sums over 70 k rows),
* ``ops.cross_entropy`` (loss, and the in-place gradient) incl. ``ignore_index`` rows beyond the 2^31st element,
* ``ops.linear_nll`` (the scorer's fused decoder + NLL, which never stores the logits) on the same rows,
* one training step of the model at 512 columns against the same step taken as two half-batches.

Tolerances are the suite's (1e-4 relative on losses / logits, 5e-4 on gradients; BASELINE's bar is 1e-3): K = 512 sums of
fp32 in another order than the vendor GEMM's.  ~45 GB of device memory, a few seconds (4 tests, 2.6 s on the box)."""
import math

import pytest
import torch

from bayeslms_amd import ops
from bayeslms_amd import model as M

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROWS, V, D = 70016, 33000, 512  # 2,310,528,000 logits: row 65,076 starts beyond element 2^31


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


@pytest.fixture(scope="module")
def big():
    assert ROWS * V > 2 ** 31 and ROWS * V * 4 > 2 ** 33
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(ROWS, D, device=DEV, generator=g) * 0.5
    w = torch.randn(V, D, device=DEV, generator=g) * 0.05
    b = torch.randn(V, device=DEV, generator=g) * 0.1
    t = torch.randint(0, V, (ROWS,), device=DEV, generator=g)
    t[::97] = -100  # rows without a target on both sides of the 2^31st element
    t[-1] = V - 1   # the very last logit of the tensor is a target
    yield x, w, b, t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("path", ["level1", "trainer"])
def test_decoder_products_and_cross_entropy_beyond_2_31_elements(big, path):
    """level1: the model's ``Logits`` under the loop's own ``F.cross_entropy`` (logits kept, torch's mean
    over the rows whose target is not -100).  trainer: ``ops.cross_entropy(unit_grad=True)``, the engine trainers' form --
    the gradient is written over the logits in the forward pass; its mean is over ALL rows (the trainers' targets are
    corpus words, train.py:299-304: there is no ignore_index on that path), so it runs on the clamped targets."""
    x, w, b, t = big
    if path == "trainer":
        t = t.clamp(min=0)
    xr, wr, br = (v.detach().clone().requires_grad_(True) for v in (x, w, b))
    xo, wo, bo = (v.detach().clone().requires_grad_(True) for v in (x, w, b))
    # torch's own kernels on the same device
    ref_logits = torch.nn.functional.linear(xr, wr, br)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, t)  # ignore_index = -100, mean over the others
    ref_loss.backward()
    ref_tail = ref_logits.detach()[-3:].clone()
    ref_mid = ref_logits.detach()[65070:65080].clone()  # the rows around element 2^31
    del ref_logits
    torch.cuda.empty_cache()
    # the engine: NT forward, one-pass cross entropy, NN + TN backward
    logits = ops.linear(xo, wo, bo)
    assert logits.numel() > 2 ** 31
    assert rel(logits.detach()[-3:], ref_tail) < 1e-4 and rel(logits.detach()[65070:65080], ref_mid) < 1e-4
    if path == "level1":
        loss = torch.nn.functional.cross_entropy(ops.as_logits(logits), t)
    else:
        loss, _ = ops.cross_entropy(logits, t, unit_grad=True)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-4 * abs(float(ref_loss.detach())), (float(loss.detach()), float(ref_loss.detach()))
    loss.backward()
    assert rel(xo.grad, xr.grad) < 5e-4, "dX = dlogits @ W (NN product, A operand of 9.2 GB)"
    assert rel(xo.grad[-64:], xr.grad[-64:]) < 5e-4
    if path == "level1":
        assert float(xo.grad[::97].abs().max()) == 0.0  # rows without a target: exact zeros, on both sides of element 2^31
    assert rel(wo.grad, wr.grad) < 5e-4, "dW = dlogits^T @ X (TN product over 70 k rows)"
    assert rel(bo.grad, br.grad) < 5e-4, "bias gradient: column sums of 70 k rows"


def test_fused_decoder_nll_on_the_same_rows(big):
    x, w, b, t = big
    tt = t.clamp(min=0)
    with torch.no_grad():
        got = ops.linear_nll(x, w, b, tt)
        want = torch.empty(ROWS, device=DEV)
        for lo in range(0, ROWS, 8192):  # torch in slabs: the whole log_softmax would be a second 9.2 GB tensor
            lg = torch.nn.functional.linear(x[lo:lo + 8192], w, b)
            want[lo:lo + 8192] = torch.nn.functional.cross_entropy(lg, tt[lo:lo + 8192], reduction="none")
    assert got.shape == (ROWS,)
    assert rel(got, want) < 1e-4


def test_training_step_at_512_columns_equals_two_half_batches():
    """configs[3]'s global batch on one GPU: logits (128, 512, 33000) = 2.16 G elements.  The same step as TWO halves
    of 256 columns (dropout and the noise are keyed by GLOBAL column and step, so the halves see the masks the whole
    batch sees; gradients are means over the global batch) must give the same loss and the same gradients."""
    T, B, Vv = 128, 512, 33000
    assert T * B * Vv > 2 ** 31
    torch.manual_seed(11)
    model = M.BayesTransformerModel(Vv, 256, 4, 512, 2, 0.2, True, "FFN").to(DEV)  # a narrow model: the extent is T * B * V
    g = torch.Generator(device=DEV).manual_seed(3)
    data = torch.randint(0, Vv, (T, B), device=DEV, generator=g)
    tgt = torch.randint(0, Vv, (T * B,), device=DEV, generator=g)

    def grads(cols_lo, cols_hi):
        model.zero_grad(set_to_none=True)
        model.train()
        model.set_seed(1111)
        model.set_step(7)
        model.set_columns(cols_lo, B)
        n = cols_hi - cols_lo
        out = model(data[:, cols_lo:cols_hi].contiguous())
        tg = tgt.view(T, B)[:, cols_lo:cols_hi].contiguous().view(-1)
        loss, _ = ops.cross_entropy(out.view(-1, Vv), tg)
        (loss * (n / B)).backward()
        gs = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        return float(loss.detach()) * n / B, gs
    whole, gw = grads(0, B)
    torch.cuda.empty_cache()
    l1, g1 = grads(0, B // 2)
    l2, g2 = grads(B // 2, B)
    assert math.isfinite(whole) and abs(whole - (l1 + l2)) < 1e-4 * abs(whole), (whole, l1 + l2)
    assert set(gw) == set(g1) == set(g2)
    for k in gw:
        assert rel(g1[k] + g2[k], gw[k]) < 5e-4, k


def test_longest_sequence_the_positional_table_allows():
    """T = 5000 = PositionalEncoding's max_len (model.py:93-105): the longest input the reference's Transformers accept
    (one more row and its ``x + self.pe[:x.size(0)]`` fails to broadcast, model.py:116).  Eval NLL and one training
    loss + gradients (dropout 0, eps from the Philox stream) against the CPU oracle; T = 5001 raises here too."""
    from oracle import bayes_oracle as O, philox as P
    V, d, h, ff, nl, T, B = 60, 128, 2, 64, 2, 5000, 2
    torch.manual_seed(1111)
    m = M.BayesTransformerModel(V, d, h, ff, nl, 0.0, True, "FFN")
    for mod in m.modules():  # layer 0 is built with a hard-coded dropout of 0.2 (model.py:1202)
        if hasattr(mod, "p"):
            mod.p = 0.0
        if hasattr(mod, "dropout") and isinstance(mod.dropout, float):
            mod.dropout = 0.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    gen = torch.Generator().manual_seed(5)
    src = torch.randint(0, V, (T, B), generator=gen)
    tgt = torch.randint(0, V, (T * B,), generator=gen)
    torch.set_num_threads(16)
    m.eval()
    with torch.no_grad():
        _, nll = ops.cross_entropy(m(src.to(DEV)).view(-1, V), tgt.to(DEV))
        ref_nll = O.token_nll(O.transformer_lm(src, sd, h, None), tgt)
    assert rel(nll.cpu(), ref_nll) < 1e-4
    m.train()
    m.set_seed(1111)
    m.set_step(3)
    lin2 = m.transformerlayers[0].linear2
    loss, _ = ops.cross_entropy(m(src.to(DEV)).view(-1, V), tgt.to(DEV))
    loss.backward()
    eps = torch.from_numpy(P.normal(d * ff, 1111, P.STREAM_WEIGHT + lin2._site_base, 3)).view(d, ff)
    leaf = {k: v.clone().requires_grad_(k.endswith(("weight_mean", "weight_lgstd", "qkv_net.weight"))) for k, v in sd.items()}
    leaf["decoder.weight"] = leaf["encoder.weight"]
    rl = O.cross_entropy_mean(O.transformer_lm(src, leaf, h, eps), tgt)
    rl.backward()
    assert abs(float(loss.detach()) - float(rl.detach())) < 1e-4 * abs(float(rl.detach()))
    cur = dict(m.named_parameters())
    for k in ("transformerlayers.0.linear2.weight_mean", "transformerlayers.0.linear2.weight_lgstd",
              "transformerlayers.1.self_attn.qkv_net.weight"):
        assert rel(cur[k].grad.cpu(), leaf[k].grad) < 1e-3, k  # attention over 5000 positions: BASELINE's bar
    with pytest.raises(Exception):
        m(torch.randint(0, V, (T + 1, 1), device=DEV))


def test_empty_inputs_end_in_empties_or_in_the_engines_own_error():
    """The other end of the range.  The reference's flows never produce an empty batch (train.py:299-304 always cuts
    at least one row, the scorer wraps every hypothesis in <s> ... </s>): products over zero rows return empties as
    torch's do, everything else refuses with BayesLMError -- no launch with an empty grid, no arithmetic error."""
    from bayeslms_amd._lib import BayesLMError
    w, b = torch.randn(32, 16, device=DEV), torch.randn(32, device=DEV)
    none = torch.empty(0, dtype=torch.long, device=DEV)
    assert ops.linear(torch.empty(0, 16, device=DEV), w, b).shape == (0, 32)
    assert ops.linear(torch.empty(0, 3, 16, device=DEV), w, b).shape == (0, 3, 32)
    with torch.no_grad():
        assert ops.linear_nll(torch.empty(0, 16, device=DEV), w, b, none).shape == (0,)
    refused = {
        "embed": lambda: ops.embed(torch.empty(0, 2, dtype=torch.long, device=DEV), torch.randn(10, 16, device=DEV)),
        "cross_entropy": lambda: ops.cross_entropy(torch.empty(0, 32, device=DEV), none),
        "attention": lambda: ops.attention(torch.empty(0, 2, 3 * 128, device=DEV), 2),
        "dropout": lambda: ops.dropout(torch.empty(0, 16, device=DEV), ops.Drop(0.5, 1, 0, 0, 0, 1)),
    }
    tlm = M.BayesTransformerModel(50, 32, 2, 64, 2, 0.2, True, "FFN").to(DEV).eval()
    rnn = M.RNNModel("LSTM", 50, 32, 32, 2, 0.2, True).to(DEV).eval()
    for shape in ((0, 2), (3, 0)):
        ids = torch.empty(*shape, dtype=torch.long, device=DEV)
        refused["tlm %r" % (shape,)] = lambda ids=ids: tlm(ids)
        refused["lstm %r" % (shape,)] = lambda ids=ids: rnn(ids, rnn.init_hidden(ids.shape[1]))
    with torch.no_grad():
        for name, fn in refused.items():
            with pytest.raises(BayesLMError):
                fn()
    torch.cuda.synchronize()  # nothing was left behind on the stream
