"""GPU parity of the individual C-ABI operators against fp32 CPU math (torch CPU / the oracle).

Floating point: tolerance 1e-3 relative is BASELINE.json's bar; the GEMMs are checked far tighter
(the f32 MFMA is an exact fp32 fma chain, only the summation order differs from the CPU BLAS)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bayes_oracle as O  # noqa: E402
from oracle import philox as P  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def ops_mod():
    from bayeslms_amd import ops
    return ops


def L():
    from bayeslms_amd import _lib
    return _lib


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ------------------------------------------------------------------ library / device
def test_library_is_native_and_gfx950(dev):
    lib = L()
    lib.require_gfx950()
    assert lib.lib().blm_abi_version() == lib.ABI_VERSION


def test_errors_are_reported_not_swallowed(dev):
    import ctypes as C
    lib = L()
    a = lib.GemmArgs()
    a.abi_version = 99
    assert lib.lib().blm_gemm(C.byref(a), None) == -2
    assert b"abi_version" in lib.lib().blm_last_error()
    with pytest.raises(lib.BayesLMError):
        ops_mod().linear(torch.zeros(2, 3), torch.zeros(4, 3))  # CPU tensors: no fallback


# ------------------------------------------------------------------ GEMM family
GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (64, 64, 32), (65, 130, 33), (128, 128, 64), (200, 50, 16), (37, 300, 129),
               (256, 512, 96), (10, 33000 // 50, 512), (8, 4096, 1024)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_nn_tn(dev, M, N, K):
    ops = ops_mod()
    lib = L()
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    Ad, Bd = A.to(dev), Bm.to(dev)
    C1 = torch.empty(M, N, device=dev)
    ops.gemm(lib.GEMM_NT, Ad, Bd, C1, M, N, K, K, K, N)
    assert rel(C1, A.double() @ Bm.double().t()) < 1e-5
    # NN: C[M,K'] = D[M,N] * Bm[N,K]
    D = torch.randn(M, N, generator=g)
    C2 = torch.empty(M, K, device=dev)
    ops.gemm(lib.GEMM_NN, D.to(dev), Bd, C2, M, K, N, N, K, K)
    assert rel(C2, D.double() @ Bm.double()) < 1e-5
    # TN: C[N,K] = D[M,N]^T * A[M,K]
    C3 = torch.empty(N, K, device=dev)
    ops.gemm(lib.GEMM_TN, D.to(dev), Ad, C3, N, K, M, N, K, K)
    assert rel(C3, D.double().t() @ A.double()) < 1e-5
    # accumulate + alpha
    ops.gemm(lib.GEMM_TN, D.to(dev), Ad, C3, N, K, M, N, K, K, alpha=0.5, accumulate=True)
    assert rel(C3, 1.5 * (D.double().t() @ A.double())) < 1e-5


def test_gemm_random_shapes_all_paths(dev):
    """Seeded random sweep over the launch paths: 128x128 deep-prefetch loop (1, 2, 3, many K tiles, K
    tails), small tiles, split-K (long K, small outputs, tall TN), padded leading dimensions, bias
    epilogue, accumulate -- every result against fp64 on the device."""
    ops, lib = ops_mod(), L()
    rng = np.random.RandomState(2024)
    dims = [1, 3, 31, 32, 33, 64, 96, 127, 128, 129, 160, 255, 256, 300, 512, 640]
    ks = [1, 31, 32, 33, 64, 65, 96, 100, 128, 160, 512, 2048, 2080, 4096, 4100]
    cases = [(int(rng.choice(dims)), int(rng.choice(dims)), int(rng.choice(ks))) for _ in range(36)]
    cases += [(1536, 512, 8192), (512, 512, 8192), (256, 256, 33000), (2048, 128, 4096)]
    for idx, (M, N, K) in enumerate(cases):
        op = (lib.GEMM_NT, lib.GEMM_NN, lib.GEMM_TN)[idx % 3]
        pad_a, pad_b, pad_c = (0, 4, 8)[idx % 3], (0, 4)[idx % 2], (0, 4)[(idx // 2) % 2]
        g = torch.Generator(device=dev).manual_seed(idx)
        rn = lambda r, c, pad: torch.randn(r, c + pad, device=dev, generator=g)[:, :c]  # noqa: E731
        if op == lib.GEMM_NT:
            A, B = rn(M, K, pad_a), rn(N, K, pad_b)
            ref = A.double() @ B.double().t()
        elif op == lib.GEMM_NN:
            A, B = rn(M, K, pad_a), rn(K, N, pad_b)
            ref = A.double() @ B.double()
        else:
            A, B = rn(K, M, pad_a), rn(K, N, pad_b)
            ref = A.double().t() @ B.double()
        Cfull = torch.randn(M, N + pad_c, device=dev, generator=g)
        C = Cfull[:, :N]
        acc = idx % 4 == 1
        bias = torch.randn(N, device=dev, generator=g) if (idx % 5 == 2 and not acc) else None
        want = ref + (C.double() if acc else 0) + (bias.double() if bias is not None else 0)
        ops.gemm(op, A, B, C, M, N, K, A.stride(0), B.stride(0), C.stride(0), accumulate=acc,
                 epilogue=lib.EPI_BIAS if bias is not None else lib.EPI_NONE, bias=bias)
        err = float((C.double() - want).abs().max() / (want.abs().max() + 1e-30))
        assert err < 2e-5, (idx, op, M, N, K, pad_a, pad_b, pad_c, acc, bias is not None, err)


@pytest.mark.parametrize("op,M,N,K,acc,bias", [
    ("TN", 4096, 1024, 2240, True, False),    # medium-K weight gradient whose 128x128 grid is half a round -> 64x128 tiles
    ("TN", 8200, 1024, 2240, True, False),    # ... and one that leaves a nearly empty last round
    ("NN", 2240, 1024, 4096, False, False),   # input gradient under one round -> 128x64 tiles + the fill rule's K slices
    ("NN", 2240, 1024, 33000, False, False),  # very long reduction: stays on split 128x128 tiles
    ("NT", 2240, 4096, 1024, False, True),    # short K: 64x64 tiles fill the rounds better than 64x128
    ("NT", 2240, 9000, 1024, False, True),    # short K, large grid: 64x128 instead of 128x128
    ("NT", 8192, 512, 4096, False, False),    # the roofline shape: 128x64 tiles = 512 workgroups
    ("NT", 8192, 512, 4096, False, True),
    ("NN", 8192, 512, 1536, True, False),     # K = 1536: unsplittable, 64x128 tiles
    ("TN", 512, 4096, 8192, True, False),
    ("NT", 3200, 512, 4096, False, True),     # recipe-shape output projection: the bias epilogue under K slices
    ("NT", 1600, 512, 8192, True, True),      # ... and accumulating
    ("TN", 33000, 512, 8192, True, False),    # the headline's decoder weight gradient: tail slicing from the plan table (1024 whole tiles + 8 x 32 slices)
    ("TN", 33000, 1024, 2240, True, False),   # ... cfg2's (64x64 tiles, 4 slices in the tail)
])
def test_gemm_tile_rule_shapes(dev, op, M, N, K, acc, bias):
    """The shapes the round-2 tile / split rules single out (gemm_f32_mfma.h launch_op), against fp64 on the device."""
    ops, lib = ops_mod(), L()
    g = torch.Generator(device=dev).manual_seed(M + 3 * N + 7 * K)
    rn = lambda r, c: torch.randn(r, c, device=dev, generator=g)  # noqa: E731
    if op == "NT":
        A, B, code = rn(M, K), rn(N, K), lib.GEMM_NT
        ref = A.double() @ B.double().t()
    elif op == "NN":
        A, B, code = rn(M, K), rn(K, N), lib.GEMM_NN
        ref = A.double() @ B.double()
    else:
        A, B, code = rn(K, M), rn(K, N), lib.GEMM_TN
        ref = A.double().t() @ B.double()
    C = rn(M, N)
    bv = torch.randn(N, device=dev, generator=g) if bias else None
    want = ref + (C.double() if acc else 0) + (bv.double() if bias else 0)
    ops.gemm(code, A, B, C, M, N, K, A.stride(0), B.stride(0), N, accumulate=acc,
             epilogue=lib.EPI_BIAS if bias else lib.EPI_NONE, bias=bv)
    assert rel(C, want) < 1e-5


@pytest.mark.parametrize("acc", [False, True])
def test_gemm_bias_epilogue_forced_k_slices(dev, acc):
    """Bias epilogue under split-K (forced through blm_gemm_plan_override): the bias is added by the first slice only."""
    ops, lib = ops_mod(), L()
    g = torch.Generator(device=dev).manual_seed(77)
    M, N, K = 200, 136, 1056
    A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    bv = torch.randn(N, device=dev, generator=g) * 10
    for tile, splits in ((11, 4), (22, 3), (28, 2), (12, 8)):
        C = torch.randn(M, N, device=dev, generator=g)
        want = A.double() @ B.double().t() + bv.double() + (C.double() if acc else 0)
        lib.check(lib.lib().blm_gemm_plan_override(tile, splits), "override")
        try:
            ops.gemm(lib.GEMM_NT, A, B, C, M, N, K, K, K, N, accumulate=acc, epilogue=lib.EPI_BIAS, bias=bv)
        finally:
            lib.check(lib.lib().blm_gemm_plan_override(0, 0), "override")
        assert rel(C, want) < 1e-5, (tile, splits)


@pytest.mark.parametrize("op", ["NT", "NN", "TN"])
@pytest.mark.parametrize("tile,M,N", [(22, 3000, 2900), (28, 3000, 2900), (11, 2624, 2100), (12, 2000, 5000), (21, 5000, 2000),
                                      (22, 2048, 4096),   # 512 tiles = one whole round of the 128x128 tile: nothing to slice
                                      (22, 700, 900)])    # no whole round: every tile is sliced (the uniform form)
def test_gemm_tail_slicing(dev, op, tile, M, N):
    """Plans with splits < 0: the tiles of the whole rounds are computed in one piece, only the tiles of the last, partly
    filled round are cut along K (float atomics) -- plain, bias and accumulating epilogues, every layout."""
    ops, lib = ops_mod(), L()
    K = 1056
    g = torch.Generator(device=dev).manual_seed(M + N + tile)
    rn = lambda r, c: torch.randn(r, c, device=dev, generator=g)  # noqa: E731
    if op == "NT":
        A, B, code = rn(M, K), rn(N, K), lib.GEMM_NT
        ref = A.double() @ B.double().t()
    elif op == "NN":
        A, B, code = rn(M, K), rn(K, N), lib.GEMM_NN
        ref = A.double() @ B.double()
    else:
        A, B, code = rn(K, M), rn(K, N), lib.GEMM_TN
        ref = A.double().t() @ B.double()
    bv = torch.randn(N, device=dev, generator=g) * 5
    for acc, bias, S in ((False, False, -4), (True, False, -8), (False, True, -2), (True, True, -3)):
        C = rn(M, N)
        want = ref + (C.double() if acc else 0) + (bv.double() if bias else 0)
        lib.check(lib.lib().blm_gemm_plan_override(tile, S), "override")
        try:
            ops.gemm(code, A, B, C, M, N, K, A.stride(0), B.stride(0), N, accumulate=acc,
                     epilogue=lib.EPI_BIAS if bias else lib.EPI_NONE, bias=bv if bias else None)
        finally:
            lib.check(lib.lib().blm_gemm_plan_override(0, 0), "override")
        assert rel(C, want) < 1e-5, (acc, bias, S)


def test_bayes_wgrad_tail_slicing(dev):
    """The Bayesian weight-gradient epilogue (two outputs, KL terms once) under a tail plan."""
    ops, lib = ops_mod(), L()
    M, K, N = 1056, 2624, 2100          # dW is (N, K) = 2100 x 2624: 1353 tiles of 64 x 64 on 1280 slots
    g = torch.Generator().manual_seed(31)
    mu = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.1).to(dev))
    lg = torch.nn.Parameter((torch.rand(N, K, generator=g) - 3.0).to(dev))
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev)
    gy = (torch.randn(M, N, generator=g) * 0.1).to(dev)
    eps = torch.randn(N, K, generator=g)
    lam = 0.37
    lib.check(lib.lib().blm_gemm_plan_override(11, -4), "override")
    try:
        for _ in range(2):
            y = ops.bayes_linear(x, mu, lg, ops.NoiseSpec(eps=eps.to(dev)), kl_lambda=lam, fused=False)
            (y * gy).sum().backward()
    finally:
        lib.check(lib.lib().blm_gemm_plan_override(0, 0), "override")
    mu_c, lg_c = mu.detach().cpu().double().requires_grad_(True), lg.detach().cpu().double().requires_grad_(True)
    yr = O.bayes_linear(x.cpu().double(), mu_c, lg_c, eps.double())
    ((yr * gy.cpu().double()).sum() + lam * O.kl_mean_form(mu_c, lg_c)).backward()
    assert rel(mu.grad, 2 * mu_c.grad.float()) < 2e-5
    assert rel(lg.grad, 2 * lg_c.grad.float()) < 2e-5


@pytest.mark.parametrize("M,V,K,with_bias", [(2560, 33000, 512, True), (700, 33000, 1024, True), (77, 1000, 60, True), (5, 52, 18, True),
                                             (300, 4096, 64, False), (1, 8, 4, True), (129, 260, 33, True)])
def test_linear_nll_equals_log_softmax_of_the_materialised_logits(dev, M, V, K, with_bias):
    """blm_linear_nll (decoder product with softmax partials per column tile in its epilogue + a folding kernel; the logits
    are never stored) against log_softmax + gather of the fp64 logits -- every tile, aligned and guarded loaders."""
    ops, lib = ops_mod(), L()
    g = torch.Generator(device=dev).manual_seed(M + V + K)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(V, K, device=dev, generator=g) * (4.0 / K ** 0.5)
    b = torch.randn(V, device=dev, generator=g) if with_bias else None
    tgt = torch.randint(0, V, (M,), device=dev, generator=g)
    tgt[0] = V - 1                      # last column of the last (partial) tile
    tgt[-1] = 0
    logits = x.double() @ w.double().t() + (b.double() if with_bias else 0)
    want = -(torch.log_softmax(logits, 1).gather(1, tgt.view(-1, 1)).squeeze(1))
    tiles = (0,) if (K % 4 or V < 64) else (0, 11, 12, 21, 22, 28)
    for tile in tiles:
        lib.check(lib.lib().blm_gemm_plan_override(tile, 0), "override")
        try:
            with torch.no_grad():
                got = ops.linear_nll(x, w, b, tgt)
        finally:
            lib.check(lib.lib().blm_gemm_plan_override(0, 0), "override")
        assert float((got.double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())), tile
    with pytest.raises(Exception, match="inference-only"):
        ops.linear_nll(x.clone().requires_grad_(True), w, b, tgt)


@pytest.mark.parametrize("M,V,K1,K2,alpha", [(2560, 33000, 1024, 512, 0.8), (300, 33000, 512, 512, 0.5), (77, 1000, 12, 16, 0.8),
                                             (5, 52, 12, 16, 0.3), (129, 260, 64, 32, 1.0), (64, 4096, 32, 64, 0.0), (40, 30, 16, 12, 0.7),
                                             (200, 33001, 32, 64, 0.6)])
def test_linear_nll_interp_equals_log_softmax_of_the_interpolated_logits(dev, M, V, K1, K2, alpha):
    """blm_linear_nll2 (reference scorer :157-168): alpha (x1 W1^T + b1) + (1 - alpha) (x2 W2^T + b2) as ONE decoder +
    cross-entropy launch over the packed operands, against log_softmax + gather of the fp64 interpolated logits and against the
    materialised two-matrix kernel (blm_ce_interp_fwd); the packed weights are reused by a second call."""
    ops = ops_mod()
    g = torch.Generator(device=dev).manual_seed(M + V + K1)
    x1, x2 = torch.randn(M, K1, device=dev, generator=g), torch.randn(M, K2, device=dev, generator=g)
    w1 = torch.randn(V, K1, device=dev, generator=g) * (4.0 / K1 ** 0.5)
    w2 = torch.randn(V, K2, device=dev, generator=g) * (4.0 / K2 ** 0.5)
    b1, b2 = torch.randn(V, device=dev, generator=g), torch.randn(V, device=dev, generator=g)
    tgt = torch.randint(0, V, (M,), device=dev, generator=g)
    tgt[0], tgt[-1] = V - 1, 0
    l1 = x1.double() @ w1.double().t() + b1.double()
    l2 = x2.double() @ w2.double().t() + b2.double()
    want = -(torch.log_softmax(alpha * l1 + (1 - alpha) * l2, 1).gather(1, tgt.view(-1, 1)).squeeze(1))
    assert ops.linear_nll_interp_supported(w1, b1, w2, b2)
    dec = ops.InterpDecoder(w1, b1, w2, b2, alpha)
    with torch.no_grad():
        got = ops.linear_nll_interp(x1, x2, dec, tgt)
        assert dec.packed
        again = ops.linear_nll_interp(x1, x2, dec, tgt)  # [W1 | W2] and the mixed bias are reused
        _, mat = ops.cross_entropy_interp(ops.linear(x1, w1, b1), ops.linear(x2, w2, b2), alpha, tgt)
    tol = 2e-5 * max(1.0, float(want.abs().max()))
    assert float((got.double() - want).abs().max()) < tol
    assert torch.equal(got, again)
    assert float((mat.double() - want).abs().max()) < tol
    with pytest.raises(Exception, match="inference-only"):
        ops.linear_nll_interp(x1.clone().requires_grad_(True), x2, dec, tgt)


def test_linear_nll_out_of_range_target_is_nan_not_garbage(dev):
    """A target outside [0, V) (a padding id, -1) used to leave its logit slot uninitialised: the row's NLL is NaN."""
    ops = ops_mod()
    g = torch.Generator(device=dev).manual_seed(7)
    x, w = torch.randn(9, 16, device=dev, generator=g), torch.randn(64, 16, device=dev, generator=g)
    tgt = torch.randint(0, 64, (9,), device=dev, generator=g)
    tgt[2], tgt[5] = -1, 64
    with torch.no_grad():
        torch.full((1 << 16,), 3.25, device=dev)  # whatever the allocator hands out next is not zeros
        nll = ops.linear_nll(x, w, None, tgt)
    bad = torch.isnan(nll).cpu()
    assert bad.tolist() == [i in (2, 5) for i in range(9)]


def test_injected_eps_of_the_wrong_extent_is_refused(dev):
    """The kernels read an injected eps with lgstd's extent on trust; a mis-shaped, non-contiguous or CPU eps raises on the host."""
    ops, lib = ops_mod(), L()
    mu, lg = torch.zeros(8, 12, device=dev), torch.zeros(8, 12, device=dev)
    ok = ops.sample_weight(mu, lg, ops.NoiseSpec(eps=torch.ones(8, 12, device=dev)))
    assert torch.equal(ok, torch.ones(8, 12, device=dev))
    for bad in (torch.ones(8, 11, device=dev), torch.ones(12, 8, device=dev).t(), torch.ones(8, 12), torch.ones(8, 12, device=dev).double()):
        with pytest.raises(lib.BayesLMError, match="injected eps"):
            ops.sample_weight(mu, lg, ops.NoiseSpec(eps=bad))


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric B catches a transposed C write (cdna guide section 3)."""
    ops, lib = ops_mod(), L()
    n = 96
    Bm = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    out = torch.empty(n, n, device=dev)
    ops.gemm(lib.GEMM_NN, torch.eye(n, device=dev), Bm.to(dev), out, n, n, n, n, n, n)
    assert torch.equal(out.cpu(), Bm)
    ops.gemm(lib.GEMM_NT, torch.eye(n, device=dev), Bm.to(dev), out, n, n, n, n, n, n)
    assert torch.equal(out.cpu(), Bm.t())


def test_gemm_epilogues(dev):
    ops, lib = ops_mod(), L()
    M, N, K = 70, 90, 40
    g = torch.Generator().manual_seed(3)
    A, Bm, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    z_ref = A @ Bm.t() + bias
    out, aux = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(lib.GEMM_NT, A.to(dev), Bm.to(dev), out, M, N, K, K, K, N, epilogue=lib.EPI_BIAS, bias=bias.to(dev))
    assert rel(out, z_ref) < 1e-5
    ops.gemm(lib.GEMM_NT, A.to(dev), Bm.to(dev), out, M, N, K, K, K, N, epilogue=lib.EPI_BIAS_GELU, bias=bias.to(dev), aux=aux)
    assert rel(out, torch.nn.functional.gelu(z_ref)) < 5e-6
    # aux = gelu'(z) (times the dropout keep factor, 1 here); MUL_DGELU: C = (A B^T) * aux
    zz = z_ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(zz).sum().backward()
    assert rel(aux, zz.grad) < 1e-5
    ops.gemm(lib.GEMM_NT, A.to(dev), Bm.to(dev), out, M, N, K, K, K, N, epilogue=lib.EPI_MUL_DGELU, aux=aux)
    assert rel(out, (A @ Bm.t()) * zz.grad) < 1e-5
    # fused bias gradient: column sums of the A operand of a wgrad-shaped TN GEMM (also with split-K)
    for Mw, Nw, Kw in ((90, 40, 70), (512, 512, 4096), (1536, 64, 300)):
        dY, X = torch.randn(Kw, Mw, generator=g), torch.randn(Kw, Nw, generator=g)
        dW, db = torch.zeros(Mw, Nw, device=dev), torch.ones(Mw, device=dev)
        ops.gemm(lib.GEMM_TN, dY.to(dev), X.to(dev), dW, Mw, Nw, Kw, Mw, Nw, Nw, accumulate=True, colsum_a=db)
        assert rel(dW, dY.double().t() @ X.double()) < 1e-5
        assert rel(db, 1.0 + dY.double().sum(0)) < 1e-5
    # GP mixture (tanh, sigmoid, relu, gelu) and its derivative
    coef = torch.rand(4, N, generator=g)
    ops.gemm(lib.GEMM_NT, A.to(dev), Bm.to(dev), out, M, N, K, K, K, N, epilogue=lib.EPI_GP_MIX, bias=bias.to(dev), aux=aux,
             coef=coef.to(dev))
    assert rel(out, O.gp_mixture(z_ref, coef, ["tanh", "sigmoid", "relu", "gelu"])) < 1e-5
    zz = z_ref.clone().requires_grad_(True)
    O.gp_mixture(zz, coef, ["tanh", "sigmoid", "relu", "gelu"]).sum().backward()
    ops.gemm(lib.GEMM_NT, A.to(dev), Bm.to(dev), out, M, N, K, K, K, N, epilogue=lib.EPI_MUL_DGP_MIX, aux=aux, coef=coef.to(dev))
    assert rel(out, (A @ Bm.t()) * zz.grad) < 2e-5


# ------------------------------------------------------------------ Philox / sampling / KL
def test_philox_stream_matches_numpy_oracle(dev):
    ops = ops_mod()
    for n, seed, stream, step in ((1, 1, 0x1000, 0), (1003, 1111, 0x1005, 7), (1 << 16, 2 ** 40 + 5, 0x1fff, 123456)):
        z = ops.philox_normal(n, seed, stream, step).cpu().numpy()
        ref = P.normal(n, seed, stream, step)
        # integer Philox is bit exact; Box-Muller differs by the transcendental approximations
        np.testing.assert_allclose(z, ref, rtol=0, atol=3e-5)


def test_sample_weight_injected_and_philox(dev):
    ops = ops_mod()
    g = torch.Generator().manual_seed(5)
    for rows, cols, lo, srows in ((8, 12, 0, 8), (16, 12, 4, 4), (12, 1, 3, 3), (7, 5, 0, 7), (512, 4096, 0, 512)):
        mu = torch.randn(rows, cols, generator=g).squeeze(-1) if cols == 1 else torch.randn(rows, cols, generator=g)
        lg = (torch.rand(srows, cols, generator=g) - 2.0)
        lg = lg.squeeze(-1) if cols == 1 else lg
        eps = torch.randn_like(lg)
        kl = torch.zeros((), device=dev)
        W = ops.sample_weight(mu.to(dev), lg.to(dev), ops.NoiseSpec(eps=eps.to(dev)), lo, srows, kl_out=kl, kl_weight=2.0)
        ref = mu.clone()
        ref[lo:lo + srows] += torch.exp(lg) * eps
        assert rel(W, ref) < 1e-6
        assert abs(float(kl) - 2.0 * float(O.kl_mean_form(mu[lo:lo + srows], lg))) < 1e-4 * abs(float(kl)) + 1e-6
        # Philox mode equals the numpy stream
        W2 = ops.sample_weight(mu.to(dev), lg.to(dev), ops.NoiseSpec(None, 77, 3, 9), lo, srows)
        z = torch.from_numpy(P.normal(srows * cols, 77, P.STREAM_WEIGHT + 3, 9)).view_as(lg)
        ref2 = mu.clone()
        ref2[lo:lo + srows] += torch.exp(lg) * z
        assert float((W2.cpu() - ref2).abs().max()) < 1e-4


def test_integration_md_ctypes_stub_runs_as_written(dev):
    """INTEGRATION.md section 2 shows the binding a maintainer would add: that first code block is executed VERBATIM here (from the
    repository root, as the snippet's relative library path assumes) and its `sample_weight` held to the numpy Philox oracle
    and to the KL formula of BayesLinear.kl_divergence (model.py:1109-1125)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    sec = doc[doc.index("## 2. Binding the C ABI directly"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "def sample_weight(" in code and "blm_sample_weight" in code
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)
    try:
        exec(compile(code, "INTEGRATION.md:section2", "exec"), ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(9)
    mu, lg = torch.randn(24, 20, generator=g), torch.rand(24, 20, generator=g) - 2.0
    kl = torch.zeros((), device=dev)
    W = ns["sample_weight"](mu.to(dev), lg.to(dev), 77, 3, 9, kl_out=kl)
    z = torch.from_numpy(P.normal(24 * 20, 77, P.STREAM_WEIGHT + 3, 9)).view_as(lg)
    assert float((W.cpu() - (mu + torch.exp(lg) * z)).abs().max()) < 1e-4
    assert abs(float(kl) - float(O.kl_mean_form(mu, lg))) < 1e-4 * abs(float(kl)) + 1e-6
    # a bad call comes back as a status with a message, which is what the stub turns into its RuntimeError
    rc = ns["lib"].blm_sample_weight(None, 1, 1, None, None, None, ns["C"].c_float(1.0), None)
    assert rc != 0 and b"blm_sample_weight" in ns["lib"].blm_last_error()


@pytest.mark.parametrize("sizes", [(), (250, 100, 75), (1024, 512, 4096), (64, 4096, 512)])
def test_c_host_without_python_or_torch_runs_a_bayes_linear_step(dev, tmp_path, sizes):
    """The drop-in boundary is the C ABI, not the Python package: examples/c_host/bayes_linear_step.c (plain C + the HIP runtime,
    built by gcc) runs one BayesLinear training step (model.py:1083-1129 and its autograd) through blm_sample_weight /
    blm_gemm NT, NN, TN + BLM_EPI_BAYES_WGRAD and holds y, dx, dmu, dlgstd and the KL to a double-precision loop over the
    same Philox noise (1e-4), at the default size, an odd one and the headline layer's two shapes."""
    import subprocess
    from conftest import build_c_host
    exe = build_c_host(tmp_path)
    r = subprocess.run([exe] + [str(v) for v in sizes], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK (worst relative difference" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "gfx950" in r.stdout and "returns -2" in r.stdout


@pytest.mark.parametrize("fused", [False, True])
def test_bayes_linear_golden(dev, fused):
    """BayesLinear fwd/bwd/KL against the reference's own numbers (tests/golden/bayes_linear.npz)."""
    from conftest import load_golden
    ops = ops_mod()
    g, _, _ = load_golden("bayes_linear")
    x = g["x"].to(dev).requires_grad_(True)
    mu = torch.nn.Parameter(g["mu"].to(dev))
    lg = torch.nn.Parameter(g["lgstd"].to(dev))
    lam = float(g["kl_scale"])
    noise = ops.NoiseSpec(eps=g["eps"].to(dev))
    y = ops.bayes_linear(x, mu, lg, noise, kl_lambda=0.0, fused=fused)
    kl = ops.kl_mean(mu, lg)
    assert rel(y, g["y_train"]) < 1e-5
    assert abs(float(kl) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    ((y * g["g"].to(dev)).sum() + kl * lam).backward()
    assert rel(x.grad, g["dx"]) < 1e-5
    assert rel(mu.grad, g["dmu"]) < 1e-5
    assert rel(lg.grad, g["dlgstd"]) < 1e-5
    # fused KL gradient in the wgrad epilogue gives the same totals without backprop through kl
    mu.grad = None
    lg.grad = None
    x2 = g["x"].to(dev).requires_grad_(True)
    y2 = ops.bayes_linear(x2, mu, lg, noise, kl_lambda=lam, fused=fused)
    (y2 * g["g"].to(dev)).sum().backward()
    assert rel(mu.grad, g["dmu"]) < 1e-5
    assert rel(lg.grad, g["dlgstd"]) < 1e-5
    # eval mode = mean weights
    assert rel(ops.bayes_linear(x, mu, lg, None), g["y_eval"]) < 1e-5


@pytest.mark.parametrize("fused", [False, True])
def test_bayes_linear_philox_fwd_bwd_consistent(dev, fused):
    """Philox mode: forward, dgrad and the wgrad epilogue all regenerate the SAME eps from the counter."""
    ops = ops_mod()
    N, K, M = 64, 128, 96
    g = torch.Generator().manual_seed(8)
    mu = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.1).to(dev))
    lg = torch.nn.Parameter((torch.rand(N, K, generator=g) - 3.0).to(dev))
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(M, N, generator=g).to(dev)
    noise = ops.NoiseSpec(None, 1111, 5, 42)
    y = ops.bayes_linear(x, mu, lg, noise, 0.0, fused)
    (y * gy).sum().backward()
    eps = torch.from_numpy(P.normal(N * K, 1111, P.STREAM_WEIGHT + 5, 42)).view(N, K)
    mu_c, lg_c, x_c = mu.detach().cpu().requires_grad_(True), lg.detach().cpu().requires_grad_(True), x.detach().cpu().requires_grad_(True)
    yr = O.bayes_linear(x_c, mu_c, lg_c, eps)
    (yr * gy.cpu()).sum().backward()
    assert rel(y, yr) < 2e-4
    assert rel(x.grad, x_c.grad) < 2e-4
    assert rel(mu.grad, mu_c.grad) < 2e-4
    assert rel(lg.grad, lg_c.grad) < 5e-4


@pytest.mark.parametrize("M,K,N", [(4096, 256, 512), (8192, 512, 1024)])
def test_bayes_linear_wgrad_split_k(dev, M, K, N):
    """Long reduction (M rows of the batch): the Bayesian wgrad epilogue runs under split-K -- both
    gradients accumulate through atomics, the KL terms are added by the first K slice only.  Two
    backward passes into the same .grad must give exactly twice the CPU gradient (KL included once
    per pass)."""
    ops = ops_mod()
    g = torch.Generator().manual_seed(M + N)
    mu = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.1).to(dev))
    lg = torch.nn.Parameter((torch.rand(N, K, generator=g) - 3.0).to(dev))
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev)
    gy = (torch.randn(M, N, generator=g) * 0.1).to(dev)
    eps = torch.randn(N, K, generator=g)
    lam = 0.37
    for _ in range(2):
        y = ops.bayes_linear(x, mu, lg, ops.NoiseSpec(eps=eps.to(dev)), kl_lambda=lam, fused=False)
        (y * gy).sum().backward()
    mu_c, lg_c = mu.detach().cpu().double().requires_grad_(True), lg.detach().cpu().double().requires_grad_(True)
    yr = O.bayes_linear(x.cpu().double(), mu_c, lg_c, eps.double())
    ((yr * gy.cpu().double()).sum() + lam * O.kl_mean_form(mu_c, lg_c)).backward()
    assert rel(mu.grad, 2 * mu_c.grad.float()) < 2e-5
    assert rel(lg.grad, 2 * lg_c.grad.float()) < 2e-5


def test_kl_mean_window_and_minus_one(dev):
    ops = ops_mod()
    g = torch.Generator().manual_seed(2)
    mu = torch.nn.Parameter(torch.randn(16, 6, generator=g).to(dev))
    lg = torch.nn.Parameter((torch.rand(4, 6, generator=g) - 1).to(dev))
    kl = ops.kl_mean(mu, lg, row_lo=8, minus_one=True)
    mu_c, lg_c = mu.detach().cpu().requires_grad_(True), lg.detach().cpu().requires_grad_(True)
    ref = O.kl_mean_form_minus1(mu_c[8:12], lg_c)
    assert abs(float(kl) - float(ref)) < 1e-5
    (kl * 3.0).backward()
    (ref * 3.0).backward()
    assert rel(mu.grad, mu_c.grad) < 1e-5 and rel(lg.grad, lg_c.grad) < 1e-5


@pytest.mark.parametrize("inject", [True, False])
def test_variational_group_equals_single_tensor_ops(dev, inject):
    """blm_variational_group_fwd/_bwd (all tensors of a module in one launch, KL and its gradient folded in) against
    the single-tensor entry points on the same items: float4 and scalar items, row windows, a bias vector, items with
    and without a KL weight, an item whose sample is not used downstream (no dW)."""
    ops = ops_mod()
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 32, 16, 16, 0.5, 0.0), (64, 40, 16, 16, 1.5, 0.0), (64, 1, 16, 16, 0.25, 0.0), (7, 5, 0, 7, 2.0, 1.0),
              (48, 8, 0, 48, 0.0, 0.0), (4096, 1024, 2048, 1024, 0.75, 0.0), (33, 3, 30, 3, 0.0, 0.0)]
    def build():
        gen = torch.Generator().manual_seed(12)
        items = []
        for k, (rows, cols, lo, srows, klw, klm) in enumerate(shapes):
            mu = torch.randn(rows, cols, generator=gen)
            lg = torch.rand(srows, cols, generator=gen) - 2.0
            if cols == 1:
                mu, lg = mu.squeeze(-1), lg.squeeze(-1)
            eps = torch.randn(lg.shape, generator=gen)
            noise = ops.NoiseSpec(eps=eps.to(dev)) if inject else ops.NoiseSpec(None, 1234, 40 + k, 3)
            items.append((torch.nn.Parameter(mu.to(dev)), torch.nn.Parameter(lg.to(dev)), noise, lo, klw, klm))
        return items
    gy = [torch.randn(r, c, generator=g).squeeze(-1).to(dev) if c == 1 else torch.randn(r, c, generator=g).to(dev)
          for r, c, *_ in shapes]
    unused = 4  # this item's sample gets no gradient
    # grouped
    a = build()
    Ws, kl = ops.variational_group(a)
    loss = sum((W * y).sum() for k, (W, y) in enumerate(zip(Ws, gy)) if k != unused) + 3.0 * kl
    loss.backward()
    # single-tensor ops
    b = build()
    Ws1 = [ops.sampled(mu, lg, noise, lo) for mu, lg, noise, lo, _, _ in b]
    kl1 = sum(ops.kl_mean(mu, lg, lo, minus_one=bool(klm)) * klw for mu, lg, _, lo, klw, klm in b if klw != 0.0)
    loss1 = sum((W * y).sum() for k, (W, y) in enumerate(zip(Ws1, gy)) if k != unused) + 3.0 * kl1
    loss1.backward()
    for W, W1 in zip(Ws, Ws1):
        assert torch.equal(W, W1)
    assert abs(float(kl) - float(kl1)) < 1e-5 * abs(float(kl1))
    for k, ((mu, lg, *_), (mu1, lg1, *_)) in enumerate(zip(a, b)):
        if k == unused and shapes[k][4] == 0.0:
            assert mu.grad is None or float(mu.grad.abs().max()) == 0.0
            continue
        assert rel(mu.grad, mu1.grad) < 1e-6, k
        assert rel(lg.grad, lg1.grad) < 1e-6, k
    # the group refuses what it cannot hold
    with pytest.raises(Exception):
        ops.variational_group([a[0]] * 17)


# ------------------------------------------------------------------ surrounding ops
def test_embed_pe_dropout0(dev):
    ops = ops_mod()
    V, D, T, B = 50, 24, 7, 3
    g = torch.Generator().manual_seed(4)
    W = torch.nn.Parameter(torch.randn(V, D, generator=g).to(dev))
    pe = O.positional_table(64, D).view(64, D)
    ids = torch.randint(0, V, (T, B), generator=g)
    out = ops.embed(ids.to(dev), W, pe.to(dev), 3.0)
    ref_w = W.detach().cpu().requires_grad_(True)
    ref = torch.nn.functional.embedding(ids, ref_w) * 3.0 + pe[:T].unsqueeze(1)
    assert rel(out, ref) < 1e-6
    gy = torch.randn(T, B, D, generator=g)
    out.backward(gy.to(dev))
    ref.backward(gy)
    assert rel(W.grad, ref_w.grad) < 1e-5


def test_dropout_mask_is_the_philox_mask_and_column_sharding(dev):
    ops = ops_mod()
    T, B, D, p = 5, 8, 16, 0.3
    x = torch.ones(T, B, D, device=dev)
    d = ops.Drop(p, 99, 4, 7, 0, B)
    y = ops.dropout(x, d).cpu()
    keep = torch.from_numpy(P.keep_mask(T * B * D, p, 99, P.STREAM_DROPOUT + 4, 7)).view(T, B, D)
    assert torch.equal(y != 0, keep)
    assert rel(y[keep], torch.full_like(y[keep], 1 / (1 - p))) < 1e-6
    # rank 1 of 2 (columns 4..7) sees exactly its slice of the global mask
    y1 = ops.dropout(x[:, 4:].contiguous(), ops.Drop(p, 99, 4, 7, 4, B)).cpu()
    assert torch.equal(y1, y[:, 4:])


@pytest.mark.parametrize("D", [16, 18, 100, 200, 384, 512, 640, 768, 1000, 1024, 1100, 1280, 1536])
def test_add_dropout_ln(dev, D):
    ops = ops_mod()
    T, B = 6, 5
    g = torch.Generator().manual_seed(D)
    x, y = torch.randn(T, B, D, generator=g), torch.randn(T, B, D, generator=g)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    gd, bd = torch.nn.Parameter(gamma.to(dev)), torch.nn.Parameter(beta.to(dev))
    xd, yd = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    out = ops.add_dropout_ln(xd, yd, gd, bd, 1e-5)
    xr, yr, gr, br = (t.clone().requires_grad_(True) for t in (x, y, gamma, beta))
    ref = torch.nn.functional.layer_norm(xr + yr, (D,), gr, br, 1e-5)
    assert rel(out, ref) < 1e-5
    go = torch.randn(T, B, D, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    assert rel(xd.grad, xr.grad) < 2e-5 and rel(yd.grad, yr.grad) < 2e-5
    assert rel(gd.grad, gr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5
    # with dropout: s = x + keep*y/(1-p), the mask being the Philox mask
    p = 0.25
    drop = ops.Drop(p, 5, 1, 2, 0, B)
    out2 = ops.add_dropout_ln(x.to(dev), y.to(dev), gd, bd, 1e-5, drop)
    keep = torch.from_numpy(P.keep_mask(T * B * D, p, 5, P.STREAM_DROPOUT + 1, 2)).view(T, B, D).float() / (1 - p)
    assert rel(out2, torch.nn.functional.layer_norm(x + y * keep, (D,), gamma, beta, 1e-5)) < 1e-5


@pytest.mark.parametrize("T,B,nhead,hd", [(6, 3, 4, 4), (100, 2, 2, 32), (128, 3, 2, 64), (33, 2, 1, 16), (33, 2, 2, 64),
                                          (1, 1, 1, 64), (97, 1, 3, 64),
                                          # T <= 32: the forward runs one wave per head (attn_fwd_short_kernel)
                                          (20, 5, 3, 64), (32, 64, 8, 64), (7, 3, 2, 64), (31, 2, 5, 64),
                                          # 512 heads and more: two heads per workgroup (below: one, to cover more CUs)
                                          (128, 64, 8, 64), (100, 65, 8, 64), (37, 128, 4, 64),
                                          # longer than 128: the chunked (flash-style) matrix-core kernels
                                          (129, 2, 2, 64), (200, 1, 3, 64), (256, 2, 1, 64), (301, 1, 2, 64), (515, 1, 1, 64),
                                          # any other head size / length: the untiled one-wave-per-row kernels
                                          # (train.py's defaults give head_dim 100)
                                          (35, 2, 2, 100), (20, 1, 3, 7), (150, 2, 2, 32), (140, 1, 1, 100), (9, 2, 1, 130),
                                          # head_dim 128, T <= 128: two lanes per row (attn_*_wide_kernel); beyond 128 steps: untiled again
                                          (128, 3, 2, 128), (1, 2, 1, 128), (37, 5, 3, 128), (100, 64, 4, 128), (127, 1, 1, 128), (129, 1, 2, 128),
                                          # every other head size up to 128 takes the same kernels with its tail zero-filled
                                          (128, 2, 2, 100), (64, 3, 2, 96), (33, 2, 3, 48), (50, 2, 2, 65), (17, 1, 2, 127), (40, 2, 1, 3), (128, 1, 3, 63)])
def test_attention_matches_oracle(dev, T, B, nhead, hd):
    ops = ops_mod()
    d = nhead * hd
    g = torch.Generator().manual_seed(T + hd)
    qkv = torch.randn(T, B, 3 * d, generator=g)
    qd = qkv.to(dev).requires_grad_(True)
    out = ops.attention(qd, nhead)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.chunk(3, dim=-1)
    ref = O.attention_core(q, k, v, nhead, O.causal_mask(T))
    assert rel(out, ref) < 1e-5
    go = torch.randn(T, B, d, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    assert rel(qd.grad, qr.grad) < 2e-5
    # separate q/k/v entry point
    q3 = [t.contiguous().to(dev).requires_grad_(True) for t in qkv.chunk(3, dim=-1)]
    out3 = ops.attention_qkv(q3[0], q3[1], q3[2], nhead)
    assert rel(out3, ref) < 1e-5
    out3.backward(go.to(dev))
    assert rel(torch.cat([t.grad for t in q3], -1), qr.grad) < 2e-5


@pytest.mark.parametrize("T,B,nhead,hd", [(8, 2, 2, 4), (96, 2, 2, 64), (50, 3, 1, 64), (128, 1, 2, 64), (160, 2, 1, 64),
                                          (203, 1, 2, 64), (30, 2, 2, 100), (131, 1, 1, 32), (66, 64, 8, 64), (30, 3, 2, 64), (16, 2, 4, 64),
                                          (128, 2, 2, 128), (45, 3, 1, 128), (77, 2, 2, 100)])
def test_attention_dropout_uses_philox_mask(dev, T, B, nhead, hd):
    """Probability dropout (VALU kernels for small heads, MFMA kernels for head_dim 64; T % 4 != 0
    takes the per-element Philox path)."""
    ops = ops_mod()
    p = 0.4
    d = nhead * hd
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(T, B, 3 * d, generator=g)
    drop = ops.Drop(p, 321, 9, 3, 0, B)
    qd = qkv.to(dev).requires_grad_(True)
    out = ops.attention(qd, nhead, drop)
    keep = torch.from_numpy(P.keep_mask(B * nhead * T * T, p, 321, P.STREAM_DROPOUT + 9, 3)).view(B * nhead, T, T).float() / (1 - p)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.chunk(3, dim=-1)
    qh = (q * hd ** -0.5).contiguous().view(T, B * nhead, hd).transpose(0, 1)
    kh = k.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    vh = v.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    pr = torch.softmax(torch.bmm(qh, kh.transpose(1, 2)) + O.causal_mask(T), -1) * keep
    ref = torch.bmm(pr, vh).transpose(0, 1).contiguous().view(T, B, d)
    assert rel(out, ref) < 1e-5
    go = torch.randn(T, B, d, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    assert rel(qd.grad, qr.grad) < 2e-5


@pytest.mark.parametrize("T,B,nhead,hd", [(8, 2, 2, 4), (96, 2, 2, 64), (128, 1, 2, 64), (160, 2, 1, 64), (30, 2, 2, 100), (131, 1, 1, 32),
                                          (7, 4, 4, 4), (90, 2, 2, 128)])
@pytest.mark.parametrize("window", [False, True])
def test_attention_with_the_dropout_mask_handed_over(dev, T, B, nhead, hd, window):
    """blm_attn_fwd_keep / blm_attn_bwd_keep (``Drop.keep``): the dropout factors of the probabilities are an operand -- the
    (B_global * nhead, T, T) tensor torch's CPU dropout drew for the reference's probabilities (NoiseState.source "torch") --
    instead of the Philox stream.  LDS-tiled and one-wave-per-query vector-ALU kernels (head_dim 64 included: the matrix-core
    kernels only know the stream); ``window``: this rank holds columns [1, 1 + B) of a global batch of B + 2."""
    ops = ops_mod()
    p = 0.3
    d = nhead * hd
    G, off = (B + 2, 1) if window else (B, 0)
    g = torch.Generator().manual_seed(T + hd)
    torch.manual_seed(T)
    keep_all = torch.dropout(torch.ones(G * nhead, T, T), p, True)
    keep = keep_all[off * nhead:(off + B) * nhead]
    qkv = torch.randn(T, B, 3 * d, generator=g)
    qd = qkv.to(dev).requires_grad_(True)
    out = ops.attention(qd, nhead, ops.Drop(p, col_offset=off, global_cols=G, keep=keep_all.to(dev)))
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.chunk(3, dim=-1)
    qh = (q * hd ** -0.5).contiguous().view(T, B * nhead, hd).transpose(0, 1)
    kh = k.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    vh = v.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    pr = torch.softmax(torch.bmm(qh, kh.transpose(1, 2)) + O.causal_mask(T), -1) * keep
    ref = torch.bmm(pr, vh).transpose(0, 1).contiguous().view(T, B, d)
    assert rel(out, ref) < 1e-5
    go = torch.randn(T, B, d, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    assert rel(qd.grad, qr.grad) < 2e-5
    with pytest.raises(Exception):  # a mask of another size is refused
        ops.attention(qd, nhead, ops.Drop(p, col_offset=off, global_cols=G, keep=keep_all[:-1].contiguous().to(dev)))


@pytest.mark.parametrize("T,B,nhead,p", [(128, 4, 8, 0.2), (128, 3, 1, 0.0), (97, 1, 3, 0.3), (50, 3, 2, 0.4), (33, 2, 2, 0.0),
                                         (1, 1, 1, 0.0), (64, 2, 2, 0.1), (130, 2, 2, 0.2), (100, 64, 8, 0.2)])
def test_attention_backward_workspace_path_equals_recomputation(dev, monkeypatch, T, B, nhead, p):
    """blm_attn_bwd_ws (dK/dV pass leaves dS in the scratch buffer, dQ = dS K by the same workgroup) against blm_attn_bwd
    (two recomputations): one or two heads per workgroup, partial tiles, T % 4 != 0, dropout on/off; T > 128 asks for no
    workspace and runs the chunked kernels either way."""
    ops = ops_mod()
    hd, d = 64, nhead * 64
    assert int(L().lib().blm_attn_bwd_ws_floats(T, B, nhead, hd)) == (B * nhead * T * T if T <= 128 else 0)
    g = torch.Generator().manual_seed(T * 7 + B)
    qkv = torch.randn(T, B, 3 * d, generator=g).to(dev)
    go = torch.randn(T, B, d, generator=g).to(dev)
    drop = ops.Drop(p, 99, 4, 6, 0, B) if p > 0 else ops.NO_DROP
    grads = []
    for ws in (True, False):
        monkeypatch.setattr(ops, "_ATTN_WS", ws)
        x = qkv.clone().requires_grad_(True)
        ops.attention(x, nhead, drop).backward(go)
        grads.append(x.grad)
    assert rel(grads[0], grads[1]) < 2e-6


@pytest.mark.parametrize("M,V", [(5, 7), (18, 50), (64, 33000), (3, 1001), (9, 40), (4, 12288), (5, 12284), (3, 50000), (2, 70000)])
def test_cross_entropy(dev, M, V):
    ops = ops_mod()
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(M, V, generator=g) * 3
    tgt = torch.randint(0, V, (M,), generator=g)
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, tgt)
    (ref * 1.7).backward()
    for unit in (False, True):
        ld = logits.to(dev).requires_grad_(True)
        loss, nll = ops.cross_entropy(ld * 1.0, tgt.to(dev), unit_grad=unit)
        assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
        assert rel(nll, O.token_nll(logits, tgt)) < 1e-5
        (loss * (1.0 if unit else 1.7)).backward()
        assert rel(ld.grad, lr.grad / (1.7 if unit else 1.0)) < 1e-5
    with torch.no_grad():
        loss, _ = ops.cross_entropy(logits.to(dev), tgt.to(dev))
        assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))


@pytest.mark.parametrize("V", [33278, 1001, 67, 10002])
def test_a_vocabulary_that_is_not_a_multiple_of_4_keeps_the_vector_paths(dev, V):
    """wikitext-2 has 33278 words: rows of V floats are not 16-byte aligned, which used to take the decoder's epilogue, the cross
    entropy and both backward products off their vector paths (headline step 19.9 -> 26.4 ms).  ops.linear pads the rows of such an
    output to a multiple of 4 floats and hands out the (..., V) view; the cross entropy and the backward products take the row
    stride.  Against torch: logits, loss (engine trainer form, the reference loop's form and `.view(-1, V)` as train.py:404 does),
    every gradient; the padding columns are never read as data."""
    ops = ops_mod()
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(V)
    T, B, K = 6, 5, 32
    x = (torch.randn(T, B, K, generator=g) * 0.5)
    w = (torch.randn(V, K, generator=g) * 0.1)
    b = torch.randn(V, generator=g) * 0.1
    t = torch.randint(0, V, (T * B,), generator=g)
    ref_leaves = [a.clone().requires_grad_(True) for a in (x, w, b)]
    ref_logits = F.linear(*ref_leaves)
    ref = F.cross_entropy(ref_logits.view(-1, V), t)
    ref.backward()
    for form in ("engine", "engine_unit", "torch_loop"):
        xs, ws, bs = [a.to(dev).requires_grad_(True) for a in (x, w, b)]
        y = ops.linear(xs, ws, bs)
        assert y.shape == (T, B, V) and y.stride(-1) == 1 and y.stride(-2) % 4 == 0 and y.stride(-2) >= V and y.data_ptr() % 16 == 0
        assert (y.stride(-2) == V) == (V % 4 == 0)
        assert rel(y, ref_logits) < 1e-5
        if form == "torch_loop":   # the reference's loop: nn.CrossEntropyLoss on output.view(-1, ntokens) of the model's Logits
            y = ops.as_logits(y)
            before = y.detach().clone()
            loss = torch.nn.CrossEntropyLoss()(y.view(-1, V), t.to(dev))
        else:
            loss, nll = ops.cross_entropy(y.view(-1, V), t.to(dev), unit_grad=form == "engine_unit")
            assert nll.shape == (T * B,)
        assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
        loss.backward()
        for a, r, name in zip((xs, ws, bs), ref_leaves, ("x", "w", "b")):
            assert rel(a.grad, r.grad) < 2e-5, (form, name)
        if form == "torch_loop":
            assert torch.equal(y.detach().as_subclass(torch.Tensor), before)  # the caller's logits are untouched
    # a strided tensor the helper cannot take as it is (rows not 16-byte aligned) is copied, not misread
    odd = torch.randn(T * B, V + 1, generator=g).to(dev)[:, 1:]
    loss2, _ = ops.cross_entropy(odd.clone().requires_grad_(True)[:, :], t.to(dev))
    assert abs(float(loss2) - float(F.cross_entropy(odd.cpu(), t))) < 1e-5 * abs(float(loss2))


def test_colsum_and_clip_sgd(dev):
    ops, lib = ops_mod(), L()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(300, 70, generator=g)
    out = torch.zeros(70, device=dev)
    lib.check(lib.lib().blm_colsum(x.to(dev).data_ptr(), 70, out.data_ptr(), 300, 70, 0, lib.stream()))
    assert rel(out, x.sum(0)) < 1e-5
    # clip + SGD momentum, two steps, vs the oracle restatement of train.py:419-420
    ps = [torch.randn(n, generator=g) for n in (1000, 37, 4096)]
    gs = [torch.randn(n, generator=g) * 3 for n in (1000, 37, 4096)]
    pd, gd = [p.to(dev) for p in ps], [x.to(dev) for x in gs]
    md = [torch.zeros_like(p) for p in pd]
    table = ops.PtrTable(pd, gd, md)
    pr = [torch.nn.Parameter(p.clone()) for p in ps]
    bufs = [None] * 3
    for step in range(2):
        sq = ops.clip_sgd(table, 0.25, 0.1, 0.9, step == 0)
        total = O.clip_and_sgd(pr, gs, bufs, 0.1, 0.25)
        assert abs(float(sq.sqrt()) - float(total)) < 1e-4 * float(total)
        for a, b in zip(pd, pr):
            assert rel(a, b.data) < 1e-5


def test_lstm_layer_matches_oracle(dev):
    ops = ops_mod()
    T, B, E, H = 7, 5, 24, 16
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=g) * 0.3  # noqa: E731
    x, h0, c0 = mk(T, B, E), mk(B, H), mk(B, H)
    w_ih, w_hh, b_ih, b_hh = mk(4 * H, E), mk(4 * H, H), mk(4 * H), mk(4 * H)
    dl = [t.to(dev).requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    y, hT, cT = ops.lstm_layer(*dl)
    cl = [t.clone().requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    yr, hr, cr = O.lstm_layer(*cl)
    assert rel(y, yr) < 1e-5 and rel(hT, hr) < 1e-5 and rel(cT, cr) < 1e-5
    gy, gh, gc = mk(T, B, H), mk(B, H), mk(B, H)
    ((y * gy.to(dev)).sum() + (hT * gh.to(dev)).sum() + (cT * gc.to(dev)).sum()).backward()
    ((yr * gy).sum() + (hr * gh).sum() + (cr * gc).sum()).backward()
    for a, b, name in zip(dl, cl, "x h0 c0 w_ih w_hh b_ih b_hh".split()):
        assert rel(a.grad, b.grad) < 5e-5, name


@pytest.mark.parametrize("T,B,E,H", [(5, 64, 48, 256), (4, 20, 32, 96), (3, 70, 40, 320), (3, 1, 16, 32), (6, 2, 16, 64),
                                     (6, 3, 24, 96), (4, 4, 32, 1024), (5, 1, 64, 1024), (3, 5, 16, 64),
                                     # the software-pipelined K loops: whole chunks only (H % 256 == 0: no zero fill, scalar
                                     # chunk offsets) with ragged batch tiles, and the general form with a K tail (H = 1856)
                                     (3, 37, 32, 1024), (2, 33, 32, 2048), (2, 9, 32, 1536), (2, 5, 32, 1856), (2, 8, 32, 1280)])
def test_lstm_layer_fused_step_matches_oracle(dev, T, B, E, H):
    """H % 32 == 0 takes blm_lstm_step_fwd (one launch per step: MFMA recurrent product + cell):
    ragged batch tiles (20, 70, 1), K tails (H/8 = 12, 40), and the accumulate-in-place backward.  B <= 4 forwards run
    the tiny-batch kernel (one wave per hidden unit, the scorer's carry chain), B = 5 the MFMA one again."""
    ops = ops_mod()
    g = torch.Generator().manual_seed(12)
    mk = lambda *s: torch.randn(*s, generator=g) * 0.2  # noqa: E731
    x, h0, c0 = mk(T, B, E), mk(B, H), mk(B, H)
    w_ih, w_hh, b_ih, b_hh = mk(4 * H, E), mk(4 * H, H) * 0.5, mk(4 * H), mk(4 * H)
    dl = [t.to(dev).requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    y, hT, cT = ops.lstm_layer(*dl)
    cl = [t.clone().requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    yr, hr, cr = O.lstm_layer(*cl)
    assert rel(y, yr) < 1e-5 and rel(hT, hr) < 1e-5 and rel(cT, cr) < 1e-5
    gy, gh, gc = mk(T, B, H), mk(B, H), mk(B, H)
    ((y * gy.to(dev)).sum() + (hT * gh.to(dev)).sum() + (cT * gc.to(dev)).sum()).backward()
    ((yr * gy).sum() + (hr * gh).sum() + (cr * gc).sum()).backward()
    for a, b, name in zip(dl, cl, "x h0 c0 w_ih w_hh b_ih b_hh".split()):
        assert rel(a.grad, b.grad) < 5e-5, name


@pytest.mark.parametrize("T,B,E,H", [(5, 20, 40, 200), (4, 7, 72, 72), (3, 33, 24, 650), (6, 2, 100, 100), (2, 64, 16, 65)])
@pytest.mark.parametrize("with_noise", [False, True])
def test_lstm_layer_pads_a_hidden_size_that_is_not_a_multiple_of_32(dev, monkeypatch, T, B, E, H, with_noise):
    """The classic word-language-model sizes (200, 650, 1500; train.py's default is 200) do not fit the fused step kernels' tiles.
    ops.lstm_layer zero-pads them to the next multiple of 32 (padded units: zero weights and bias, so cell and output stay 0 and feed
    nothing back) and slices the padding off again: the fused kernels must be the ones that run, and outputs, final states and
    every gradient -- of the UNPADDED parameters -- equal the oracle's; with VLSTMCell's per-step noise rows too."""
    ops = ops_mod()
    g = torch.Generator().manual_seed(13)
    mk = lambda *s: torch.randn(*s, generator=g) * 0.2  # noqa: E731
    x, h0, c0 = mk(T, B, E), mk(B, H), mk(B, H)
    w_ih, w_hh, b_ih, b_hh = mk(4 * H, E), mk(4 * H, H) * 0.5, mk(4 * H), mk(4 * H)
    rows = mk(T, H) if with_noise else None
    seen = []
    real = ops._LSTMLayer.apply
    monkeypatch.setattr(ops._LSTMLayer, "apply", staticmethod(lambda *a: (seen.append(tuple(a[4].shape)), real(*a))[1]))
    dl = [t.to(dev).requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    dr = rows.to(dev).requires_grad_(True) if with_noise else None
    y, hT, cT = ops.lstm_layer(*dl, dr)
    Hp = (H + 31) // 32 * 32
    assert seen == [(4 * Hp, Hp)] and y.shape == (T, B, H) and hT.shape == (B, H)   # the padded recurrent weight reached the layer
    if with_noise:  # reference: the unpadded layer (skinny GEMM + cell kernel + row add per step), itself pinned by the variational_rnn fixtures
        monkeypatch.setattr(ops, "_PAD_HIDDEN_FROM", 1 << 30)
        cl = [t.to(dev).requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
        cr_ = rows.to(dev).requires_grad_(True)
        yr, hr, cr = ops.lstm_layer(*cl, cr_)
        assert seen[-1] == (4 * H, H)
        to = lambda t: t.to(dev)  # noqa: E731
    else:
        cl = [t.clone().requires_grad_(True) for t in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
        cr_ = None
        yr, hr, cr = O.lstm_layer(*cl)
        to = lambda t: t  # noqa: E731
    assert rel(y, yr) < 1e-5 and rel(hT, hr) < 1e-5 and rel(cT, cr) < 1e-5
    gy, gh, gc = mk(T, B, H), mk(B, H), mk(B, H)
    ((y * gy.to(dev)).sum() + (hT * gh.to(dev)).sum() + (cT * gc.to(dev)).sum()).backward()
    ((yr * to(gy)).sum() + (hr * to(gh)).sum() + (cr * to(gc)).sum()).backward()
    for a, b, name in zip(dl + ([dr] if with_noise else []), cl + ([cr_] if with_noise else []), "x h0 c0 w_ih w_hh b_ih b_hh rows".split()):
        assert a.grad.shape == b.grad.shape and rel(a.grad, b.grad) < 5e-5, name


def test_lstm_step_fwd_is_deterministic_and_rejects_bad_shapes(dev):
    lib = L().lib()
    from bayeslms_amd._lib import ptr, stream, ERR_UNSUPPORTED
    B, H = 64, 1024
    g = torch.Generator(device=dev).manual_seed(3)
    xw = torch.randn(B, 4 * H, device=dev, generator=g)
    w = torch.randn(4 * H, H, device=dev, generator=g) * 0.03
    hp, cp = torch.randn(B, H, device=dev, generator=g), torch.randn(B, H, device=dev, generator=g)
    outs = []
    for _ in range(2):
        h, c, ga = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, 4 * H, device=dev)
        assert lib.blm_lstm_step_fwd(ptr(xw), ptr(w), ptr(hp), ptr(cp), ptr(h), ptr(c), ptr(ga), None, B, H, stream()) == 0
        outs.append((h, c, ga))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    gates = xw.double() + hp.double() @ w.double().t()
    i, f, gg, o = gates.split(H, dim=1)
    cn = torch.sigmoid(f) * cp.double() + torch.sigmoid(i) * torch.tanh(gg)
    hn = torch.sigmoid(o) * torch.tanh(cn)
    assert rel(outs[0][0], hn.float()) < 1e-5 and rel(outs[0][1], cn.float()) < 1e-5
    # H not a multiple of 32: refused, nothing launched
    assert lib.blm_lstm_step_fwd(ptr(xw), ptr(w), ptr(hp), ptr(cp), ptr(h), ptr(c), ptr(ga), None, 4, 40, stream()) == ERR_UNSUPPORTED


def test_lstm_step_bwd_full_size_matches_composition(dev):
    """cfg2 shape (B = 64, H = 1024): the fused backward step == blm_gemm (fp64 reference here) followed
    by the cell backward kernel; run twice: bit-identical (fixed summation order, no atomics)."""
    lib = L().lib()
    from bayeslms_amd._lib import ptr, stream
    B, H = 64, 1024
    g = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    dg_t, w = rn(B, 4 * H) * 0.1, rn(4 * H, H) * 0.03
    dy, dcn, cp, c = rn(B, H) * 0.1, rn(B, H) * 0.1, rn(B, H), rn(B, H)
    ga = torch.sigmoid(rn(B, 4 * H))
    ga[:, 2 * H:3 * H] = torch.tanh(rn(B, H))
    w_t = torch.empty(H, 4 * H, device=dev)
    assert lib.blm_transpose(ptr(w), ptr(w_t), 4 * H, H, stream()) == 0
    assert torch.equal(w_t, w.t().contiguous())
    outs = []
    for _ in range(2):
        dgo, dcp, dh = torch.empty(B, 4 * H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
        assert lib.blm_lstm_step_bwd(ptr(dg_t), ptr(w_t), ptr(dy), ptr(dcn), ptr(cp), ptr(c), ptr(ga), ptr(dgo), ptr(dcp),
                                     ptr(dh), B, H, stream()) == 0
        outs.append((dgo, dcp, dh))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    dh_ref = (dg_t.double() @ w.double()).float()
    assert rel(outs[0][2], dh_ref) < 1e-5
    dgo_ref, dcp_ref = torch.empty(B, 4 * H, device=dev), torch.empty(B, H, device=dev)
    assert lib.blm_lstm_cell_bwd2(ptr(dh_ref), ptr(dy), ptr(dcn), ptr(cp), ptr(c), ptr(ga), ptr(dgo_ref), ptr(dcp_ref), B, H,
                                  stream()) == 0
    assert rel(outs[0][0], dgo_ref) < 1e-5 and rel(outs[0][1], dcp_ref) < 1e-5
    # dh-only mode (last step of a layer)
    dh2 = torch.empty(B, H, device=dev)
    assert lib.blm_lstm_step_bwd(ptr(dg_t), ptr(w_t), None, None, None, None, None, None, None, ptr(dh2), B, H, stream()) == 0
    assert torch.equal(dh2, outs[0][2])


@pytest.mark.parametrize("T,N,nhead,fused", [(17, 9, 8, True), (32, 5, 4, False), (1, 3, 8, True), (40, 4, 8, True)])
def test_attention_on_packed_rows_equals_scatter_attend_gather(dev, T, N, nhead, fused):
    """ops.packed_tokens.attention: the attention core on the REAL tokens' rows of a padded (T, N) batch of hypotheses
    (blm_attn_fwd_rows: the kernel finds a token's row through rowmap) == scatter into the padded layout, ordinary attention,
    gather -- bit for bit (same kernel, same operands); T > 32 takes that fallback by itself."""
    ops = ops_mod()
    g = torch.Generator().manual_seed(T * 7 + N)
    d = nhead * 64
    lens = torch.randint(1, T + 1, (N,), generator=g)
    lens[0] = T
    sel = torch.tensor([t * N + n for n in range(N) for t in range(int(lens[n]))], dtype=torch.int64)
    sel = sel[torch.randperm(sel.numel(), generator=g)].to(dev)  # the caller's row order is arbitrary
    R = sel.numel()
    with torch.no_grad():
        if fused:
            qkv = (torch.randn(R, 1, 3 * d, generator=g) * 0.5).to(dev)
            args = (qkv, None, None)
        else:
            args = tuple((torch.randn(R, 1, d, generator=g) * 0.5).to(dev) for _ in range(3))
        with ops.packed_tokens(sel, T, N) as pk:
            got = pk.attention(*args, nhead)
            took_rows = pk._rows_ok
            if fused:
                want = pk.pack(ops.attention(pk.unpack(args[0]), nhead))
            else:
                want = pk.pack(ops.attention_qkv(pk.unpack(args[0]), pk.unpack(args[1]), pk.unpack(args[2]), nhead))
    assert took_rows == (T <= 32)
    assert got.shape == (R, 1, d) and torch.equal(got, want)


def test_init_multi_and_colsum2(dev):
    """blm_init_multi: dst = (src or 0) + (src2 or 0) for up to eight vectors per launch (vector and scalar paths, more than eight
    split by the host, in-place sums, one buffer twice = two launches); blm_colsum2: the same column sums into two vectors."""
    ops = ops_mod()
    lib = L().lib()
    from bayeslms_amd._lib import BayesLMError, ptr, stream
    g = torch.Generator(device=dev).manual_seed(3)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    sizes = [4096, 20 * 1024, 7, 1, 33, 4 * 1024, 12, 64, 5, 2048, 3]
    srcs = [rn(n) for n in sizes]
    src2 = [rn(n) if i % 3 == 0 else None for i, n in enumerate(sizes)]
    dsts = [torch.full((n,), 9.0, device=dev) for n in sizes]
    items = []
    for i, n in enumerate(sizes):
        items.append((dsts[i], None if i % 4 == 1 else srcs[i], src2[i]))
    ops._init_multi(items)
    for i, n in enumerate(sizes):
        want = (torch.zeros(n, device=dev) if i % 4 == 1 else srcs[i]) + (src2[i] if src2[i] is not None else 0)
        assert torch.equal(dsts[i], want), i
    # a non-contiguous source is converted, an unaligned view takes the scalar path
    base = rn(10, 6)
    d = torch.empty(10, device=dev)
    big = torch.empty(4097, device=dev)
    ops._init_multi([(d, base[:, 2], None), (big[1:], srcs[0], None)])
    assert torch.equal(d, base[:, 2]) and torch.equal(big[1:], srcs[0])
    with pytest.raises(BayesLMError):
        ops._init_multi([(d, srcs[0], None)])
    # in place (grad += db), and the same buffer twice through ops._bias_pair_grads: two launches, both applied
    b = torch.nn.Parameter(rn(64))
    b.grad = rn(64)
    g0, db = b.grad.clone(), rn(64)
    assert ops._bias_pair_grads((b, b), [db], [True, True]) == [None, None]
    assert torch.allclose(b.grad, g0 + 2 * db, rtol=0, atol=1e-6)
    nb = b * 2.0  # a non-leaf bias (a sampled one) gets db itself
    out = ops._bias_pair_grads((b, nb), [db], [False, True])
    assert out[0] is None and out[1] is db
    # nine destinations and a NULL in the middle through the C entry point's own checks
    assert lib.blm_init_multi(9, None, None, None, None, stream()) != 0
    # colsum2
    x = rn(300, 130)
    o1, o2 = torch.full((130,), 5.0, device=dev), torch.full((130,), 7.0, device=dev)
    assert lib.blm_colsum2(ptr(x), 130, ptr(o1), ptr(o2), 300, 130, 0, stream()) == 0
    ref = x.double().sum(0).float()
    assert rel(o1, ref) < 1e-5 and rel(o2, ref) < 1e-5
    assert lib.blm_colsum2(ptr(x), 130, ptr(o1), ptr(o2), 300, 130, 1, stream()) == 0
    assert rel(o1, 2 * ref) < 1e-5 and rel(o2, 2 * ref) < 1e-5
    assert lib.blm_colsum2(ptr(x), 130, ptr(o1), ptr(o1), 300, 130, 1, stream()) != 0  # the same vector twice is refused


@pytest.mark.parametrize("T,B,H", [(7, 20, 64), (5, 3, 1024)])
def test_lstm_seq_bwd_is_the_per_step_chain(dev, T, B, H):
    """blm_lstm_seq_bwd (all backward steps of a layer from one call; two chunked calls as the layer wavefront issues them) ==
    blm_lstm_cell_bwd2 for step T-1 + one blm_lstm_step_bwd per earlier step, bit for bit, incl. the per-step dh rows."""
    lib = L().lib()
    from bayeslms_amd._lib import ptr, stream
    g = torch.Generator(device=dev).manual_seed(T * 100 + B)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    w_t = (rn(H, 4 * H) * 0.05).contiguous()
    dy, dh_T, cs = rn(T, B, H) * 0.1, rn(B, H) * 0.1, rn(T + 1, B, H)
    ga = torch.sigmoid(rn(T, B, 4 * H))
    ga[:, :, 2 * H:3 * H] = torch.tanh(rn(T, B, H))
    dc0 = rn(B, H) * 0.1
    # per-step reference chain
    dg_r, dcs_r, dhr_r = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(2, B, H, device=dev), torch.zeros(T, B, H, device=dev)
    dcs_r[0].copy_(dc0)
    k = 0
    for t in range(T - 1, -1, -1):
        if t == T - 1:
            rc = lib.blm_lstm_cell_bwd2(ptr(dh_T), ptr(dy[t]), ptr(dcs_r[k]), ptr(cs[t]), ptr(cs[t + 1]), ptr(ga[t]), ptr(dg_r[t]),
                                        ptr(dcs_r[k ^ 1]), B, H, stream())
        else:
            rc = lib.blm_lstm_step_bwd(ptr(dg_r[t + 1]), ptr(w_t), ptr(dy[t]), ptr(dcs_r[k]), ptr(cs[t]), ptr(cs[t + 1]), ptr(ga[t]),
                                       ptr(dg_r[t]), ptr(dcs_r[k ^ 1]), ptr(dhr_r[t]), B, H, stream())
        assert rc == 0
        k ^= 1
    for cuts in ([T, 0], [T, T // 2, 0], [T, T - 1, 1, 0]):
        dg, dcs, dhr = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(2, B, H, device=dev), torch.zeros(T, B, H, device=dev)
        dcs[0].copy_(dc0)
        kk = 0
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            assert lib.blm_lstm_seq_bwd(ptr(dh_T), ptr(dy), ptr(cs), ptr(ga), ptr(w_t), ptr(dg), ptr(dcs), kk, ptr(dhr), T, hi, lo,
                                        B, H, stream()) == 0
            kk ^= (hi - lo) & 1
        torch.cuda.synchronize()
        assert kk == k and torch.equal(dg, dg_r) and torch.equal(dcs[kk], dcs_r[k]) and torch.equal(dhr, dhr_r), cuts
    # argument checks: a chunk outside [0, T], a missing dh_T for the top chunk
    assert lib.blm_lstm_seq_bwd(ptr(dh_T), ptr(dy), ptr(cs), ptr(ga), ptr(w_t), ptr(dg), ptr(dcs), 0, None, T, T + 1, 0, B, H, stream()) != 0
    assert lib.blm_lstm_seq_bwd(None, ptr(dy), ptr(cs), ptr(ga), ptr(w_t), ptr(dg), ptr(dcs), 0, None, T, T, 0, B, H, stream()) != 0
    assert lib.blm_lstm_seq_bwd(None, ptr(dy), ptr(cs), ptr(ga), ptr(w_t), ptr(dg), ptr(dcs), 0, None, T, T - 1, 0, B, H, stream()) == 0


@pytest.mark.parametrize("M,V", [(7, 50), (33, 33000), (5, 1001)])
def test_ce_interp_matches_torch(dev, M, V):
    """two-model scoring: NLL of alpha*a + (1-alpha)*b (reference scorer :157-168), odd V = scalar tail"""
    ops = ops_mod()
    g = torch.Generator().manual_seed(M + V)
    a, b = torch.randn(M, V, generator=g) * 3, torch.randn(M, V, generator=g) * 3
    tgt = torch.randint(0, V, (M,), generator=g)
    for alpha in (0.0, 0.3, 1.0):
        ref = torch.nn.functional.cross_entropy((alpha * a + (1 - alpha) * b).double(), tgt, reduction="none")
        mean, nll = ops.cross_entropy_interp(a.to(dev), b.to(dev), alpha, tgt.to(dev))
        assert rel(nll, ref.float()) < 1e-5
        assert abs(float(mean) - float(ref.mean())) < 1e-4 * max(1.0, float(ref.mean()))


# ------------------------------------------------------------------ full-size, size-independent properties
def test_sampled_gemm_full_size_properties(dev):
    """cfg3 shape (M=8192, N=512, K=4096): fused-in-loader sampling == materialise-then-GEMM (same
    Philox counters), linearity in X, and a row subset against fp64 on the CPU."""
    ops, lib = ops_mod(), L()
    M, N, K = 8192, 512, 4096
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn(M, K, device=dev, generator=g)
    mu = torch.randn(N, K, device=dev, generator=g) * 0.05
    lg = torch.rand(N, K, device=dev, generator=g) * 2 - 5
    noise = ops.NoiseSpec(None, 1111, 16, 3)
    y_f = ops.bayes_linear(X, mu, lg, noise, 0.0, True)
    y_m = ops.bayes_linear(X, mu, lg, noise, 0.0, False)
    assert rel(y_f, y_m) < 1e-5  # split-K partial sums (materialised path) vs one chain (fused path)
    y2 = ops.bayes_linear(2.0 * X, mu, lg, noise, 0.0, True)
    assert rel(y2, 2.0 * y_f) < 1e-5
    W = ops.sample_weight(mu, lg, noise)
    rows = torch.tensor([0, 1, 127, 128, 4095, 8191])
    ref = X[rows.to(dev)].double().cpu() @ W.double().cpu().t()
    assert rel(y_f[rows.to(dev)], ref) < 1e-5
    # a different step draws different noise; the same step is reproducible bit for bit
    assert torch.equal(ops.bayes_linear(X, mu, lg, noise, 0.0, True), y_f)
    assert not torch.equal(ops.bayes_linear(X, mu, lg, ops.NoiseSpec(None, 1111, 16, 4), 0.0, True), y_f)


@pytest.mark.parametrize("shape", [(8192, 512, 4096), (2240, 1024, 1024), (512, 4096, 8192), (384, 256, 160)])
def test_gemm_split_bf16_modes_opt_in(dev, shape):
    """OPT-IN split-bf16 arithmetic (never the default): all three layouts against fp64.  bf16x3 (two parts) within
    2e-5 of the output scale; bf16x6 (three parts = the exact 24-bit mantissa) within 1.5x of the fp32 MFMA mode's
    own error (which is 3e-7 .. 3e-6: sequential fp32 accumulation over K); the switch back restores fp32."""
    from bayeslms_amd import _lib as L, ops
    M, N, K = shape
    torch.manual_seed(11)
    if ops.get_gemm_mode() != "f32":
        pytest.skip("suite is running under a BLM_GEMM_MODE override")
    for op in (L.GEMM_NT, L.GEMM_NN, L.GEMM_TN):
        if op == L.GEMM_NT:
            A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
            ref = A.double() @ B.double().t()
            lda, ldb = K, K
        elif op == L.GEMM_NN:
            A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) / K ** 0.5
            ref = A.double() @ B.double()
            lda, ldb = K, N
        else:
            A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev) / K ** 0.5
            ref = A.double().t() @ B.double()
            lda, ldb = M, N
        scale = float(ref.abs().max())

        def run(mode):
            try:
                ops.set_gemm_mode(mode)
                c = torch.empty(M, N, device=dev)
                ops.gemm(op, A, B, c, M, N, K, lda, ldb, N)
            finally:
                ops.set_gemm_mode("f32")
            return c, float((c.double() - ref).abs().max()) / scale
        c32, e32 = run("f32")
        c3, e3 = run("bf16x3")
        c6, e6 = run("bf16x6")
        assert e32 < 1e-5 and e3 < 2e-5 and e6 < 1.5 * e32 + 2e-7, (op, e32, e3, e6)
        again, _ = run("f32")
        # back in fp32 mode (split-K launches add their slices with float atomics: equal up to summation order)
        assert float((again.double() - c32.double()).abs().max()) / scale < 1e-6 and ops.get_gemm_mode() == "f32"


def test_gemm_bf16x6_is_exact_on_exactly_representable_products(dev):
    """Three bf16 parts hold an fp32 value exactly: a product with an identity matrix returns the input bit for bit
    in bf16x6 mode (bf16x3 keeps 16 mantissa bits and does not)."""
    from bayeslms_amd import _lib as L, ops
    if ops.get_gemm_mode() != "f32":
        pytest.skip("suite is running under a BLM_GEMM_MODE override")
    torch.manual_seed(12)
    M, K = 256, 256
    A = torch.randn(M, K, device=dev)
    eye = torch.eye(K, device=dev)
    out = {}
    for mode in ("bf16x6", "bf16x3"):
        try:
            ops.set_gemm_mode(mode)
            c = torch.empty(M, K, device=dev)
            ops.gemm(L.GEMM_NT, A, eye, c, M, K, K, K, K, K)
        finally:
            ops.set_gemm_mode("f32")
        out[mode] = c
    assert torch.equal(out["bf16x6"], A)
    assert not torch.equal(out["bf16x3"], A) and float((out["bf16x3"] - A).abs().max()) < 1e-4


def test_attention_unsupported_head_dim_is_loud(dev):
    """head_dim above 512 is refused with a message instead of computing something else."""
    ops = ops_mod()
    from bayeslms_amd._lib import BayesLMError
    qkv = torch.randn(4, 1, 3 * 520, device=dev, requires_grad=True)
    with pytest.raises(BayesLMError, match="head_dim"):
        ops.attention(qkv, 1)


def test_lstm_wide_inference_batch_takes_the_gemm_path_with_the_same_result(dev):
    """No-grad forward at B >= 256 (the scorer's packed hypotheses): recurrent product through the tiled GEMM + cell kernel
    instead of the fused step kernel (ops._LSTMLayer) -- same states as the fused path (taken when gradients are on)."""
    ops = ops_mod()
    T, B, E, H = 7, 300, 48, 64
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)  # noqa: E731
    x, h0, c0 = mk(T, B, E), mk(B, H), mk(B, H)
    w = [mk(4 * H, E), mk(4 * H, H), mk(4 * H), mk(4 * H)]
    with torch.no_grad():
        y0, h1, c1 = ops.lstm_layer(x, h0, c0, *w)
    y1, h2, c2 = ops.lstm_layer(x.clone().requires_grad_(True), h0, c0, *w)
    for a, b in ((y0, y1), (h1, h2), (c1, c2)):
        assert rel(a, b.detach()) < 1e-5
    xr, hr, cr = x.cpu().double(), h0.cpu().double(), c0.cpu().double()
    wc = [t.cpu().double() for t in w]
    for t in range(T):
        z = xr[t] @ wc[0].t() + hr @ wc[1].t() + wc[2] + wc[3]
        i, f, gg, o = z.chunk(4, 1)
        cr = torch.sigmoid(f) * cr + torch.sigmoid(i) * torch.tanh(gg)
        hr = torch.sigmoid(o) * torch.tanh(cr)
    assert rel(h1, hr.float()) < 1e-5 and rel(c1, cr.float()) < 1e-5


@pytest.mark.parametrize("T,B,E,H,p", [(12, 5, 24, 32, 0.0), (17, 3, 16, 64, 0.3), (35, 64, 1024, 1024, 0.0), (40, 1, 32, 64, 0.0),
                                         (100, 32, 32, 64, 0.2), (36, 20, 48, 1024, 0.1)])
def test_lstm_stack2_wavefront_equals_two_sequential_layers(dev, T, B, E, H, p):
    """ops.lstm_stack2 (two layers as a wavefront on two streams, layer 2 one time chunk behind, inter-layer dropout
    applied chunk by chunk with the whole tensor's mask) runs the same kernels on the same operands as two
    ops.lstm_layer calls with ops.dropout between them: outputs, final states and gradients are equal to summation order
    (a chunk's input product and the whole window's may run under different tile / K-slice plans)."""
    ops = ops_mod()
    ops.set_lstm_wavefront(None)
    probe = torch.empty(T, B, E, device=dev)
    wa = torch.empty(4 * H, H, device=dev)
    # the measured rule: on by itself at the reference recipes' shape (seq_len 100, batch 32) and at BASELINE configs[0]
    # (seq_len 35, batch 20), off where the step kernels fill the chip and for small layers (H < 640: host-bound, the second stream costs more than it returns)
    assert ops.lstm_stack2_ok(probe, wa, wa, wa) == (B <= 32 and T >= 32 and H >= 640)
    ops.set_lstm_wavefront(True)
    g = torch.Generator().manual_seed(21)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)  # noqa: E731
    x, h0, c0 = mk(T, B, E), mk(2, B, H), mk(2, B, H)
    l1 = [mk(4 * H, E), mk(4 * H, H) * 0.5, mk(4 * H), mk(4 * H)]
    l2 = [mk(4 * H, H), mk(4 * H, H) * 0.5, mk(4 * H), mk(4 * H)]
    gy, gh, gc = mk(T, B, H), mk(2, B, H), mk(2, B, H)
    drop = ops.Drop(p, 77, 3, 5, 0, B) if p > 0 else ops.NO_DROP

    def run(wavefront):
        leaves = [t.clone().requires_grad_(True) for t in [x, h0, c0] + l1 + l2]
        xx, hh, cc = leaves[:3]
        a1, a2 = leaves[3:7], leaves[7:11]
        if wavefront:
            assert ops.lstm_stack2_ok(xx, a1[1], a2[1], a2[0])
            y, (h1, h2), (c1, c2) = ops.lstm_stack2(xx, hh, cc, a1, a2, drop)
        else:
            y1, h1, c1 = ops.lstm_layer(xx, hh[0], cc[0], *a1)
            y, h2, c2 = ops.lstm_layer(ops.dropout(y1, drop), hh[1], cc[1], *a2)
        loss = (y * gy).sum() + (h1 * gh[0]).sum() + (h2 * gh[1]).sum() + (c1 * gc[0]).sum() + (c2 * gc[1]).sum()
        loss.backward()
        torch.cuda.synchronize()
        return [y.detach(), h1.detach(), c1.detach(), h2.detach(), c2.detach()] + [t.grad for t in leaves]
    a, b = run(True), run(False)
    ops.set_lstm_wavefront(None)  # back to the measured rule
    names = "y h1 c1 h2 c2 dx dh0 dc0 dw_ih1 dw_hh1 db_ih1 db_hh1 dw_ih2 dw_hh2 db_ih2 db_hh2".split()
    for u, v, n in zip(a, b, names):
        # layer 1's dy comes out of a dgrad GEMM per chunk instead of one over all T (different tile / split-K summation
        # order); the batched wgrad GEMMs use float atomics; since the bias epilogue takes K slices, so may the input products
        # (a rounding difference in an input product is carried through the recurrence: 6e-6 on y after 35 steps x 2 layers at H = 1024)
        assert rel(u, v) < 2e-5, n


# ------------------------------------------------------------------ residual link
@pytest.mark.parametrize("branch", ["linear", "ffn", "ffn_gp"])
@pytest.mark.parametrize("case", ["block", "third_consumer", "armed_twice", "other_tensor"])
def test_residual_link_gradients_equal_the_unlinked_path(dev, branch, case):
    """out = LN(x + f(x)) with the branch's dgrad GEMM folding into the LayerNorm backward's dx (ops.ResidualLink) must
    give the same gradients as plain autograd -- also when x has a THIRD consumer (a tap on the layer input), when two
    ops arm one link, and when the LayerNorm's residual is not the tensor the branch read (the link then stands down)."""
    ops = ops_mod()
    torch.manual_seed(3)
    T, B, D, F_ = 6, 5, 32, 64
    x0 = torch.randn(T, B, D, device=dev)
    w1, b1 = torch.randn(F_, D, device=dev) * 0.2, torch.randn(F_, device=dev) * 0.1
    w2, b2 = torch.randn(D, F_, device=dev) * 0.2, torch.randn(D, device=dev) * 0.1
    wl, bl = torch.randn(D, D, device=dev) * 0.2, torch.randn(D, device=dev) * 0.1
    coef = torch.randn(4, F_, device=dev) * 0.5
    gamma, beta = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
    r = torch.randn(T, B, D, device=dev)
    tap = torch.randn(T, B, D, device=dev)

    def run(linked):
        ps = [t.clone().requires_grad_(True) for t in (x0, w1, b1, w2, b2, wl, bl, coef, gamma, beta)]
        x, W1, B1, W2, B2, WL, BL, CF, G, Bt = ps
        xs = x * 1.0  # non-leaf layer input, as inside a stack
        lk = ops.ResidualLink() if linked else None

        def f(inp, link):
            if branch == "linear":
                return ops.linear(inp, WL, BL, link)
            if branch == "ffn":
                return ops.ffn(inp, W1, B1, W2, B2, link=link)
            return ops.ffn_gp(inp, W1, B1, CF, W2, B2, link=link)
        y = f(xs, lk)
        res = xs
        if case == "armed_twice":
            y = y + f(xs, lk)
        if case == "other_tensor":
            res = xs * 2.0
        out = ops.add_dropout_ln(res, y, G, Bt, 1e-5, ops.NO_DROP, lk)
        loss = (out * r).sum()
        if case == "third_consumer":
            loss = loss + (xs * tap).sum() + (xs.tanh() * tap).sum()
        loss.backward()
        return [p.grad for p in ps]
    got, want = run(True), run(False)
    used = {"linear": (0, 5, 6, 8, 9), "ffn": (0, 1, 2, 3, 4, 8, 9), "ffn_gp": (0, 1, 2, 3, 4, 7, 8, 9)}[branch]
    for i in used:
        assert got[i] is not None and want[i] is not None, i
        assert rel(got[i], want[i]) < 2e-5, (i, rel(got[i], want[i]))
    # and against torch's own autograd for the plain block (fp32 CPU)
    if case == "third_consumer" and branch == "linear":
        xc = x0.cpu().clone().requires_grad_(True)
        xs = xc * 1.0
        out = torch.nn.functional.layer_norm(xs + torch.nn.functional.linear(xs, wl.cpu(), bl.cpu()), (D,), gamma.cpu(), beta.cpu())
        ((out * r.cpu()).sum() + (xs * tap.cpu()).sum() + (xs.tanh() * tap.cpu()).sum()).backward()
        assert rel(got[0], xc.grad) < 1e-4
