"""CPU: the launch planner of the fp32 MFMA GEMM (csrc/gemm_plan.hip) is host code -- every rule of it can be checked
without a GPU through blm_gemm_plan_query / _model_us / _override / _set / _clear (include/bayeslm.h)."""
import ctypes as C
import re

import pytest

from bayeslms_amd import _lib as L


def args(op, m, n, k, epi=L.EPI_NONE, acc=False, ldc=None, misaligned=False):
    a = L.GemmArgs()
    a.abi_version = L.ABI_VERSION
    a.op, a.M, a.N, a.K = op, m, n, k
    a.lda = m if op == L.GEMM_TN else k
    a.ldb = k if op == L.GEMM_NT else n
    a.ldc = n if ldc is None else ldc
    a.epilogue = epi
    a.flags = L.GEMM_ACCUMULATE if acc else 0
    a.A = 4 if misaligned else 0  # only inspected for alignment
    return a


def plan(a):
    out = L.GemmPlan()
    L.check(L.lib().blm_gemm_plan_query(C.byref(a), C.byref(out)), "blm_gemm_plan_query")
    return out.tile, out.splits, out.source, out.model_us


def launch_plan(a):
    """The planning step of a launch (blm_gemm does the same in front of its kernel): takes its time off an open comm window."""
    out = L.GemmPlan()
    L.check(L.lib().blm_gemm_plan_launch(C.byref(a), C.byref(out)), "blm_gemm_plan_launch")
    return out.tile, out.splits, out.source, out.model_us


@pytest.fixture(autouse=True)
def _clean():
    lib = L.lib()
    lib.blm_gemm_plan_override(0, 0)
    lib.blm_gemm_plan_clear(1)
    yield
    lib.blm_gemm_plan_override(0, 0)
    lib.blm_gemm_plan_clear(1)


def table_entries():
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayeslms_amd", "csrc", "gemm_plans.inc")
    return [tuple(int(v) for v in m.groups()) for m in re.finditer(r"^\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (-?\d+)\}", open(path).read(), re.M)]


def test_every_plan_is_legal_and_deterministic():
    dims = [1, 7, 64, 100, 512, 2240, 3200, 8192, 33000]
    for op in (L.GEMM_NT, L.GEMM_NN, L.GEMM_TN):
        for m in dims:
            for n in dims:
                for k in (1, 32, 500, 512, 4096, 33000):
                    for epi, acc in ((L.EPI_NONE, False), (L.EPI_NONE, True), (L.EPI_BIAS, False), (L.EPI_BIAS_GELU, False)):
                        a = args(op, m, n, k, epi, acc)
                        t, s, src, us = plan(a)
                        assert t in (11, 12, 21, 22, 28) and (1 <= s <= 16 or -32 <= s <= -2) and src in (0, 1) and us > 0
                        assert plan(a)[:2] == (t, s)
                        if epi not in (L.EPI_NONE, L.EPI_BIAS):
                            assert s == 1  # partial sums cannot pass through a non-linear epilogue (the bias rides on slice 0)
                        if abs(s) > 1:
                            assert k // abs(s) >= 32
                        fast = all(v % 4 == 0 and v >= 4 for v in ((m if op == L.GEMM_TN else k), (k if op == L.GEMM_NT else n)))
                        if not fast:
                            assert t == 11  # odd extents: only the guarded 64x64 kernel exists
                        if t == 28:
                            assert k % 32 == 0  # the eight-wave tile has no K-tail path


def test_tail_plans_are_accepted_clamped_and_modelled():
    lib = L.lib()
    a = args(L.GEMM_TN, 33000, 512, 8192, acc=True)
    L.check(lib.blm_gemm_plan_set(L.GEMM_TN, 33000, 512, 8192, 0, 1, 28, -32), "set")
    assert plan(a)[:3] == (28, -32, 1)                       # splits < 0: only the tail round is sliced
    us = C.c_float()
    L.check(lib.blm_gemm_plan_model_us(C.byref(a), 28, -32, C.byref(us)), "model")
    us6 = C.c_float()
    L.check(lib.blm_gemm_plan_model_us(C.byref(a), 28, 6, C.byref(us6)), "model")
    assert 0 < us.value < us6.value                          # 16 MB instead of 405 MB through atomics
    lib.blm_gemm_plan_clear(1)
    L.check(lib.blm_gemm_plan_override(0, -8), "override")
    assert plan(args(L.GEMM_NT, 8192, 512, 4096, L.EPI_BIAS_GELU))[1] == 1   # not splittable: tail plans are clamped too
    assert plan(args(L.GEMM_NN, 3000, 2900, 64))[1] == -2                     # K / slices >= 32
    assert lib.blm_gemm_plan_override(0, -1) != 0 and lib.blm_gemm_plan_set(0, 1, 1, 1, 0, 0, 11, 0) != 0


def test_split_k_needs_a_dense_or_accumulated_c_and_aligned_operands_get_all_tiles():
    assert plan(args(L.GEMM_NN, 64, 4096, 8192))[1] > 1                      # skinny output, long K: slices fill the chip
    assert plan(args(L.GEMM_NN, 64, 4096, 8192, ldc=8192))[1] == 1            # a strided C cannot be zeroed in one pass ...
    assert plan(args(L.GEMM_NN, 64, 4096, 8192, ldc=8192, acc=True))[1] > 1   # ... unless it is accumulated into
    assert plan(args(L.GEMM_NT, 64, 4096, 8192, L.EPI_BIAS))[1] > 1           # the bias epilogue is linear: slices allowed
    assert plan(args(L.GEMM_NT, 64, 4096, 8192, L.EPI_BIAS, ldc=8192))[1] == 1
    assert plan(args(L.GEMM_NT, 8192, 4096, 4096, misaligned=True))[0] == 11
    assert plan(args(L.GEMM_NT, 8192, 4096, 4096))[0] != 11                   # a big aligned product is not run on 64x64 tiles


def test_plan_table_override_and_runtime_entries():
    lib = L.lib()
    ent = table_entries()
    assert ent, "gemm_plans.inc holds the in-situ tuned shapes"
    for op, m, n, k, epi, acc, t, s in ent:
        a = args(op, m, n, k, epi, bool(acc))
        if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU):
            a.bias = 16
        got = plan(a)
        assert got[:3] == (t, s, 1), (op, m, n, k, got)
    op, m, n, k, epi, acc, t, s = ent[0]
    a = args(op, m, n, k, epi, bool(acc))
    lib.blm_gemm_plan_clear(0)  # cost model only
    assert plan(a)[2] == 0
    lib.blm_gemm_plan_clear(1)
    assert plan(a)[2] == 1
    L.check(lib.blm_gemm_plan_set(op, m, n, k, epi, acc, 22, 1), "set")  # a run-time entry supersedes the built-in one
    assert plan(a)[:3] == (22, 1, 1)
    L.check(lib.blm_gemm_plan_override(12, 0), "override")                   # the override beats both
    assert plan(a)[0] == 12 and plan(a)[2] == 2
    assert lib.blm_gemm_plan_override(13, 0) == L.ERR_INVALID and lib.blm_gemm_plan_set(0, 1, 1, 1, 0, 0, 22, 0) == L.ERR_INVALID
    lib.blm_gemm_plan_override(0, 0)
    lib.blm_gemm_plan_clear(1)
    assert plan(a)[:3] == (t, s, 1)


def test_cost_model_shape():
    """Round-fill x per-tile efficiency: time grows with K, a second (partial) round costs, the model is finite everywhere."""
    lib = L.lib()

    def us(a, tile, s):
        v = C.c_float()
        L.check(lib.blm_gemm_plan_model_us(C.byref(a), tile, s, C.byref(v)), "model_us")
        return v.value
    a1, a2 = args(L.GEMM_NT, 8192, 512, 2048), args(L.GEMM_NT, 8192, 512, 4096)
    for t in (11, 12, 21, 22):
        assert 1.5 < us(a2, t, 1) / us(a1, t, 1) < 2.1
    full, over = args(L.GEMM_NT, 128 * 32, 128 * 16, 4096), args(L.GEMM_NT, 128 * 33, 128 * 16, 4096)  # 512 / 528 tiles of 128x128
    assert us(over, 22, 1) > 1.2 * us(full, 22, 1)  # 16 tiles spill into a second round of the 512 workgroup slots
    lib.blm_gemm_plan_clear(0)
    # the fitted model reproduces the measured preferences the round-2 rules were written around
    assert plan(args(L.GEMM_NT, 8192, 512, 4096))[0] in (21, 12, 11)          # not 128x128: 256 tiles = one wave per SIMD
    assert plan(args(L.GEMM_TN, 512, 512, 8192, acc=True))[1] >= 8            # o_net weight gradient: 16 / 64 tiles need K slices
    assert plan(args(L.GEMM_NN, 2240, 1024, 33000))[1] >= 4                   # decoder input gradient at M = 2240
    lib.blm_gemm_plan_clear(1)


def test_planner_counts_on_the_compute_units_it_is_given():
    """blm_gemm_plan_set_cus(n): data-parallel training narrows the planner while gradient buckets are in flight (RCCL's
    channel workgroups hold CUs beside the backward GEMMs).  The roofline launch -- NT 8192 x 512 x 4096 -- is exactly ONE
    round of 256 one-per-CU eight-wave workgroups on the whole chip (table entry); with 8 CUs taken that plan would spill a
    second round of 8 workgroups, so the planner must leave it: table off, cost model on 248 CUs, no one-per-CU tile."""
    lib = L.lib()
    a = args(L.GEMM_NT, 8192, 512, 4096)
    assert lib.blm_gemm_plan_get_cus() == 256
    t256, s256, src256, us256 = plan(a)
    assert (t256, src256) == (28, 1)
    try:
        for cus in (248, 240, 224, 192):
            L.check(lib.blm_gemm_plan_set_cus(cus), "set_cus")
            assert lib.blm_gemm_plan_get_cus() == cus
            t, s, src, us = plan(a)
            assert src == 0, "the plan table was measured on the whole chip"
            assert t != 28 or s > 1, "one workgroup per CU on every CU of the whole chip does not fit a narrowed one"
            # the model prices the whole-chip plan on the narrowed chip as the two rounds it would be, and finds a plan whose
            # cost stays near the lost share of the chip (equal workgroups quantise: 512 on 248 CUs means 3 on some)
            us28 = C.c_float()
            L.check(lib.blm_gemm_plan_model_us(C.byref(a), 28, 1, C.byref(us28)), "model_us")
            assert us28.value > 1.3 * us, (cus, us28.value, us)
            assert us < us256 * (256.0 / cus) * 1.2, (cus, us, us256)
        L.check(lib.blm_gemm_plan_set_cus(0), "set_cus")
        assert lib.blm_gemm_plan_get_cus() == 256 and plan(a)[:3] == (t256, s256, src256)
        assert lib.blm_gemm_plan_set_cus(7) != 0 and lib.blm_gemm_plan_set_cus(257) != 0
        assert lib.blm_gemm_plan_get_cus() == 256
    finally:
        lib.blm_gemm_plan_set_cus(0)


def test_tail_sliced_plans_count_their_rounds_with_the_narrowed_chip():
    """A tail plan slices only the tiles beyond the last WHOLE round of workgroup slots; the number of slots follows the CUs
    the plan was made for (Plan.cus travels to the launcher), so every plan stays legal on a narrowed chip."""
    lib = L.lib()
    try:
        for cus in (256, 248, 200):
            L.check(lib.blm_gemm_plan_set_cus(cus), "set_cus")
            for (op, m, n, k, epi, acc, tile, splits) in table_entries():
                t, s, src, us = plan(args(op, m, n, k, epi, bool(acc)))
                assert t in (11, 12, 21, 22, 28) and (1 <= s <= 16 or -32 <= s <= -2) and us > 0
                assert src == (1 if cus == 256 else 0)
    finally:
        lib.blm_gemm_plan_set_cus(0)


def test_plans_are_memoised_per_cu_count_and_invalidated_by_table_changes():
    lib = L.lib()
    a = args(L.GEMM_NT, 1234, 520, 768)
    base = plan(a)[:2]
    L.check(lib.blm_gemm_plan_set(L.GEMM_NT, 1234, 520, 768, L.EPI_NONE, 0, 11, 2), "set")
    assert plan(a)[:3] == (11, 2, 1)  # a cached model plan must not survive a new table entry
    lib.blm_gemm_plan_clear(1)
    assert plan(a)[:2] == base
    L.check(lib.blm_gemm_plan_override(12, 0), "override")
    assert plan(a)[0] == 12
    lib.blm_gemm_plan_override(0, 0)
    assert plan(a)[:2] == base


def test_comm_window_is_kept_in_modelled_device_time():
    """blm_gemm_plan_comm_window(us): every plan made while the window is open takes its modelled time off it; inside the
    window a shape listed in the comm table (csrc/gemm_plans_comm.inc or blm_gemm_plan_set_comm) takes that plan (source 3),
    every other shape keeps its whole-chip plan; 0 closes the window."""
    lib = L.lib()
    a = args(L.GEMM_TN, 4096, 512, 8192, acc=True)   # FFN linear1 weight gradient: in the built-in comm table
    b = args(L.GEMM_NT, 8192, 512, 4096)             # the roofline launch: not in it
    whole_a, whole_b = plan(a), plan(b)
    assert whole_a[2] == 1 and whole_b[:3] == (28, 1, 1)
    assert lib.blm_gemm_plan_comm_window_left() == 0.0
    L.check(lib.blm_gemm_plan_comm_window(C.c_float(1000.0)), "window")
    try:
        left0 = lib.blm_gemm_plan_comm_window_left()
        assert plan(a)[2] == 3 and plan(b)[:3] == whole_b[:3]
        assert lib.blm_gemm_plan_comm_window_left() == left0    # a query is not a launch -- the window keeps its time
        under_a = launch_plan(a)
        left1 = lib.blm_gemm_plan_comm_window_left()
        assert under_a[2] == 3 and under_a[:2] != whole_a[:2] and under_a[0] != 28
        assert abs((left0 - left1) - under_a[3]) < 1e-2 * under_a[3]
        assert launch_plan(b)[:3] == whole_b[:3]                # no comm entry: the whole-chip plan, and its time comes off
        assert lib.blm_gemm_plan_comm_window_left() < left1
        L.check(lib.blm_gemm_plan_set_comm(L.GEMM_NT, 8192, 512, 4096, L.EPI_NONE, 0, 11, 4), "set_comm")
        assert plan(b)[:3] == (11, 4, 3)
        for _ in range(64):                                     # the window runs out by itself
            launch_plan(b)
        assert lib.blm_gemm_plan_comm_window_left() == 0.0 and plan(b)[:3] == whole_b[:3]
        L.check(lib.blm_gemm_plan_comm_window(C.c_float(500.0)), "window")
        L.check(lib.blm_gemm_plan_comm_window(C.c_float(0.0)), "window")
        assert lib.blm_gemm_plan_comm_window_left() == 0.0 and plan(a)[:3] == whole_a[:3]
    finally:
        lib.blm_gemm_plan_comm_window(C.c_float(0.0))
        lib.blm_gemm_plan_clear(1)


def test_deterministic_mode_legalises_every_plan_to_one_k_slice():
    """blm_set_option("deterministic", 1): partial sums never meet through float atomics -- table plans (K-sliced and tail-sliced),
    overrides, run-time entries and the cost model's picks all come back with splits == 1; switching it off restores them
    (the memo is keyed by can_split, nothing has to be cleared)."""
    lib = L.lib()
    sliced = [e for e in table_entries() if e[7] != 1]
    assert sliced, "the built-in table has K-sliced and tail-sliced plans"
    shapes = [args(e[0], e[1], e[2], e[3], e[4], bool(e[5])) for e in sliced]
    shapes.append(args(L.GEMM_TN, 1024, 1024, 70000, acc=True))  # the cost model slices this one
    before = [plan(a) for a in shapes]
    assert any(abs(b[1]) > 1 for b in before)
    L.check(lib.blm_set_option(b"deterministic", 1), "set_option")
    try:
        assert all(plan(a)[1] == 1 for a in shapes)
        L.check(lib.blm_gemm_plan_override(0, 8), "override")
        assert all(plan(a)[1] == 1 for a in shapes)
        L.check(lib.blm_gemm_plan_override(0, 0), "override")
        L.check(lib.blm_gemm_plan_set(L.GEMM_NT, 320, 4096, 1024, L.EPI_NONE, 0, 11, -4), "set")
        assert plan(args(L.GEMM_NT, 320, 4096, 1024))[1] == 1
    finally:
        L.check(lib.blm_set_option(b"deterministic", 0), "set_option")
    assert [plan(a)[:2] for a in shapes] == [b[:2] for b in before]
