#!/usr/bin/env python3
"""Child process of tests/test_sanitizer_cpu.py (run under LD_PRELOAD of the ASan runtime with BLM_LIB = the host-only
AddressSanitizer + UBSan build of the library; no GPU involved, a launch that is reached returns BLM_ERR_HIP).

    sanitizer_sweep.py sweep    every blm_* entry point of include/bayeslm.h with NULL / misaligned / negative / zero / huge
                                arguments, driven by the ctypes prototypes: each call must RETURN (a status, a size or a
                                version) -- never crash, never trip a sanitizer
    sanitizer_sweep.py hammer   4 threads on the planner's shared state (choose_plan's memo, the run-time tables, the
                                override, the comm window, the option registry) for a few seconds
"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd import _lib as L  # noqa: E402

# `void*` parameters the HOST reads (arrays copied into the launch): they get real host arrays, never a made-up address
HOST_TABLES = {"blm_init_multi": (1, 2, 3, 4)}
BOGUS = 0x10004  # a device-looking, 4-byte-aligned (not 16-byte-aligned) address: the host must never dereference it


def struct_of(ptr_type, variant):
    t = ptr_type._type_
    s = t()
    if hasattr(s, "abi_version"):
        s.abi_version = L.ABI_VERSION if variant != "bad_abi" else 77
    if variant in ("small", "bad_abi"):
        for name, ft in t._fields_:
            if name == "abi_version":
                continue
            if ft in (C.c_int32, C.c_int64, C.c_uint32):
                setattr(s, name, 8)
            elif ft is C.c_void_p:
                setattr(s, name, BOGUS)
            elif ft is C.c_float:
                setattr(s, name, 0.5)
    elif variant == "negative":
        for name, ft in t._fields_:
            if ft in (C.c_int32, C.c_int64) and name != "abi_version":
                setattr(s, name, -3)
    elif variant in ("huge", "large"):
        for name, ft in t._fields_:
            if name == "abi_version":
                continue
            if ft is C.c_int32:
                setattr(s, name, 2 ** 31 - 1 if variant == "huge" else 70000)
            elif ft is C.c_int64:
                setattr(s, name, 2 ** 40 if variant == "huge" else 2 ** 20)
            elif ft is C.c_void_p:
                setattr(s, name, BOGUS)
    return s


def build(name, argtypes, variant, keep):
    out = []
    for i, t in enumerate(argtypes):
        host = i in HOST_TABLES.get(name, ())
        if t is C.c_void_p:
            if host:
                arr = (C.c_int64 * 8)(*([0] * 8 if variant in ("null", "negative") else [BOGUS if i < 4 else 16] * 8))
                if i == 4:
                    arr = (C.c_int64 * 8)(*([-1] * 8 if variant == "negative" else [16] * 8))
                keep.append(arr)
                out.append(C.cast(arr, C.c_void_p))
            else:
                out.append(None if variant in ("null", "negative", "null_pos") else BOGUS)
        elif t is C.c_char_p:
            out.append(None if variant in ("null", "negative") else b"no_such_option")
        elif isinstance(t, type) and issubclass(t, C._Pointer):
            if variant in ("null", "null_pos"):
                out.append(None)
            elif issubclass(t._type_, C.Structure):
                s = struct_of(t, variant)
                keep.append(s)
                out.append(C.byref(s))
            else:
                v = t._type_()
                keep.append(v)
                out.append(C.byref(v))
        elif t in (C.c_int, C.c_int64):
            out.append({"null": 1, "null_pos": 4, "negative": -1, "zero": 0, "small": 8, "bad_abi": 8,
                        "huge": (2 ** 31 - 1) if t is C.c_int else 2 ** 40,
                        "large": 70000 if t is C.c_int else 2 ** 20}[variant])  # 70000^2 overflows an int, 70000^3 a sane tensor
        elif t is C.c_float:
            out.append({"negative": -1.0, "huge": 3e38}.get(variant, 0.5))
        else:
            raise SystemExit("sweep: unhandled argument type %r of %s" % (t, name))
    return out


def sweep():
    lib = L.lib()
    n_calls, by_status = 0, {}
    for name, (res, argtypes) in sorted(L.SIGNATURES.items()):
        fn = getattr(lib, name)
        for variant in ("null", "null_pos", "negative", "zero", "small", "bad_abi", "large", "huge"):
            keep = []
            args = build(name, argtypes, variant, keep)
            rc = fn(*args)
            n_calls += 1
            if res is C.c_int and name not in ("blm_get_gemm_mode", "blm_gemm_plan_get_cus"):
                assert rc in (L.OK, L.ERR_INVALID, L.ERR_ABI, L.ERR_HIP, L.ERR_UNSUPPORTED), (name, variant, rc)
                by_status[rc] = by_status.get(rc, 0) + 1
                if rc != L.OK:
                    assert lib.blm_last_error(), (name, variant)  # every failure leaves a message
                if variant == "negative" and argtypes and name not in STATE_SETTERS:
                    assert rc != L.OK or name in NOOP_OK, (name, "negative sizes / NULL accepted", rc)
        # whatever a variant left behind in the process-wide planner / option state
        lib.blm_gemm_plan_override(0, 0)
        lib.blm_gemm_plan_clear(1)
        lib.blm_gemm_plan_set_cus(0)
        lib.blm_gemm_plan_comm_window(C.c_float(0.0))
        lib.blm_set_gemm_mode(0)
    print("SWEEP_OK calls=%d statuses=%s" % (n_calls, sorted(by_status.items())))


# functions for which -1 is a legal value of every integer they take (or that take none that matter)
STATE_SETTERS = {"blm_gemm_plan_clear"}
# entry points whose documented behaviour for empty / NULL work is "nothing to do": BLM_OK
NOOP_OK = set()


def hammer(seconds=4.0):
    lib = L.lib()
    stop = time.time() + seconds
    errors = []

    def args(op, m, n, k, acc):
        a = L.GemmArgs()
        a.abi_version = L.ABI_VERSION
        a.op, a.M, a.N, a.K = op, m, n, k
        a.lda = m if op == L.GEMM_TN else k
        a.ldb = k if op == L.GEMM_NT else n
        a.ldc = n
        a.flags = L.GEMM_ACCUMULATE if acc else 0
        return a

    def planner(seed):
        out = L.GemmPlan()
        i = seed
        while time.time() < stop:
            i = (i * 1103515245 + 12345) & 0x7FFFFFFF
            a = args(i % 3, 64 + (i >> 3) % 4000, 64 + (i >> 7) % 4000, 32 + (i >> 11) % 9000, (i >> 5) & 1)
            fn = lib.blm_gemm_plan_launch if i & 16 else lib.blm_gemm_plan_query
            if fn(C.byref(a), C.byref(out)) != 0 or out.tile not in (11, 12, 21, 22, 28) or out.splits == 0:
                errors.append(("plan", out.tile, out.splits))

    def mutator(seed):
        i = seed
        while time.time() < stop:
            i = (i * 1103515245 + 12345) & 0x7FFFFFFF
            what = i % 7
            if what == 0:
                lib.blm_gemm_plan_set(i % 3, 64 + i % 500, 64 + (i >> 4) % 500, 64 + (i >> 8) % 500, 0, 0, (11, 12, 21, 22, 28)[i % 5], 1 + i % 4)
            elif what == 1:
                lib.blm_gemm_plan_clear(i & 1)
            elif what == 2:
                lib.blm_gemm_plan_override((0, 11, 22)[i % 3], (0, 2, -4)[(i >> 3) % 3])
            elif what == 3:
                lib.blm_gemm_plan_set_cus((0, 240, 128)[i % 3])
            elif what == 4:
                lib.blm_gemm_plan_comm_window(C.c_float(float(i % 500)))
            elif what == 5:
                lib.blm_set_option(b"deterministic", i & 1)
                v = C.c_int(0)
                lib.blm_get_option(b"lstm_pipe", C.byref(v))
            else:
                lib.blm_gemm_plan_set_comm(i % 3, 64 + i % 500, 64, 64, 0, 0, 11, 2)
    threads = [threading.Thread(target=planner, args=(s,)) for s in (1, 2, 3)] + [threading.Thread(target=mutator, args=(9,))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    lib.blm_gemm_plan_override(0, 0)
    lib.blm_gemm_plan_clear(1)
    lib.blm_gemm_plan_set_cus(0)
    lib.blm_gemm_plan_comm_window(C.c_float(0.0))
    lib.blm_set_option(b"deterministic", 0)
    print("HAMMER_OK")


if __name__ == "__main__":
    {"sweep": sweep, "hammer": hammer}[sys.argv[1]]()
