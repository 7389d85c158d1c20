"""Debug: per-phase cycles of the GEMM steady-state loop (library built with -DBLM_GEMM_PROF).
   make -C bayeslms_amd/csrc clean && make -C bayeslms_amd/csrc CXXFLAGS+=' -DBLM_GEMM_PROF'"""
import ctypes, os, sys
sys.path.insert(0, ".")
import torch
from bayeslms_amd import _lib as L, ops

def main():
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(L.LIB_PATH)
    shapes = [("nt", L.GEMM_NT, 8192, 512, 4096), ("nt", L.GEMM_NT, 8192, 4096, 512), ("nn", L.GEMM_NN, 8192, 512, 33000),
              ("tn", L.GEMM_TN, 4096, 512, 8192)]
    for tag, op, m, n, k in shapes:
        if op == L.GEMM_NT:
            A, B, lda, ldb = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), k, k
        elif op == L.GEMM_NN:
            A, B, lda, ldb = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), k, n
        else:
            A, B, lda, ldb = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), m, n
        C = torch.zeros(m, n, device=dev)
        acc = op == L.GEMM_TN
        fn = getattr(lib, "blm_debug_prof_" + tag)
        for _ in range(3):
            ops.gemm(op, A, B, C, m, n, k, lda, ldb, n, accumulate=acc)
        torch.cuda.synchronize()
        fn(None, 1)
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(op, A, B, C, m, n, k, lda, ldb, n, accumulate=acc)
        e1.record(); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 4)()
        fn(out, 1)
        comp, stash, bar, nn = [float(x) for x in out]
        ms = e0.elapsed_time(e1) / reps
        print(f"{tag} {m}x{n}x{k}: {ms*1000:.1f} us {2*m*n*k/ms/1e9:.1f} TF | per k-tile per wave (cycles): compute {comp/nn:.0f}  stash(vmcnt+ds_write) {stash/nn:.0f}  barrier {bar/nn:.0f}  total {(comp+stash+bar)/nn:.0f}  (own MFMA issue = 4096)")

if __name__ == "__main__":
    main()
