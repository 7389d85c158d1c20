import sys, torch
sys.path.insert(0, ".")
from bayeslms_amd import _lib as L, ops
dev = torch.device("cuda:0")
def run(name, op, m, n, k, acc):
    if op == L.GEMM_NT: A, B, lda, ldb = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), k, k
    elif op == L.GEMM_NN: A, B, lda, ldb = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), k, n
    else: A, B, lda, ldb = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), m, n
    C = torch.zeros(m, n, device=dev)
    for _ in range(5): ops.gemm(op, A, B, C, m, n, k, lda, ldb, n, accumulate=acc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(op, A, B, C, m, n, k, lda, ldb, n, accumulate=acc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-28s %8.1f us %6.1f TF" % (name, ms * 1e3, 2.0 * m * n * k / ms / 1e9))
for rep in range(2):
    run("NT 8192x4096x512", L.GEMM_NT, 8192, 4096, 512, False)
    run("NN 8192x4096x512", L.GEMM_NN, 8192, 4096, 512, False)
    run("TN 8192x4096x512 (no split)", L.GEMM_TN, 8192, 4096, 512, True)
    run("NT 8192x4096x2048", L.GEMM_NT, 8192, 4096, 2048, False)
    run("NN 8192x4096x2048", L.GEMM_NN, 8192, 4096, 2048, False)
    run("TN 8192x4096x2048 (no split)", L.GEMM_TN, 8192, 4096, 2048, True)
