#!/usr/bin/env python3
"""CU contention between RCCL's channel workgroups and the backward GEMMs, rehearsed on ONE GPU.

No multi-GPU run is needed for the question "what does a collective that holds k CUs cost the step, and does a planner that
knows about it get the time back?": tools/comm_occupier.hip launches k workgroups with the footprint of RCCL's channel kernel
on gfx950 (256 threads, 280 registers per lane, 19,744 B LDS -- read off this image's librccl.so) WHERE engine.GradReducer
launches a bucket's all-reduce (`collective=` hook, same side stream, same event ordering), streaming the bucket through HBM
(dst += 0) and pacing itself to the time the bucket would take on the links at a given bus bandwidth.

Part 1: the roofline launch (NT 8192 x 512 x 4096, one round of 256 one-per-CU eight-wave workgroups) and the other cfg3
        backward shapes stand-alone beside a resident occupier, planned for the whole chip and for 256 - k CUs.
Part 2: the cfg3 training step (headline configuration) with the occupier in the reducer's place: step time, exposed
        communication time, for k in {0, 8, 16, 32, 64} x bus bandwidth {150, 300} GB/s x planner {whole chip, comm window,
        256 - k CUs for all of backward}.

    python3 tools/comm_occupancy_rehearsal.py [--steps 12] [--out profiles/r04_comm_occupancy_rehearsal.txt]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcomm_occupier.so")


def occupier_lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB,
                               os.path.join(HERE, "comm_occupier.hip")])
    lib = C.CDLL(LIB)
    lib.occupier_launch.restype = C.c_int
    lib.occupier_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
    return lib


class Occupier:
    """Stands in for dist.all_reduce inside GradReducer: k channel workgroups per bucket, paced to bytes / bus bandwidth."""

    def __init__(self, k, gbps, dev):
        self.k, self.gbps = int(k), float(gbps)
        self.lib = occupier_lib()
        self.zeros = torch.zeros(48 << 20, device=dev, dtype=torch.float32)  # src of dst += src: the gradients stay what they are
        self.launches = 0
        self.us_total = 0.0

    def __call__(self, view, comm_stream):
        if self.k <= 0:
            return
        n = view.numel()
        off = 0
        while off < n:  # a bucket larger than the zero buffer goes in pieces (the tied 67.6 MB weight)
            m = min(n - off, self.zeros.numel())
            us = m * 4 / (self.gbps * 1e3)
            rc = self.lib.occupier_launch(view[off:].data_ptr(), self.zeros.data_ptr(), m, self.k, us, 32, 1, comm_stream.cuda_stream)
            if rc:
                raise RuntimeError("occupier_launch failed (%d)" % rc)
            self.launches += 1
            self.us_total += us
            off += m


def standalone(dev, ks, out):
    """cfg3 backward GEMM shapes beside a RESIDENT occupier (one long launch on a side stream), whole-chip plan vs 256 - k."""
    from bayeslms_amd import _lib as L, ops
    side = torch.cuda.Stream()
    lib = occupier_lib()
    shapes = [("roofline fwd  NT 8192x512x4096", L.GEMM_NT, 8192, 512, 4096, False),
              ("ffn2 dgrad    NN 8192x4096x512", L.GEMM_NN, 8192, 4096, 512, False),
              ("ffn2 wgrad    TN 512x4096x8192 acc", L.GEMM_TN, 512, 4096, 8192, True),
              ("ffn1 wgrad    TN 4096x512x8192 acc", L.GEMM_TN, 4096, 512, 8192, True),
              ("qkv dgrad     NN 8192x512x1536", L.GEMM_NN, 8192, 512, 1536, False),
              ("decoder dgrad NN 8192x512x33000", L.GEMM_NN, 8192, 512, 33000, False)]
    zeros = torch.zeros(16 << 20, device=dev)
    sink = torch.zeros(16 << 20, device=dev)
    rows = []
    for name, op, M, N, K, acc in shapes:
        if op == L.GEMM_NT:
            A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            lda, ldb = K, K
        elif op == L.GEMM_NN:
            A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
            lda, ldb = K, N
        else:
            A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            lda, ldb = M, N
        Cout = torch.zeros(M, N, device=dev)
        for k in ks:
            for narrowed in ((False,) if k == 0 else (False, True)):
                ops.set_gemm_cus(256 - k if narrowed else 0)
                for _ in range(3):
                    ops.gemm(op, A, B, Cout, M, N, K, lda, ldb, N, accumulate=acc)
                torch.cuda.synchronize()
                reps = 8
                if k > 0:  # resident for the whole measurement: reps x ~2.2 ms at most
                    rc = lib.occupier_launch(sink.data_ptr(), zeros.data_ptr(), zeros.numel(), k, reps * 2600.0 + 500.0, 512, 1,
                                             side.cuda_stream)
                    assert rc == 0
                    torch.cuda._sleep(200000)  # ~0.1 ms: the occupier is on its CUs before the first GEMM is dispatched
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.gemm(op, A, B, Cout, M, N, K, lda, ldb, N, accumulate=acc)
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / reps
                a = L.GemmArgs()
                a.abi_version = L.ABI_VERSION
                a.op, a.M, a.N, a.K, a.lda, a.ldb, a.ldc = op, M, N, K, lda, ldb, N
                a.flags = L.GEMM_ACCUMULATE if acc else 0
                pl = L.GemmPlan()
                L.lib().blm_gemm_plan_query(C.byref(a), C.byref(pl))
                rows.append({"shape": name, "k": k, "planner_cus": 256 - k if narrowed else 256, "tile": pl.tile, "splits": pl.splits,
                             "us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)})
                print(json.dumps(rows[-1]), file=out, flush=True)
    ops.set_gemm_cus(0)
    return rows


def step_runs(dev, ks, rates, steps, warm, out):
    import bench as Bn
    from bayeslms_amd import engine, model as M, ops
    from bayeslms_amd.data import batchify, get_batch, synthetic_corpus
    import torch.distributed as dist
    if not dist.is_initialized():  # a one-rank group: engine.LateRows (compact embedding-row exchange) runs as in a DP job
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    # gloo moves device tensors through the host (a synchronisation per step that RCCL does not have): with one rank the
    # id all-gather of LateRows.begin is a device copy
    dist.all_gather_into_tensor = lambda out, inp, group=None: out.copy_(inp)
    V, T, Bc = Bn.V, Bn.T, Bn.B_PER_GPU
    stream = synthetic_corpus(V, Bc * ((warm + steps) * T + 1) + 17, seed=1111)
    train = batchify(stream, Bc, dev)
    rows = []
    for k in ks:
        for gbps in (rates if k > 0 else rates[:1]):
            for mode in (("off",) if k == 0 else ("off", "window", "narrow")):
                torch.manual_seed(1111)
                model = M.BayesTransformerModel(V, Bn.D_MODEL, Bn.NHEAD, Bn.D_FF, Bn.NLAYERS, Bn.DROPOUT, True, "FFN").to(dev)
                occ = Occupier(k, gbps, dev)
                kl_scale = float(T) / float(len(train))
                tr = engine.Trainer(model, lr=Bn.LR, clip=1.0, kl_scale=kl_scale, collective=occ, comm_cus=k, comm_plan=mode,
                                    comm_gbps=gbps)
                tr.reducer.measure = True
                kl_fn = Bn._kl_fn
                for i in range(warm + steps):
                    data, tgt = get_batch(train, i * T, T)
                    if i == warm:
                        torch.cuda.synchronize()
                        tr.reducer.exposed_events = []
                        occ.launches, occ.us_total = 0, 0.0
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    tr.step(data, tgt, None, kl_fn)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / steps
                rows.append({"k": k, "bus_GBps": gbps if k else None,
                             "planner": {"off": "whole chip", "window": "comm-table plans while a bucket's window is open",
                                         "narrow": "cost model on %d CUs from the first bucket to the end of backward" % (256 - k)}[mode],
                             "ms_per_step": round(ms, 3), "tokens_per_s": round(T * Bc / ms * 1e3, 0),
                             "comm_exposed_ms": None if k == 0 else round(tr.reducer.comm_exposed_ms(), 3),
                             "occupier_launches_per_step": occ.launches / steps,
                             "occupier_busy_ms_per_step": round(occ.us_total / steps / 1e3, 3)})
                print(json.dumps(rows[-1]), file=out, flush=True)
                del tr, model, occ
                ops.set_grad_ready_hook(None)
                ops.set_gemm_cus(0)
                torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ks", type=str, default="0,8,16,32,64")
    ap.add_argument("--rates", type=str, default="150,300")
    ap.add_argument("--out", type=str, default="")
    ap.add_argument("--skip-standalone", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ks = [int(v) for v in args.ks.split(",")]
    rates = [float(v) for v in args.rates.split(",")]
    out = open(args.out, "w") if args.out else sys.stdout
    print("# comm occupancy rehearsal: occupier = RCCL channel-kernel footprint (256 threads, 280 VGPR+AGPR, 19744 B LDS), one GPU", file=out)
    if not args.skip_standalone:
        print("# part 1: stand-alone cfg3 GEMM shapes beside a resident occupier of k workgroups", file=out)
        standalone(dev, ks, out)
    print("# part 2: cfg3 training step, occupier launched per bucket where GradReducer launches the all-reduce", file=out)
    step_runs(dev, ks, rates, args.steps, args.warmup, out)
    if args.out:
        out.close()
        print(open(args.out).read())


if __name__ == "__main__":
    main()
