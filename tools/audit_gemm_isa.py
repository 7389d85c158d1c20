#!/usr/bin/env python3
"""Audit of the GEMM kernels' ISA: in front of every hand-counted `s_waitcnt lgkmcnt(N > 0)` of the
operand double-buffering (gemm_f32_mfma.h compute()) the N youngest LGKM-counted instructions must
all be ds_reads (no scalar load, LDS write or message among them), or the count no longer says
which ds_read has landed.  Also flags scratch traffic inside loops (spills around a loop are
harmless).  Run after touching the kernel:  python tools/audit_gemm_isa.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    bad = 0
    for f in ("gemm_nt", "gemm_nn", "gemm_tn"):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, f + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I",
                                   os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-Wno-pass-failed",
                                   os.path.join(ROOT, "bayeslms_amd", "csrc", f + ".hip"), "-o", out],
                                  stderr=subprocess.DEVNULL)
            lines = open(out).read().split("\n")
        kern, regions = None, 0
        lgkm = ("ds_", "s_load", "s_buffer_load", "s_memtime", "s_sendmsg")
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []  # (first line, last line) of every backward branch target .. branch
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                a = labels[m.group(1)]
                nm = sum(1 for k in range(a, i) if "v_mfma" in lines[k])
                if nm >= 64 and i - a < 1500:  # a steady-state K loop (the tail loops run once or twice)
                    loops.append((a, i))
        for i, l in enumerate(lines):
            m = re.match(r"^(_ZN3blm15gemm_f32_kernel\S+):", l)
            if m:
                kern = m.group(1)
            t = l.strip()
            m = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", t)
            if kern and m and int(m.group(1)) > 0 and "vmcnt" not in t:
                # a counted wait: the N youngest LGKM-counted instructions in front of it must all be
                # ds_reads of the fragment double buffer, or the count no longer says which read landed
                n, j, regions = int(m.group(1)), i - 1, regions + 1
                while n > 0 and j >= 0 and not lines[j].startswith("_ZN3blm"):
                    u = lines[j].strip()
                    if u.startswith(lgkm):
                        if not u.startswith("ds_read"):
                            bad += 1
                            print("SUSPECT", kern, j, u, "in front of", t)
                        n -= 1
                    j -= 1
            if t.startswith("scratch_") and any(a <= i <= b for a, b in loops):
                bad += 1
                print("SCRATCH IN LOOP", kern, i, t)
        print(f, "counted-wait regions:", regions)
    print("OK" if bad == 0 else "%d problems" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
