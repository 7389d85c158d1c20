#!/usr/bin/env python3
"""Audit of the GEMM kernels' ISA: inside the hand-counted `s_waitcnt lgkmcnt(2)` regions of the
operand double-buffering (gemm_f32_mfma.h compute()) there must be no other LGKM-counted
instruction (scalar loads, LDS writes, messages), or the count no longer says which ds_read has
landed.  Also reports scratch use.  Run after touching the kernel:  python tools/audit_gemm_isa.py"""
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
        kern, inreg, regions = None, False, 0
        for i, l in enumerate(lines):
            m = re.match(r"^(_ZN3blm15gemm_f32_kernel\S+):", l)
            if m:
                kern, inreg = m.group(1), False
            t = l.strip()
            if "s_waitcnt lgkmcnt(2)" in t and not inreg:
                inreg, regions = True, regions + 1
            elif inreg and t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                inreg = False
            elif inreg and t.startswith(("s_load", "s_buffer_load", "ds_write", "s_sendmsg", "s_memtime")):
                bad += 1
                print("SUSPECT", kern, i, t)
            if "ScratchSize:" in t and not t.endswith(" 0"):
                bad += 1
                print("SCRATCH", kern, t)
        print(f, "counted-wait regions:", regions)
    print("OK" if bad == 0 else "%d problems" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
