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
        bad += valu_in_loops(lines, r"^(_ZN3blm15gemm_f32_kernelILi\dELi\dELi\dELb0ELb1ELi0E\S+):", "global_load_lds", 4 if f != "gemm_tn" else 16)  # TN: + the bias-gradient column sums' branch and the second tile's read bases
    bad += audit_lstm()
    print("OK" if bad == 0 else "%d problems" % bad)
    return 1 if bad else 0


def valu_in_loops(lines, kern_re, must_have, limit):
    """On gfx950 a vector instruction in a matrix loop is paid in matrix time (tools/mfma_valu_overlap.hip): in every
    steady-state loop (a backward branch spanning >= 32 MFMAs and at least one `must_have` instruction) of the
    kernels matching kern_re, count the non-MFMA v_* instructions and flag more than `limit` per trip."""
    bad, kern = 0, None
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    starts = [(i, re.match(r"^(_Z\S+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^(_Z\S+):", l)]
    back, best = [], {}
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            back.append((labels[m.group(1)], i))
    def mfmas(a, i):
        return sum(1 for x in lines[a:i] if x.strip().startswith("v_mfma"))
    heavy = [(a, i) for a, i in back if mfmas(a, i) >= 32]
    for a, i in heavy:
        if any((a2, i2) != (a, i) and a <= a2 and i2 <= i for a2, i2 in heavy):  # innermost matrix loops only
            continue
        owner = [k for s0, k in starts if s0 < a]
        if not owner or not re.match(kern_re, owner[-1] + ":"):
            continue
        body = [x.strip() for x in lines[a:i]]
        nm = sum(1 for x in body if x.startswith("v_mfma"))
        if nm < 32 or not any(x.startswith(must_have) for x in body):
            continue
        valu = sum(1 for x in body if x.startswith("v_") and not x.startswith("v_mfma"))
        if owner[-1] not in best or nm > best[owner[-1]][0]:
            best[owner[-1]] = (nm, valu)
    for k, (nm, valu) in best.items():  # the steady-state loop = the innermost one with the most MFMAs (remainder steps run once)
        print("  %-70s loop of %3d MFMAs: %d vector instructions" % (k[:70], nm, valu))
        if valu > limit:
            bad += 1
            print("VECTOR INSTRUCTIONS IN A MATRIX LOOP", k, valu)
    return bad


def audit_lstm():
    """The software-pipelined LSTM step kernels without a K tail must have NO vector instruction in their K loops."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "lstm_step.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I",
                               os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-Wno-pass-failed",
                               os.path.join(ROOT, "bayeslms_amd", "csrc", "lstm_step.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    # <RING 2, ..., NW 4, PIPE true, TAIL false>
    return (valu_in_loops(lines, r"^(_ZN3blm20lstm_step_fwd_kernelILi2ELi\dELb1ELi4ELb1ELb0E\S+):", "buffer_load", 0) +
            valu_in_loops(lines, r"^(_ZN3blm20lstm_step_bwd_kernelILi2ELb1ELi4ELb1ELb0E\S+):", "buffer_load", 0))


if __name__ == "__main__":
    sys.exit(main())
