#!/usr/bin/env python3
"""The LDS-DMA of the GEMM and attention kernels is issued from inline asm (`s_mov_b32 m0, sL; s_nop 0; global_load_lds_dwordx4
vOFF, s[B:B+1]`).  hipcc pads hazards only between its OWN instructions, so one hazard is ours to rule out: a VALU write of an
SGPR (v_readfirstlane_b32 / v_readlane_b32 / v_cmp writing an SGPR pair) needs 5 wait states before a VMEM instruction reads
that SGPR as its base (guide §5.7 item 2).  An `s_nop 4` in front of every DMA costs 2-6 % of GEMM throughput (measured, same
box), so instead this script PROVES the hazard absent in the generated code: it compiles the sources to assembly and checks
that no SGPR used as the base of an asm-issued global_load_lds was written by a VALU instruction within the 5 preceding
instructions.  Run after every edit of gemm.hip / attention.hip (tests/test_abi.py runs it on the built sources).

    python tools/check_asm_dma_hazards.py            # exit code 0 = clean"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "molly_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assembly(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
               "-ffp-contract=fast", "-Wno-inline-asm", "-x", "hip", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return open(out).read().split("\n")


def sgprs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(src):
    lines = assembly(src)
    insts = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    bad, n_dma = [], 0
    for k, (ln, ins) in enumerate(insts):
        if not ins.startswith("global_load_lds"):
            continue
        ops = [t.strip() for t in ins.split(None, 1)[1].split(",")]
        base = set()
        for t in ops:
            base |= sgprs(t.split()[0])
        if not base:
            continue                                   # `off` form: the address is in VGPRs
        n_dma += 1
        for back in range(1, 7):                       # the asm string itself holds 2 instructions before the load
            if k - back < 0:
                break
            prev = insts[k - back][1]
            if prev.startswith(("v_readfirstlane", "v_readlane")) or (prev.startswith("v_cmp") and "s[" in prev.split(",")[0]):
                dst = sgprs(prev.split(None, 1)[1].split(",")[0].strip())
                if dst & base:
                    bad.append((src, ln + 1, prev, ins))
    return n_dma, bad


def main():
    total, bad = 0, []
    for src in ("gemm.hip", "attention.hip", "lora.hip"):
        n, b = check(src)
        total += n
        bad += b
    for b in bad:
        print("HAZARD %s:%d  %s  ->  %s" % b)
    print(f"{total} asm-issued LDS-DMA instructions with an SGPR base checked, {len(bad)} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
