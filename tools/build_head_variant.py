#!/usr/bin/env python3
"""Build tools/variants/libmolly_head.so from the COMMITTED sources (git HEAD: molly_amd/csrc + include), so that an uncommitted
change can be timed against its predecessor on one box in one call:
    python tools/build_head_variant.py [rev]     then     MOLLY_LIB_PATH=tools/variants/libmolly_head.so python tools/bench_attn.py"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molly_amd import build as B  # noqa: E402


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
    with tempfile.TemporaryDirectory() as td:
        tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "molly_amd/csrc", "include"], capture_output=True, check=True).stdout
        subprocess.run(["tar", "-x", "-C", td], input=tar, check=True)
        csrc, inc = os.path.join(td, "molly_amd", "csrc"), os.path.join(td, "include")
        flags = [f for f in B.FLAGS if not f.startswith("-I")] + ["-I" + csrc, "-I" + inc]
        srcs = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))

        def comp(src):
            obj = os.path.join(td, src.rsplit(".", 1)[0] + ".o")
            cmd = [B.HIPCC] + flags + B.EXTRA.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(csrc, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-3000:])
            return obj
        with ThreadPoolExecutor(max_workers=6) as ex:
            objs = list(ex.map(comp, srcs))
        out = os.path.join(ROOT, "tools", "variants", "libmolly_head.so")
        subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
        print(out)


if __name__ == "__main__":
    main()
