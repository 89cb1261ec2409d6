#!/usr/bin/env python3
"""Build a variant of libmolly_hip.so with extra -D flags, for same-box A/B runs (MOLLY_LIB_PATH=<variant> python bench.py ...).
    python tools/build_variant.py nt0 -DMOLLY_NT_LOAD=0 -DMOLLY_NT_STORE=0      ->  tools/variants/libmolly_nt0.so
The variant libraries are git-ignored (*.so) and travel to the GPU box with the snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import build as B  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out = os.path.join(B.ROOT, "tools", "variants")
    objd = os.path.join(out, name)
    os.makedirs(objd, exist_ok=True)

    def comp(src):
        obj = os.path.join(objd, src.rsplit(".", 1)[0] + ".o")
        cmd = [B.HIPCC] + B.FLAGS + extra + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(B.CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return obj
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(comp, B._sources()))
    lib = os.path.join(out, f"libmolly_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
