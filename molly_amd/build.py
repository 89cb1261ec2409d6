"""Build libmolly_hip.so (the C-ABI HIP library, include/molly_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repo snapshot.  Staleness is decided by CONTENT (a digest of the source, every header and the flags,
kept beside each object), not by modification times: a snapshot copy reshuffles mtimes, and a rebuild on the GPU box
would cost box minutes and print into the benchmark's stdout.  Messages go to stderr for the same reason."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libmolly_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
         "-Wall", "-Wno-unused-function", "-Wno-inline-asm", "-ffp-contract=fast"]


# per-file extra flags
# gemm.hip: the ticket of the dynamic tile fetch is ONE lane's returning atomic add whose result must stay in flight for a whole
# K-tile; hipcc's atomic optimizer rewrites it into a wave reduction + readfirstlane that waits for the result on the spot
EXTRA = {"gemm.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}
                # (tried for attention.hip: -fno-slp-vectorize — hipcc packs the softmax's scalar adds into v_pk_add_f32 behind v_mov
                # shuffles — 168-170 us either way at B8 T2048 H16 D128: not kept)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(src):
    h = hashlib.sha256()
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "molly_hip.h")]
    for f in [os.path.join(CSRC, src)] + hs:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join([f for f in FLAGS if not f.startswith("-I")] + EXTRA.get(src, [])).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
    stamp = obj + ".sha256"
    sp = os.path.join(CSRC, src)
    dg = _digest(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == dg:
        return obj, False
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dg)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile what is stale and link.  Serialised across processes by a file lock: the ranks of a multi-GPU launch may all call
    it (before they touch the GPU); the first one in builds, the others find every digest current and return."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    if force:
        for f in os.listdir(OBJ):
            if f != ".lock":
                os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in res]
    want = hashlib.sha256("".join(open(o + ".sha256").read() for o in objs).encode()).hexdigest()
    lstamp = os.path.join(OBJ, "libmolly_hip.sha256")
    linked = os.path.exists(LIB) and os.path.exists(lstamp) and open(lstamp).read().strip() == want
    if any(c for _, c in res) or not linked:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(lstamp, "w") as f:
            f.write(want)
        if verbose:
            print(f"[molly_amd.build] linked {LIB} from {len(objs)} objects", file=sys.stderr)
    return LIB


# ---- host-sanitizer build (CPU box only) ------------------------------------------------------------------
HOST_ASAN_DIR = os.path.join(HERE, "csrc", "build_host_asan")
HOST_ASAN_FLAGS = ["--cuda-host-only", "-O1", "-g", "-fPIC", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                   "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-DMOLLY_HOST_DRY=1",
                   "-Wno-unused-value", "-Wno-inline-asm"]


def build_host_asan(verbose: bool = True) -> str:
    """The HOST half of every csrc translation unit under AddressSanitizer + UBSan, launches recorded instead of issued
    (common.h MOLLY_HOST_DRY), linked with tools/host_asan_driver.cpp into an executable that walks the entry points' argument
    space.  No device code is compiled (--cuda-host-only: seconds per file); the code-object symbols the host objects refer to
    are satisfied by empty stand-ins — nothing here can run a kernel, and nothing here is ever run on the GPU box.
    Returns the path of the driver executable."""
    os.makedirs(HOST_ASAN_DIR, exist_ok=True)

    def comp(src):
        path = os.path.join(CSRC, src) if not os.path.isabs(src) else src
        obj = os.path.join(HOST_ASAN_DIR, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        lang = ["-x", "hip"] if src.endswith(".hip") else ["-x", "c++"]
        r = subprocess.run([HIPCC] + HOST_ASAN_FLAGS + lang + ["-c", path, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"host-asan compile failed for {src}:\n{r.stderr[-4000:]}")
        return obj
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(comp, _sources() + [os.path.join(ROOT, "tools", "host_asan_driver.cpp")]))
    nm = subprocess.run(["nm"] + objs, capture_output=True, text=True, check=True).stdout
    syms = sorted({ln.split()[-1] for ln in nm.splitlines() if " U __hip_fatbin_" in ln})
    stub = os.path.join(HOST_ASAN_DIR, "fatbin_stub.c")
    with open(stub, "w") as f:
        f.write("/* generated: the device code objects a --cuda-host-only build refers to do not exist */\n")
        for sy in syms:
            f.write(f'__attribute__((section(".hip_fatbin"), aligned(4096))) const char {sy}[4096] = {{0}};\n')
    subprocess.run(["gcc", "-c", stub, "-o", stub[:-2] + ".o"], check=True)
    exe = os.path.join(HOST_ASAN_DIR, "host_asan_driver")
    r = subprocess.run([HIPCC, "-fsanitize=address,undefined"] + objs + [stub[:-2] + ".o", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host-asan link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[molly_amd.build] host-sanitizer driver: {exe}", file=sys.stderr)
    return exe


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        exe = build_host_asan()
        sys.exit(subprocess.run([exe] + [a for a in sys.argv[1:] if a.isdigit()]).returncode)
    build(force="--force" in sys.argv)
