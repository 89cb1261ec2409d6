"""ctypes binding of libmolly_hip.so.  The prototypes are read from include/molly_hip.h — the header is the
single source of truth for the C ABI (tests/test_abi.py checks every declared symbol is exported).

The product path fails loudly when the library is missing: there is NO eager/CPU fallback anywhere in
`molly_amd` (the CPU oracle lives in /oracle and is test infrastructure only).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "molly_hip.h")
# MOLLY_LIB_PATH: a differently built library for same-box A/B runs (tools/); the product default is the in-tree build
LIB_PATH = os.environ.get("MOLLY_LIB_PATH") or os.path.join(HERE, "libmolly_hip.so")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[Tuple[str, object]]]]:
    """-> {name: (restype, [(argname, ctype), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(molly_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        al = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    an = a.split("*")[-1].strip()
                    al.append((an, ctypes.c_void_p))
                else:
                    parts = a.split()
                    ty = [p for p in parts[:-1] if p != "const"][-1]
                    al.append((parts[-1], _CT[ty]))
        protos[name] = (restype, al)
    return protos


def _bind_torch_hip_runtime():
    """ONE HIP runtime per process.  libmolly_hip.so needs `libamdhip64.so.7`; PyTorch's ROCm wheels bundle their own copy under
    torch/lib with the same soname and load it lazily (at the first torch.cuda call, not at `import torch`).  Loaded before
    that, this library pulled in the system runtime (/opt/rocm) and the process ended up with two: kernels launched through one
    onto streams and buffers owned by the other — which happens to work for plain launches, and fails where a launch consults
    the runtime's device state (rocPRIM's radix sort inside molly_batch_assemble: "no ROCm-capable device is detected"; found by
    the world-2 smoke in round 2).  Loading torch's copy first (RTLD_GLOBAL, no device is touched) makes the dynamic loader
    resolve our dependency to it, whatever the import order of the caller."""
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except (ImportError, OSError):
        pass                      # no torch / no bundled runtime: the system runtime is the only one there is


class MollyLib:
    def __init__(self, path: str = LIB_PATH, strict: bool = True):
        """strict=False (tools only: an OLDER build loaded beside the in-tree one for an A/B) binds what that library exports."""
        if not os.path.exists(path):
            raise RuntimeError(
                f"molly_amd: HIP library {path} is missing — build it with `python -m molly_amd.build` "
                f"(or __graft_entry__.build()).  There is no CPU fallback.")
        self.path = path
        _bind_torch_hip_runtime()
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        self.fn = {}
        for name, (restype, al) in self.protos.items():
            try:
                f = getattr(self.cdll, name)       # AttributeError = header/library mismatch: fail loudly
            except AttributeError:
                if strict:
                    raise
                continue
            f.restype = restype
            f.argtypes = [t for _, t in al]
            self.fn[name] = f

    def last_error(self) -> str:
        return self.fn["molly_last_error"]().decode()

    def call(self, name: str, *args):
        """Call an int-returning entry point; tensors are passed as device pointers; raises on non-zero."""
        f = self.fn[name]
        conv = []
        for a, (an, ty) in zip(args, self.protos[name][1]):
            if ty is ctypes.c_void_p:
                if a is None:
                    conv.append(None)
                elif hasattr(a, "data_ptr"):
                    conv.append(a.data_ptr())
                elif isinstance(a, ctypes.Array):            # a host array of pointers (molly_p2p_*)
                    conv.append(ctypes.cast(a, ctypes.c_void_p))
                else:
                    conv.append(int(a))
            else:
                conv.append(a)
        if len(args) != len(self.protos[name][1]):
            raise TypeError(f"{name}: expected {len(self.protos[name][1])} args, got {len(args)}")
        rc = f(*conv)
        if rc != 0:
            raise RuntimeError(f"{name} failed (rc={rc}): {self.last_error()}")
        return rc

    def query(self, name: str, *args) -> int:
        """Call an entry point whose int return value is an answer, not a status (e.g. *_blocks)."""
        return self.fn[name](*args)


_LIB = None


def lib() -> MollyLib:
    global _LIB
    if _LIB is None:
        _LIB = MollyLib()
    return _LIB
