"""CPU: the C-ABI library loads and exports every symbol include/molly_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from molly_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libmolly_hip.so not built (run __graft_entry__.build())"
    protos = _lib.parse_header()
    assert len(protos) >= 20
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(cdll, n)]
    assert not missing, missing


def test_header_has_no_torch_types_and_cites_reference():
    src = open(_lib.HEADER).read()
    assert "torch" not in src.lower().replace("pytorch", "") or "at::" not in src
    assert 'extern "C"' in src
    assert "reference" in src and "HF:" in src


def test_error_channel_and_abi_version():
    m = _lib.lib()
    assert m.fn["molly_abi_version"]() >= 1
    # argument validation happens before any launch: callable without a GPU
    rc = m.fn["molly_gemm_nt_bf16"](None, None, None, None, None, None, 128, 128, 48, 48, 48, 128, 0, 0)
    assert rc != 0 and "multiple of 64" in m.last_error()
    # grouped launch: the problem table is validated on the host before anything touches the GPU
    rc = m.fn["molly_gemm_grouped_bf16"](None, None, 0, 1024, 0)
    assert rc != 0 and "1..16 problems" in m.last_error()
    prob = (ctypes.c_int64 * 6)(16, 32, 48, 256 | (250 << 32), 1024 | (256 << 32), 256)       # N = 250: not a multiple of 8
    rc = m.fn["molly_gemm_grouped_bf16"](None, ctypes.cast(prob, ctypes.c_void_p), 1, 1024, 0)
    assert rc != 0 and "multiple of 8" in m.last_error()
    rc = m.fn["molly_cls_loss_fwd_bwd"](None, None, None, None, None, None, 4, 8, 4, 0, -100, 1)  # ld < V
    assert rc != 0 and "cls_loss" in m.last_error()


def test_product_package_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle\.molly_ref|importlib.*oracle", re.M)
    for dp, _, fns in os.walk(os.path.join(ROOT, "molly_amd")):
        for fn in fns:
            if fn.endswith(".py"):
                txt = open(os.path.join(dp, fn)).read()
                assert not pat.search(txt), f"{fn} references the oracle"


def test_asm_issued_lds_dma_has_no_sgpr_hazard():
    """The GEMM / attention LDS-DMA is issued from inline asm, where hipcc pads no hazards: prove on the generated assembly
    that no SGPR base of such a load was written by a VALU instruction within the 5 preceding instructions."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_dma_hazards.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 hazards" in r.stdout


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """Loading the library before torch has loaded its bundled HIP runtime must not pull in a second one (round 2: with two
    runtimes rocPRIM's radix sort inside molly_batch_assemble found "no ROCm-capable device")."""
    import subprocess
    import sys
    code = "\n".join(["import sys; sys.path.insert(0, %r)" % ROOT, "from molly_amd import _lib", "_lib.lib()",
                      "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})",
                      "print('LIBS', len(libs), libs)"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("LIBS")][-1]
    assert line.split()[1] == "1", line
