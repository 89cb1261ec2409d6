"""GPU: randomised shapes through the GEMM forms / launch shapes / epilogue flags and through attention forward + backward, each
against an fp32 torch computation (tools/gemm_diag/fuzz_gemm.py, tools/fuzz_attn.py: the same code, fewer cases)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(path, argv):
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_" + os.path.basename(path)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    sys.argv = [path] + argv
    try:
        with pytest.raises(SystemExit) as e:
            mod.main()
    finally:
        sys.argv = old
    assert e.value.code == 0


@pytest.mark.parametrize("seed", [11, 12])
def test_gemm_forms_and_launch_shapes_random(seed):
    _run(os.path.join(ROOT, "tools", "gemm_diag", "fuzz_gemm.py"), ["--cases", "250", "--seed", str(seed)])


@pytest.mark.parametrize("seed", [21, 22])
def test_attention_fwd_bwd_random(seed):
    _run(os.path.join(ROOT, "tools", "fuzz_attn.py"), ["--cases", "60", "--seed", str(seed)])
