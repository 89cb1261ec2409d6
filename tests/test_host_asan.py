"""CPU: the host half of libmolly_hip under AddressSanitizer + UBSan (SURVEY.md §5 row 2 — the reference's only
CI is a linter, /root/reference/.github/workflows/pylint.yml:30-33).  `python -m molly_amd.build --host-asan` compiles every csrc
translation unit host-only with -fsanitize=address,undefined and MOLLY_HOST_DRY (launches recorded and checked against gfx950's
limits instead of issued) and links tools/host_asan_driver.cpp, which walks fuzz_gemm.py's shape space, the attention argument
space and the elementwise / optimizer entry points at edge sizes through every entry point's validation, launch_cfg's cost model
and the stream-K / split-K range arithmetic.  Never run on the GPU box (no `gpu` marker; the build has no device code)."""
import subprocess

import pytest


def test_host_half_of_the_library_is_clean_under_asan_and_ubsan():
    from molly_amd import build as B
    import shutil
    if shutil.which(B.HIPCC) is None:
        pytest.skip("hipcc not present")
    exe = B.build_host_asan(verbose=False)
    r = subprocess.run([exe, "6000", "3000"], capture_output=True, text=True, timeout=600,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
                            "PATH": "/usr/bin:/bin"})
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "host-asan: ok" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    # the walk really went through the launchers: thousands of recorded launches, GEMM calls accepted
    line = next(l for l in r.stdout.splitlines() if l.startswith("host-asan:") and "dry launches" in l)
    assert int(line.split("dry launches")[0].split(",")[-1].strip()) > 5000, line
    acc = {l.split()[0]: int(l.split()[-2]) for l in r.stdout.splitlines() if l.startswith("  ") and l.rstrip().endswith("accepted")}
    assert acc["gemm_bf16_ctx"] > 1000 and acc["gemm_grouped"] > 20 and acc["attn_bwd_ws"] > 300, acc
