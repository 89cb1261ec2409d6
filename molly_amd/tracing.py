"""Named ranges for rocprofv3's marker trace around the phases of the step (tracing row)."""
from __future__ import annotations

import os

import torch


class roctx:
    """`with ops.roctx("decoder fwd"):` — a named range in rocprofv3's marker trace (torch.cuda.nvtx is roctx on ROCm) around a phase of
    the step when MOLLY_ROCTX=1; otherwise nothing (the reference has no tracing beyond a wall-clock context manager,
    src/utils/tools.py:36-42).  Collect with `rocprofv3 --marker-trace --kernel-trace -- python3 bench.py ...` (never together with
    --pmc on this pool)."""
    ON = os.environ.get("MOLLY_ROCTX", "0") == "1"

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if roctx.ON:
            torch.cuda.nvtx.range_push(self.name)
        return self

    def __exit__(self, *exc):
        if roctx.ON:
            torch.cuda.nvtx.range_pop()
        return False
