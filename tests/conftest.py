import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _release_gpu_memory(request):
    """Full-size models (up to Molly-8B with gradients: 130 GB) must not pile up across tests."""
    yield
    if request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.empty_cache()


@pytest.fixture(scope="session", autouse=True)
def _gemm_launch_shape():
    """MOLLY_TEST_GEMM_BLOCKS=dyn (or -3, 0) runs the whole GPU suite with the 256x256 GEMM launched the way a rank of a multi-GPU job
    launches it (resident blocks that draw their tiles: the N > 1 default; or round 2's blocks of three tiles / one tile per block)
    instead of the static walk: every GemmContext reads the variable (ops.GemmContext), this fixture sets the thread's default one."""
    mode = os.environ.get("MOLLY_TEST_GEMM_BLOCKS")
    if mode and torch.cuda.is_available():
        from molly_amd import ops
        from molly_amd._lib import lib
        ops.ensure_gemm_workspace(1 << 28)
        lib().call("molly_gemm_set_persistent_blocks", 256 if mode == "dyn" else int(mode))
        lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["dynamic"], 1 if mode == "dyn" else 0)
    yield


@pytest.fixture(scope="session")
def tiny_meta():
    with open(os.path.join(GOLD, "tiny_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def tiny_gold():
    return dict(np.load(os.path.join(GOLD, "tiny_fp32.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def tiny_gold_bf16():
    return dict(np.load(os.path.join(GOLD, "tiny_bf16.npz"), allow_pickle=False))


WIDE_TAGS = ("wide", "wide4b", "wide8b")      # Molly-1.7B / 4B / 8B decoder widths (tests/golden/gen_golden_wide.py)


def wide_fixture(tag):
    """G4: one layer of every stack at real widths -> (meta, golden arrays)."""
    with open(os.path.join(GOLD, f"{tag}_meta.json")) as f:
        meta = json.load(f)
    return meta, dict(np.load(os.path.join(GOLD, f"{tag}_fp32.npz"), allow_pickle=False))


def steps_fixture():
    """G5: four optimizer steps x GA 2 of the reference's loop pieces on the tiny model (tests/golden/gen_golden_steps.py)."""
    g = dict(np.load(os.path.join(GOLD, "tiny_steps.npz"), allow_pickle=False))
    types = [["protein", "dna"], ["rna", "pad"]]                     # make_batch's span layout (gen_golden.py)

    def batch(s, k):
        b = {key: torch.from_numpy(g[f"in/{s}/{k}/{key}"]) for key in ("input_ids", "labels", "attention_mask", "omic_ids")}
        b["omic_info_list"] = [[{"type": t, "start": int(st)} for t, st in zip(tr, sr)]
                               for tr, sr in zip(types, g[f"in/{s}/{k}/starts"])]
        return b
    return g, batch


def tiny_batch(gold, meta):
    return {
        "input_ids": torch.from_numpy(gold["in/input_ids"]),
        "labels": torch.from_numpy(gold["in/labels"]),
        "attention_mask": torch.from_numpy(gold["in/attention_mask"]),
        "omic_ids": torch.from_numpy(gold["in/omic_ids"]),
        "omic_info_list": meta["omic_info_list"],
    }


def tiny_state_dict(meta, dtype=torch.float32):
    from molly_amd.synth import synth_state_dict
    shapes = {k: tuple(v) for k, v in meta["state_dict_shapes"].items()}
    sd = synth_state_dict(shapes, meta["config"]["seed_w"])
    return {k: v.to(dtype) for k, v in sd.items()}
