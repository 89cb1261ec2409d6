"""The remainder carving of the grouped weight-gradient launch (molly_amd/qwen3.py::_carve_remainder): geometry on CPU tensors — every
output element belongs to exactly one of the launches, the grouped launch gets whole rounds of 256 tiles."""
import pytest
import torch

from molly_amd.qwen3 import _carve_remainder


def _problem(ma, nb, k, to):
    a = torch.empty(ma, k, dtype=torch.bfloat16)
    b = torch.empty(k, nb, dtype=torch.bfloat16)
    out = torch.zeros((nb, ma) if to else (ma, nb), dtype=torch.float32)
    return (a, b, out, to)


def _tiles(ps):
    return sum((-(-a.shape[0] // 256)) * (-(-b.shape[1] // 256)) for a, b, _, _ in ps)


@pytest.mark.parametrize("shapes,k,expect", [
    # Qwen3-4B at one sample per GPU (3,072 tokens): 1,540 tiles = 6 rounds + 4
    ([(2560, 6144, True), (2560, 4096, False), (2560, 19456, True), (2560, 9728, False)], 3072, 4),
    # Qwen3-1.7B at 16 samples: 768 tiles, whole rounds: nothing to carve
    ([(2048, 4096, True), (2048, 2048, False), (2048, 12288, True), (2048, 6144, False)], 32768, 0),
    # a remainder of 40: too many to be worth a launch of their own
    ([(2560, 6144, True), (2560, 4096, False), (2560, 19456, True), (2560, 9728, False), (256, 9216, False)], 3072, 0),
    # short contraction: a tile-time is not worth a launch
    ([(2560, 6144, True), (2560, 4096, False), (2560, 19456, True), (2560, 9728, False)], 1024, 0),
    # one wide single-row problem carries the remainder
    ([(256, 65536 + 768, False)], 4096, 3),
])
def test_carve_covers_every_output_once(shapes, k, expect):
    probs = [_problem(ma, nb, k, to) for ma, nb, to in shapes]
    new, carved = _carve_remainder(probs)
    if expect == 0:
        assert carved is None and new is probs
        return
    assert _tiles([carved]) == expect and _tiles(new) % 256 == 0 and _tiles(new) + expect == _tiles(probs)
    assert len(new) <= 16
    for a, b, out, to in new + [carved]:
        assert tuple(out.shape) == ((b.shape[1], a.shape[0]) if to else (a.shape[0], b.shape[1]))
        assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
        out += 1
    for _, _, out, _ in probs:
        assert float(out.min()) == 1.0 and float(out.max()) == 1.0


def test_carve_switch(monkeypatch):
    probs = [_problem(ma, nb, 3072, to) for ma, nb, to in [(2560, 6144, True), (2560, 4096, False), (2560, 19456, True), (2560, 9728, False)]]
    monkeypatch.setenv("MOLLY_WGRAD_CARVE", "0")
    new, carved = _carve_remainder(probs)
    assert carved is None and new is probs
