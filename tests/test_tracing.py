"""The roctx phase ranges (molly_amd/tracing.py): nothing when off; balanced push/pop in program order when on."""
import torch

from molly_amd import tracing


def _record(monkeypatch):
    calls = []
    monkeypatch.setattr(torch.cuda.nvtx, "range_push", lambda name: calls.append(("push", name)))
    monkeypatch.setattr(torch.cuda.nvtx, "range_pop", lambda: calls.append(("pop",)))
    return calls


def test_ranges_are_absent_by_default(monkeypatch):
    calls = _record(monkeypatch)
    monkeypatch.setattr(tracing.roctx, "ON", False)
    with tracing.roctx("outer"):
        with tracing.roctx("inner"):
            pass
    assert calls == []


def test_ranges_nest_and_close_on_an_exception(monkeypatch):
    calls = _record(monkeypatch)
    monkeypatch.setattr(tracing.roctx, "ON", True)
    try:
        with tracing.roctx("outer"):
            with tracing.roctx("inner"):
                raise ValueError("x")
    except ValueError:
        pass
    assert calls == [("push", "outer"), ("push", "inner"), ("pop",), ("pop",)]


def test_the_optimizer_step_is_bracketed(monkeypatch):
    """Zero2Optimizer.step names its three phases (checked on the source: the class needs a device to run)."""
    import inspect

    from molly_amd.trainer.zero2 import Zero2Optimizer
    src = inspect.getsource(Zero2Optimizer.step)
    for name in ("reduce-scatter", "grad norm + clip", "AdamW + all-gather"):
        assert name in src
