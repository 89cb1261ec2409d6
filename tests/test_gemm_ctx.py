"""CPU: the GEMM's launch state is per CONTEXT (include/molly_hip.h molly_gemm_ctx_*), not process-global — round 2's review found
`Zero2Optimizer.__init__` flipping a global launch mode that every later GEMM of the process inherited.  The context calls are
host-only code, so they run without a GPU."""
import threading

import pytest

from molly_amd import ops
from molly_amd._lib import lib


def test_contexts_keep_their_own_knobs_and_the_default_stays_untouched():
    L = lib()
    default_before = {k: L.query("molly_gemm_ctx_get", None, v) for k, v in ops.GEMM_KEYS.items()}
    a, b = ops.GemmContext(), ops.GemmContext()
    assert a.handle and b.handle and a.handle != b.handle
    a.set("persistent_blocks", -3)
    b.set("persistent_blocks", 0)
    a.set("streamk", 0)
    b.set("schedule", 1)
    assert (a.get("persistent_blocks"), a.get("streamk"), a.get("schedule")) == (-3, 0, -1)
    assert (b.get("persistent_blocks"), b.get("streamk"), b.get("schedule")) == (0, 1, 1)
    assert {k: L.query("molly_gemm_ctx_get", None, v) for k, v in ops.GEMM_KEYS.items()} == default_before
    assert default_before["persistent_blocks"] == 256 and default_before["streamk"] == 1
    with pytest.raises(RuntimeError, match="persistent_blocks"):
        a.set("persistent_blocks", 7)                                 # neither a multiple of 8 nor -t
    assert a.get("persistent_blocks") == -3                           # a rejected value changes nothing
    # the current-context stack of the Python layer
    assert ops._ctx() is None
    with ops.use_gemm_context(a):
        assert ops._ctx() == a.handle
        with ops.use_gemm_context(b):
            assert ops._ctx() == b.handle
        assert ops._ctx() == a.handle
    assert ops._ctx() is None


def test_default_context_is_per_host_thread():
    """The setters without a context edit the CALLING thread's default context: another host thread never inherits them."""
    L = lib()
    L.call("molly_gemm_set_persistent_blocks", 0)
    try:
        seen = {}

        def other():
            seen["mode"] = L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["persistent_blocks"])
            L.call("molly_gemm_set_persistent_blocks", -2)
            seen["own"] = L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["persistent_blocks"])
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"mode": 256, "own": -2}
        assert L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["persistent_blocks"]) == 0
    finally:
        L.call("molly_gemm_set_persistent_blocks", 256)


def test_two_optimizers_in_one_process_do_not_leak_a_launch_mode(monkeypatch):
    """What `Zero2Optimizer` wants beside its collectives is recorded on the optimizer and applied by `attach_optimizer` to the
    context of the ONE model it steps (molly_amd/model.py); constructing optimizers changes no context at all."""
    import torch
    from molly_amd.trainer import Zero2Optimizer
    from molly_amd.trainer import zero2 as Z
    L = lib()

    class K:                                                       # shard arithmetic stand-in: the constructor needs none of it
        pass
    before = L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["persistent_blocks"])
    c1, c2 = ops.GemmContext(), ops.GemmContext()
    p, g = torch.zeros(512, dtype=torch.bfloat16), torch.zeros(512, dtype=torch.bfloat16)
    o1 = Zero2Optimizer(p, g, 256, kernels=K())
    monkeypatch.setenv("MOLLY_GEMM_PERSISTENT_MULTI", "0")
    o2 = Zero2Optimizer(p.clone(), g.clone(), 256, kernels=K())
    assert L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["persistent_blocks"]) == before == 256
    assert (c1.get("persistent_blocks"), c2.get("persistent_blocks")) == (256, 256)
    # what attach_optimizer does with an optimizer that runs beside collectives (world > 1 on the GPU)
    o1.gemm_blocks_mode, o2.gemm_blocks_mode = -3, 0
    for ctx, o in ((c1, o1), (c2, o2)):
        ctx.set("persistent_blocks", o.gemm_blocks_mode)
    assert (c1.get("persistent_blocks"), c2.get("persistent_blocks")) == (-3, 0)
    assert L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["persistent_blocks"]) == 256
    c1.set("dynamic", 1)                                           # 'dyn', the N > 1 default: 256 resident blocks that draw their tiles
    assert (c1.get("dynamic"), c2.get("dynamic"), L.query("molly_gemm_ctx_get", None, ops.GEMM_KEYS["dynamic"])) == (1, 0, 0)
    assert not hasattr(Z, "_GEMM_MODE_GLOBAL")
