"""vvm_exec_batch keeps the plan of its last call.  The plan may be reused only for what it was made from (states, sizes, options);
everything else a call hands over -- stream addresses, constants -- must be read on every call, and a state released and another
made in between must end the reuse."""
import numpy as np
import pytest

from batch_epoch_case import Emitter
from renderer_amd import abi, scripts
from renderer_amd.runtime import VectorVMState

pytestmark = pytest.mark.gpu
f32 = np.float32
CAP = 120_000


def _plans(be):
    return int(be.lib.vvm_debug_batch_plans_made())


def _new_inputs(em, seed, kill_half):
    """Fresh input streams in NEW device buffers for an emitter, and a new call frame that names them."""
    em.host_in = scripts.fountain_streams_np(seed, CAP, kill_half=kill_half)
    new = []
    for a in em.host_in:
        p = em.be.lib.vvm_device_alloc(a.nbytes)
        assert p
        em.be._check(em.be.lib.vvm_device_upload(p, a.ctypes.data, a.nbytes), "upload")
        new.append(p)
    old = em.ins
    em.ins = new
    em.frame = em.state.frame([abi.DataSet(em.ins, em.outs, 0)], em.n, [em.w.constants])
    return old


def test_plan_is_reused_only_for_what_it_was_made_from():
    be = abi.load_hip_backend()
    sizes = [100_000, 16_384, 3, 50_000, 2_049, 70_001]
    ems = [Emitter(be, v, n, v % 2 == 1, 10 + v, cap=CAP) for v, n in enumerate(sizes)]
    stale = []
    try:
        opts = [abi.ExecOptions(0, 0, 0, None, 0) for _ in ems]

        def run(tag, expect_new_plan):
            before = _plans(be)
            be.exec_batch([e.frame for e in ems], opts)
            assert (_plans(be) - before == 1) == expect_new_plan, tag
            for k, e in enumerate(ems):
                assert e.check("%s / emitter %d" % (tag, k), tail=False), tag

        run("first call", True)
        run("same again", False)
        # other stream addresses and other data, same states and sizes: the plan holds, the addresses must be read anew
        stale += _new_inputs(ems[1], 77, False)
        stale += _new_inputs(ems[3], 78, True)
        run("new input streams", False)
        # other constants (another time step): read anew as well
        for e in (ems[0], ems[4]):
            e.w.constants = scripts.fountain_constants(dt=1.0 / 30.0, drag=0.9)
            e.frame = e.state.frame([abi.DataSet(e.ins, e.outs, 0)], e.n, [e.w.constants])
        run("new constants", False)
        # a size changes: planned again (the Exec moves to another kernel shape)
        ems[2].n = 90_000
        ems[2].frame = ems[2].state.frame([abi.DataSet(ems[2].ins, ems[2].outs, 0)], ems[2].n, [ems[2].w.constants])
        run("new size", True)
        run("same again after the new size", False)
        # an option changes
        opts[0] = abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_NO_EIGHT_WIDE)
        run("new option", True)
        assert ems[0].state.last_exec_width() == 4
        # a state is released and another made (its address may be the old one): planned again, and the new script is what runs
        e = ems[5]
        e.state.close()
        e.w = scripts.fountain(variant=9)
        e.state = VectorVMState(e.w.script.blob, be)
        e.frame = e.state.frame([abi.DataSet(e.ins, e.outs, 0)], e.n, [e.w.constants])
        run("another state", True)
        run("same again after the other state", False)
        # fewer Execs
        ems_saved = list(ems)
        del ems[1]
        del opts[1]
        run("one Exec fewer", True)
        ems[:] = ems_saved
    finally:
        for e in ems:
            e.free()
        for p in stale:
            be.lib.vvm_device_free(p)
