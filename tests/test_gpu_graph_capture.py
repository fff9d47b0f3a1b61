"""Execs enqueued while their stream is being captured become nodes of a HIP graph, and replays of that graph are Execs like any
other.  What makes that work: vvm_exec_async of a state that has run on the stream before issues kernel launches only (the
parameter block travels as a kernel argument, buffers are sized, nothing is copied or waited for), every launch takes its
look-back tag from the launch words on the device and leaves the next one there, and vvm_exec_finish reads the counts of the
last replay and brings the host's twin of that tag up to date.

The second test replays across the wrap of the tag (see test_gpu_epoch_wrap.py for the construction): the host has seen one of
the five launches, finds at the finish that the device's tag is below its own, and clears the look-back words ahead of the
next launch."""
import os

import pytest

from batch_epoch_case import Emitter
from renderer_amd import abi, hip

pytestmark = pytest.mark.gpu
LAST_EPOCH = (1 << 30) - 1


def _fill(be, e, byte):
    import batch_epoch_case
    for p in e.outs:
        be._check(be.lib.vvm_device_memset(p, byte, batch_epoch_case.CAP * 4), "memset")


@pytest.mark.parametrize("n, flags, width", [(300_000, 0, 8), (50_000, 0, 4), (300_000, abi.VVM_EXEC_NO_LIGHT_KERNEL, 8)],
                         ids=["light eight-wide", "four-wide", "general eight-wide"])
def test_replays_of_captured_execs_equal_the_oracle(n, flags, width):
    be = abi.load_hip_backend()
    e = Emitter(be, 0, n, True, 11, cap=400_000)
    s = hip.Stream()
    graph = None
    try:
        opt = abi.ExecOptions(0, 0, 0, s.value, flags)
        be.exec_ex(e.frame, opt)          # the state's first Exec on this stream: sizes its buffers, outside the capture
        assert e.state.last_exec_width() == width and e.check("before the capture", tail=False)
        s.begin_capture(hip.hipStreamCaptureModeGlobal)
        for _ in range(3):
            be.exec_async(e.frame, opt)
        graph = s.end_capture()
        assert graph.num_nodes() == 3          # three kernel launches and nothing else
        _fill(be, e, 0xCD)          # (nothing has run since the capture began: the streams still hold this after it)
        e.frame.ctx.DataSets.data[0].DataSetAccessIndex = -7
        for _ in range(4):
            graph.launch(s)
        s.synchronize()
        be.exec_finish(e.frame)
        assert e.check("after twelve replayed Execs")
        # ordinary Execs of the same state afterwards, a larger one included
        e.n = e.frame.ctx.NumInstances = 390_000
        be.exec_ex(e.frame, opt)
        assert e.check("a larger Exec after the replays", tail=False)
    finally:
        if graph:
            graph.destroy()
        e.free()
        s.destroy()


def test_replays_across_the_wrap_of_the_look_back_tag():
    be = abi.load_hip_backend()
    cap = 600_000
    b = Emitter(be, 0, 250_000, True, 1, cap=cap)
    c = Emitter(be, 0, 40_000, True, 7, cap=cap)
    c.frame = b.state.frame([abi.DataSet(c.ins, c.outs, 0)], 40_000, [c.w.constants])          # C's streams through B's state
    s = hip.Stream()
    opt = abi.ExecOptions(0, 0, 0, s.value, abi.VVM_EXEC_NO_EIGHT_WIDE)
    saved = os.environ.get("VVM_DEBUG_SKIP")
    graph = None
    try:
        # B leaves its inclusive prefixes in 245 look-back words, tagged 3
        be._check(be.lib.vvm_debug_set_launch_epoch(b.state.ptr, 3, 0), "set epoch")
        be.exec_ex(b.frame, opt)
        assert b.check("B", tail=False)
        # W, two tiles, captured at tag 2^30 - 3 and replayed five times: tags 2^30 - 3, - 2, - 1, 1, 2; the next one is 3 again
        be._check(be.lib.vvm_debug_set_launch_epoch(b.state.ptr, LAST_EPOCH - 2, 0), "set epoch")
        b.n = b.frame.ctx.NumInstances = 2_000
        s.begin_capture(hip.hipStreamCaptureModeGlobal)
        be.exec_async(b.frame, opt)
        graph = s.end_capture()
        for _ in range(5):
            graph.launch(s)
        s.synchronize()
        be.exec_finish(b.frame)
        assert b.check("W", tail=False)
        # C: even tiles poll words that only B has written (odd tiles publish late); B's words carry C's tag unless they were cleared
        os.environ["VVM_DEBUG_SKIP"] = "1024"
        be.exec_ex(c.frame, opt)
        assert c.check("C", tail=False)
        os.environ.pop("VVM_DEBUG_SKIP", None)
        b.n = b.frame.ctx.NumInstances = 250_000
        be.exec_ex(b.frame, opt)
        assert b.check("B again", tail=False)
    finally:
        if saved is None:
            os.environ.pop("VVM_DEBUG_SKIP", None)
        else:
            os.environ["VVM_DEBUG_SKIP"] = saved
        if graph:
            graph.destroy()
        c.state.close()
        b.free()
        for p in c.ins + c.outs:
            be.lib.vvm_device_free(p)
        s.destroy()
