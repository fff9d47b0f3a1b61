"""vvm_exec_noalloc (the entry point that allocates nothing) against vvm_exec_ex, through the C ABI: bit-identical streams and
counts on two calls in a row, the second with a smaller input, so that the second call finds the first one's words in the
device's scratch array."""
import numpy as np
import pytest

from renderer_amd import abi, scripts
from renderer_amd.runtime import VectorVMState
from vm_harness import PAD

pytestmark = pytest.mark.gpu
f32 = np.float32


class _Streams:
    def __init__(self, be, arrays, n, outs=8):
        self.be, self.n = be, n
        self.ins, self.outs = [], []
        for a in arrays:
            buf = np.zeros(n + PAD, f32)
            buf[:n] = a[:n]
            p = be.lib.vvm_device_alloc(buf.nbytes)
            assert p
            be._check(be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
            self.ins.append(p)
        for _ in range(outs):
            p = be.lib.vvm_device_alloc((n + PAD) * 4)
            assert p
            be._check(be.lib.vvm_device_memset(p, 0xCD, (n + PAD) * 4), "memset")
            self.outs.append(p)

    def download(self, k):
        h = np.empty(self.n + PAD, np.uint32)
        self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], h.nbytes), "download")
        return h

    def free(self):
        for p in self.ins + self.outs:
            self.be.lib.vvm_device_free(p)


def test_no_alloc_entry_point_equals_exec_ex():
    be = abi.load_hip_backend()
    w = scripts.fountain()
    st_ref, st_new = VectorVMState(w.script.blob, be), VectorVMState(w.script.blob, be)
    held = []
    try:
        for n, width in ((300_000, 8), (70_001, 4)):          # the second call is smaller, and runs on the other kernel width
            host = scripts.fountain_streams_np(n, n, kill_half=True)
            a, b = _Streams(be, host, n), _Streams(be, host, n)
            held += [a, b]
            fa = st_ref.frame([abi.DataSet(a.ins, a.outs, 0)], n, [w.constants])
            fb = st_new.frame([abi.DataSet(b.ins, b.outs, 0)], n, [w.constants])
            be.exec_ex(fa, abi.ExecOptions(7, 0, 0, None, 0))
            be.exec_noalloc(fb, abi.ExecOptions(7, 0, 0, None, 0))
            assert st_new.last_exec_width() == st_ref.last_exec_width() == width
            surv = fa.survivors(0)
            assert fb.survivors(0) == surv and 0 < surv < n
            for k in range(8):
                got, want = b.download(k), a.download(k)
                assert np.array_equal(got, want), "n = %d, stream %d" % (n, k)          # (the untouched tail included)
            ia, ib = fa.ctx.Internal, fb.ctx.Internal
            assert (ia.MaxChunksPerBatch, ia.MaxInstancesPerChunk) == (ib.MaxChunksPerBatch, ib.MaxInstancesPerChunk)
    finally:
        st_ref.close()
        st_new.close()
        for s in held:
            s.free()


def test_no_alloc_entry_point_refuses_what_does_not_fit():
    be = abi.load_hip_backend()
    w = scripts.fountain()
    st = VectorVMState(w.script.blob, be)
    s = _Streams(be, scripts.fountain_streams_np(0, 64), 64)
    try:
        frame = st.frame([abi.DataSet(s.ins, s.outs, 0)], 300_000_000, [w.constants])          # 146485 tiles of 2048: refused before any launch
        with pytest.raises(abi.VVMError) as e:
            be.exec_noalloc(frame)
        assert e.value.code == abi.VVM_E_TOO_LARGE and "131072" in str(e.value)
    finally:
        st.close()
        s.free()


def test_no_alloc_entry_point_with_two_datasets_in_one_pass():
    """A script that acquires two datasets (two chains, each with look-back words of its own in the scratch array) through both entry
    points: same counts and streams for both datasets."""
    import dataset_cases
    be = abi.load_hip_backend()
    blob = dataset_cases.two_dataset_script().blob
    consts = dataset_cases.constants(0.5, 0.7)
    st_ref, st_new = VectorVMState(blob, be), VectorVMState(blob, be)
    held = []
    try:
        assert st_new.info().Flags & 0x40000
        for n in (300_000, 9_001):
            rng = np.random.default_rng(n)
            host = [rng.random(n).astype(f32), rng.random(n).astype(f32)]
            a, b = _Streams(be, host, n, outs=4), _Streams(be, host, n, outs=4)
            held += [a, b]
            frames = []
            for st, s in ((st_ref, a), (st_new, b)):
                frames.append(st.frame([abi.DataSet(s.ins, s.outs[:2], 0), abi.DataSet([], s.outs[2:], 0, [0, 0, 0], [0, 1, 2])], n, [consts]))
            be.exec_ex(frames[0], abi.ExecOptions(0, 0, 0, None, 0))
            be.exec_noalloc(frames[1], abi.ExecOptions(0, 0, 0, None, 0))
            assert frames[0].survivors(0) == frames[1].survivors(0) == int((host[0] < f32(0.5)).sum())
            assert frames[0].survivors(1) == frames[1].survivors(1) == int((host[1] > f32(0.7)).sum())
            for k in range(4):
                assert np.array_equal(b.download(k), a.download(k)), "n = %d, stream %d" % (n, k)
    finally:
        st_ref.close()
        st_new.close()
        for s in held:
            s.free()


def test_no_alloc_entry_point_runs_the_lean_form_of_a_wide_record():
    """A wide record (twelve attributes handed through beside the fountain's eight): up to 4096 tiles the entry point runs the lean
    program and the copy kernel with the ride words in the scratch array, beyond that the whole script; streams and counts equal
    vvm_exec_ex's either way.  The second call is smaller and finds the first one's ride words in the array."""
    import through_cases as sc
    be = abi.load_hip_backend()
    blob = sc.wide_record_script(12, 0).blob
    st_ref, st_new = VectorVMState(blob, be), VectorVMState(blob, be)
    held = []
    try:
        assert st_new.info().Flags & 0x80000
        for n, width in ((300_000, 8), (70_001, 8), (4096 * 2048 + 1, 4)):
            host = sc.streams(n, 12, 0, 0.5, seed=n % 1000)
            a, b = _Streams(be, host, n, outs=20), _Streams(be, host, n, outs=20)
            held += [a, b]
            fa = st_ref.frame([abi.DataSet(a.ins, a.outs, 0)], n, [sc.CONSTANTS])
            fb = st_new.frame([abi.DataSet(b.ins, b.outs, 0)], n, [sc.CONSTANTS])
            be.exec_ex(fa, abi.ExecOptions(0, 0, 0, None, 0))
            be.exec_noalloc(fb, abi.ExecOptions(0, 0, 0, None, 0))
            assert st_ref.last_exec_width() == 8 and st_new.last_exec_width() == width
            surv = fa.survivors(0)
            assert fb.survivors(0) == surv and 0 < surv < n
            for k in range(20):
                assert np.array_equal(b.download(k), a.download(k)), "n = %d, stream %d" % (n, k)
            for s in (a, b):
                s.free()
            held = []
    finally:
        st_ref.close()
        st_new.close()
        for s in held:
            s.free()
