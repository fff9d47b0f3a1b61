"""vvm_batch_create / vvm_batch_exec: a batch kept over several ticks, with sizes, constants and streams changing in between;
every emitter's result equals its own Exec on the oracle at every tick."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import abi, scripts
from renderer_amd.runtime import VectorVMState
from vm_harness import PAD, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32
CAP = 320_000


class _Emitter:
    def __init__(self, be, e, n):
        self.be, self.n = be, n
        self.w = scripts.fountain(variant=e % 64)
        self.consts = bytearray(self.w.constants)
        self.host_in = scripts.fountain_streams_np(e * 1_000_003, CAP, kill_half=(e % 3 != 0))
        self.ins = [self._upload(a) for a in self.host_in]
        self.outs = [self._fresh() for _ in range(8)]
        self.state = VectorVMState(self.w.script.blob, be)
        self.ds = abi.DataSet(self.ins, self.outs, 0)
        self.frame = self.state.frame([self.ds], n, [bytes(self.consts)])

    def _upload(self, a):
        buf = np.zeros(CAP + PAD, f32)
        buf[:CAP] = a
        p = self.be.lib.vvm_device_alloc(buf.nbytes)
        assert p
        self.be._check(self.be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
        return p

    def _fresh(self):
        p = self.be.lib.vvm_device_alloc((CAP + PAD) * 4)
        assert p
        self.be._check(self.be.lib.vvm_device_memset(p, 0xCD, (CAP + PAD) * 4), "memset")
        return p

    def set_constants(self, blob):
        """New contents in the constant table the call frame points to (same address)."""
        self.consts[:] = blob
        C.memmove(self.frame._const_bufs[0], bytes(blob), len(blob))

    def swap_outputs(self):
        """Fresh output streams, written into the pointer array the dataset meta refers to."""
        self.old_outs, self.outs = self.outs, [self._fresh() for _ in range(8)]
        for k, p in enumerate(self.outs):
            self.ds._out[k] = p

    def check(self, tag):
        ref, ref_n, _ = run_oracle(self.w.script, [bytes(self.consts)], [a[:max(self.n, 1)] for a in self.host_in], [f32] * 8, self.n)
        assert self.frame.survivors(0) == ref_n, tag
        for k in range(8):
            h = np.empty(CAP + PAD, np.uint32)
            self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], h.nbytes), "download")
            assert np.array_equal(h[:ref_n], ref[k][:ref_n].view(np.uint32)), (tag, k)

    def free(self):
        self.state.close()
        for p in self.ins + self.outs + getattr(self, "old_outs", []):
            self.be.lib.vvm_device_free(p)


def test_a_batch_kept_over_ticks():
    be = abi.load_hip_backend()
    sizes = [0, 1, 1023, 1024, 1025, 5000, 300_001, 70_000] + [2000 + 977 * e for e in range(30)]
    ems = [_Emitter(be, e, n) for e, n in enumerate(sizes)]
    handle = None
    try:
        arr, oarr = be.pack_batch([em.frame for em in ems], [abi.ExecOptions(0, 0, 0, None, 0) for _ in ems])
        handle = be.batch_create(arr, oarr)

        def tick(changed, tag):
            be.batch_exec(handle, changed)
            for i, em in enumerate(ems):
                em.check("%s emitter %d" % (tag, i))

        tick(0, "first")
        assert ems[6].state.last_exec_width() == 8 and ems[5].state.last_exec_width() == 4
        tick(0, "again")
        # sizes change (tile counts move; nobody changes kernel)
        for i, em in enumerate(ems):
            if i % 3 == 1:
                em.n = max(0, em.n // 2 + i)
            elif i % 3 == 2:
                em.n = em.n + 3000 + i
            arr[i].NumInstances = em.n
        tick(abi.VVM_BATCH_CHANGED_INSTANCES, "sizes")
        # constants change
        for i, em in enumerate(ems):
            c = np.frombuffer(bytes(em.consts), f32).copy()
            c = np.where(np.abs(c) < 1.0, c * f32(1.5 + 0.01 * i), c).astype(f32)          # (the time step and the drag, whichever slot they sit in)
            em.set_constants(c.tobytes())
        tick(abi.VVM_BATCH_CHANGED_CONSTANTS, "constants")
        # one emitter grows past the eight-wide threshold, another one shrinks below it: the batch is planned again
        ems[7].n, ems[6].n = 290_000, 5_000          # (in a batch, scripts that fit the light kernel run eight-wide from 8 Ki instances on)
        arr[7].NumInstances, arr[6].NumInstances = 290_000, 5_000
        tick(abi.VVM_BATCH_CHANGED_INSTANCES, "kernel change")
        assert ems[7].state.last_exec_width() == 8 and ems[6].state.last_exec_width() == 4
        # new output streams
        for em in ems:
            em.swap_outputs()
        tick(abi.VVM_BATCH_CHANGED_STREAMS, "streams")
        # everything at once, then an unchanged tick
        for i, em in enumerate(ems):
            em.n = min(CAP, em.n + 17)
            arr[i].NumInstances = em.n
            em.swap_outputs()
        tick(abi.VVM_BATCH_CHANGED_INSTANCES | abi.VVM_BATCH_CHANGED_STREAMS | abi.VVM_BATCH_CHANGED_CONSTANTS, "all")
        tick(0, "unchanged")
        # the states are free for other calls between ticks
        be.exec_ex(ems[5].frame)
        tick(0, "after a plain Exec")
    finally:
        if handle:
            be.batch_destroy(handle)
        for em in ems:
            em.free()


def test_a_batch_of_eight_wide_emitters():
    """Every member forced onto the eight-wide kernels: the members whose scripts fit the light kernel share one launch, so a
    workgroup finishes one emitter's tile (its two outputs that wait in registers included) while it runs the next emitter's
    first tile; the members kept off it share a launch of the general kernel.  All against the oracle, over two ticks."""
    be = abi.load_hip_backend()
    sizes = [1, 2047, 2048, 2049, 4096, 6000, 10_000, 33_333, 70_001, 5, 3000, 12_289] * 3
    ems = [_Emitter(be, e, n) for e, n in enumerate(sizes)]
    handle = None
    try:
        # (every third member is kept off the light kernel, so the batch has a launch of each eight-wide kernel)
        opts = [abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_FORCE_EIGHT_WIDE | (abi.VVM_EXEC_NO_LIGHT_KERNEL if i % 3 == 2 else 0)) for i in range(len(ems))]
        arr, oarr = be.pack_batch([em.frame for em in ems], opts)
        be.exec_batch_packed(arr, oarr)
        for i, em in enumerate(ems):
            em.check("plain batch, emitter %d" % i)
            assert em.state.last_exec_width() == 8
        assert all(em.state.info().Flags & 0x20000 for em in ems)      # (every variant of the script fits the light kernel)
        handle = be.batch_create(arr, oarr)
        for tick in range(2):
            for i, em in enumerate(ems):
                em.n = max(1, (em.n * 3) // 2 + 7 * i) if tick else em.n
                arr[i].NumInstances = em.n
            be.batch_exec(handle, abi.VVM_BATCH_CHANGED_INSTANCES)
            for i, em in enumerate(ems):
                em.check("handle tick %d, emitter %d" % (tick, i))
    finally:
        if handle:
            be.batch_destroy(handle)
        for em in ems:
            em.free()
