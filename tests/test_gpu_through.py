"""The lean form of a script with a wide record (what the script computes on the light eight-wide kernel, the streams it hands
through unchanged by a copy kernel behind it) against the oracle and against the same script run whole."""
import numpy as np
import pytest

import through_cases as sc
from renderer_amd import abi
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
EIGHT = abi.VVM_EXEC_FORCE_EIGHT_WIDE


def _compare(n, extra_float, extra_int, death, flags=EIGHT, touch_one=False, instance_offset=0, expect_lean=True, lean_from=1024, more_math=0, trig=False, lean_width=8):
    script = sc.wide_record_script(extra_float, extra_int, touch_one, more_math, trig)
    ins = sc.streams(n, extra_float, extra_int, death)
    dts = sc.out_dtypes(extra_float, extra_int)
    want, want_n, _ = run_oracle(script, [sc.CONSTANTS], ins, dts, n, instance_offset=instance_offset)
    width = []
    got, got_n, info = run_hip(script, [sc.CONSTANTS], ins, dts, n, instance_offset=instance_offset, flags=flags, last_width=width)
    assert bool(info.Flags & 0x80000) == expect_lean
    runs_lean = expect_lean and not (flags & abi.VVM_EXEC_NO_THROUGH_KERNEL) and ((flags & EIGHT) or n >= lean_from)
    if n > 0:
        whole_width = 8 if (extra_float + extra_int <= 8 and ((flags & EIGHT) or n >= 256 * 1024)) else 4      # (up to sixteen streams: the general eight-wide kernel)
        assert width == [lean_width if runs_lean else whole_width]
    assert got_n == want_n
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g[instance_offset:instance_offset + got_n].view(np.uint32), w[instance_offset:instance_offset + want_n].view(np.uint32))
        assert (g[instance_offset + got_n:].view(np.uint8) == 0xCD).all()
    return info


@pytest.mark.parametrize("n", [0, 1, 5, 2047, 2048, 2049, 70001])
@pytest.mark.parametrize("death", [0.0, 0.5, 1.0])
def test_wide_record_against_the_oracle(n, death):
    info = _compare(n, 14, 2, death)
    assert (info.Flags >> 24) == 17      # handed-through streams: the sixteen extra ones and the lifetime


def test_the_same_script_whole():
    _compare(70001, 14, 2, 0.5, flags=EIGHT | abi.VVM_EXEC_NO_THROUGH_KERNEL)
    _compare(1001, 14, 2, 0.5, flags=0)      # below the size from which the lean form runs: whole, four-wide


def test_a_handed_through_stream_that_the_arithmetic_reads_too():
    _compare(50000, 12, 0, 0.3, touch_one=True, instance_offset=7)


def test_large_and_by_its_own_choice():
    """From 1 Ki instances on the library takes the lean form by itself; below, the whole script."""
    _compare(3 * 1024 * 1024 + 5, 10, 0, 0.5, flags=0)
    _compare(70001, 10, 2, 0.25, flags=0)
    _compare(1024, 10, 0, 0.5, flags=0)
    _compare(1023, 10, 0, 0.5, flags=0)


def test_scripts_that_stay_whole():
    _compare(5000, 0, 0, 0.5, expect_lean=False)      # fits the light kernel as it is


def test_a_script_that_fits_the_general_eight_wide_kernel_goes_lean_later():
    """Fourteen streams: the whole script runs eight-wide on the general kernel, and the lean form takes over from 4 Mi instances."""
    _compare(5000, 6, 0, 0.5)                                        # forced
    _compare(300000, 6, 0, 0.5, flags=0, lean_from=4 * 1024 * 1024)   # whole, eight-wide by the library's choice
    _compare(4 * 1024 * 1024 + 3, 6, 0, 0.5, flags=0, lean_from=4 * 1024 * 1024)


@pytest.mark.parametrize("n,death", [(1, 0.0), (2049, 0.5), (70001, 0.3), (70001, 1.0), (600001, 0.5)])
def test_what_the_script_computes_needs_the_general_eight_wide_kernel(n, death):
    """Six of the extra streams are computed with: fourteen streams in the lean program, which then runs on the general eight-wide
    kernel (which leaves the ride words as the light one does); twelve more handed through."""
    info = _compare(n, 18, 1, death, more_math=6)
    assert (info.Flags >> 24) == 14


def test_more_than_eight_computed_outputs_flush_in_the_middle_of_a_tile():
    """Eight fountain outputs and six computed ones: two groups of staging slots in the lean program, twelve streams handed through."""
    _compare(70001, 18, 0, 0.5, more_math=6, instance_offset=3)


# ---- lean members of a batch: one launch of the light kernel for all programs, one launch of the copy kernel for all handed-through streams


class _WideEmitter:
    """An emitter with a wide record (through_cases.wide_record_script) with the interface of test_gpu_batch_handle._Emitter."""
    CAP = 90_000

    def __init__(self, be, e, n, extra_float, extra_int, cap=None):
        if cap:
            self.CAP = cap
        from renderer_amd.runtime import VectorVMState
        from vm_harness import PAD
        self.be, self.n, self.PAD = be, n, PAD
        self.script = sc.wide_record_script(extra_float, extra_int)
        self.dts = sc.out_dtypes(extra_float, extra_int)
        self.consts = bytearray(sc.CONSTANTS)
        self.host_in = sc.streams(self.CAP, extra_float, extra_int, death=(0.0, 0.5, 0.9)[e % 3], seed=100 + e)
        self.ins = [self._upload(a) for a in self.host_in]
        self.outs = [self._fresh() for _ in self.dts]
        self.state = VectorVMState(self.script.blob, be)
        nf = 8 + extra_float
        self.ds = abi.DataSet(self.ins, self.outs, 0, [0, nf, nf + extra_int], [0, nf, nf + extra_int])
        self.frame = self.state.frame([self.ds], n, [bytes(self.consts)])

    def _upload(self, a):
        buf = np.zeros(self.CAP + self.PAD, a.dtype)
        buf[:self.CAP] = a
        p = self.be.lib.vvm_device_alloc(buf.nbytes)
        assert p
        self.be._check(self.be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
        return p

    def _fresh(self):
        p = self.be.lib.vvm_device_alloc((self.CAP + self.PAD) * 4)
        assert p
        self.be._check(self.be.lib.vvm_device_memset(p, 0xCD, (self.CAP + self.PAD) * 4), "memset")
        return p

    def swap_outputs(self):
        self.old_outs = getattr(self, "old_outs", []) + self.outs
        self.outs = [self._fresh() for _ in self.dts]
        for k, p in enumerate(self.outs):
            self.ds._out[k] = p

    def check(self, tag, fresh=False):
        """fresh: the output streams have not been written before this Exec, so nothing may lie behind the survivors."""
        ref, ref_n, _ = run_oracle(self.script, [bytes(self.consts)], [a[:max(self.n, 1)] for a in self.host_in], self.dts, self.n)
        assert self.frame.survivors(0) == ref_n, tag
        for k in range(len(self.dts)):
            h = np.empty(self.CAP + self.PAD, np.uint32)
            self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], h.nbytes), "download")
            assert np.array_equal(h[:ref_n], ref[k][:ref_n].view(np.uint32)), (tag, k)
            assert not fresh or (h[ref_n:self.CAP] == 0xCDCDCDCD).all(), (tag, k, "past the survivors")

    def free(self):
        self.state.close()
        for p in self.ins + self.outs + getattr(self, "old_outs", []):
            self.be.lib.vvm_device_free(p)


def test_lean_members_of_a_batch():
    """Wide-record emitters beside fountain emitters in vvm_exec_batch and in a vvm_batch handle over ticks: from 1 Ki instances on a
    wide-record member runs lean (its program in the light kernel's launch, its handed-through streams in the batch's one
    copy-kernel launch), below it whole on the four-wide kernel.  Sizes cross that line between ticks, outputs are swapped."""
    from test_gpu_batch_handle import _Emitter
    be = abi.load_hip_backend()
    wide_sizes = [20_000, 1_024, 1_023, 70_001, 1, 33_333, 2_049, 50_000, 900]
    ems = []
    for e, n in enumerate(wide_sizes):
        ems.append(_WideEmitter(be, e, n, 10 + (e % 3) * 4, e % 3))
        ems.append(_Emitter(be, e, 3000 + 4000 * e))
    handle = None
    try:
        opts = [abi.ExecOptions(0, 0, 0, None, 0) for _ in ems]
        arr, oarr = be.pack_batch([em.frame for em in ems], opts)
        be.exec_batch_packed(arr, oarr)
        for i, em in enumerate(ems):
            em.check("plain batch, emitter %d" % i, *([True] if isinstance(em, _WideEmitter) else []))
        widths = [em.state.last_exec_width() for em in ems[0::2]]
        assert widths == [8 if n >= 1024 else 4 for n in wide_sizes]
        handle = be.batch_create(arr, oarr)
        for tick in range(4):
            changed = 0
            if tick in (1, 3):      # sizes: some cross the line in either direction, some only change their tile counts
                for i, em in enumerate(ems):
                    em.n = [em.n, max(1, em.n // 2 + 13 * i), em.n, min(em.CAP if hasattr(em, "CAP") else 320_000, em.n * 3 + 100)][tick]
                    arr[i].NumInstances = em.n
                changed |= abi.VVM_BATCH_CHANGED_INSTANCES
            if tick == 2:
                for em in ems:
                    em.swap_outputs()
                changed |= abi.VVM_BATCH_CHANGED_STREAMS
            be.batch_exec(handle, changed)
            for i, em in enumerate(ems):
                em.check("handle tick %d, emitter %d" % (tick, i), *([tick == 2] if isinstance(em, _WideEmitter) else []))
        # a tick in which only tile counts change (nobody changes kernel): every wide member already above the line grows a little
        for i, em in enumerate(ems):
            if em.n >= 1024:
                em.n = min(em.n + 2048 + 7 * i, getattr(em, "CAP", 320_000))
                arr[i].NumInstances = em.n
        be.batch_exec(handle, abi.VVM_BATCH_CHANGED_INSTANCES)
        for i, em in enumerate(ems):
            em.check("tile counts only, emitter %d" % i)
        be.batch_exec(handle, 0)
        for i, em in enumerate(ems):
            em.check("unchanged, emitter %d" % i)
    finally:
        if handle:
            be.batch_destroy(handle)
        for em in ems:
            em.free()


def _two_dataset_wide_script(extra):
    """The fountain update on a wide record, with a second dataset that receives position and age of the particles that die."""
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    dt, drag = s.const(0), s.const(4)
    pos = [s.input(0, FLOAT, k) for k in range(3)]
    vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
    age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
    rest = [s.input(0, FLOAT, 8 + k) for k in range(extra)]
    age1 = s.temp(); s.op("add", age, dt, age1)
    alive, dead = s.temps(2)
    s.op("cmplt", age1, life, alive)
    s.op("cmpge", age1, life, dead)
    s.acquireindex(alive, 0)
    s.acquireindex(dead, 1)
    v1, p1 = s.temps(3), s.temps(3)
    for c in range(3):
        s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
        s.op("mad", v1[c], dt, pos[c], p1[c])
    s.output_data(FLOAT, 0, [(p1[c], s.output(0, FLOAT, c)) for c in range(3)] + [(v1[c], s.output(0, FLOAT, 3 + c)) for c in range(3)] +
                  [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))] + [(rest[k], s.output(0, FLOAT, 8 + k)) for k in range(extra)])
    s.output_data(FLOAT, 1, [(p1[c], s.output(1, FLOAT, c)) for c in range(3)] + [(age1, s.output(1, FLOAT, 3)), (rest[0], s.output(1, FLOAT, 4))])
    return s.build()


@pytest.mark.parametrize("n,death", [(3, 0.5), (2049, 0.0), (70001, 0.5), (70001, 1.0), (300001, 0.3)])
def test_a_wide_record_with_a_second_dataset(n, death):
    """Two acquired datasets in one pass of the general eight-wide kernel; the first dataset's handed-through streams go to the copy
    kernel (it works from the first chain's ranks and look-back words), the second dataset's outputs, one of them an input handed
    through, stay in the program."""
    extra = 16
    script = _two_dataset_wide_script(extra)
    ins = sc.streams(n, extra, 0, death)
    dts = [np.float32] * (8 + extra)
    more = [[np.float32] * 5]
    want_more, got_more = [], []
    want, want_n, _ = run_oracle(script, [sc.CONSTANTS], ins, dts, n, more_datasets=more, more_results=want_more)
    width = []
    got, got_n, info = run_hip(script, [sc.CONSTANTS], ins, dts, n, flags=EIGHT, last_width=width, more_datasets=more, more_results=got_more)
    assert info.Flags & 0x80000 and (info.Flags >> 24) == extra + 1 and info.Flags & 0x40000 and width == [8]
    assert got_n == want_n and got_more[0][1] == want_more[0][1] == n - want_n
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g[:got_n].view(np.uint32), w[:want_n].view(np.uint32))
        assert (g[got_n:].view(np.uint8) == 0xCD).all()
    for g, w in zip(got_more[0][0], want_more[0][0]):
        np.testing.assert_array_equal(g[:n - want_n].view(np.uint32), w[:n - want_n].view(np.uint32))


def test_a_compiler_format_wide_record_script_through_the_optimizer():
    """Compiler-format bytecode (all attributes read at the top, all written at the bottom) -> vvm_optimize_script -> the blob runs
    lean on the GPU and whole on the oracle: same bits."""
    from types import SimpleNamespace
    from renderer_amd.assembler import FLOAT
    from renderer_amd.original import K, OriginalScript
    o = OriginalScript()
    attrs = [o.input(0, FLOAT, i, o.temp()) for i in range(24)]
    px, py, pz, vx, vy, vz, age, life = attrs[:8]
    dt, g, drag = K(0), [K(1), K(2), K(3)], K(4)
    age2 = o.temp(); o.op("add", age, dt, age2)
    alive = o.temp(); o.op("cmplt", age2, life, alive)
    vel, pos = [], []
    for v, p, gi in zip((vx, vy, vz), (px, py, pz), g):
        t = o.temp(); o.op("mad", gi, dt, v, t)
        v2 = o.temp(); o.op("mul", t, drag, v2)
        p2 = o.temp(); o.op("mad", v2, dt, p, p2)
        vel.append(v2); pos.append(p2)
    idx = o.temp(); o.acquireindex(alive, idx, 0)
    for i, r in enumerate(pos + vel + [age2] + attrs[7:]):
        o.output(FLOAT, 0, idx, r, i)
    script = SimpleNamespace(blob=abi.load_hip_backend().optimize_script(o.done(), o.ext))
    n = 150_001
    ins = sc.streams(n, 16, 0, 0.4)
    dts = [np.float32] * 24
    want, want_n, _ = run_oracle(script, [sc.CONSTANTS], ins, dts, n)
    width = []
    got, got_n, info = run_hip(script, [sc.CONSTANTS], ins, dts, n, last_width=width)
    assert info.Flags & 0x80000 and width == [8] and got_n == want_n
    for g_, w_ in zip(got, want):
        np.testing.assert_array_equal(g_[:got_n].view(np.uint32), w_[:want_n].view(np.uint32))


@pytest.mark.parametrize("n,death", [(1, 0.0), (1023, 0.5), (1025, 0.5), (70001, 0.3), (70001, 1.0), (70001, 0.0), (600001, 0.5)])
def test_what_the_script_computes_needs_the_transcendental_functions(n, death):
    """A sine and an exponential in what the script computes: the lean program runs four instances per lane on the kernel that has
    them (sixteen registers), which leaves the ride words in its own layout, and the copy kernel works on its tiles of 1024."""
    info = _compare(n, 16, 1, death, trig=True, lean_width=4, flags=0, lean_from=1 if n >= 1024 else 10 ** 9)
    assert (info.Flags >> 24) == 18


@pytest.mark.parametrize("n,death", [(5000, 0.5), (300001, 0.2)])
def test_a_record_with_more_streams_than_one_launch_of_the_copy_kernel_takes(n, death):
    """120 handed-through streams: the whole script would run one instance per lane; lean, the copy kernel is launched three times."""
    info = _compare(n, 110, 9, death, flags=0)
    assert (info.Flags >> 24) == 120


@pytest.mark.parametrize("n,death,offset", [(1, 0.0, 0), (2047, 0.5, 1), (70001, 0.5, 3), (70001, 0.0, 0), (70001, 1.0, 5), (400001, 0.25, 7)])
def test_sixteen_bit_streams_are_handed_through_too(n, death, offset):
    """Three half streams among the handed-through ones, at odd instance offsets (the streams are aligned to their elements only)."""
    script = sc.wide_record_script(12, 1, extra_half=3)
    ins = sc.streams(n, 12, 1, death, extra_half=3)
    dts = sc.out_dtypes(12, 1, 3)
    want, want_n, _ = run_oracle(script, [sc.CONSTANTS], ins, dts, n, instance_offset=offset)
    width = []
    got, got_n, info = run_hip(script, [sc.CONSTANTS], ins, dts, n, instance_offset=offset, flags=EIGHT, last_width=width)
    assert info.Flags & 0x80000 and (info.Flags >> 24) == 17 and width == [8] and got_n == want_n
    for g, w, dt in zip(got, want, dts):
        view = np.uint16 if dt == np.float16 else np.uint32
        np.testing.assert_array_equal(g[offset:offset + got_n].view(view), w[offset:offset + want_n].view(view))
        assert (g[offset + got_n:].view(np.uint8) == 0xCD).all()


def test_lean_members_in_every_slice_of_a_large_batch():
    """1100 emitters in one vvm_exec_batch call (three slices that are filled, uploaded and launched one after the other), every
    fourth one with a wide record: each slice has its own row of blocks for the copy kernel and its own launch of it."""
    be = abi.load_hip_backend()
    ems = []
    try:
        for e in range(1100):
            n = 1 + (e * 37) % 2900
            # (the others run the same update on the plain eight-stream record, which fits the light kernel whole)
            ems.append(_WideEmitter(be, e, n, 10, 1, cap=3000) if e % 4 == 1 else _WideEmitter(be, e, n, 0, 0, cap=3000))
    except AssertionError:
        for em in ems:
            em.free()
        raise
    try:
        arr, oarr = be.pack_batch([em.frame for em in ems], [abi.ExecOptions(0, 0, 0, None, 0) for _ in ems])
        be.exec_batch_packed(arr, oarr)
        for i in list(range(0, 1100, 7)) + [1, 5, 1097]:
            ems[i].check("emitter %d" % i)
        lean = [em.state.last_exec_width() == 8 for em in ems[1::4]]
        assert lean == [em.n >= 1024 for em in ems[1::4]] and any(lean) and not all(lean)
    finally:
        for em in ems:
            em.free()
