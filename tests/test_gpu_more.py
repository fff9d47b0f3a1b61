"""GPU parity beyond the single-op cases: long look-back chains, untouched tails, the transcendental script with the noise
built-ins, batched Execs with distinct bytecode, asynchronous Execs, and full-size (64 Mi) properties.
"""
import ctypes as C
import os

import numpy as np
import pytest

from op_cases import ALL_ONES
from renderer_amd import abi, scripts
from renderer_amd.assembler import FLOAT, Script
from renderer_amd.runtime import VectorVMState
from vm_harness import PAD, run_hip, run_oracle, ulp_diff

pytestmark = pytest.mark.gpu
f32 = np.float32


def _bits(a):
    return a.view(np.uint32)


# ---- compaction at scale -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pattern", ["half", "sparse", "dense", "bursts", "none_alive", "all_alive"])
def test_long_look_back_chains(pattern):
    """About 2000 tiles: many more than one look-back window, with death patterns that make tile counts vary wildly."""
    w = scripts.fountain()
    n = 2_000_003
    ins = scripts.fountain_streams_np(0, n)
    idx = np.arange(n)
    rng = np.random.default_rng(11)
    dead = {"half": rng.random(n) < 0.5, "sparse": rng.random(n) < 0.999, "dense": rng.random(n) < 0.001,
            "bursts": (idx // 37123) % 3 != 0, "none_alive": np.ones(n, bool), "all_alive": np.zeros(n, bool)}[pattern]
    ins[6] = np.where(dead, ins[7], ins[6] * f32(0.0)).astype(f32)
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
    got, got_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n)
    assert got_n == ref_n == int((~dead).sum())
    for k in range(8):
        assert np.array_equal(_bits(got[k][:got_n]), _bits(ref[k][:ref_n])), "stream %d" % k
        assert np.all(_bits(got[k][got_n:]) == 0xCDCDCDCD), "stream %d written past its survivors" % k


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 5000])
def test_nothing_is_written_past_the_survivors(n):
    w = scripts.fountain()
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    for off in (0, 3):
        got, got_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, instance_offset=off)
        for k in range(8):
            assert np.all(_bits(got[k][:off]) == 0xCDCDCDCD) and np.all(_bits(got[k][off + got_n:]) == 0xCDCDCDCD)


def test_more_outputs_than_staging_slots():
    """Twelve output streams: the kernel stages eight per flush and writes the rest in a second one."""
    s = Script()
    x = [s.input(0, FLOAT, i) for i in range(3)]
    keep = s.temp(); s.op("cmplt", x[0], x[1], keep)
    s.acquireindex(keep, 0)
    outs = []
    for i in range(12):
        t = s.temp()
        s.op("mad", x[i % 3], s.const(i % 4), x[(i + 1) % 3], t)
        outs.append(t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, i)) for i, t in enumerate(outs)])
    built = s.build()
    consts = np.array([0.5, 1.5, -2.0, 3.25], f32).tobytes()
    n = 7001
    rng = np.random.default_rng(2)
    ins = [rng.standard_normal(n).astype(f32) for _ in range(3)]
    ref, ref_n, _ = run_oracle(built, [consts], ins, [f32] * 12, n)
    got, got_n, _ = run_hip(built, [consts], ins, [f32] * 12, n)
    assert got_n == ref_n
    for k in range(12):
        assert np.array_equal(_bits(got[k][:got_n]), _bits(ref[k][:ref_n])), k
        assert np.all(_bits(got[k][got_n:]) == 0xCDCDCDCD)


def test_outputs_without_acquireindex_write_nothing():
    """No acquireindex for the dataset: the reference's output handlers see zero instances (VectorVMRuntime.cpp:565, 769)."""
    s = Script()
    t = s.temp()
    s.op("add", s.input(0, FLOAT, 0), s.const(0), t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
    built = s.build()
    n = 3000
    x = np.ones(n, f32)
    ref, ref_n, _ = run_oracle(built, [bytes(16)], [x], [f32], n)
    got, got_n, _ = run_hip(built, [bytes(16)], [x], [f32], n)
    assert got_n == ref_n == 0
    assert np.all(_bits(got[0]) == 0xCDCDCDCD)


# ---- the transcendental script -------------------------------------------------------------------------------------------------
def test_transcendental_script_with_curl_noise():
    """sin_cos, exp and the curl-noise built-in.  The kernel evaluates the same polynomials and the same Perlin lattice code in
    the oracle's operation order, so the bound is tight: <= 2 ulp on the displaced position (observed 0)."""
    w = scripts.transcendental(True)
    n = 100_003
    ins = scripts.transcendental_streams_np(0, n)
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=list(w.ext_fns))
    got, got_n, info = run_hip(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=list(w.ext_fns))
    assert got_n == ref_n == n
    worst = 0
    for k in range(4):
        d = ulp_diff(got[k][:n], ref[k][:n])
        worst = max(worst, int(d.max()))
    assert worst <= 2, "max ulp distance %d" % worst


def test_perlin_builtin():
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = [s.input(0, FLOAT, i) for i in range(3)]
    t = s.temp()
    f = s.ext_fn(3, 1)
    s.call(f, x, [t])
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
    built = s.build()
    n = 20000
    rng = np.random.default_rng(4)
    ins = [(rng.standard_normal(n) * 40).astype(f32) for _ in range(3)]
    ins[0][:4] = [0.0, 1.0, -3.0, 255.5]
    ref, _, _ = run_oracle(built, [ALL_ONES], ins, [f32], n, ext_fns=[abi.VVM_EXTFN_PERLIN3])
    got, got_n, _ = run_hip(built, [ALL_ONES], ins, [f32], n, ext_fns=[abi.VVM_EXTFN_PERLIN3])
    assert got_n == n
    assert ulp_diff(got[0][:n], ref[0][:n]).max() <= 2


def test_perlin_builtin_on_coordinates_outside_the_int32_range():
    """The cell index is (int32)floor(p) & 255 with the x86 cast (UnrealMath.cpp:3665-3667): NaN and what does not fit land in
    cell 0.  The device's own conversion saturates, which would pick cell 255 above the range."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = [s.input(0, FLOAT, i) for i in range(3)]
    t = s.temp()
    f = s.ext_fn(3, 1)
    s.call(f, x, [t])
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
    built = s.build()
    special = np.array([3e9, -3e9, 1e30, -1e30, np.inf, -np.inf, np.nan, 2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0,
                        4294967296.0, 1e10, 2.5e9, 16777216.0, -16777217.0], f32)
    n = 3 * len(special) * 64
    rng = np.random.default_rng(11)
    ins = [(rng.standard_normal(n) * 40).astype(f32) for _ in range(3)]
    for axis in range(3):                                                 # every special value on every axis, the others ordinary
        for k, v in enumerate(special):
            ins[axis][(axis * len(special) + k) * 64:(axis * len(special) + k + 1) * 64:2] = v
    ins[0][1::7] = special[np.arange(len(ins[0][1::7])) % len(special)]    # and mixed on several axes at once
    ins[1][1::7] = special[(np.arange(len(ins[1][1::7])) * 5 + 3) % len(special)]
    ref, _, _ = run_oracle(built, [ALL_ONES], ins, [f32], n, ext_fns=[abi.VVM_EXTFN_PERLIN3])
    got, got_n, _ = run_hip(built, [ALL_ONES], ins, [f32], n, ext_fns=[abi.VVM_EXTFN_PERLIN3])
    assert got_n == n
    assert np.array_equal(np.isnan(got[0][:n]), np.isnan(ref[0][:n]))
    assert ulp_diff(got[0][:n], ref[0][:n]).max() <= 2


def test_host_callback_is_refused_by_the_gpu_executor():
    w = scripts.transcendental(True)
    n = 16
    ins = scripts.transcendental_streams_np(0, n)
    with pytest.raises(abi.VVMError) as e:
        run_hip(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=[lambda ctx: None])
    assert e.value.code == abi.VVM_E_UNSUPPORTED_OP


# ---- batched and asynchronous Execs -------------------------------------------------------------------------------------------
class _DeviceStreams:
    def __init__(self, be, arrays, out_count, n, fill=0xCD):
        self.be, self.n = be, n
        self.ins, self.outs = [], []
        for a in arrays:
            buf = np.zeros(n + PAD, a.dtype); buf[:n] = a[:n]
            p = be.lib.vvm_device_alloc(buf.nbytes); assert p
            be._check(be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
            self.ins.append(p)
        for _ in range(out_count):
            p = be.lib.vvm_device_alloc((n + PAD) * 4); assert p
            be._check(be.lib.vvm_device_memset(p, fill, (n + PAD) * 4), "memset")
            self.outs.append(p)

    def download(self, k):
        h = np.empty(self.n + PAD, f32)
        self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], h.nbytes), "download")
        return h

    def free(self):
        for p in self.ins + self.outs:
            self.be.lib.vvm_device_free(p)


def test_batched_execs_with_distinct_bytecode():
    """One launch for many emitters, each with its own script variant, constants and size; same results as one Exec each."""
    be = abi.load_hip_backend()
    emitters = 48
    # (two emitters are large enough for the eight-instances-per-lane kernel: the batch then runs as two groups, one per kernel)
    sizes = [0, 1, 7, 1024, 1025, 3000, 300_001, 262_144] + [2000 + 97 * e for e in range(emitters - 8)]
    states, frames, streams, refs, opts = [], [], [], [], []
    try:
        for e in range(emitters):
            w = scripts.fountain(variant=e % 64)
            n = sizes[e]
            ins = scripts.fountain_streams_np(e * 100000, max(n, 1), kill_half=(e % 2 == 0))
            ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
            refs.append((ref, ref_n))
            ds = _DeviceStreams(be, ins, 8, max(n, 1))
            streams.append(ds)
            st = VectorVMState(w.script.blob, be)
            states.append(st)
            frames.append(st.frame([abi.DataSet(ds.ins, ds.outs, 0)], n, [w.constants]))
            opts.append(abi.ExecOptions(0, 0, 0, None, 0))
        be.exec_batch(frames, opts)
        for e in range(emitters):
            ref, ref_n = refs[e]
            assert frames[e].survivors(0) == ref_n, "emitter %d" % e
            for k in range(8):
                got = streams[e].download(k)
                assert np.array_equal(_bits(got[:ref_n]), _bits(ref[k][:ref_n])), "emitter %d stream %d" % (e, k)
                assert np.all(_bits(got[ref_n:]) == 0xCDCDCDCD)
        # a second batch on the same states re-uses the launch words
        be.exec_batch(frames, opts)
        for e in range(emitters):
            assert frames[e].survivors(0) == refs[e][1]
    finally:
        for st in states:
            st.close()
        for ds in streams:
            ds.free()


def test_large_batch_is_launched_in_slices():
    """1100 emitters: the batch is cut into two slices that are filled and launched one after the other; every emitter's result
    equals its own Exec on the oracle.  The streams of all emitters are ranges of one device allocation per attribute."""
    import torch
    be = abi.load_hip_backend()
    dev = torch.device("cuda", 0)
    emitters = 1100
    sizes = [(e * 37) % 500 for e in range(emitters)]
    sizes[700] = 270_000                       # one emitter on the eight-wide kernel, in the second slice
    starts, total = [], 0
    for n in sizes:
        starts.append(total)
        total += (n + PAD + 63) // 64 * 64
    host_in = scripts.fountain_streams_np(7, total, kill_half=True)
    d_in = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in host_in]
    d_out = [torch.full((total,), -3.0, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    states, frames = [], []
    try:
        for e in range(emitters):
            w = scripts.fountain(variant=e % 64)
            st = VectorVMState(w.script.blob, be)
            states.append(st)
            off = starts[e] * 4
            frames.append(st.frame([abi.DataSet([t.data_ptr() + off for t in d_in], [t.data_ptr() + off for t in d_out], 0)], sizes[e], [w.constants]))
        be.exec_batch(frames, [abi.ExecOptions(0, 0, 0, None, 0) for _ in range(emitters)])
        got = [t.cpu().numpy() for t in d_out]
        for e in range(0, emitters, 7):
            w = scripts.fountain(variant=e % 64)
            n = sizes[e]
            ins = [a[starts[e]:starts[e] + max(n, 1)] for a in host_in]
            ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
            assert frames[e].survivors(0) == ref_n, "emitter %d" % e
            for k in range(8):
                assert np.array_equal(_bits(got[k][starts[e]:starts[e] + ref_n]), _bits(ref[k][:ref_n])), "emitter %d stream %d" % (e, k)
        assert states[700].last_exec_width() == 8 and states[699].last_exec_width() == 4
    finally:
        for st in states:
            st.close()


def test_id_and_multi_dataset_scripts_inside_a_batch():
    """A batch may hold scripts with update_id / acquire_id and scripts that acquire several datasets: they run behind the batched
    kernels on the same stream, with the results of an Exec of their own; the plain emitters around them are unaffected."""
    import dataset_cases
    import id_cases
    from vm_harness import IdState
    be = abi.load_hip_backend()
    plain, states, streams, refs = [], [], [], []
    try:
        for e in range(4):
            w = scripts.fountain(variant=e)
            n = 3000 + 511 * e
            ins = scripts.fountain_streams_np(e * 7777, n, kill_half=True)
            refs.append(run_oracle(w.script, [w.constants], ins, [f32] * 8, n))
            ds = _DeviceStreams(be, ins, 8, n)
            streams.append(ds)
            st = VectorVMState(w.script.blob, be)
            states.append(st)
            plain.append(st.frame([abi.DataSet(ds.ins, ds.outs, 0)], n, [w.constants]))

        def check_plain():
            for e in range(4):
                ref, ref_n, _ = refs[e]
                assert plain[e].survivors(0) == ref_n
                for k in range(8):
                    assert np.array_equal(_bits(streams[e].download(k)[:ref_n]), _bits(ref[k][:ref_n]))

        # update_id
        n = 70001
        rng = np.random.default_rng(11)
        x = rng.random(n).astype(f32)
        ids = rng.permutation(n + 7)[:n].astype(np.int32)
        st0 = IdState(np.full(n + 7, -1, np.int32), np.full(2 * n + 16, -7, np.int32), num_free=5, max_used=3)
        st0.free_table[:5] = [100001, 100002, 100003, 100004, 100005]
        script, consts = id_cases.update_script(), [id_cases.update_constants(f32(0.5))]
        want, got = st0.copy(), st0.copy()
        ref, ref_n, _ = run_oracle(script, consts, [x, ids], [f32, np.int32], n, ids=want, page_kb=2)
        out, out_n, _ = run_hip(script, consts, [x, ids], [f32, np.int32], n, ids=got, page_kb=2, batch_with=plain)
        assert out_n == ref_n
        for a, b in zip(out, ref):
            assert np.array_equal(_bits(a[:out_n]), _bits(b[:ref_n]))
        assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
        assert (got.num_free, got.max_used) == (want.num_free, want.max_used)
        check_plain()

        # acquire_id
        m = 4099
        sp = IdState(np.zeros(0, np.int32), np.full(3 * m + 40, -3, np.int32), num_free=m + 11, max_used=0, num_spawned=6, tag=77)
        sp.free_table[:m + 11] = rng.permutation(10 * m + 100)[:m + 11]
        want, got = sp.copy(), sp.copy()
        sconsts = [bytes([255] * 4) + bytes(12)]
        ref, ref_n, _ = run_oracle(id_cases.spawn_script(), sconsts, [], [np.int32, np.int32], m, ids=want)
        out, out_n, _ = run_hip(id_cases.spawn_script(), sconsts, [], [np.int32, np.int32], m, ids=got, batch_with=plain)
        assert out_n == ref_n == m and np.array_equal(out[0][:m], ref[0][:m]) and np.array_equal(out[1][:m], ref[1][:m])
        assert np.array_equal(got.free_table, want.free_table) and (got.num_free, got.num_spawned) == (want.num_free, want.num_spawned)

        # two acquired datasets
        k = 5003
        xx, yy = rng.random(k).astype(f32), rng.random(k).astype(f32)
        two = dataset_cases.two_dataset_script(True, False)
        tconsts = [dataset_cases.constants(0.5, 0.7)]
        ref_more, got_more = [], []
        ref, ref_n, _ = run_oracle(two, tconsts, [xx, yy], [f32] * 2, k, more_datasets=[[f32, np.int32]], more_results=ref_more)
        out, out_n, _ = run_hip(two, tconsts, [xx, yy], [f32] * 2, k, more_datasets=[[f32, np.int32]], more_results=got_more, batch_with=plain)
        assert out_n == ref_n and got_more[0][1] == ref_more[0][1]
        for a, b in zip(out, ref):
            assert np.array_equal(_bits(a[:out_n]), _bits(b[:ref_n]))
        for a, b in zip(got_more[0][0], ref_more[0][0]):
            assert np.array_equal(a[:got_more[0][1]].view(np.uint32), b[:ref_more[0][1]].view(np.uint32))
        check_plain()
    finally:
        for st in states:
            st.close()
        for ds in streams:
            ds.free()


def test_async_execs_in_a_row():
    be = abi.load_hip_backend()
    w = scripts.fountain()
    n = 200_000
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
    ds = _DeviceStreams(be, ins, 8, n)
    st = VectorVMState(w.script.blob, be)
    try:
        frame = st.frame([abi.DataSet(ds.ins, ds.outs, 0)], n, [w.constants])
        for _ in range(12):                      # more Execs in flight than parameter slots
            st.exec_async(frame)
        assert st.finish(frame) == ref_n
        for k in range(8):
            assert np.array_equal(_bits(ds.download(k)[:ref_n]), _bits(ref[k][:ref_n]))
    finally:
        st.close()
        ds.free()


# ---- full size ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kill_half", [False, True])
def test_full_size_against_torch(kill_half):
    """64 Mi particles (the benchmark's size).  The oracle would take minutes here; the fountain script is four IEEE operations
    per output, which torch evaluates unfused on the device, and boolean-mask indexing keeps order: bit-exact expected values."""
    import torch
    n = 64 * 1024 * 1024
    dev = torch.device("cuda", 0)
    be = abi.load_hip_backend()
    w = scripts.fountain()
    ins = scripts.fountain_streams_torch(0, n, dev, kill_half=kill_half, pad=64)
    outs = [torch.full((n + 64,), -7.0, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    st = VectorVMState(w.script.blob, be)
    try:
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants])
        surv = st.exec(frame)
        dt, g, drag = torch.tensor(1.0 / 60.0, dtype=torch.float32), torch.tensor(-980.0, dtype=torch.float32), torch.tensor(0.98, dtype=torch.float32)
        age1 = ins[6][:n] + dt
        keep = age1 < ins[7][:n]
        assert surv == int(keep.sum().item())
        if not kill_half:
            assert surv == n
        for c in range(3):
            gc = g if c == 2 else torch.tensor(0.0)
            v1 = (gc.to(dev) * dt.to(dev) + ins[3 + c][:n]) * drag.to(dev)
            p1 = v1 * dt.to(dev) + ins[c][:n]
            assert torch.equal(outs[3 + c][:surv].view(torch.int32), v1[keep].view(torch.int32)), "velocity %d" % c
            assert torch.equal(outs[c][:surv].view(torch.int32), p1[keep].view(torch.int32)), "position %d" % c
        assert torch.equal(outs[6][:surv].view(torch.int32), age1[keep].view(torch.int32))
        assert torch.equal(outs[7][:surv].view(torch.int32), ins[7][:n][keep].view(torch.int32))
        for k in range(8):
            assert bool((outs[k][surv:] == -7.0).all()), "stream %d written past its survivors" % k
        # idempotence: a second Exec on the same inputs gives the same outputs
        first = [o[:surv].clone() for o in outs]
        assert st.exec(frame) == surv
        for k in range(8):
            assert torch.equal(first[k].view(torch.int32), outs[k][:surv].view(torch.int32))
    finally:
        st.close()


def test_streams_larger_than_4_gib():
    """1.2 G particles, half of them dying: every stream is 4.8 GB, so instance and output positions pass 2^30 and byte offsets
    pass 2^32.  Checked like the full-size test, one attribute at a time to bound the memory of the check."""
    import torch
    n = 1_200_000_123
    dev = torch.device("cuda", 0)
    be = abi.load_hip_backend()
    w = scripts.fountain()
    ins = scripts.fountain_streams_torch(0, n, dev, kill_half=True, pad=64)
    outs = [torch.full((n + 64,), -7.0, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    st = VectorVMState(w.script.blob, be)
    try:
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants])
        surv = st.exec(frame)
        dt, drag = torch.tensor(1.0 / 60.0, dtype=torch.float32, device=dev), torch.tensor(0.98, dtype=torch.float32, device=dev)
        age1 = ins[6][:n] + dt
        keep = age1 < ins[7][:n]
        assert surv == int(keep.sum().item()) and n // 3 < surv < 2 * (n // 3)
        assert torch.equal(outs[6][:surv].view(torch.int32), age1[keep].view(torch.int32))
        del age1
        assert torch.equal(outs[7][:surv].view(torch.int32), ins[7][:n][keep].view(torch.int32))
        for c in range(3):
            g = torch.tensor(-980.0 if c == 2 else 0.0, dtype=torch.float32, device=dev)
            v1 = (g * dt + ins[3 + c][:n]) * drag
            assert torch.equal(outs[3 + c][:surv].view(torch.int32), v1[keep].view(torch.int32)), "velocity %d" % c
            p1 = v1 * dt + ins[c][:n]
            del v1
            assert torch.equal(outs[c][:surv].view(torch.int32), p1[keep].view(torch.int32)), "position %d" % c
            del p1
        for k in range(8):
            assert bool((outs[k][surv:] == -7.0).all()), "stream %d written past its survivors" % k
    finally:
        st.close()


# ---- two datasets ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,secondary_first,many", [(1, False, False), (1000, False, False), (5003, True, False), (300007, False, True), (300007, True, False)])
@pytest.mark.parametrize("mode", ["chains", "fused", "separate"])
def test_two_datasets_match_the_oracle(n, secondary_first, many, mode, monkeypatch):
    """acquireindex on two datasets, each with its own mask, count and output positions.  chains: both in one pass of the kernel
    (the default for scripts like this one); fused / separate: one program per dataset, run tile by tile in one launch / as
    launches of their own (what scripts get whose kernels have no second chain)."""
    if mode != "chains":
        monkeypatch.setenv("VVM_NO_CHAINS", "1")      # (read when the state is allocated)
    flags = abi.VVM_EXEC_SEPARATE_PASSES if mode == "separate" else 0
    import dataset_cases
    rng = np.random.default_rng(n)
    x, y = rng.random(n).astype(f32), rng.random(n).astype(f32)
    script = dataset_cases.two_dataset_script(secondary_first, many)
    consts = [dataset_cases.constants(0.5, 0.7)]
    dt0 = [f32] * (11 if many else 2)
    ref_more, got_more = [], []
    ref, ref_n, _ = run_oracle(script, consts, [x, y], dt0, n, more_datasets=[[f32, np.int32]], more_results=ref_more)
    got, got_n, info = run_hip(script, consts, [x, y], dt0, n, more_datasets=[[f32, np.int32]], more_results=got_more, flags=flags)
    assert bool(info.Flags & 0x40000) == (mode == "chains")      # both datasets in one pass
    assert got_n == ref_n and got_more[0][1] == ref_more[0][1]
    for a, b in zip(got, ref):
        assert np.array_equal(_bits(a[:got_n]), _bits(b[:ref_n]))
        assert np.all(_bits(a[got_n:]) == 0xCDCDCDCD)
    m = got_more[0][1]
    for a, b in zip(got_more[0][0], ref_more[0][0]):
        assert np.array_equal(a[:m].view(np.uint32), b[:m].view(np.uint32))
        assert np.all(a[m:].view(np.uint32) == 0xCDCDCDCD)
    d0, d1 = dataset_cases.expected(x, y, 0.5, 0.7, many)
    assert got_n == len(d0[0]) and m == len(d1[0])


def test_a_later_pass_with_smaller_tiles_gets_its_own_look_back_words():
    """Dataset 0 needs two values, dataset 1 sixty at once: the first program runs four instances per lane, the second one, so the
    second pass has four times the tiles.  Its look-back words are sized for it (sized for the first pass they end short, and the
    words past the end land in whatever lies behind the array)."""
    s = Script()
    xs = [s.input(0, FLOAT, i) for i in range(60)]
    k0, k1 = s.temps(2)
    s.op("cmplt", xs[0], s.const(1), k0)
    s.acquireindex(k0, 0)
    s.output_data(FLOAT, 0, [(xs[1], s.output(0, FLOAT, 0))])
    s.op("cmplt", xs[1], s.const(2), k1)
    s.acquireindex(k1, 1)
    prods = []
    for x in xs:
        t = s.temp()
        s.op("mul", x, s.const(3), t)
        prods.append(t)
    acc = s.temp()
    s.op("add", prods[0], prods[1], acc)
    for t in prods[2:]:
        s.op("add", acc, t, acc)
    s.output_data(FLOAT, 1, [(acc, s.output(1, FLOAT, 0)), (prods[59], s.output(1, FLOAT, 1))])
    built = s.build()
    n = 300_000
    rng = np.random.default_rng(5)
    ins = [rng.random(n).astype(f32) for _ in range(60)]
    consts = [ALL_ONES[:4] + np.array([0.5, 0.7, 1.25], f32).tobytes()]
    for flags in (0, abi.VVM_EXEC_SEPARATE_PASSES):
        ref_more, got_more, widths = [], [], []
        ref, ref_n, _ = run_oracle(built, consts, ins, [f32], n, more_datasets=[[f32, f32]], more_results=ref_more)
        got, got_n, info = run_hip(built, consts, ins, [f32], n, more_datasets=[[f32, f32]], more_results=got_more, flags=flags, last_width=widths)
        assert info.InstancesPerLane == 4 and widths == [1]      # the first pass's program, the kernel of the last launch
        assert got_n == ref_n and got_more[0][1] == ref_more[0][1] and 0 < got_n < n
        assert np.array_equal(_bits(got[0][:got_n]), _bits(ref[0][:ref_n]))
        m = got_more[0][1]
        for a, b in zip(got_more[0][0], ref_more[0][0]):
            assert np.array_equal(a[:m].view(np.uint32), b[:m].view(np.uint32))


def test_internal_fields_follow_the_reference_chunking():
    """ctx->Internal.* is what the reference's chunking computes for the Exec (VectorVMRuntime.cpp:2210-2261), at any page size."""
    w = scripts.fountain(alias_outputs=False)
    n = 50_000
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    for page_kb in (64, 4):
        _, _, ref_frame = run_oracle(w.script, [w.constants], ins, [f32] * 8, n, page_kb=page_kb)
        be = abi.load_hip_backend()
        be.set_page_size_kb(page_kb)
        try:
            st = VectorVMState(w.script.blob, be)
            ptrs = [be.lib.vvm_device_alloc(4 * (n + PAD)) for _ in range(16)]
            frame = st.frame([abi.DataSet(ptrs[:8], ptrs[8:], 0)], n, [w.constants])
            for p, a in zip(ptrs[:8], ins):
                buf = np.zeros(n + PAD, f32); buf[:n] = a
                be._check(be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
            st.exec(frame)
            a, b = frame.ctx.Internal, ref_frame.ctx.Internal
            assert (a.MaxChunksPerBatch, a.MaxInstancesPerChunk, a.PerBatchRegisterDataBytesRequired, a.NumBytesRequiredPerBatch) == \
                   (b.MaxChunksPerBatch, b.MaxInstancesPerChunk, b.PerBatchRegisterDataBytesRequired, b.NumBytesRequiredPerBatch)
            if page_kb == 4:
                assert a.MaxChunksPerBatch > 1
            st.close()
            for p in ptrs:
                be.lib.vvm_device_free(p)
        finally:
            be.set_page_size_kb(64)


@pytest.mark.parametrize("block", range(3))
def test_random_scripts_in_the_optimized_format(block):
    """Scripts from the generator of tools/gpu_fuzz_linker.py (merged opcodes, two-output ops, transcendentals, random draws, half and
    no-advance inputs, up to 24 outputs of three kinds, dead values, up to 48 live streams: every kernel shape): GPU against oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gpu_fuzz_linker as fz
    for seed in range(5000 + 40 * block, 5000 + 40 * block + 40):
        built, nf, ni, nh, dts = fz.gen(seed)
        if fz.gen.with_ids:                 # (scripts with update_id need ID tables: tools/gpu_fuzz_linker.py runs those)
            continue
        n = 2048 + 1 + (seed * 37) % 3000
        rng = np.random.default_rng(seed)
        ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, n).astype(np.int32) for _ in range(ni)] + \
              [rng.uniform(-2, 2, n).astype(np.float16) for _ in range(nh)]
        consts = bytes([255] * 4) + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
        more = fz.gen.more_datasets          # (every fourth script acquires a second dataset, some of those a third and a fourth)
        ref_more, got_more = [], []
        ref = run_oracle(built, [consts], ins, dts, n, seed=seed, more_datasets=more, more_results=ref_more, ext_fns=list(fz.gen.ext))
        got = run_hip(built, [consts], ins, dts, n, seed=seed, flags=abi.VVM_EXEC_FORCE_EIGHT_WIDE if seed % 2 else 0,
                      more_datasets=more, more_results=got_more, ext_fns=list(fz.gen.ext))
        assert got[1] == ref[1], seed
        for d in range(len(more)):
            assert got_more[d][1] == ref_more[d][1], seed
            for a, b, dt in zip(got_more[d][0], ref_more[d][0], more[d]):
                a, b = a[:got_more[d][1]], b[:ref_more[d][1]]
                nan = (np.isnan(a) & np.isnan(b)) if dt != np.int32 else np.zeros(len(a), bool)
                assert np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan]), seed
        for k, dt in enumerate(dts):
            a, b = got[0][k][:got[1]], ref[0][k][:ref[1]]
            view = np.uint16 if dt == np.float16 else np.uint32
            nan = (np.isnan(a) & np.isnan(b)) if dt != np.int32 else np.zeros(len(a), bool)
            assert np.array_equal(a.view(view)[~nan], b.view(view)[~nan]), (seed, k)


def test_random_scripts_in_one_batch():
    """Thirty scripts of the random generator, every kernel shape among them, in ONE vvm_exec_batch call: each emitter's result equals
    its own Exec on the oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gpu_fuzz_linker as fz
    from vm_harness import _type_offsets
    be = abi.load_hip_backend()
    L = be.lib
    frames, opts, states, allocs, expect = [], [], [], [], []
    try:
        for seed in [v for v in range(8000, 8040) if v % 4 != 1]:          # (single-dataset scripts)
            built, nf, ni, nh, dts = fz.gen(seed)
            if fz.gen.with_ids:
                continue
            n = 1500 + (seed * 131) % 4000
            rng = np.random.default_rng(seed)
            ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, n).astype(np.int32) for _ in range(ni)] + \
                  [rng.uniform(-2, 2, n).astype(np.float16) for _ in range(nh)]
            consts = bytes([255] * 4) + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
            expect.append(run_oracle(built, [consts], ins, dts, n, seed=seed, ext_fns=list(fz.gen.ext)) + (dts, n))
            d_in, d_out = [], []
            for a in ins:
                buf = np.zeros(n + PAD, a.dtype); buf[:n] = a
                p = L.vvm_device_alloc(buf.nbytes); assert p
                be._check(L.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
                d_in.append(p)
            for dt in dts:
                p = L.vvm_device_alloc((n + PAD) * np.dtype(dt).itemsize); assert p
                be._check(L.vvm_device_memset(p, 0xCD, (n + PAD) * np.dtype(dt).itemsize), "memset")
                d_out.append(p)
            allocs.append((d_in, d_out))
            st = VectorVMState(built.blob, be)
            states.append(st)
            ds = abi.DataSet(d_in, d_out, 0, _type_offsets(ins), _type_offsets([np.empty(0, dt) for dt in dts]))
            frames.append(st.frame([ds], n, [consts], list(fz.gen.ext)))
            opts.append(abi.ExecOptions(seed, 0, 0, None, 0))
        be.exec_batch(frames, opts)
        shapes = set()
        for e, (ref, ref_n, _, dts, n) in enumerate(expect):
            assert frames[e].survivors(0) == ref_n, "emitter %d" % e
            info = states[e].info()
            shapes.add((info.InstancesPerLane, info.NumVirtualRegisters > 16, info.LdsSpillRegisters > 0))
            for k, dt in enumerate(dts):
                h = np.empty(n + PAD, dtype=dt)
                be._check(L.vvm_device_download(h.ctypes.data, allocs[e][1][k], h.nbytes), "download")
                a, b = h[:ref_n], ref[k][:ref_n]
                view = np.uint16 if dt == np.float16 else np.uint32
                nan = (np.isnan(a) & np.isnan(b)) if dt != np.int32 else np.zeros(len(a), bool)
                assert np.array_equal(a.view(view)[~nan], b.view(view)[~nan]), (e, k)
        assert len(shapes) >= 3, shapes
    finally:
        for st in states:
            st.close()
        for d_in, d_out in allocs:
            for p in d_in + d_out:
                L.vvm_device_free(p)


def test_noise_call_that_only_a_second_dataset_needs():
    """The first pass has no use for the noise call (it is not linked there); the second dataset's pass needs the noise table."""
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    x, y, z = [s.input(0, FLOAT, i) for i in range(3)]
    fn = s.ext_fn(3, 1)
    keep = s.temp(); s.op("cmplt", x, s.const(1), keep)
    s.acquireindex(s.const(0), 0)
    s.acquireindex(keep, 1)
    nz = s.temp(); s.call(fn, [x, y, z], [nz])
    s.output_data(FLOAT, 0, [(y, s.output(0, FLOAT, 0))])
    s.output_data(FLOAT, 1, [(nz, s.output(1, FLOAT, 0))])
    built = s.build()
    n = 5000
    rng = np.random.default_rng(2)
    ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(3)]
    consts = [bytes([255] * 4) + np.array([0.5], f32).tobytes()]
    ref_more, got_more = [], []
    ref = run_oracle(built, consts, ins, [f32], n, ext_fns=[abi.VVM_EXTFN_PERLIN3], more_datasets=[[f32]], more_results=ref_more)
    got = run_hip(built, consts, ins, [f32], n, ext_fns=[abi.VVM_EXTFN_PERLIN3], more_datasets=[[f32]], more_results=got_more)
    assert got[1] == ref[1] == n and got_more[0][1] == ref_more[0][1] and 0 < got_more[0][1] < n
    m = got_more[0][1]
    assert np.array_equal(_bits(got_more[0][0][0][:m]), _bits(ref_more[0][0][0][:m]))


def test_null_streams_and_short_constant_tables_are_refused_before_any_launch():
    be = abi.load_hip_backend()
    w = scripts.fountain()
    n = 1000
    ins = scripts.fountain_streams_np(0, n)
    ds = _DeviceStreams(be, ins, 8, n)
    st = VectorVMState(w.script.blob, be)
    try:
        ok = st.frame([abi.DataSet(ds.ins, ds.outs, 0)], n, [w.constants])
        assert st.exec(ok) == n
        broken = list(ds.ins); broken[3] = 0
        with pytest.raises(abi.VVMError) as e:
            st.exec(st.frame([abi.DataSet(broken, ds.outs, 0)], n, [w.constants]))
        assert e.value.code == abi.VVM_E_INVALID_ARG and "null" in str(e.value)
        with pytest.raises(abi.VVMError) as e:
            st.exec(st.frame([abi.DataSet(ds.ins, ds.outs, 0)], n, [w.constants[:8]]))
        assert e.value.code == abi.VVM_E_INVALID_ARG and "constant" in str(e.value)
        assert st.exec(ok) == n                              # the state is still usable
    finally:
        st.close()
        ds.free()
