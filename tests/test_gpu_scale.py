"""GPU parity at the sizes and call patterns of the benchmark configurations that the smaller tests do not reach:
the full batched configuration (4096 emitters x 16 Ki particles, 64 bytecode variants, four slices), instance ranges sharded
on one device through libvvm_hip.so, and two large Execs of different states in flight at the same time."""
import os
import sys

import numpy as np
import pytest

from renderer_amd import abi, scripts
from renderer_amd.runtime import ShardedExec, VectorVMState, shard_range
from vm_harness import PAD, run_hip, run_oracle

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_sharded_gloo import _script_with_index_and_random  # noqa: E402

pytestmark = pytest.mark.gpu
f32 = np.float32


def _bits(a):
    return a.view(np.uint32)


def test_full_batched_configuration_against_the_oracle():
    """4096 emitters x 16 Ki particles, emitter e running fountain variant e mod 64, half of the particles dying: one
    vvm_exec_batch call (cut into four slices).  Every emitter's survivor count is checked against a count made with torch, and
    every 89th emitter's streams (47 emitters, all 64 variants... 47 of them) bit for bit against its own Exec on the oracle."""
    import torch
    be = abi.load_hip_backend()
    dev = torch.device("cuda", 0)
    emitters, per = 4096, 16384
    total = emitters * per
    ins = scripts.fountain_streams_torch(0, total, dev, kill_half=True, pad=64)
    outs = [torch.full((total + 64,), -5.0, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    workloads = [scripts.fountain(variant=v) for v in range(64)]
    states, frames = [], []
    try:
        for e in range(emitters):
            w = workloads[e % 64]
            st = VectorVMState(w.script.blob, be)
            states.append(st)
            off = e * per * 4
            frames.append(st.frame([abi.DataSet([t.data_ptr() + off for t in ins], [t.data_ptr() + off for t in outs], 0)], per, [w.constants]))
        arr, oarr = be.pack_batch(frames, [abi.ExecOptions(0, 0, 0, None, 0) for _ in range(emitters)])
        for _ in range(2):                        # the second call re-uses the workspace and the look-back words of the first
            be.exec_batch_packed(arr, oarr)
        got_counts = np.array([f.survivors(0) for f in frames])
        host_out = [t.cpu().numpy() for t in outs]
        checked = 0
        for e in list(range(0, emitters, 89)) + [emitters - 1]:
            w = workloads[e % 64]
            host_in = scripts.fountain_streams_np(e * per, per, kill_half=True)
            ref, ref_n, _ = run_oracle(w.script, [w.constants], host_in, [f32] * 8, per)
            assert got_counts[e] == ref_n, "emitter %d" % e
            for k in range(8):
                seg = host_out[k][e * per:(e + 1) * per]
                assert np.array_equal(_bits(seg[:ref_n]), _bits(ref[k][:ref_n])), "emitter %d stream %d" % (e, k)
                assert np.all(seg[ref_n:] == f32(-5.0)), "emitter %d stream %d written past its survivors" % (e, k)
            checked += 1
        assert checked >= 47
        # all emitters: the survivor counts.  A variant scales the age step, so count per emitter with that variant's rule on the device
        # is not uniform; the untouched tail of every emitter's output range gives the count instead.
        untouched = (outs[6][:total].view(emitters, per) == -5.0).sum(dim=1).cpu().numpy()
        assert np.array_equal(per - untouched, got_counts)
        assert 0.3 * total < got_counts.sum() < 0.7 * total
    finally:
        for st in states:
            st.close()


@pytest.mark.parametrize("n,world", [(300_003, 2), (70_001, 3), (5, 2)])
def test_sharded_exec_through_the_hip_library(n, world):
    """The ranks' Execs of a sharded run, one after the other on device 0, each over its contiguous range with ExecIndexBase =
    RandInstanceBase = first global instance: the concatenation of the per-rank survivor prefixes equals one Exec over all
    instances on the oracle (exec_index and the random draws are keyed by the global index)."""
    built = _script_with_index_and_random()
    age_all = scripts.u01_np(0, n, 3).astype(f32)
    life_all = np.full(n, 0.5, f32)
    ref, ref_n, _ = run_oracle(built, [bytes(16)], [age_all, life_all], [f32, f32, np.int32], n, seed=77)
    counts = [0] * world
    pieces = [None] * world
    infos = []
    for rank in range(world):

        def local_exec(lo, hi, rank=rank):
            m = hi - lo
            outs, surv, _ = run_hip(built, [bytes(16)], [age_all[lo:lo + max(m, 1)], life_all[lo:lo + max(m, 1)]], [f32, f32, np.int32], m, seed=77,
                                    exec_index_base=lo, rand_instance_base=lo)
            pieces[rank] = [o[:surv].copy() for o in outs]
            counts[rank] = surv
            return surv

        # (the all-gather of a one-process rehearsal: ranks run in order, so rank r sees the counts of the ranks before it; the full
        # list is checked after the loop)
        infos.append(ShardedExec(rank, world, local_exec, lambda c: list(counts)).run(n))
    assert sum(counts) == ref_n
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    for rank in range(world):
        assert infos[rank]["range"] == shard_range(n, rank, world)
        assert infos[rank]["global_offset"] == offs[rank]
    for k in range(3):
        joined = np.concatenate([p[k] for p in pieces])
        assert np.array_equal(_bits(joined), _bits(ref[k][:ref_n])), "output %d" % k


def test_two_large_execs_of_different_states_in_flight_together():
    """Two states, each on its own stream, each with a 64 Mi-particle Exec enqueued before either is finished: both grids are
    sized for the whole GPU and compete for it.  Both results must be what each Exec gives alone."""
    import torch
    be = abi.load_hip_backend()
    dev = torch.device("cuda", 0)
    n = 64 * 1024 * 1024
    w = scripts.fountain()
    sets = []
    for s in range(2):
        ins = scripts.fountain_streams_torch(s * n, n, dev, kill_half=(s == 1), pad=64)
        outs = [torch.full((n + 64,), -7.0, dtype=torch.float32, device=dev) for _ in range(8)]
        sets.append((ins, outs))
    torch.cuda.synchronize()
    states = [VectorVMState(w.script.blob, be) for _ in range(2)]
    try:
        frames = [st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants])
                  for st, (ins, outs) in zip(states, sets)]
        for rounds in range(3):
            for st, fr in zip(states, frames):
                st.exec_async(fr)
        survs = [st.finish(fr) for st, fr in zip(states, frames)]
        dt, drag = torch.tensor(1.0 / 60.0, dtype=torch.float32, device=dev), torch.tensor(0.98, dtype=torch.float32, device=dev)
        for (ins, outs), surv in zip(sets, survs):
            age1 = ins[6][:n] + dt
            keep = age1 < ins[7][:n]
            assert surv == int(keep.sum().item())
            assert torch.equal(outs[6][:surv].view(torch.int32), age1[keep].view(torch.int32))
            assert torch.equal(outs[7][:surv].view(torch.int32), ins[7][:n][keep].view(torch.int32))
            for c in range(3):
                g = torch.tensor(-980.0 if c == 2 else 0.0, dtype=torch.float32, device=dev)
                v1 = (g * dt + ins[3 + c][:n]) * drag
                assert torch.equal(outs[3 + c][:surv].view(torch.int32), v1[keep].view(torch.int32))
                p1 = v1 * dt + ins[c][:n]
                assert torch.equal(outs[c][:surv].view(torch.int32), p1[keep].view(torch.int32))
            for k in range(8):
                assert bool((outs[k][surv:] == -7.0).all())
        assert survs[0] == n and n // 3 < survs[1] < 2 * (n // 3)
    finally:
        for st in states:
            st.close()


# ---- the C ABI's sharded Exec ----------------------------------------------------------------------------------------------------------
class _Shard:
    """One rank's streams of _script_with_index_and_random on the device, and its call frame."""

    def __init__(self, be, built, age, life, lo, hi):
        from vm_harness import PAD as pad
        self.be, self.m = be, hi - lo
        self.ptrs = []
        ins = []
        for a in (age, life):
            buf = np.zeros(max(self.m, 1) + pad, f32)
            buf[:self.m] = a[lo:hi]
            p = be.lib.vvm_device_alloc(buf.nbytes)
            assert p
            be._check(be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
            ins.append(p)
        self.out_bytes = (max(self.m, 1) + pad) * 4
        outs = []
        for _ in range(3):
            p = be.lib.vvm_device_alloc(self.out_bytes)
            assert p
            be._check(be.lib.vvm_device_memset(p, 0xCD, self.out_bytes), "memset")
            outs.append(p)
        self.ptrs = ins + outs
        self.outs = outs
        self.state = VectorVMState(built.blob, be)
        # (streams ordered float, float | float, float, int32)
        self.frame = self.state.frame([abi.DataSet(ins, [outs[0], outs[1], outs[2]], 0, [0, 2, 2], [0, 2, 3])], self.m, [bytes(16)])
        self.opt = abi.ExecOptions(77, lo, lo, None, 0)

    def download(self, k, count, dtype):
        h = np.empty(self.out_bytes // 4, dtype)
        self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], self.out_bytes), "download")
        return h[:count]

    def free(self):
        self.state.close()
        for p in self.ptrs:
            self.be.lib.vvm_device_free(p)


def test_exec_sharded_with_a_host_all_gather_callback():
    """vvm_exec_sharded with the callback form, two ranks rehearsed one after the other on device 0: the callback checks what the
    rank sends and supplies both ranks' counts (known from the oracle); the library's figures and the concatenated survivor
    prefixes must equal one Exec over all instances."""
    be = abi.load_hip_backend()
    built = _script_with_index_and_random()
    n, world = 200_001, 2
    age = scripts.u01_np(0, n, 3).astype(f32)
    life = np.full(n, 0.5, f32)
    ref, ref_n, _ = run_oracle(built, [bytes(16)], [age, life], [f32, f32, np.int32], n, seed=77)
    ranges = [shard_range(n, r, world) for r in range(world)]
    want = [int((age[lo:hi] < 0.5).sum()) for lo, hi in ranges]
    assert sum(want) == ref_n
    pieces = []
    for rank, (lo, hi) in enumerate(ranges):
        sh = _Shard(be, built, age, life, lo, hi)
        sent = []

        def gather(send, recv, count, stream, rank=rank):
            assert count == 1 and not stream
            sent.append(send[0])
            for r in range(world):
                recv[r] = want[r]
            return 0

        comm, keep = be.comm_create_callback(rank, world, gather, abi.VVM_COMM_BUFFERS_HOST)
        try:
            res = be.exec_sharded(sh.frame, sh.opt, comm)
            assert sent == [want[rank]]
            assert (res.NumDataSets, res.LocalSurvivors[0], res.GlobalOffset[0], res.GlobalSurvivors[0]) == (1, want[rank], sum(want[:rank]), ref_n)
            assert sh.frame.survivors(0) == want[rank]          # the context holds the local count
            pieces.append([sh.download(k, want[rank], dt) for k, dt in enumerate((f32, f32, np.int32))])
        finally:
            be.comm_destroy(comm)
            sh.free()
    for k in range(3):
        joined = np.concatenate([p[k] for p in pieces])
        assert np.array_equal(_bits(joined), _bits(ref[k][:ref_n])), "output %d" % k


def test_exec_sharded_over_rccl_and_a_device_callback_with_one_rank():
    """The two device-side forms with world size 1 (all a one-GPU box can hold): RCCL loaded on demand with a communicator of one
    rank, and a device callback that copies send to recv on the stream it is given.  Three pipelined steps each."""
    import ctypes
    be = abi.load_hip_backend()
    built = _script_with_index_and_random()
    n = 150_000
    age = scripts.u01_np(0, n, 3).astype(f32)
    life = np.full(n, 0.5, f32)
    want = int((age < 0.5).sum())
    hip = ctypes.CDLL("libamdhip64.so")          # (the copy the process already has loaded)
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    hip.hipMemcpyAsync.restype = ctypes.c_int

    def device_gather(send, recv, count, stream):
        assert stream
        return hip.hipMemcpyAsync(ctypes.cast(recv, ctypes.c_void_p), ctypes.cast(send, ctypes.c_void_p), 4 * count, 3, stream)      # device to device

    comm_cb, keep = be.comm_create_callback(0, 1, device_gather, abi.VVM_COMM_BUFFERS_DEVICE)
    comm_rccl = be.comm_create_rccl(0, 1, be.comm_unique_id())
    sh = _Shard(be, built, age, life, 0, n)
    try:
        for comm in (comm_cb, comm_rccl):
            for _ in range(3):
                be.exec_sharded_async(sh.frame, sh.opt, comm)
            for _ in range(3):
                res = be.exec_sharded_wait(comm)
                assert (res.LocalSurvivors[0], res.GlobalOffset[0], res.GlobalSurvivors[0]) == (want, 0, want)
            be.exec_finish(sh.frame)
            assert sh.frame.survivors(0) == want
            with pytest.raises(abi.VVMError):
                be.exec_sharded_wait(comm)          # nothing pending
            res = be.exec_sharded(sh.frame, sh.opt, comm)
            assert res.GlobalSurvivors[0] == want
    finally:
        be.comm_destroy(comm_cb)
        be.comm_destroy(comm_rccl)
        sh.free()
