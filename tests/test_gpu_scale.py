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
