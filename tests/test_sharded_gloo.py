"""The N > 1 path on CPU: two processes over gloo run renderer_amd.runtime.ShardedExec.

Each rank executes its contiguous instance range (here through the CPU oracle, injected as `local_exec`; on GPUs the same
ShardedExec drives libvvm_hip.so and RCCL) and all-gathers its survivor count.  The concatenation of the per-rank compacted
prefixes, in rank order, must be bit-identical to one Exec over all instances: concatenating stable compactions of contiguous
shards in shard order is the stable compaction of the whole (the reference keeps survivor order, VectorVMRuntime.cpp:786-804).
The random streams and exec_index are keyed by the GLOBAL instance index, so they do not depend on the sharding either.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _script_with_index_and_random():
    """fountain-like kill plus outputs that depend on the global instance index and on the random stream."""
    from renderer_amd.assembler import FLOAT, INT32, Script
    s = Script()
    age, life = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)
    alive = s.temp(); s.op("cmplt", age, life, alive)
    s.acquireindex(alive, 0)
    idx = s.temp(); s.op("exec_index", idx)
    r = s.temp(); s.op("random", life, r)
    s.output_data(FLOAT, 0, [(age, s.output(0, FLOAT, 0)), (r, s.output(0, FLOAT, 1))])
    s.output_data(INT32, 0, [(idx, s.output(0, INT32, 0))])
    return s.build()


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from renderer_amd import scripts
        from renderer_amd.runtime import ShardedExec
        from vm_harness import run_oracle
        f32 = np.float32
        built = _script_with_index_and_random()
        result = {}

        def local_exec(lo, hi):
            m = hi - lo
            age = scripts.u01_np(lo, max(m, 1), 3).astype(f32)
            life = np.full(max(m, 1), 0.5, f32)
            outs, surv, _ = run_oracle(built, [bytes(16)], [age, life], [f32, f32, np.int32], m, seed=77,
                                       exec_index_base=lo, rand_instance_base=lo)
            result["outs"] = [o[:surv].copy() for o in outs]
            return surv

        def all_gather(count):
            t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(t, torch.tensor([count], dtype=torch.int64))
            return [int(x.item()) for x in t]

        info = ShardedExec(rank, world, local_exec, all_gather).run(n)
        # the same figures from the C ABI's own arithmetic (vvm_shard_offsets of libvvm_hip.so, the part of vvm_exec_sharded that
        # needs no device), fed with what this rank gathered
        from renderer_amd import abi
        res = abi.load_hip_backend().shard_offsets(rank, world, 1, info["counts"])
        info["c_abi"] = (int(res.LocalSurvivors[0]), int(res.GlobalOffset[0]), int(res.GlobalSurvivors[0]))
        q.put((rank, info, [o.tobytes() for o in result["outs"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [10007, 1])
def test_two_rank_sharded_exec_matches_single_exec(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    from renderer_amd import scripts
    from vm_harness import run_oracle
    f32 = np.float32
    built = _script_with_index_and_random()
    age = scripts.u01_np(0, n, 3).astype(f32)
    life = np.full(n, 0.5, f32)
    ref, ref_n, _ = run_oracle(built, [bytes(16)], [age, life], [f32, f32, np.int32], n, seed=77)

    infos = [g[1] for g in got]
    assert infos[0]["counts"] == infos[1]["counts"]
    assert sum(infos[0]["counts"]) == ref_n == infos[0]["global_survivors"]
    assert infos[0]["dataset_access_index"] == ref_n - 1
    assert infos[0]["global_offset"] == 0 and infos[1]["global_offset"] == infos[0]["local_survivors"]
    assert infos[0]["range"][1] == infos[1]["range"][0]
    for info in infos:
        assert info["c_abi"] == (info["local_survivors"], info["global_offset"], info["global_survivors"])
    for k in range(3):
        joined = b"".join(g[2][k] for g in got)
        assert joined == ref[k][:ref_n].tobytes(), "output %d differs from the single Exec" % k
