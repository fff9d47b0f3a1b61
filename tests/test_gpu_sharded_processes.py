"""The sharded Exec of the C ABI with a real exchange: two (and three) live processes, each holding one contiguous range of the
instances on device 0, each calling vvm_exec_sharded with a communicator whose all-gather is a gloo exchange between the
processes (tests/sharded_rank_process.py).  The concatenation of the ranks' survivor prefixes in rank order and every rank's
vvm_shard_result must equal one Exec over all instances on the oracle: concatenating stable compactions of contiguous ranges
in range order is the stable compaction of the whole (the reference keeps survivor order, VectorVMRuntime.cpp:786-804)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from renderer_amd import scripts
from renderer_amd.runtime import shard_range
from vm_harness import run_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from sharded_rank_process import secondary_only_script  # noqa: E402
from test_sharded_gloo import _script_with_index_and_random  # noqa: E402

pytestmark = pytest.mark.gpu
f32 = np.float32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n,world", [(300_003, 2), (70_001, 3), (3, 2)])
def test_rank_processes_exchange_counts_through_vvm_exec_sharded(n, world, tmp_path):
    port = _free_port()
    files = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if os.environ.get("VVM_TEST_RANK_DEBUG"):      # (diagnostic runs: the HIP runtime's and the kernel driver interface's logging in the ranks)
        env.update(HSAKMT_DEBUG_LEVEL="7", AMD_LOG_LEVEL="4")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_rank_process.py"), str(r), str(world), str(port), str(n), files[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            logs.append(out.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    keep = os.environ.get("VVM_TEST_RANK_LOGS")
    if keep:
        for r, text in enumerate(logs):
            with open(os.path.join(keep, "rank_n%d_w%d_r%d.log" % (n, world, r)), "w") as f:
                f.write(text)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-8000:])
    reports = [json.load(open(f)) for f in files]
    blobs = [open(f + ".bin", "rb").read() for f in files]

    age = scripts.u01_np(0, n, 3).astype(f32)
    life = np.full(n, 0.5, f32)
    # script 1: dataset 0, outputs age / random draw / exec_index
    ref, ref_n, _ = run_oracle(_script_with_index_and_random(), [bytes(16)], [age, life], [f32, f32, np.int32], n, seed=77)
    # script 2: dataset 1 only
    more = []
    run_oracle(secondary_only_script(), [bytes(16)], [age, life], [], n, seed=77, more_datasets=[[f32, np.int32]], more_results=more)
    (ref1, ref1_n), = more

    ranges = [shard_range(n, r, world) for r in range(world)]
    want = [int((age[lo:hi] < f32(0.5)).sum()) for lo, hi in ranges]
    want1 = [(hi - lo) - w for (lo, hi), w in zip(ranges, want)]
    assert sum(want) == ref_n and sum(want1) == ref1_n
    pieces = [[], [], []]
    pieces1 = [[], []]
    for r, (rep, blob) in enumerate(zip(reports, blobs)):
        assert rep["range"] == list(ranges[r])
        before, before1 = sum(want[:r]), sum(want1[:r])
        assert rep["sync"] == [1, want[r], before, ref_n, want[r]]
        assert rep["pipelined"] == [[want[r], before, ref_n]] * 3
        # dataset 0 is not acquired by the second script: zero everywhere, whatever the first script's Execs left in the buffers
        assert rep["secondary"] == [2, 0, 0, 0, want1[r], before1, ref1_n, 0, want1[r]]
        # what the rank handed to its all-gather callback: four exchanges of the first script, one of the second
        assert [g[0] for g in rep["gathers"][:4]] == [want[r]] * 4 and rep["gathers"][4][1] == want1[r]
        assert len(blob) == 12 * want[r] + 8 * want1[r]
        at = 0
        for k in range(3):
            pieces[k].append(blob[at:at + 4 * want[r]]); at += 4 * want[r]
        for k in range(2):
            pieces1[k].append(blob[at:at + 4 * want1[r]]); at += 4 * want1[r]
    for k in range(3):
        assert b"".join(pieces[k]) == ref[k][:ref_n].tobytes(), "output %d differs from the single Exec" % k
    for k in range(2):
        assert b"".join(pieces1[k]) == ref1[k][:ref1_n].tobytes(), "dataset 1 output %d differs from the single Exec" % k
