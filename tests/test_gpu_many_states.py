"""A process that holds states by the thousand (one per emitter) must leave the device usable for other processes: the states share
a few streams per device.  (With a stream per state, a second process failed inside the HIP runtime at its first queue as soon
as the first one held about a thousand of them.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from renderer_amd import abi, scripts
from renderer_amd.runtime import VectorVMState
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, %r)
from renderer_amd import abi
be = abi.load_hip_backend()
p = be.lib.vvm_device_alloc(4096)
assert p
be._check(be.lib.vvm_device_memset(p, 0, 4096), "memset")
be.lib.vvm_device_free(p)
print("child ok", flush=True)
""" % ROOT


def test_a_second_process_starts_while_this_one_holds_two_thousand_states():
    be = abi.load_hip_backend()
    w = scripts.fountain()
    states = [VectorVMState(w.script.blob, be) for _ in range(2000)]
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
        assert r.returncode == 0 and b"child ok" in r.stdout, r.stdout.decode(errors="replace")[-3000:]
    finally:
        for s in states:
            s.close()


def test_states_that_share_a_stream_keep_their_results_apart():
    """Eight states (two per shared stream), all enqueued before any is finished, each on its own input range."""
    be = abi.load_hip_backend()
    w = scripts.fountain()
    f32 = np.float32
    n = 40_003
    import torch
    dev = torch.device("cuda", 0)
    runs = []
    states = [VectorVMState(w.script.blob, be) for _ in range(8)]
    try:
        for k, st in enumerate(states):
            ins = scripts.fountain_streams_torch(k * n, n, dev, kill_half=(k % 2 == 1), pad=64)
            outs = [torch.full((n + 64,), -3.0, dtype=torch.float32, device=dev) for _ in range(8)]
            fr = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants])
            runs.append((st, fr, ins, outs))
        torch.cuda.synchronize()
        for st, fr, _, _ in runs:
            st.exec_async(fr)
        for k, (st, fr, ins, outs) in enumerate(runs):
            surv = st.finish(fr)
            host_in = scripts.fountain_streams_np(k * n, n, kill_half=(k % 2 == 1))
            ref, ref_n, _ = run_oracle(w.script, [w.constants], host_in, [f32] * 8, n)
            assert surv == ref_n
            for j in range(8):
                got = outs[j].cpu().numpy()
                assert np.array_equal(got[:surv].view(np.uint32), ref[j][:ref_n].view(np.uint32)), "state %d stream %d" % (k, j)
                assert np.all(got[surv:] == f32(-3.0))
    finally:
        for st in states:
            st.close()
