"""The two routes by which a single Exec's parameter block reaches the kernel -- as a kernel argument (blocks of up to 1 KB)
or through a host-to-device copy -- give the same streams bit for bit, and both agree with the CPU oracle."""
import numpy as np
import pytest

from renderer_amd import abi, scripts
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32
IN_MEMORY = abi.VVM_EXEC_PARAMS_IN_MEMORY


def _same(a_outs, b_outs, count, off=0):
    for k, (a, b) in enumerate(zip(a_outs, b_outs)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "stream %d differs between the routes" % k
        assert count + off <= a.size


@pytest.mark.parametrize("n", [0, 1, 5, 1024, 4099, 300_000])
@pytest.mark.parametrize("kill_half", [False, True])
@pytest.mark.parametrize("width", [0, abi.VVM_EXEC_NO_EIGHT_WIDE, abi.VVM_EXEC_FORCE_EIGHT_WIDE, abi.VVM_EXEC_NO_VGPR_WIDE])
def test_fountain_both_routes(n, kill_half, width):
    w = scripts.fountain()
    ins = scripts.fountain_streams_np(0, max(n, 1), age_fraction=0.5, kill_half=kill_half)
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n, instance_offset=3)
    arg, arg_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, flags=width, instance_offset=3)
    mem, mem_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, flags=width | IN_MEMORY, instance_offset=3)
    assert arg_n == ref_n and mem_n == ref_n
    _same(arg, mem, arg_n, 3)
    for a, b in zip(arg, ref):
        assert np.array_equal(a[3:3 + ref_n].view(np.uint32), b[3:3 + ref_n].view(np.uint32))


def test_transcendental_both_routes():
    """A script with external functions, no-advance-free, on the kernels that carry the noise table."""
    w = scripts.transcendental(True)
    n = 10_001
    ins = scripts.transcendental_streams_np(0, n)
    arg, arg_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=w.ext_fns)
    mem, mem_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=w.ext_fns, flags=IN_MEMORY)
    assert arg_n == mem_n == n
    _same(arg, mem, arg_n)


@pytest.mark.parametrize("n", [0, 1, 4099, 300_000])
@pytest.mark.parametrize("kill_half", [False, True])
def test_counts_left_in_the_callers_device_buffer(n, kill_half):
    """vvm_exec_options.DeviceCounts: the Exec writes the survivor count there (what the multi-GPU all-gather reads), the call
    still reports it through the context, and the streams are the same as without."""
    w = scripts.fountain()
    ins = scripts.fountain_streams_np(0, max(n, 1), age_fraction=0.5, kill_half=kill_half)
    plain, plain_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n)
    counts = []
    got, got_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, device_counts=counts, repeat=2)
    assert got_n == plain_n
    assert counts[0] == plain_n and all(c == -1 for c in counts[1:])
    _same(got, plain, got_n)
