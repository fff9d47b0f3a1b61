"""GPU parity of the fountain / kill-heavy scripts: libvvm_hip.so through the C ABI vs the CPU oracle.

Bar: survivor count, survivor order and every output bit identical (all ops of these scripts are IEEE-exact
adds/multiplies/compares, VectorVMRuntime.cpp:1906-1948, :2032)."""
import numpy as np
import pytest

from renderer_amd import abi, scripts
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 4099, 14289, 100003]


def _check(w, consts, ins, n, **kw):
    f32 = np.float32
    ref, ref_n, _ = run_oracle(w.script, [consts], ins, [f32] * 8, n, **{k: v for k, v in kw.items() if k != "flags"})
    got, got_n, info = run_hip(w.script, [consts], ins, [f32] * 8, n, **kw)
    assert got_n == ref_n, "survivor count %d != %d (n=%d)" % (got_n, ref_n, n)
    off = kw.get("instance_offset", 0)
    for k in range(8):
        a = got[k][off:off + got_n].view(np.uint32)
        b = ref[k][off:off + ref_n].view(np.uint32)
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, "stream %d differs at %s (n=%d)" % (k, bad[:8], n)
        # nothing before InstanceOffset or past the survivors is touched
        assert np.all(got[k][:off].view(np.uint32) == 0xCDCDCDCD)
        assert np.all(got[k][off + got_n:].view(np.uint32) == 0xCDCDCDCD)
    return got_n, info


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kill_half", [False, True])
def test_fountain_parity(n, kill_half):
    w = scripts.fountain()
    ins = scripts.fountain_streams_np(0, max(n, 1), age_fraction=0.5, kill_half=kill_half)
    got_n, info = _check(w, w.constants, ins, n)
    assert info.InstancesPerLane == 4
    if not kill_half:
        assert got_n == n


@pytest.mark.parametrize("variant", [0, 1, 5, 8, 16, 24, 37, 63])
def test_fountain_variants(variant):
    w = scripts.fountain(variant=variant)
    n = 5000
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    _check(w, w.constants, ins, n)


@pytest.mark.parametrize("alias,merged", [(False, False), (False, True), (True, False)])
def test_fountain_forms(alias, merged):
    w = scripts.fountain(merged=merged, alias_outputs=alias)
    n = 3001
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    _check(w, w.constants, ins, n)


def test_fountain_known_answers():
    """dt=0.5, g=(0,0,-9.8), drag=1, p=(0,0,i), v=(0,0,10-i), age=0.25 i, lifetime=1.3: survivors are i = 0..3."""
    f32 = np.float32
    w = scripts.fountain()
    n = 10
    i = np.arange(n, dtype=f32)
    z = np.zeros(n, f32)
    ins = [z, z, i.copy(), z, z, f32(10) - i, f32(0.25) * i, np.full(n, 1.3, f32)]
    consts = scripts.fountain_constants(dt=0.5, gravity=(0, 0, -9.8), drag=1.0)
    got, surv, _ = run_hip(w.script, [consts], ins, [f32] * 8, n)
    assert surv == 4
    vz = (f32(-9.8) * f32(0.5) + (f32(10) - i[:4])) * f32(1.0)
    np.testing.assert_array_equal(got[5][:4], vz)
    np.testing.assert_array_equal(got[2][:4], vz * f32(0.5) + i[:4])
    np.testing.assert_array_equal(got[6][:4], np.array([0.5, 0.75, 1.0, 1.25], f32))
    np.testing.assert_array_equal(got[7][:4], np.full(4, 1.3, f32))


@pytest.mark.parametrize("offset", [1, 3, 64, 1001])
def test_instance_offset(offset):
    w = scripts.fountain()
    n = 2500
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    _check(w, w.constants, ins, n, instance_offset=offset)


@pytest.mark.parametrize("pattern", ["all_dead", "alternate", "first_only", "last_only", "blocks"])
def test_compaction_patterns(pattern):
    f32 = np.float32
    w = scripts.fountain()
    n = 5003
    ins = scripts.fountain_streams_np(0, n)
    idx = np.arange(n)
    dead = {"all_dead": np.ones(n, bool), "alternate": (idx % 2 == 1), "first_only": idx != 0, "last_only": idx != n - 1,
            "blocks": (idx // 1024) % 2 == 0}[pattern]
    ins[6] = np.where(dead, ins[7], ins[6] * f32(0.0)).astype(f32)
    got_n, _ = _check(w, w.constants, ins, n)
    assert got_n == int((~dead).sum())


def test_narrow_kernel_matches():
    w = scripts.fountain()
    n = 7777
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    _check(w, w.constants, ins, n, flags=abi.VVM_EXEC_NO_VGPR_WIDE)


def test_repeated_exec_reuses_state():
    """Several Execs on one state: the launch words re-arm themselves between launches."""
    w = scripts.fountain()
    n = 30000
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    f32 = np.float32
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
    got, got_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, repeat=5)
    assert got_n == ref_n
    for k in range(8):
        np.testing.assert_array_equal(got[k][:got_n].view(np.uint32), ref[k][:ref_n].view(np.uint32))
