"""GPU parity of the eight-instances-per-lane kernel (the one large plain Execs run on), forced here at small sizes so that
partial tiles, both halves of a tile, death patterns and every kind of operand are covered against the CPU oracle."""
import struct

import numpy as np
import pytest

from op_cases import ALL_ONES, operand_streams, single_op_script
from renderer_amd import abi, scripts
from renderer_amd.assembler import FLOAT, HALF, INT32, Script
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32
EIGHT = abi.VVM_EXEC_FORCE_EIGHT_WIDE


def _check8(script, consts, ins, dtypes, n, **kw):
    """Scripts that fit the light eight-wide kernel run on it by default: those are checked on the general kernel as well."""
    ref, ref_n, _ = run_oracle(script, consts, ins, dtypes, n, **kw)
    got, got_n, info = run_hip(script, consts, ins, dtypes, n, flags=EIGHT, **kw)
    _compare8(got, got_n, ref, ref_n, n, kw)
    if info.Flags & 0x20000:
        got, got_n, _ = run_hip(script, consts, ins, dtypes, n, flags=EIGHT | abi.VVM_EXEC_NO_LIGHT_KERNEL, **kw)
        _compare8(got, got_n, ref, ref_n, n, kw)
    return got_n


def _compare8(got, got_n, ref, ref_n, n, kw):
    assert got_n == ref_n, "survivor count %d != %d (n=%d)" % (got_n, ref_n, n)
    off = kw.get("instance_offset", 0)
    for k, (a, b) in enumerate(zip(got, ref)):
        bits = np.uint16 if a.dtype == np.float16 else np.uint32
        fill = 0xCDCD if a.dtype == np.float16 else 0xCDCDCDCD
        x, y = a[off:off + got_n].view(bits), b[off:off + ref_n].view(bits)
        both_nan = np.isnan(a[off:off + got_n]) & np.isnan(b[off:off + ref_n]) if a.dtype in (np.float32, np.float16) else False
        bad = np.nonzero((x != y) & ~both_nan)[0]                 # (NaN payloads are not compared)
        assert bad.size == 0, "stream %d differs at %s (n=%d)" % (k, bad[:8], n)
        assert np.all(a[:off].view(bits) == fill) and np.all(a[off + got_n:].view(bits) == fill), "stream %d written outside its survivors" % k
    return got_n


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 3073, 4099, 14289, 100003])
@pytest.mark.parametrize("kill_half", [False, True])
def test_fountain_parity_eight_wide(n, kill_half):
    w = scripts.fountain()
    ins = scripts.fountain_streams_np(0, max(n, 1), age_fraction=0.5, kill_half=kill_half)
    got_n = _check8(w.script, [w.constants], ins, [f32] * 8, n)
    if not kill_half:
        assert got_n == n


@pytest.mark.parametrize("pattern", ["all_dead", "alternate", "first_only", "last_only", "first_half_of_tile", "second_half_of_tile", "blocks"])
def test_compaction_patterns_eight_wide(pattern):
    w = scripts.fountain()
    n = 9001
    ins = scripts.fountain_streams_np(0, n)
    idx = np.arange(n)
    dead = {"all_dead": np.ones(n, bool), "alternate": (idx % 2 == 1), "first_only": idx != 0, "last_only": idx != n - 1,
            "first_half_of_tile": (idx % 2048) >= 1024, "second_half_of_tile": (idx % 2048) < 1024, "blocks": (idx // 700) % 2 == 0}[pattern]
    ins[6] = np.where(dead, ins[7], ins[6] * f32(0.0)).astype(f32)
    got_n = _check8(w.script, [w.constants], ins, [f32] * 8, n, instance_offset=5)
    assert got_n == int((~dead).sum())


def test_long_look_back_chain_eight_wide():
    w = scripts.fountain()
    n = 3_000_001
    ins = scripts.fountain_streams_np(0, n)
    rng = np.random.default_rng(3)
    dead = rng.random(n) < 0.5
    ins[6] = np.where(dead, ins[7], ins[6] * f32(0.0)).astype(f32)
    assert _check8(w.script, [w.constants], ins, [f32] * 8, n) == int((~dead).sum())


@pytest.mark.parametrize("name", ["add", "mad", "select", "f2i", "cmplt_select", "mul_2x", "b2i_2x", "div", "sqrt", "mad_sqrt", "bit_rshift", "divi", "clamp", "round"])
def test_op_parity_eight_wide(name):
    script, n_in, n_out = single_op_script(name)
    n = 5000
    ins = operand_streams(name, n_in, n, seed=5)
    _check8(script, [ALL_ONES], ins, [f32] * n_out, n)


def test_exec_index_and_random_eight_wide():
    """Per-instance values that depend on the instance index: both halves of the tile must see their own indices."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    i0, i1, r0, r1 = s.temps(4)
    s.op("exec_index", i0); s.op("exec_indexf", i1); s.op("random", x, r0); s.op("random_add", x, x, r1)
    s.output_data(INT32, 0, [(i0, s.output(0, INT32, 0))])
    s.output_data(FLOAT, 0, [(i1, s.output(0, FLOAT, 0)), (r0, s.output(0, FLOAT, 1)), (r1, s.output(0, FLOAT, 2))])
    n = 6007
    x = np.linspace(0.5, 4.0, n).astype(f32)
    _check8(s.build(), [ALL_ONES], [x], [f32, f32, f32, np.int32], n, seed=1234, exec_index_base=100, rand_instance_base=7)


def test_half_streams_noadvance_and_many_outputs_eight_wide():
    s = Script()
    x = s.input(0, FLOAT, 0)
    h = s.input(0, HALF, 0)
    na = s.input(0, FLOAT, 1, noadvance=True)
    keep = s.temp(); s.op("cmplt", x, s.const(0), keep)
    s.acquireindex(keep, 0)
    hf = s.temp(); s.op("half_to_float", h, hf)
    outs = []
    for i in range(11):
        t = s.temp()
        s.op("mad", x, na, hf, t) if i % 2 else s.op("add", x, hf, t)
        outs.append(t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, i)) for i, t in enumerate(outs)])
    s.output_data(HALF, 0, [(x, s.output(0, HALF, 0)), (hf, s.output(0, HALF, 1))])
    n = 4500
    rng = np.random.default_rng(9)
    x = rng.random(n).astype(f32)
    h = rng.standard_normal(n).astype(np.float16)
    na = np.full(n, 2.5, f32)
    _check8(s.build(), [struct.pack("<f", 0.6) + bytes(12)], [x, na, h], [f32] * 11 + [np.float16] * 2, n)


def test_large_exec_takes_the_eight_wide_kernel_by_itself():
    """At VVM_EIGHT_WIDE_MIN_INSTANCES and above no flag is needed; the results stay those of the oracle."""
    w = scripts.fountain()
    n = 4 * 1024 * 1024 + 77
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
    got, got_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n)
    alt, alt_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, flags=abi.VVM_EXEC_NO_EIGHT_WIDE)
    assert got_n == ref_n == alt_n
    for k in range(8):
        assert np.array_equal(got[k][:got_n].view(np.uint32), ref[k][:ref_n].view(np.uint32))
        assert np.array_equal(alt[k][:got_n].view(np.uint32), ref[k][:ref_n].view(np.uint32))
