"""GPU parity of every op of the optimized format: libvvm_hip.so through the C ABI against the CPU oracle, one op per script,
on edge values (all pairs of them) and random bit patterns.

Bar: bit-identical for every op except
  * NaN payload bits of float results (the reference pins none; x86 and the GPU propagate payloads by different rules),
  * the libm-class functions tan / atan / exp2 / log2 (glibc on the CPU, the device math library on the GPU): the ULP bounds
    in op_cases.ULP_TOLERANCE,
  * rcp: both sides are hardware reciprocal ESTIMATES (VectorReciprocalEstimate, UnrealMathSSE.h:2121-2125): relative 2^-11.
The polynomial functions (sin cos exp log pow sin_cos asin acos atan2) are evaluated in the oracle's operation order and are
held to bit equality as well.
"""
import numpy as np
import pytest

import np_semantics as nps
from op_cases import ALL_ONES, RANDOM_OPS, REL_TOLERANCE, ULP_TOLERANCE, all_plain_ops, operand_streams, single_op_script
from renderer_amd import abi
from renderer_amd.assembler import FLOAT, HALF, INT32, Script
from vm_harness import run_hip, run_oracle, ulp_diff

pytestmark = pytest.mark.gpu
f32 = np.float32
N = 6000      # six tiles of the wide kernel, the last one partial


def _compare(name, got, ref):
    g, r = got.view(np.uint32), ref.view(np.uint32)
    if name in REL_TOLERANCE:
        gf, rf = got.view(f32).astype(np.float64), ref.view(f32).astype(np.float64)
        ok = (np.abs(gf - rf) <= np.abs(rf) * REL_TOLERANCE[name]) | (g == r) | (np.isnan(gf) & np.isnan(rf))
        assert ok.all(), "%s: relative error %g" % (name, np.nanmax(np.abs(gf - rf) / np.abs(rf)))
        return
    if name in ULP_TOLERANCE:
        d = ulp_diff(got.view(f32), ref.view(f32))
        finite = np.isfinite(ref.view(f32)) & (np.abs(ref.view(f32)) > f32(1e-30))
        assert d[finite].max(initial=0) <= ULP_TOLERANCE[name], "%s: %d ulp" % (name, d[finite].max())
        return
    if name not in nps.INTEGER_RESULT:
        g, r = nps.canon_nan(g), nps.canon_nan(r)
    bad = np.nonzero(g != r)[0]
    assert bad.size == 0, "%s: %d of %d differ, first at %d: gpu %08x oracle %08x" % (name, bad.size, g.size, bad[0], int(g[bad[0]]), int(r[bad[0]]))


def _domain(name, ins):
    """The operand streams are raw bit patterns; tan gets several periods around zero instead of arguments up to 1e38 (whose
    reduction mod pi is a property of the math library, not of the executor), exp2 its whole finite range."""
    if name == "tan":
        x = np.nan_to_num(ins[0], nan=0.3, posinf=1.0, neginf=-1.0)
        return [np.where(np.abs(x) > 20.0, np.fmod(x, f32(20.0)), x).astype(f32)]
    if name in ("exp2",):
        return [np.clip(np.nan_to_num(ins[0], nan=0.3, posinf=1.0, neginf=-1.0), -150, 128).astype(f32)]
    if name in ("sin", "cos", "sin_cos"):
        # the oracle's range reduction multiplies by 4/pi and converts to int32: keep below 2^31 * pi / 4 as the reference's
        # own tests do (UnrealMathTest.cpp:1395-1397 sample [-2pi, 2pi]); edge values beyond are still compared when finite
        return ins
    return ins


@pytest.mark.parametrize("name", all_plain_ops())
def test_op_parity(name):
    script, n_in, n_out = single_op_script(name)
    ins = _domain(name, operand_streams(name, n_in, N))
    ref, ref_n, _ = run_oracle(script, [ALL_ONES], ins, [f32] * n_out, N)
    got, got_n, _ = run_hip(script, [ALL_ONES], ins, [f32] * n_out, N)
    assert got_n == ref_n == N
    for k in range(n_out):
        _compare(name, got[k][:N], ref[k][:N])


@pytest.mark.parametrize("name", ["add", "mad", "sin_cos", "mul_2x", "select", "f2i"])
def test_op_parity_with_outputs_aliased_onto_temps_and_narrow_kernel(name):
    script, n_in, n_out = single_op_script(name, alias_outputs=True)
    ins = operand_streams(name, n_in, N, seed=99)
    ref, _, _ = run_oracle(script, [ALL_ONES], ins, [f32] * n_out, N)
    for flags in (0, abi.VVM_EXEC_NO_VGPR_WIDE):
        got, got_n, info = run_hip(script, [ALL_ONES], ins, [f32] * n_out, N, flags=flags)
        assert got_n == N
        for k in range(n_out):
            _compare(name, got[k][:N], ref[k][:N])


def test_random_ops_match_the_oracle_bit_for_bit():
    """The generator is this project's own counter-based one (seed, global instance, ordinal of the draw): both sides compute
    the same integers, and the float mapping (VectorVMRuntime.cpp:953-956) is exact."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    x, m = s.input(0, FLOAT, 0), s.input(0, INT32, 0)
    r0, r1, r2, r3, r4 = s.temps(5)
    s.op("random", x, r0); s.op("random_2x", x, r1, r2); s.op("random_add", x, x, r3); s.op("randomi", m, r4)
    s.output_data(FLOAT, 0, [(r, s.output(0, FLOAT, i)) for i, r in enumerate((r0, r1, r2, r3))])
    s.output_data(INT32, 0, [(r4, s.output(0, INT32, 0))])
    built = s.build()
    n = 50000
    scale = np.linspace(0.5, 9.0, n).astype(f32)
    mod = np.full(n, 1000, np.int32)
    kw = dict(seed=0x1234567890ABCDEF, rand_instance_base=777)
    ref, _, _ = run_oracle(built, [ALL_ONES], [scale, mod], [f32] * 4 + [np.int32], n, **kw)
    got, got_n, _ = run_hip(built, [ALL_ONES], [scale, mod], [f32] * 4 + [np.int32], n, **kw)
    assert got_n == n
    for k in range(5):
        np.testing.assert_array_equal(got[k][:n].view(np.uint32), ref[k][:n].view(np.uint32))
    assert 0 <= got[4][:n].min() and got[4][:n].max() < 1000


def test_exec_index_family_with_base():
    s = Script()
    s.acquireindex(s.const(0), 0)
    a, b, c = s.temps(3)
    s.op("exec_index", a); s.op("exec_indexf", b); s.op("exec_index_addi", s.input(0, INT32, 0), c)
    s.output_data(INT32, 0, [(a, s.output(0, INT32, 0)), (c, s.output(0, INT32, 1))])
    s.output_data(FLOAT, 0, [(b, s.output(0, FLOAT, 0))])
    built = s.build()
    n = 33333
    add = (np.arange(n, dtype=np.int32) * 7) % 1000
    got, got_n, _ = run_hip(built, [ALL_ONES], [add], [f32, np.int32, np.int32], n, exec_index_base=250_000_000)
    assert got_n == n
    np.testing.assert_array_equal(got[1][:n], np.arange(n, dtype=np.int32) + 250_000_000)
    np.testing.assert_array_equal(got[0][:n], (np.arange(n, dtype=np.int64) + 250_000_000).astype(f32))
    np.testing.assert_array_equal(got[2][:n], np.arange(n, dtype=np.int32) + 250_000_000 + add)


def test_half_streams_in_and_out():
    s = Script()
    h = s.input(0, HALF, 0)
    keep = s.input(0, FLOAT, 0)
    s.acquireindex(keep, 0)
    f = s.temp(); s.op("half_to_float", h, f)
    g = s.temp(); s.op("mul", f, s.const(1), g)
    s.output_data(HALF, 0, [(g, s.output(0, HALF, 0))])
    s.output_data(HALF, 0, [(h, s.output(0, HALF, 1))], from_half=True)
    s.output_data(FLOAT, 0, [(f, s.output(0, FLOAT, 0))])
    built = s.build()
    consts = ALL_ONES[:4] + np.array([1.7], f32).tobytes() + bytes(8)
    for n in (1, 5, 1003, 4097):
        rng = np.random.default_rng(n)
        x = np.concatenate([(rng.standard_normal(n) * 10.0 ** rng.integers(-6, 5, n)).astype(np.float16)])
        x[: min(n, 6)] = np.array([0, -0.0, 65504, 6e-8, np.inf, -np.inf], np.float16)[: min(n, 6)]
        keepv = np.where(rng.random(n) < 0.6, f32(-1.0), f32(1.0)).astype(f32)      # sign bit = keep
        ref, ref_n, _ = run_oracle(built, [consts], [keepv, x], [f32, np.float16, np.float16], n)
        got, got_n, _ = run_hip(built, [consts], [keepv, x], [f32, np.float16, np.float16], n)
        assert got_n == ref_n == int((keepv < 0).sum())
        np.testing.assert_array_equal(got[0][:got_n].view(np.uint32), ref[0][:ref_n].view(np.uint32))
        np.testing.assert_array_equal(got[1][:got_n].view(np.uint16), ref[1][:ref_n].view(np.uint16))
        np.testing.assert_array_equal(got[2][:got_n].view(np.uint16), ref[2][:ref_n].view(np.uint16))
        # nothing past the survivors is written
        for k, dt in enumerate((np.uint32, np.uint16, np.uint16)):
            tail = got[k][got_n:].view(dt)
            assert np.all(tail == (0xCDCDCDCD if dt == np.uint32 else 0xCDCD))


def test_noadvance_inputs_and_many_constants():
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    na = s.input(0, FLOAT, 1, noadvance=True)
    nah = s.input(0, HALF, 0, noadvance=True)
    t = s.temp(); s.op("mul", x, na, t)
    hf = s.temp(); s.op("half_to_float", nah, hf)
    acc = s.temp(); s.op("add", t, hf, acc)
    for i in range(1, 70):           # constants 64.. are read from the table, the first 64 from the uniform register
        nxt = s.temp()
        s.op("add", acc, s.const(i), nxt)
        acc = nxt
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0))])
    built = s.build()
    consts = ALL_ONES[:4] + np.arange(1, 70, dtype=f32).tobytes() + bytes(8)
    n = 3000
    xs = np.linspace(-3, 3, n).astype(f32)
    nav = np.full(n, 2.5, f32)
    nahv = np.full(n, 0.75, np.float16)
    for off in (0, 5):
        ref, ref_n, _ = run_oracle(built, [consts], [xs, nav, nahv], [f32], n, instance_offset=off)
        got, got_n, _ = run_hip(built, [consts], [xs, nav, nahv], [f32], n, instance_offset=off)
        assert got_n == ref_n == n
        np.testing.assert_array_equal(got[0][off:off + n].view(np.uint32), ref[0][off:off + n].view(np.uint32))


@pytest.mark.parametrize("n_temps,lds_registers", [(22, 0), (30, 0), (38, 0), (43, 0), (44, 0), (45, 1), (50, 6), (55, 11)])
def test_many_live_registers(n_temps, lds_registers):
    """More live temps than the 16 registers of the plain kernel: the 4-instances-per-lane kernel with 44 resident registers,
    and LDS slots for the ones beyond."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    ts = s.temps(n_temps)
    for i, t in enumerate(ts):
        s.op("mad", x if i == 0 else ts[i - 1], s.const(1), s.const(2), t)
    acc = s.temp()
    s.op("add", ts[0], ts[1], acc)
    for t in ts[2:]:
        nxt = s.temp()
        s.op("mul_add", acc, s.const(3), t, nxt)
        acc = nxt
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0)), (ts[5], s.output(0, FLOAT, 1))])
    built = s.build()
    consts = ALL_ONES[:4] + np.array([1.01, 0.1, 0.5], f32).tobytes() + bytes(16)
    n = 5000
    xs = np.linspace(-2, 2, n).astype(f32)
    ref, _, _ = run_oracle(built, [consts], [xs], [f32, f32], n)
    got, got_n, info = run_hip(built, [consts], [xs], [f32, f32], n)
    assert info.InstancesPerLane == 4 and info.LdsSpillRegisters == lds_registers and got_n == n
    for k in range(2):
        np.testing.assert_array_equal(got[k][:n].view(np.uint32), ref[k][:n].view(np.uint32))
