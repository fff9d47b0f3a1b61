"""GPU parity of the light eight-wide kernel (eight resident registers, six staging slots, three workgroups per CU; compares
folded into acquireindex) against the CPU oracle and against the general eight-wide kernel."""
import struct

import numpy as np
import pytest

from renderer_amd import abi
from renderer_amd.assembler import FLOAT, HALF, INT32, Script
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32
EIGHT = abi.VVM_EXEC_FORCE_EIGHT_WIDE
LIGHT_PROGRAM = 0x20000      # ProgramInfo.Flags: the state holds a program for the light kernel


def _both(script, consts, ins, dtypes, n, want_light=True, **kw):
    ref, ref_n, _ = run_oracle(script, consts, ins, dtypes, n, **kw)
    for flags in (EIGHT, EIGHT | abi.VVM_EXEC_NO_LIGHT_KERNEL):
        got, got_n, info = run_hip(script, consts, ins, dtypes, n, flags=flags, **kw)
        assert bool(info.Flags & LIGHT_PROGRAM) == want_light
        assert got_n == ref_n
        off = kw.get("instance_offset", 0)
        for k, (a, b) in enumerate(zip(got, ref)):
            bits, fill = (np.uint16, 0xCDCD) if a.dtype == np.float16 else (np.uint32, 0xCDCDCDCD)
            assert np.array_equal(a[off:off + got_n].view(bits), b[off:off + ref_n].view(bits)), "stream %d, flags %d" % (k, flags)
            assert np.all(a[:off].view(bits) == fill) and np.all(a[off + got_n:].view(bits) == fill), "stream %d written outside its survivors" % k
    return ref_n


def _six_out_script(cmp="cmplt"):
    """The fountain update with six of its outputs."""
    s = Script()
    dt, drag = s.const(0), s.const(4)
    pos = [s.input(0, FLOAT, k) for k in range(3)]
    vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
    age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
    age1 = s.temp(); s.op("add", age, dt, age1)
    alive = s.temp(); s.op(cmp, age1, life, alive)
    s.acquireindex(alive, 0)
    v1, p1 = s.temps(3), s.temps(3)
    for c in range(3):
        s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
        s.op("mad", v1[c], dt, pos[c], p1[c])
    s.output_data(FLOAT, 0, [(p1[0], s.output(0, FLOAT, 0)), (p1[1], s.output(0, FLOAT, 1)), (p1[2], s.output(0, FLOAT, 2)),
                             (v1[0], s.output(0, FLOAT, 3)), (v1[1], s.output(0, FLOAT, 4)), (age1, s.output(0, FLOAT, 5))])
    return s.build(alias_outputs=True)


CONSTS = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)


def _streams(n, seed=1):
    rng = np.random.default_rng(seed)
    ins = [rng.standard_normal(n).astype(f32) * f32(100.0) for _ in range(6)]
    life = rng.uniform(0.5, 2.0, n).astype(f32)
    age = np.where(rng.random(n) < 0.4, life, life * f32(1.6) * rng.random(n).astype(f32)).astype(f32)      # 40% reach their lifetime exactly, some are past it
    age[rng.random(n) < 0.01] = np.nan
    return ins + [age, life]


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 2047, 2048, 2049, 6143, 6145, 100003])
def test_six_outputs(n):
    _both(_six_out_script(), [CONSTS], _streams(n), [f32] * 6, n, instance_offset=3)


@pytest.mark.parametrize("cmp", ["cmplt", "cmple", "cmpgt", "cmpge", "cmpeq", "cmpneq"])
def test_float_compares_folded_into_acquireindex(cmp):
    n = 20011
    still = struct.pack("<6f", 0.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)      # (time step 0: an age that equals the lifetime stays equal)
    survivors = _both(_six_out_script(cmp), [still], _streams(n, seed=7), [f32] * 6, n)
    assert 0 < survivors < n


@pytest.mark.parametrize("cmp", ["cmplti", "cmplei", "cmpgti", "cmpgei", "cmpeqi", "cmpneqi"])
def test_integer_compares_folded_into_acquireindex(cmp):
    s = Script()
    a, b = s.input(0, INT32, 0), s.input(0, INT32, 1)
    keep = s.temp(); s.op(cmp, a, b, keep)
    s.acquireindex(keep, 0)
    t = s.temp(); s.op("addi", a, b, t)
    s.output_data(INT32, 0, [(t, s.output(0, INT32, 0)), (a, s.output(0, INT32, 1))])
    n = 9001
    rng = np.random.default_rng(11)
    x = rng.integers(-5, 6, n).astype(np.int32)
    y = rng.integers(-5, 6, n).astype(np.int32)
    x[:4] = [-2**31, 2**31 - 1, -1, 0]; y[:4] = [2**31 - 1, -2**31, 0, -1]
    survivors = _both(s.build(), [bytes(16)], [x, y], [np.int32] * 2, n)
    assert 0 < survivors < n


def test_compare_with_another_reader_stays_an_instruction():
    """The mask is also an output: the compare cannot be folded away, the acquireindex reads the mask's sign bit."""
    s = Script()
    x, y = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)
    keep = s.temp(); s.op("cmplt", x, y, keep)
    s.acquireindex(keep, 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, 0))])
    s.output_data(INT32, 0, [(keep, s.output(0, INT32, 0))])
    n = 7000
    rng = np.random.default_rng(5)
    survivors = _both(s.build(), [bytes(16)], [rng.random(n).astype(f32), rng.random(n).astype(f32)], [f32, np.int32], n)
    assert 0 < survivors < n


def test_predicate_from_a_constant_and_from_an_input():
    for from_input in (False, True):
        s = Script()
        x, m = s.input(0, FLOAT, 0), s.input(0, INT32, 0)
        s.acquireindex(m if from_input else s.const(0), 0)
        t = s.temp(); s.op("mul", x, x, t)
        s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
        n = 5000
        rng = np.random.default_rng(2)
        mask = np.where(rng.random(n) < 0.5, -1, 0).astype(np.int32)
        survivors = _both(s.build(), [struct.pack("<I", 0xFFFFFFFF) + bytes(12)], [rng.random(n).astype(f32), mask], [f32], n)
        assert survivors == (int((mask < 0).sum()) if from_input else n)


def _chain_script(n_out, half_at=None):
    """x + c, that + c, ...: one output per link."""
    s = Script()
    x = s.input(0, FLOAT, 0)
    keep = s.temp(); s.op("cmplt", x, s.const(2), keep)
    s.acquireindex(keep, 0)
    outs = []
    for i in range(n_out):
        t = s.temp(); s.op("add", x, s.const(1), t)
        outs.append(t)
        x = t
    for i, t in enumerate(outs):
        if i == half_at:
            s.output_data(HALF, 0, [(t, s.output(0, HALF, 0))])
        else:
            s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, i - (half_at is not None and i > half_at)))])
    return s.build()


CHAIN_CONSTS = struct.pack("<Iff", 0xFFFFFFFF, 0.5, 0.7) + bytes(4)


@pytest.mark.parametrize("n_out,light", [(6, True), (7, True), (8, True), (9, False)])
@pytest.mark.parametrize("n", [2048, 9001])
def test_seventh_and_eighth_output_wait_in_registers(n_out, light, n):
    """Six outputs are staged in LDS; two more stay in registers until the next tile knows the position; nine need a second group
    of slots, which the light kernel does not have."""
    x = np.linspace(0, 1, n).astype(f32)
    survivors = _both(_chain_script(n_out), [CHAIN_CONSTS], [x], [f32] * n_out, n, want_light=light, instance_offset=2)
    assert 0 < survivors < n


def test_half_outputs_are_staged_and_a_32_bit_output_rides_instead():
    n = 5000
    x = np.linspace(0, 1, n).astype(f32)
    for half_at in (6, 2):      # (only 32-bit outputs can wait in registers: the linker picks one of those, wherever the half output comes)
        _both(_chain_script(7, half_at), [CHAIN_CONSTS], [x], [f32] * 6 + [np.float16], n)


def test_riding_outputs_that_need_a_copy():
    """Outputs seven and eight are an input stream and a constant (nothing after the acquireindex makes them), the same value goes to
    two outputs, and a temp is written again after it was output."""
    s = Script()
    a, b = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)
    early = s.temp(); s.op("mul", a, b, early)                      # made before the acquireindex
    keep = s.temp(); s.op("cmpge", a, s.const(1), keep)
    s.acquireindex(keep, 0)
    t = s.temp(); s.op("add", a, b, t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0)), (t, s.output(0, FLOAT, 1))])
    s.op("sub", a, b, t)                                            # the temp again, after its first value was output
    u = s.temp(); s.op("mul", t, t, u)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 2)), (u, s.output(0, FLOAT, 3)), (early, s.output(0, FLOAT, 4)), (a, s.output(0, FLOAT, 5))])
    s.output_data(FLOAT, 0, [(b, s.output(0, FLOAT, 6)), (s.const(2), s.output(0, FLOAT, 7))])
    n = 7001
    rng = np.random.default_rng(8)
    ins = [rng.standard_normal(n).astype(f32), rng.standard_normal(n).astype(f32)]
    survivors = _both(s.build(), [struct.pack("<Iff", 0, 0.25, 3.5) + bytes(4)], ins, [f32] * 8, n)
    assert 0 < survivors < n


def test_eight_pass_through_streams():
    """Eight streams in, the same eight out: every register is taken by a stream, the two riding outputs are copies."""
    s = Script()
    xs = [s.input(0, FLOAT, i) for i in range(8)]
    keep = s.temp(); s.op("cmplt", xs[0], xs[1], keep)
    s.acquireindex(keep, 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, i)) for i, x in enumerate(xs)])
    n = 12345
    rng = np.random.default_rng(4)
    ins = [rng.standard_normal(n).astype(f32) for _ in range(8)]
    survivors = _both(s.build(), [bytes(16)], ins, [f32] * 8, n)
    assert 0 < survivors < n


@pytest.mark.parametrize("kill_half", [False, True])
def test_fountain_on_the_light_kernel_across_many_tiles(kill_half):
    from renderer_amd import scripts
    w = scripts.fountain()
    n = 1_000_003
    _both(w.script, [w.constants], scripts.fountain_streams_np(0, n, kill_half=kill_half), [f32] * 8, n)


def test_large_exec_takes_the_light_kernel_by_itself():
    n = 4 * 1024 * 1024 + 77
    ins = _streams(n, seed=3)
    script = _six_out_script()
    ref, ref_n, _ = run_oracle(script, [CONSTS], ins, [f32] * 6, n)
    got, got_n, info = run_hip(script, [CONSTS], ins, [f32] * 6, n)
    assert info.Flags & LIGHT_PROGRAM and got_n == ref_n
    for k in range(6):
        assert np.array_equal(got[k][:got_n].view(np.uint32), ref[k][:ref_n].view(np.uint32))


def test_arithmetic_on_constants_alone_is_left_to_the_host():
    """tmp = dt * scale; age' = age + tmp: the product is the same for every instance, the linker takes it out of the program (the
    script then fits eight registers) and the host computes it with the Exec's constants; also chained and integer forms, a derived
    value that is output directly, and a temp that holds a per-instance value after it held a derived one."""
    s = Script()
    xs = [s.input(0, FLOAT, i) for i in range(8)]
    c0, c1, c2 = s.const(0), s.const(1), s.const(2)
    t = s.temp(); s.op("mul", c0, c1, t)                 # derived
    u = s.temp(); s.op("add", t, c2, u)                  # derived from a derived one
    k = s.temp(); s.op("muli", s.const(3), s.const(3), k)    # integer
    keep = s.temp(); s.op("cmplt", xs[0], u, keep)
    s.acquireindex(keep, 0)
    a = s.temp(); s.op("mad", xs[1], t, u, a)
    s.op("sub", xs[2], a, t)                             # the temp again, now per instance
    b = s.temp(); s.op("add", t, xs[3], b)
    s.output_data(FLOAT, 0, [(a, s.output(0, FLOAT, 0)), (b, s.output(0, FLOAT, 1)), (u, s.output(0, FLOAT, 2)), (t, s.output(0, FLOAT, 3))])
    s.output_data(INT32, 0, [(k, s.output(0, INT32, 0))])
    built = s.build()
    n = 10_007
    rng = np.random.default_rng(12)
    ins = [rng.standard_normal(n).astype(f32) for _ in range(8)]
    consts = struct.pack("<fffi", 0.3, 1.7, -0.25, 70000)
    code = abi.load_hip_backend().inspect_code(built.blob, 0)
    assert sum(1 for w in code if 0 < (w[0] & 0xFF) < 190) == 3      # mad, sub, add are left: the three constant-only instructions are gone
    survivors = _both(built, [consts], ins, [f32] * 4 + [np.int32], n)
    assert 0 < survivors < n
