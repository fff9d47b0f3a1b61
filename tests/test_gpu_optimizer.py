"""GPU parity for scripts that went through vvm_optimize_script: the optimized blob on libvvm_hip.so against the same blob on the
CPU oracle, and against the instruction-by-instruction translation of the compiler-format script (tests/original_cases.py)."""
import numpy as np
import pytest

import original_cases as oc
from renderer_amd import abi
from renderer_amd.assembler import FLOAT, HALF
from renderer_amd.original import K, OriginalScript
from test_optimizer import ALL_ONES, Blob, assert_same, optimize, streams
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.mark.parametrize("n", [1, 1000, 300000])
def test_fountain_from_compiler_format(n):
    o, meta = oc.fountain()
    rng = np.random.default_rng(n)
    ins = [rng.uniform(-100, 100, n).astype(f32) for _ in range(6)] + [rng.uniform(0, 2, n).astype(f32), rng.uniform(1, 2, n).astype(f32)]
    blob = Blob(optimize(o))
    ref = run_oracle(blob, [meta["consts"]], ins, [f32] * 8, n)
    direct = run_oracle(oc.translate(o).build(), [meta["consts"]], ins, [f32] * 8, n)
    got = run_hip(blob, [meta["consts"]], ins, [f32] * 8, n)
    assert_same(got, ref, 8)
    assert_same(got, direct, 8)
    assert got[2].InstancesPerLane == 4 and got[2].LdsSpillRegisters == 0


@pytest.mark.parametrize("first,second,position,merged", oc.MERGE_PAIRS, ids=["%s-%s-%d" % p[:3] for p in oc.MERGE_PAIRS])
def test_merged_pairs_on_the_gpu(first, second, position, merged):
    o, _ = oc.merge_pair(first, second, position)
    n = 1031
    fl, it = streams(n)
    if first in ("bit_lshift", "bit_rshift") or second in ("bit_lshift", "bit_rshift"):
        it = [np.abs(a) % 9 for a in it]
    if (first, second) == ("mad", "sqrt"):
        fl = [np.abs(a) + f32(0.5) for a in fl]
    blob = Blob(optimize(o))
    ref = run_oracle(blob, [ALL_ONES], fl + it, [f32], n)
    got = run_hip(blob, [ALL_ONES], fl + it, [f32], n)
    if first == "fmod" or second == "fmod":            # fmod is exact on both sides, but keep NaN payloads out of the comparison
        keep = ~np.isnan(ref[0][0][:n])
        np.testing.assert_array_equal(got[0][0][:n][keep].view(np.uint32), ref[0][0][:n][keep].view(np.uint32))
    else:
        assert_same(got, ref, 1)


def test_half_inputs_and_constant_outputs_on_the_gpu():
    o = OriginalScript()
    h = o.input(0, HALF, 0, o.temp())
    x = o.input(0, FLOAT, 0, o.temp())
    na = o.input(0, FLOAT, 1, o.temp(), noadvance=True)
    keep = o.temp(); o.op("cmplt", x, K(2), keep)
    idx = o.temp(); o.acquireindex(keep, idx, 0)
    a1 = o.temp(); o.op("mul", h, x, a1)
    a2 = o.temp(); o.op("add", h, na, a2)
    o.output(FLOAT, 0, idx, a1, 0)
    o.output(FLOAT, 0, idx, a2, 1)
    o.output(FLOAT, 0, idx, K(1), 2)
    o.output(HALF, 0, idx, h, 0)
    n = 5000
    rng = np.random.default_rng(5)
    hs = rng.uniform(-2, 2, n).astype(np.float16)
    xs = rng.uniform(-2, 2, n).astype(f32)
    nas = np.full(n, 0.25, f32)
    consts = ALL_ONES + np.array([7.5, 0.5], f32).tobytes()
    blob = Blob(optimize(o))
    ref = run_oracle(blob, [consts], [xs, nas, hs], [f32, f32, f32, np.float16], n)
    got = run_hip(blob, [consts], [xs, nas, hs], [f32, f32, f32, np.float16], n)
    assert got[1] == ref[1] and 0 < got[1] < n
    assert_same(got, ref, 3)
    np.testing.assert_array_equal(got[0][3][:got[1]].view(np.uint16), ref[0][3][:ref[1]].view(np.uint16))


@pytest.mark.parametrize("block", range(4))
def test_random_scripts_on_the_gpu(block):
    """The generator of tests/test_optimizer_fuzz.py (register reuse, dead values, constants, compares, integer ops, a kill predicate):
    the optimized blob on libvvm_hip.so against the same blob on the oracle, bit for bit, at a size with several tiles and a ragged end."""
    from test_optimizer_fuzz import gen
    for seed in range(1000 + block * 25, 1000 + block * 25 + 25):
        o, nf, ni, fo, io = gen(seed)
        n = 2048 * 3 + 77
        rng = np.random.default_rng(seed)
        ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, n).astype(np.int32) for _ in range(ni)]
        consts = ALL_ONES + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
        dts = [f32] * fo + [np.int32] * io
        blob = Blob(optimize(o))
        ref = run_oracle(blob, [consts], ins, dts, n)
        got = run_hip(blob, [consts], ins, dts, n)
        assert got[1] == ref[1], seed
        for k in range(len(dts)):
            x, y = got[0][k][:got[1]].view(np.uint32), ref[0][k][:ref[1]].view(np.uint32)
            both_nan = np.isnan(x.view(f32)) & np.isnan(y.view(f32)) if k < fo else np.zeros(got[1], bool)
            assert np.array_equal(x[~both_nan], y[~both_nan]), (seed, k)
