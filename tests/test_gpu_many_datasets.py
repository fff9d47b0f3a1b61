"""Scripts that run acquireindex for three to thirteen datasets (the reference serves any DataSetIdx in one sweep,
VectorVMRuntime.cpp:824-878; every dataset has its own keep mask, count and output positions).  The executor takes the
datasets two to a pass, each pass one launch of a kernel with two chains (three datasets: two launches, four: two, five: three, thirteen: seven);
with VVM_NO_CHAINS one program per dataset, fused into one launch or launched one after the other.  All three against the
oracle, bit for bit, with nothing written past the survivors."""
import numpy as np
import pytest

from renderer_amd import abi
from dataset_cases import many_dataset_consts as _consts, many_dataset_script
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32


def _dtypes(d):
    return [f32, f32, np.int32] if d % 2 else [f32, f32]


def _run(k, n, mode, monkeypatch, order=None, trig_in=None, no_alloc=False):
    if mode != "paired":
        monkeypatch.setenv("VVM_NO_CHAINS", "1")      # (read when the state is allocated)
    flags = abi.VVM_EXEC_SEPARATE_PASSES if mode == "separate" else 0
    rng = np.random.default_rng(n + k)
    ins = [rng.random(n).astype(f32) for _ in range(3)]
    script = many_dataset_script(k, order, trig_in)
    consts = [_consts(k)]
    more = [_dtypes(d) for d in range(1, k)]
    ref_more, got_more, widths = [], [], []
    ref, ref_n, _ = run_oracle(script, consts, ins, _dtypes(0), n, more_datasets=more, more_results=ref_more)
    got, got_n, info = run_hip(script, consts, ins, _dtypes(0), n, more_datasets=more, more_results=got_more, flags=flags, last_width=widths, no_alloc=no_alloc)
    assert got_n == ref_n
    for a, b in zip(got, ref):
        assert np.array_equal(a[:got_n].view(np.uint32), b[:ref_n].view(np.uint32))
        assert np.all(a[got_n:].view(np.uint32) == 0xCDCDCDCD)
    survivors = [got_n]
    for d in range(k - 1):
        m = got_more[d][1]
        assert m == ref_more[d][1], "dataset %d" % (d + 1)
        survivors.append(m)
        for a, b in zip(got_more[d][0], ref_more[d][0]):
            assert np.array_equal(a[:m].view(np.uint32), b[:m].view(np.uint32)), "dataset %d" % (d + 1)
            assert np.all(a[m:].view(np.uint32) == 0xCDCDCDCD), "dataset %d written past its survivors" % (d + 1)
    if n >= 1000:
        assert len(set(survivors)) > 1 and all(0 < m < n for m in survivors)      # (the datasets keep different instances)
    return info, widths


@pytest.mark.parametrize("mode", ["paired", "fused", "separate"])
@pytest.mark.parametrize("k, n", [(3, 1), (3, 5003), (3, 300_007), (4, 77_777), (4, 600_001), (5, 300_007), (8, 40_009), (13, 9_001)])
def test_three_to_five_datasets_match_the_oracle(k, n, mode, monkeypatch):
    info, widths = _run(k, n, mode, monkeypatch)
    passes = (info.Flags >> 20) & 15
    assert passes == ((k + 1) // 2 if mode == "paired" else k)
    assert bool(info.Flags & 0x40000) == (mode == "paired")
    if mode == "paired" and n >= 262_144:
        assert widths == [8]      # (pairs and the odd one out run on the eight-wide kernels)


def test_datasets_acquired_out_of_order():
    """The pairs follow the order of the acquireindex instructions in the script, not the datasets' numbers."""
    with pytest.MonkeyPatch.context() as mp:
        info, _ = _run(4, 123_457, "paired", mp, order=[2, 0, 3, 1])
    assert (info.Flags >> 20) & 15 == 2


@pytest.mark.parametrize("trig_in", [0, 2])
def test_a_pair_that_needs_the_transcendental_kernel_sends_the_script_back_to_one_program_per_dataset(trig_in):
    """The kernels with the transcendental functions have one chain.  A sine in the first pair (datasets 0, 1) or in the odd one out
    (dataset 2: a pass of one dataset may run on any kernel, so the pairing stays)."""
    with pytest.MonkeyPatch.context() as mp:
        info, _ = _run(3, 50_021, "paired", mp, trig_in=trig_in)
    assert (info.Flags >> 20) & 15 == (3 if trig_in == 0 else 2)


def test_three_datasets_through_the_entry_point_that_allocates_nothing():
    with pytest.MonkeyPatch.context() as mp:
        info, _ = _run(3, 300_007, "paired", mp, no_alloc=True)
    assert (info.Flags >> 20) & 15 == 2


@pytest.mark.parametrize("k", [2, 3, 4])
def test_several_datasets_on_the_reference_generator(k):
    """vvm_exec_options.RandXorwowState: the kernels that read the generator's words have one chain, so a script whose datasets go
    two to a pass runs one program per dataset for such an Exec; an Exec without the state, before and after, runs the pairs."""
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    x = s.input(0, FLOAT, 0)
    draws = []
    for d in range(k):
        keep, r = s.temps(2)
        s.op("random", s.const(k), r)
        draws.append(r)
        s.op("cmplt", r, s.const(d), keep)
        s.acquireindex(keep, d)
        v = s.temp()
        s.op("mad", r, x, draws[0], v)
        s.output_data(FLOAT, d, [(v, s.output(d, FLOAT, 0)), (r, s.output(d, FLOAT, 1))])
    built = s.build()
    n = 41_003
    consts = [_consts(k)]
    rng = np.random.default_rng(k)
    ins = [rng.random(n).astype(f32)]
    more = [[f32, f32]] * (k - 1)
    words = [(0x9E3779B9 * (i + 1) + k) & 0xFFFFFFFF for i in range(24)]
    for xorwow in (None, words, None):
        ref_more, got_more = [], []
        ref, ref_n, _ = run_oracle(built, consts, ins, [f32, f32], n, seed=5, more_datasets=more, more_results=ref_more, xorwow=xorwow)
        got, got_n, info = run_hip(built, consts, ins, [f32, f32], n, seed=5, more_datasets=more, more_results=got_more, xorwow=xorwow)
        assert info.Flags & 0x40000 and (info.Flags >> 20) & 15 == (k + 1) // 2
        assert got_n == ref_n and 0 < got_n < n
        for a, b in zip(got, ref):
            assert np.array_equal(a[:got_n].view(np.uint32), b[:ref_n].view(np.uint32))
        for d in range(k - 1):
            m = got_more[d][1]
            assert m == ref_more[d][1] and 0 < m < n
            for a, b in zip(got_more[d][0], ref_more[d][0]):
                assert np.array_equal(a[:m].view(np.uint32), b[:m].view(np.uint32)), "dataset %d" % (d + 1)
