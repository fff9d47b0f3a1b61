"""libvvm_hip.so's outputdata_float_from_half against numpy's exact conversion and against the oracle, on every kernel shape
(tests/half_output_cases.py)."""
import numpy as np
import pytest

import half_output_cases as hc
from renderer_amd import abi
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu

SHAPES = {"four-wide": (abi.VVM_EXEC_NO_EIGHT_WIDE, 4), "eight-wide": (abi.VVM_EXEC_FORCE_EIGHT_WIDE, 8),
          "eight-wide general": (abi.VVM_EXEC_FORCE_EIGHT_WIDE | abi.VVM_EXEC_NO_LIGHT_KERNEL, 8), "narrow": (abi.VVM_EXEC_NO_VGPR_WIDE, 1)}


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("n", [1, 5, 1003, 4097, 70001])
@pytest.mark.parametrize("death", [0.0, 0.5, 1.0])
def test_float_from_half(shape, n, death):
    flags, lanes = SHAPES[shape]
    width = []
    hc.check(run_hip, n, death, seed=n, flags=flags, last_width=width)
    assert width == [lanes]


@pytest.mark.parametrize("shape", list(SHAPES))
def test_against_the_oracle_at_a_size_with_many_tiles(shape):
    n = 300_007
    keep, h = hc.inputs(n, 0.37, 11)
    ref, ref_n, _ = run_oracle(hc.script(), [bytes(16)], [keep, h], hc.OUT_DTYPES, n)
    got, got_n, _ = run_hip(hc.script(), [bytes(16)], [keep, h], hc.OUT_DTYPES, n, flags=SHAPES[shape][0], instance_offset=0)
    assert got_n == ref_n
    for k, dt in enumerate((np.uint32, np.uint32, np.uint16)):
        assert np.array_equal(got[k][:got_n].view(dt), ref[k][:ref_n].view(dt)), k


def test_through_the_no_alloc_entry_and_with_an_instance_offset():
    hc.check(run_hip, 20003, 0.4, seed=3, no_alloc=True)
    n, off = 5001, 37
    keep, h = hc.inputs(n, 0.5, 8)
    ref, ref_n, _ = run_oracle(hc.script(), [bytes(16)], [keep, h], hc.OUT_DTYPES, n, instance_offset=off)
    got, got_n, _ = run_hip(hc.script(), [bytes(16)], [keep, h], hc.OUT_DTYPES, n, instance_offset=off)
    assert got_n == ref_n
    assert np.array_equal(got[0][off:off + got_n].view(np.uint32), ref[0][off:off + ref_n].view(np.uint32))
