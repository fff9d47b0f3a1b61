"""The lean form of a script with a wide record (what the script computes on the light eight-wide kernel, the streams it hands
through unchanged by a copy kernel behind it) against the oracle and against the same script run whole."""
import numpy as np
import pytest

import through_cases as sc
from renderer_amd import abi
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
EIGHT = abi.VVM_EXEC_FORCE_EIGHT_WIDE


def _compare(n, extra_float, extra_int, death, flags=EIGHT, touch_one=False, instance_offset=0, expect_lean=True, lean_from=64 * 1024, more_math=0):
    script = sc.wide_record_script(extra_float, extra_int, touch_one, more_math)
    ins = sc.streams(n, extra_float, extra_int, death)
    dts = sc.out_dtypes(extra_float, extra_int)
    want, want_n, _ = run_oracle(script, [sc.CONSTANTS], ins, dts, n, instance_offset=instance_offset)
    width = []
    got, got_n, info = run_hip(script, [sc.CONSTANTS], ins, dts, n, instance_offset=instance_offset, flags=flags, last_width=width)
    assert bool(info.Flags & 0x80000) == expect_lean
    runs_lean = expect_lean and not (flags & abi.VVM_EXEC_NO_THROUGH_KERNEL) and ((flags & EIGHT) or n >= lean_from)
    if n > 0:
        whole_width = 8 if (extra_float + extra_int <= 8 and ((flags & EIGHT) or n >= 256 * 1024)) else 4      # (up to sixteen streams: the general eight-wide kernel)
        assert width == [8 if runs_lean else whole_width]
    assert got_n == want_n
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g[instance_offset:instance_offset + got_n].view(np.uint32), w[instance_offset:instance_offset + want_n].view(np.uint32))
        assert (g[instance_offset + got_n:].view(np.uint8) == 0xCD).all()
    return info


@pytest.mark.parametrize("n", [0, 1, 5, 2047, 2048, 2049, 70001])
@pytest.mark.parametrize("death", [0.0, 0.5, 1.0])
def test_wide_record_against_the_oracle(n, death):
    info = _compare(n, 14, 2, death)
    assert (info.Flags >> 24) == 17      # handed-through streams: the sixteen extra ones and the lifetime


def test_the_same_script_whole():
    _compare(70001, 14, 2, 0.5, flags=EIGHT | abi.VVM_EXEC_NO_THROUGH_KERNEL)
    _compare(30001, 14, 2, 0.5, flags=0)      # below the size from which the lean form runs: whole, four-wide


def test_a_handed_through_stream_that_the_arithmetic_reads_too():
    _compare(50000, 12, 0, 0.3, touch_one=True, instance_offset=7)


def test_large_and_by_its_own_choice():
    """From 64 Ki instances on the library takes the lean form by itself; below, the whole script."""
    _compare(3 * 1024 * 1024 + 5, 10, 0, 0.5, flags=0)
    _compare(70001, 10, 2, 0.25, flags=0)
    _compare(60000, 10, 0, 0.5, flags=0)


def test_scripts_that_stay_whole():
    _compare(5000, 0, 0, 0.5, expect_lean=False)      # fits the light kernel as it is


def test_a_script_that_fits_the_general_eight_wide_kernel_goes_lean_later():
    """Fourteen streams: the whole script runs eight-wide on the general kernel, and the lean form takes over from 4 Mi instances."""
    _compare(5000, 6, 0, 0.5)                                        # forced
    _compare(300000, 6, 0, 0.5, flags=0, lean_from=4 * 1024 * 1024)   # whole, eight-wide by the library's choice
    _compare(4 * 1024 * 1024 + 3, 6, 0, 0.5, flags=0, lean_from=4 * 1024 * 1024)


@pytest.mark.parametrize("n,death", [(1, 0.0), (2049, 0.5), (70001, 0.3), (70001, 1.0), (600001, 0.5)])
def test_what_the_script_computes_needs_the_general_eight_wide_kernel(n, death):
    """Six of the extra streams are computed with: fourteen streams in the lean program, which then runs on the general eight-wide
    kernel (which leaves the ride words as the light one does); twelve more handed through."""
    info = _compare(n, 18, 1, death, more_math=6)
    assert (info.Flags >> 24) == 14


def test_more_than_eight_computed_outputs_flush_in_the_middle_of_a_tile():
    """Eight fountain outputs and six computed ones: two groups of staging slots in the lean program, twelve streams handed through."""
    _compare(70001, 18, 0, 0.5, more_math=6, instance_offset=3)
