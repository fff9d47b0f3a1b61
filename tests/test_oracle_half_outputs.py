"""The oracle's outputdata_float_from_half (tests/half_output_cases.py), and the reference's empty stub in strict mode."""
import numpy as np
import pytest

import half_output_cases as hc
from vm_harness import run_oracle


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1003, 14289, 70001])
@pytest.mark.parametrize("death", [0.0, 0.5, 1.0])
def test_float_from_half(n, death):
    hc.check(run_oracle, n, death, seed=n)


@pytest.mark.parametrize("page_kb", [1, 64])
def test_float_from_half_over_many_chunks(page_kb):
    hc.check(run_oracle, 50001, 0.3, seed=9, page_kb=page_kb)


def test_strict_mode_writes_nothing_like_the_reference_stub():
    """VVM_Output32_from_16 returns without touching the destination (VectorVMRuntime.cpp:573-577)."""
    n = 1000
    keep, h = hc.inputs(n, 0.5, 4)
    outs, surv, _ = run_oracle(hc.script(), [bytes(16)], [keep, h], hc.OUT_DTYPES, n, strict=True)
    assert surv == int((keep < 0).sum())
    assert np.all(outs[0].view(np.uint32) == 0xCDCDCDCD)                       # untouched
    assert np.array_equal(outs[2][:surv].view(np.uint16), h[keep < 0].view(np.uint16))
