"""The oracle's sin / cos / exp / log against the published "mathfun" algorithm (tests/mathfun_cases.py): bit for bit on the
reference's own sample inputs and on sweeps of each domain, and literal for literal in the sources of the oracle and of the
kernel (the literals and the order of the Horner steps)."""
import os

import pytest

import mathfun_cases as mf
from vm_harness import run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# which functions of each source hold the steps of each header function, in the header's order
ORACLE_PARTS = {"sin": ["mf_reduce", "mf_sincos_core"], "cos": ["mf_reduce", "mf_sincos_core"], "exp": ["mf_exp"], "log": ["mf_log"]}
KERNEL_PARTS = {"sin": ["mf_reduce", "mf_polys"], "cos": ["mf_reduce", "mf_polys"], "exp": ["mf_exp"], "log": ["mf_log"]}


@pytest.mark.parametrize("fn", mf.FUNCTIONS)
def test_oracle_gives_the_headers_bits(fn):
    mf.check_function(fn, run_oracle)


@pytest.mark.parametrize("fn", mf.FUNCTIONS)
@pytest.mark.parametrize("which", ["oracle", "kernel"])
def test_source_literals_are_the_headers_constants_in_the_headers_order(fn, which):
    path, parts = {"oracle": (os.path.join(ROOT, "oracle", "vvm_oracle.c"), ORACLE_PARTS),
                   "kernel": (os.path.join(ROOT, "renderer_amd", "csrc", "vvm_kernels.hip"), KERNEL_PARTS)}[which]
    got = mf.literal_bits(open(path).read(), parts[fn])
    want = mf.expected_constant_bits(fn)
    if fn == "exp":
        # min(x, hi) and max(x, lo) name each bound twice in the kernel's scalar form (compare and select)
        dedup = [b for i, b in enumerate(got) if i == 0 or b != got[i - 1]]
        got = dedup
    assert got == want, (fn, which, [hex(b) for b in got], [hex(b) for b in want])


def test_the_header_constants_are_the_cephes_values():
    """A few of the constants against their closed forms, so that the fixture itself is not taken on trust."""
    import math
    import numpy as np
    c = mf.CONSTANTS["constants"]
    f32 = np.float32
    assert c["cephes_FOPI"]["bits"] == int(f32(4.0 / math.pi).view(np.uint32))
    assert c["cephes_LOG2EF"]["bits"] == int(f32(1.0 / math.log(2.0)).view(np.uint32))
    assert c["cephes_SQRTHF"]["bits"] == int(f32(math.sqrt(0.5)).view(np.uint32))
    # the three-part pi/4 and the two-part ln 2 sum to the exact values far below float32 resolution
    dp = sum(float(f32(np.uint32(c[k]["bits"]).view(f32))) for k in ("minus_cephes_DP1", "minus_cephes_DP2", "minus_cephes_DP3"))
    assert abs(dp + math.pi / 4) < 1e-14
    ln2 = float(np.uint32(c["cephes_exp_C1"]["bits"]).view(f32)) + float(np.uint32(c["cephes_exp_C2"]["bits"]).view(f32))
    assert abs(ln2 - math.log(2.0)) < 1e-9
