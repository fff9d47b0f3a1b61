"""libvvm_hip.so's sin / cos / exp / log against the published "mathfun" algorithm, bit for bit (tests/mathfun_cases.py)."""
import pytest

import mathfun_cases as mf
from vm_harness import run_hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fn", mf.FUNCTIONS)
def test_hip_gives_the_headers_bits(fn):
    mf.check_function(fn, run_hip)
