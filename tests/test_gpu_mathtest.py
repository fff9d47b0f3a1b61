"""libvvm_hip.so against the vectors the reference's own math test holds (tests/golden/mathtest_vectors.json), compared the way
that test compares them."""
import pytest

import mathtest_cases as mc
from vm_harness import run_hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("index", range(len(mc.RECORDS)), ids=mc.IDS)
def test_hip_meets_the_reference_vectors(index):
    mc.check_record(mc.RECORDS[index], run_hip)
