"""The oracle against the vectors the reference's own math test holds (tests/golden/mathtest_vectors.json): these pin the
arithmetic of the ops that call straight into the reference's SIMD layer to something the reference states itself."""
import pytest

import mathtest_cases as mc
from vm_harness import run_oracle


@pytest.mark.parametrize("index", range(len(mc.RECORDS)), ids=mc.IDS)
def test_oracle_meets_the_reference_vectors(index):
    mc.check_record(mc.RECORDS[index], run_oracle)


def test_the_fixture_is_what_its_generator_writes():
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    before = open(os.path.join(here, "golden", "mathtest_vectors.json")).read()
    subprocess.check_call([sys.executable, os.path.join(here, "golden", "make_mathtest_vectors.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(here, "golden", "mathtest_vectors.json")).read() == before
    assert len(json.loads(before)["records"]) >= 40
