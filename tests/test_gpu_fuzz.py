"""A short run of tools/gpu_fuzz_linker.py inside the GPU suite: 500 random scripts in the optimized format (merged opcodes, three
stream types, IDs, two datasets, noise and random ops, up to 24 outputs, 2 to 60 registers), some over hundreds of thousands to
millions of instances, some inside a batch, through vvm_exec_noalloc, or on the reference's random generator, each bit for bit
against the oracle.  The tool itself runs tens of thousands of seeds; this keeps its paths from rotting."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_five_hundred_random_scripts_match_the_oracle():
    import gpu_fuzz_linker as fuzz
    was = fuzz.LARGE
    fuzz.LARGE = True
    try:
        bad, skipped, kinds = fuzz.run(0, 500, report_every=0)
    finally:
        fuzz.LARGE = was
    assert bad == 0 and skipped == 0
    # the run did reach what it is there for
    for kind in ("fits the light kernel", "ran on the light kernel", "two chains in one pass", "inside a batch", "reference generator",
                 "through vvm_exec_noalloc", "200 K instances or more", "acquire_id and update_id on one dataset", "has a lean form (runs when eight-wide is forced)",
                 (1, True, True), (4, True, True), (4, True, False)):
        assert kinds.get(kind, 0) > 0, kind
