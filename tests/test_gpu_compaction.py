"""libvvm_hip.so's compaction against the tables the reference itself holds (tests/golden/compaction_tables.json, the numbers of
VectorVMRuntime.cpp:132-186): expected outputs come from the tables, not from the oracle."""
import numpy as np
import pytest

import compaction_cases as cc
from renderer_amd import abi
from vm_harness import run_hip

pytestmark = pytest.mark.gpu


def _check(keep, flags=0, instance_offset=0, no_alloc=False):
    t = cc.load_tables()
    ins = cc.inputs_for(keep)
    outs, surv, _ = run_hip(cc.compaction_script(), [bytes(16)], ins, cc.OUT_DTYPES, len(keep), instance_offset=instance_offset, flags=flags, no_alloc=no_alloc)
    x, _, i, h = ins
    wants = [cc.by_the_tables(x, keep, t), cc.by_the_tables(i, keep, t), cc.by_the_tables(x.astype(np.float16), keep, t), cc.by_the_tables(h, keep, t)]
    assert surv == int(keep.sum())
    for got, want in zip(outs, wants):
        np.testing.assert_array_equal(got[instance_offset:instance_offset + surv].view(np.uint8), want.view(np.uint8))
        assert (got[instance_offset + surv:].view(np.uint8) == 0xCD).all()      # nothing past the survivors


@pytest.mark.parametrize("flags", [0, abi.VVM_EXEC_FORCE_EIGHT_WIDE, abi.VVM_EXEC_NO_VGPR_WIDE])
def test_every_pair_of_masks(flags):
    _check(cc.every_pair_of_masks(), flags)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 7, 8, 9, 255, 256, 257, 1023, 1025, 2047, 2049])
def test_ragged_ends(n):
    rng = np.random.default_rng(n)
    _check(rng.random(n) < 0.5)
    _check(np.ones(n, dtype=bool), abi.VVM_EXEC_FORCE_EIGHT_WIDE)


@pytest.mark.parametrize("flags", [0, abi.VVM_EXEC_NO_EIGHT_WIDE])
def test_many_tiles(flags):
    """Several hundred tiles: the position of a tile's first survivor comes from the tiles before it (look-back)."""
    rng = np.random.default_rng(5)
    _check(rng.random(700001) < 0.37, flags, instance_offset=3)


def test_through_the_entry_point_that_does_not_allocate():
    rng = np.random.default_rng(6)
    _check(rng.random(300007) < 0.5, no_alloc=True)
