"""The oracle's compaction against the tables the reference itself holds (tests/golden/compaction_tables.json, the numbers of
VectorVMRuntime.cpp:132-186): first the oracle's own tables, which it builds from a rule, then what its Execs write."""
import ctypes as C

import numpy as np
import pytest

import compaction_cases as cc
from oracle_binding import load_oracle
from vm_harness import run_oracle


def test_the_fixture_says_survivors_move_to_the_front_in_instance_order():
    """What the reference's numbers mean, stated once in plain terms: row m lists the bytes of the instances whose bit is set in m,
    lowest instance first, then 0xFF; the advance is their size."""
    t = cc.load_tables()
    for size, sh, adv in ((4, "shuffle32", "advance32"), (2, "shuffle16", "advance16")):
        rows = t[sh].reshape(16, 16)
        for m in range(16):
            lanes = [l for l in range(4) if m & (1 << l)]
            want = [l * size + b for l in lanes for b in range(size)]
            assert list(rows[m][:len(want)]) == want
            assert all(v == 0xFF for v in rows[m][len(want):])
            assert t[adv][m] == size * len(lanes)


def test_the_oracle_builds_the_tables_the_reference_holds():
    be = load_oracle()
    bufs = [(C.c_uint8 * n)() for n in (256, 256, 16, 16)]
    be.lib.vvm_oracle_compaction_tables(*bufs)
    t = cc.load_tables()
    for buf, key in zip(bufs, ("shuffle32", "shuffle16", "advance32", "advance16")):
        assert bytes(buf) == t[key].tobytes(), key


def _check(keep, strict=False, page_kb=64, instance_offset=0):
    t = cc.load_tables()
    ins = cc.inputs_for(keep)
    outs, surv, _ = run_oracle(cc.compaction_script(), [bytes(16)], ins, cc.OUT_DTYPES, len(keep), instance_offset=instance_offset,
                               strict=strict, page_kb=page_kb)
    x, _, i, h = ins
    wants = [cc.by_the_tables(x, keep, t), cc.by_the_tables(i, keep, t), cc.by_the_tables(x.astype(np.float16), keep, t), cc.by_the_tables(h, keep, t)]
    assert surv == int(keep.sum())
    for got, want in zip(outs, wants):
        assert len(want) == surv
        np.testing.assert_array_equal(got[instance_offset:instance_offset + surv].view(np.uint8), want.view(np.uint8))


def test_every_pair_of_masks():
    _check(cc.every_pair_of_masks())


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 63, 64, 65, 1000])
def test_ragged_ends(n):
    rng = np.random.default_rng(n)
    _check(rng.random(n) < 0.5)
    _check(np.ones(n, dtype=bool))


@pytest.mark.parametrize("page_kb", [1, 64])
@pytest.mark.parametrize("strict", [False, True])
def test_many_chunks(page_kb, strict):
    """Chunks of a 1 KB page end every few groups: the destination carries on where the chunk before stopped (VectorVMRuntime.cpp:874, 2124)."""
    rng = np.random.default_rng(11)
    _check(rng.random(4099) < 0.6, strict=strict, page_kb=page_kb, instance_offset=5)
