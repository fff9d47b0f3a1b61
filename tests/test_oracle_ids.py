"""CPU tests of the oracle's update_id / acquire_id against the numpy statement in id_cases.py."""
import struct

import numpy as np
import pytest

import id_cases
from vm_harness import IdState, run_oracle

f32 = np.float32


def _update_case(n, death, seed, table_slack=7):
    rng = np.random.default_rng(seed)
    x = rng.random(n).astype(f32)
    ids = rng.permutation(n + table_slack)[:n].astype(np.int32)
    st = IdState(np.full(n + table_slack, -1, np.int32), np.full(2 * n + 16, -7, np.int32), num_free=5, max_used=3)
    st.free_table[:5] = [100001, 100002, 100003, 100004, 100005]
    return x, ids, st, f32(1.0 - death)


@pytest.mark.parametrize("n,death,page_kb", [(1, 0.0, 64), (1, 1.0, 64), (5, 0.5, 64), (1000, 0.3, 64), (1000, 0.3, 2), (4099, 0.9, 2), (4099, 0.02, 2),
                                             (20000, 0.5, 64)])
def test_update_id_matches_numpy(n, death, page_kb):
    x, ids, st, limit = _update_case(n, death, 5 + n)
    script = id_cases.update_script()
    got = st.copy()
    outs, surv, frame = run_oracle(script, [id_cases.update_constants(limit)], [x, ids], [f32, np.int32], n, ids=got, page_kb=page_kb)
    keep = x < limit
    chunk = int(frame.ctx.Internal.MaxInstancesPerChunk)
    if page_kb == 2:
        assert chunk < n                                            # the chunk-dependent part is exercised
    want = id_cases.np_update_id(st, ids, keep, chunk)
    assert surv == int(keep.sum())
    assert np.array_equal(outs[1][:surv], ids[keep])
    assert np.array_equal(got.id_table, want.id_table)
    assert np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used) == (want.num_free, want.max_used)


def test_update_id_max_is_taken_over_the_uncompacted_register():
    """Four instances, the first one dies: one survivor per... NumOutput = 3, so the maximum runs over ids[0:3] and misses ids[3]."""
    x = np.array([2.0, 0.0, 0.0, 0.0], f32)
    ids = np.array([1, 2, 3, 9], np.int32)
    st = IdState(np.full(16, -1, np.int32), np.zeros(16, np.int32), num_free=0, max_used=-1)
    run_oracle(id_cases.update_script(), [id_cases.update_constants(1.0)], [x, ids], [f32, np.int32], 4, ids=st)
    assert st.max_used == 3 and st.num_free == 1 and st.free_table[0] == 1
    assert list(st.id_table[[2, 3, 9]]) == [0, 1, 2]


@pytest.mark.parametrize("n,page_kb", [(1, 64), (7, 64), (1000, 64), (1000, 2), (4099, 2)])
def test_acquire_id_matches_numpy(n, page_kb):
    rng = np.random.default_rng(n)
    table_max = 3 * n + 40
    st = IdState(np.zeros(0, np.int32), np.full(table_max, -3, np.int32), num_free=n + 11, max_used=0, num_spawned=6, tag=77)
    st.free_table[:n + 11] = rng.permutation(10 * n + 100)[:n + 11]
    got = st.copy()
    outs, surv, frame = run_oracle(id_cases.spawn_script(), [bytes([255] * 4) + bytes(12)], [], [np.int32, np.int32], n, ids=got, page_kb=page_kb)
    chunk = int(frame.ctx.Internal.MaxInstancesPerChunk)
    if page_kb == 2:
        assert chunk < n
    want, want_ids, want_tags = id_cases.np_acquire_id(st, n, chunk)
    assert surv == n
    assert np.array_equal(outs[0][:n], want_ids) and np.array_equal(outs[1][:n], want_tags)
    assert np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.num_spawned) == (want.num_free, want.num_spawned)


# ---- several datasets with ID instructions in one script (VectorVMRuntime.cpp:1134-1135, 1406-1408: every instruction names its dataset)
def _spawn_state(n, seed, tag):
    rng = np.random.default_rng(seed)
    st = IdState(np.zeros(0, np.int32), np.full(3 * n + 40, -3, np.int32), num_free=n + 9, max_used=0, num_spawned=4, tag=tag)
    st.free_table[:n + 9] = rng.permutation(10 * n + 100)[:n + 9]
    return st


def _update_state(n, seed):
    rng = np.random.default_rng(seed)
    ids = rng.permutation(n + 5)[:n].astype(np.int32)
    st = IdState(np.full(n + 5, -1, np.int32), np.full(2 * n + 16, -7, np.int32), num_free=3, max_used=2)
    st.free_table[:3] = [900001, 900002, 900003]
    return ids, st


@pytest.mark.parametrize("n,page_kb", [(7, 64), (1000, 64), (4099, 2)])
def test_update_id_and_acquire_id_on_two_datasets_match_numpy(n, page_kb):
    rng = np.random.default_rng(n)
    x = rng.random(n).astype(f32)
    ids, upd = _update_state(n, 1)
    spawn = _spawn_state(n, 2, tag=31)
    got_upd, got_spawn = upd.copy(), spawn.copy()
    outs, surv, frame = run_oracle(id_cases.update_and_spawn_script(), [id_cases.update_constants(0.6)], [x, ids], [f32, np.int32, np.int32, np.int32], n,
                                   ids=got_upd, more_datasets=[[]], more_ids=[got_spawn], page_kb=page_kb)
    chunk = frame.ctx.Internal.MaxInstancesPerChunk
    keep = x < f32(0.6)
    want_upd = id_cases.np_update_id(upd, ids, keep, chunk)
    want_spawn, new_ids, tags = id_cases.np_acquire_id(spawn, n, chunk)
    assert surv == int(keep.sum())
    assert np.array_equal(outs[1][:surv], ids[keep]) and np.array_equal(outs[2][:surv], new_ids[keep]) and np.array_equal(outs[3][:surv], tags[keep])
    for got, want in ((got_upd, want_upd), (got_spawn, want_spawn)):
        assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
        assert (got.num_free, got.max_used, got.num_spawned) == (want.num_free, want.max_used, want.num_spawned)


def test_two_update_ids_on_two_datasets_match_numpy():
    n = 3001
    rng = np.random.default_rng(5)
    x = rng.random(n).astype(f32)
    ids0, st0 = _update_state(n, 11)
    ids1, st1 = _update_state(n, 12)
    got0, got1 = st0.copy(), st1.copy()
    more = []
    consts = [struct.pack("<ff", 0.7, 0.2) + bytes(8)]
    outs, surv, frame = run_oracle(id_cases.two_updates_script(), consts, [x, ids0, ids1], [f32, np.int32], n, ids=got0,
                                   more_datasets=[[f32, np.int32]], more_results=more, more_ids=[got1], page_kb=2)
    chunk = frame.ctx.Internal.MaxInstancesPerChunk
    for got, st, ids, limit in ((got0, st0, ids0, 0.7), (got1, st1, ids1, 0.2)):
        want = id_cases.np_update_id(st, ids, x < f32(limit), chunk)
        assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
        assert (got.num_free, got.max_used) == (want.num_free, want.max_used)
    assert surv == int((x < f32(0.7)).sum()) and more[0][1] == int((x < f32(0.2)).sum())


def _spawn_and_update_case(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.random(n).astype(f32)
    m = 3 * n + 50
    st = IdState(np.full(m, -1, np.int32), np.full(m, -7, np.int32), num_free=n + 20, max_used=3, num_spawned=2, tag=9)
    st.free_table[:n + 20] = rng.permutation(m)[:n + 20]
    return x, st


@pytest.mark.parametrize("n,death,page_kb", [(1, 0.0, 64), (1, 1.0, 64), (10, 0.4, 64), (5000, 0.4, 64), (5000, 0.4, 16), (5000, 1.0, 16), (5000, 0.0, 16),
                                             (20001, 0.35, 8), (20001, 0.9, 2)])
def test_acquire_id_and_update_id_on_one_dataset_match_numpy(n, death, page_kb):
    """A chunk takes the IDs the chunks before it have freed: the result depends on the page size (VectorVMRuntime.cpp:1175, 1411-1423)."""
    x, st = _spawn_and_update_case(n, n + page_kb)
    limit = f32(1.0 - death)
    got = st.copy()
    outs, surv, frame = run_oracle(id_cases.spawn_and_update_script(), [id_cases.update_constants(limit)], [x], [f32, np.int32, np.int32], n, ids=got, page_kb=page_kb,
                                   instance_offset=3)
    keep = x < limit
    want, ids = id_cases.np_spawn_and_update(st, keep, frame.ctx.Internal.MaxInstancesPerChunk, instance_offset=3)
    assert surv == int(keep.sum())
    assert np.array_equal(outs[0][3:3 + surv], x[keep]) and np.array_equal(outs[1][3:3 + surv], ids[keep]) and np.all(outs[2][3:3 + surv] == 9)
    assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used, got.num_spawned) == (want.num_free, want.max_used, want.num_spawned)
