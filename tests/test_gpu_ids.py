"""GPU parity of update_id / acquire_id (VectorVMRuntime.cpp:1129-1440) against the oracle: ID tables, free list, spawned-ID
log, counters, and the chunk-dependent results at a page size that cuts the Exec into many chunks of the reference."""
import numpy as np
import pytest

import id_cases
from renderer_amd import abi
from vm_harness import IdState, run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32


def _update_case(n, death, seed, table_slack=7):
    rng = np.random.default_rng(seed)
    x = rng.random(n).astype(f32)
    ids = rng.permutation(n + table_slack)[:n].astype(np.int32)
    st = IdState(np.full(n + table_slack, -1, np.int32), np.full(2 * n + 16, -7, np.int32), num_free=5, max_used=3)
    st.free_table[:5] = [100001, 100002, 100003, 100004, 100005]
    return x, ids, st, f32(1.0 - death)


@pytest.mark.parametrize("n,death,page_kb,extra,from_temp", [
    (1, 0.0, 64, 0, False), (1, 1.0, 64, 0, False), (5, 0.5, 64, 0, False), (1000, 0.3, 64, 0, False), (1023, 0.3, 2, 0, False),
    (1024, 0.5, 2, 0, True), (1025, 0.5, 2, 0, False), (4099, 0.9, 2, 0, False), (4099, 0.02, 2, 3, False), (70001, 0.5, 64, 0, False),
    (70001, 0.5, 2, 9, True), (300000, 0.999, 64, 0, False), (300000, 0.0, 64, 0, False), (300000, 1.0, 64, 0, False)])
def test_update_id_matches_the_oracle(n, death, page_kb, extra, from_temp):
    x, ids, st, limit = _update_case(n, death, 5 + n)
    script = id_cases.update_script(extra_outputs=extra, id_from_temp=from_temp)
    consts = [id_cases.update_constants(limit)]
    dtypes = [f32] * (1 + extra) + [np.int32]
    want, got = st.copy(), st.copy()
    ref, ref_n, _ = run_oracle(script, consts, [x, ids], dtypes, n, ids=want, page_kb=page_kb)
    out, out_n, _ = run_hip(script, consts, [x, ids], dtypes, n, ids=got, page_kb=page_kb)
    assert out_n == ref_n
    for a, b in zip(out, ref):
        assert np.array_equal(a[:out_n].view(np.uint32), b[:ref_n].view(np.uint32))
    assert np.array_equal(got.id_table, want.id_table)
    assert np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used) == (want.num_free, want.max_used)


def test_update_id_with_instance_offset():
    n, off = 5000, 37
    x, ids, st, limit = _update_case(n, 0.4, 3)
    script = id_cases.update_script()
    consts = [id_cases.update_constants(limit)]
    want, got = st.copy(), st.copy()
    ref, ref_n, _ = run_oracle(script, consts, [x, ids], [f32, np.int32], n, ids=want, instance_offset=off)
    out, out_n, _ = run_hip(script, consts, [x, ids], [f32, np.int32], n, ids=got, instance_offset=off)
    assert out_n == ref_n
    assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used) == (want.num_free, want.max_used)
    assert got.id_table[ids[x < limit]].min() == off


@pytest.mark.parametrize("n,page_kb", [(1, 64), (7, 64), (1000, 64), (1000, 2), (4099, 2), (200003, 64), (200003, 2)])
def test_acquire_id_matches_the_oracle(n, page_kb):
    rng = np.random.default_rng(n)
    table_max = 3 * n + 40
    st = IdState(np.zeros(0, np.int32), np.full(table_max, -3, np.int32), num_free=n + 11, max_used=0, num_spawned=6, tag=77)
    st.free_table[:n + 11] = rng.permutation(10 * n + 100)[:n + 11]
    consts = [bytes([255] * 4) + bytes(12)]
    want, got = st.copy(), st.copy()
    ref, ref_n, _ = run_oracle(id_cases.spawn_script(), consts, [], [np.int32, np.int32], n, ids=want, page_kb=page_kb)
    out, out_n, _ = run_hip(id_cases.spawn_script(), consts, [], [np.int32, np.int32], n, ids=got, page_kb=page_kb)
    assert out_n == ref_n == n
    assert np.array_equal(out[0][:n], ref[0][:n]) and np.array_equal(out[1][:n], ref[1][:n])
    assert np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.num_spawned) == (want.num_free, want.num_spawned)


def test_id_instructions_need_their_tables():
    x = np.zeros(8, f32)
    ids = np.arange(8, dtype=np.int32)
    with pytest.raises(abi.VVMError) as e:
        run_hip(id_cases.update_script(), [id_cases.update_constants(1.0)], [x, ids], [f32, np.int32], 8)
    assert e.value.code == abi.VVM_E_INVALID_ARG and "FreeIDTable" in str(e.value)
    st = IdState(np.zeros(0, np.int32), np.zeros(64, np.int32), num_free=3, num_spawned=0)
    with pytest.raises(abi.VVMError) as e:
        run_hip(id_cases.spawn_script(), [bytes([255] * 4) + bytes(12)], [], [np.int32, np.int32], 8, ids=st)
    assert e.value.code == abi.VVM_E_INVALID_ARG and "fewer free IDs" in str(e.value)
