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


@pytest.mark.parametrize("n,death,page_kb,live,from_temp", [(5, 0.5, 64, 20, False), (4099, 0.3, 2, 20, True), (70001, 0.5, 64, 28, False),
                                                         (70001, 0.9, 2, 36, True), (300000, 0.02, 64, 40, False)])
def test_update_id_with_many_live_values(n, death, page_kb, live, from_temp):
    """the same on the four-wide kernel of many registers that has the ID instructions: ordinary instructions through the hand-written
    routines, 32 registers in VGPRs, the rest (live = 36, 40) in LDS slots"""
    x, ids, st, limit = _update_case(n, death, 11 + n)
    script = id_cases.update_script(id_from_temp=from_temp, live_values=live)
    consts = [id_cases.update_constants(limit)]
    dtypes = [f32, f32, np.int32]
    want, got = st.copy(), st.copy()
    ref, ref_n, _ = run_oracle(script, consts, [x, ids], dtypes, n, ids=want, page_kb=page_kb)
    out, out_n, info = run_hip(script, consts, [x, ids], dtypes, n, ids=got, page_kb=page_kb)
    assert info.InstancesPerLane == 4 and info.NumVirtualRegisters > 16 and info.LdsSpillRegisters == max(0, info.NumVirtualRegisters - 32)
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


# ---- ID instructions on several datasets of one script ---------------------------------------------------------------------------
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


def _same_tables(got, want):
    assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used, got.num_spawned) == (want.num_free, want.max_used, want.num_spawned)


@pytest.mark.parametrize("n,page_kb", [(7, 64), (1000, 64), (4099, 2), (150001, 2), (150001, 64)])
def test_update_id_and_acquire_id_on_two_datasets(n, page_kb):
    rng = np.random.default_rng(n)
    x = rng.random(n).astype(f32)
    ids, upd = _update_state(n, 1)
    spawn = _spawn_state(n, 2, tag=31)
    script, consts, dtypes = id_cases.update_and_spawn_script(), [id_cases.update_constants(0.6)], [f32, np.int32, np.int32, np.int32]
    want_upd, want_spawn, got_upd, got_spawn = upd.copy(), spawn.copy(), upd.copy(), spawn.copy()
    ref, ref_n, _ = run_oracle(script, consts, [x, ids], dtypes, n, ids=want_upd, more_datasets=[[]], more_ids=[want_spawn], page_kb=page_kb)
    out, out_n, _ = run_hip(script, consts, [x, ids], dtypes, n, ids=got_upd, more_datasets=[[]], more_ids=[got_spawn], page_kb=page_kb)
    assert out_n == ref_n
    for a, b in zip(out, ref):
        assert np.array_equal(a[:out_n].view(np.uint32), b[:ref_n].view(np.uint32))
    _same_tables(got_upd, want_upd)
    _same_tables(got_spawn, want_spawn)


@pytest.mark.parametrize("n,page_kb", [(5, 64), (3000, 2), (100003, 64)])
def test_acquire_id_on_two_datasets(n, page_kb):
    a, b = _spawn_state(n, 3, tag=5), _spawn_state(n, 4, tag=6)
    consts = [bytes([255] * 4) + bytes(12)]
    want_a, want_b, got_a, got_b = a.copy(), b.copy(), a.copy(), b.copy()
    ref, ref_n, _ = run_oracle(id_cases.two_spawns_script(), consts, [], [np.int32] * 4, n, ids=want_a, more_datasets=[[]], more_ids=[want_b], page_kb=page_kb)
    out, out_n, _ = run_hip(id_cases.two_spawns_script(), consts, [], [np.int32] * 4, n, ids=got_a, more_datasets=[[]], more_ids=[got_b], page_kb=page_kb)
    assert out_n == ref_n == n
    for x, y in zip(out, ref):
        assert np.array_equal(x[:n], y[:n])
    _same_tables(got_a, want_a)
    _same_tables(got_b, want_b)


@pytest.mark.parametrize("flags", [0, abi.VVM_EXEC_SEPARATE_PASSES])
@pytest.mark.parametrize("n,page_kb", [(9, 64), (3001, 2), (120001, 2)])
def test_update_id_on_two_acquired_datasets(n, page_kb, flags):
    import struct
    rng = np.random.default_rng(n + 1)
    x = rng.random(n).astype(f32)
    ids0, st0 = _update_state(n, 11)
    ids1, st1 = _update_state(n, 12)
    consts = [struct.pack("<ff", 0.7, 0.2) + bytes(8)]
    script = id_cases.two_updates_script()
    want0, want1, got0, got1 = st0.copy(), st1.copy(), st0.copy(), st1.copy()
    ref_more, got_more = [], []
    ref, ref_n, _ = run_oracle(script, consts, [x, ids0, ids1], [f32, np.int32], n, ids=want0, more_datasets=[[f32, np.int32]], more_results=ref_more,
                               more_ids=[want1], page_kb=page_kb)
    out, out_n, _ = run_hip(script, consts, [x, ids0, ids1], [f32, np.int32], n, ids=got0, more_datasets=[[f32, np.int32]], more_results=got_more,
                            more_ids=[got1], page_kb=page_kb, flags=flags)
    assert out_n == ref_n and got_more[0][1] == ref_more[0][1]
    for a, b in zip(out, ref):
        assert np.array_equal(a[:out_n].view(np.uint32), b[:ref_n].view(np.uint32))
    m = ref_more[0][1]
    for a, b in zip(got_more[0][0], ref_more[0][0]):
        assert np.array_equal(a[:m].view(np.uint32), b[:m].view(np.uint32))
    _same_tables(got0, want0)
    _same_tables(got1, want1)


def _spawn_and_update_case(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.random(n).astype(f32)
    m = 3 * n + 50
    st = IdState(np.full(m, -1, np.int32), np.full(m, -7, np.int32), num_free=n + 20, max_used=3, num_spawned=2, tag=9)
    st.free_table[:n + 20] = rng.permutation(m)[:n + 20]
    return x, st


@pytest.mark.parametrize("n,death,page_kb,from_temp,offset", [(1, 0.0, 64, False, 0), (1, 1.0, 64, False, 0), (10, 0.4, 64, False, 5), (5000, 0.4, 64, True, 0),
                                                              (5000, 0.4, 16, False, 3), (5000, 1.0, 16, False, 0), (5000, 0.0, 16, True, 0), (20001, 0.35, 8, False, 7),
                                                              (20001, 0.9, 2, True, 0), (0, 0.5, 64, False, 0)])
def test_acquire_id_and_update_id_on_one_dataset(n, death, page_kb, from_temp, offset):
    """A spawn script that carries the first update: a chunk of the reference's chunking takes the IDs the chunks before it have freed
    (VectorVMRuntime.cpp:1175, 1411-1423).  The Exec runs chunk after chunk; tables, counters, the chunk-dependent MaxUsedID and the
    survivors' streams equal the oracle's at one chunk and at many."""
    x, st = _spawn_and_update_case(max(n, 1), n + page_kb)
    limit = f32(1.0 - death)
    script = id_cases.spawn_and_update_script(id_from_temp=from_temp)
    consts = [id_cases.update_constants(limit)]
    dtypes = [f32, np.int32, np.int32]
    want, got = st.copy(), st.copy()
    ref, ref_n, _ = run_oracle(script, consts, [x], dtypes, n, ids=want, page_kb=page_kb, instance_offset=offset)
    out, out_n, _ = run_hip(script, consts, [x], dtypes, n, ids=got, page_kb=page_kb, instance_offset=offset)
    assert out_n == ref_n == int((x[:n] < limit).sum())
    for a, b in zip(out, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))          # (survivors and the untouched fill around them)
    assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used, got.num_spawned) == (want.num_free, want.max_used, want.num_spawned)


@pytest.mark.parametrize("n,death,page_kb,events_first,offset", [(1, 0.0, 64, False, 0), (1, 1.0, 64, True, 0), (5000, 0.4, 64, False, 2), (5000, 0.4, 16, True, 0),
                                                                  (20001, 0.35, 8, False, 7), (20001, 0.9, 2, True, 0), (300_000, 0.5, 64, False, 0), (0, 0.5, 64, False, 0)])
def test_acquire_id_and_update_id_on_one_dataset_of_a_script_with_two(n, death, page_kb, events_first, offset):
    """The same with a second acquired dataset for the instances that die: every chunk runs one launch per dataset, the second
    dataset's pass runs the acquire_id too (it writes the IDs out) and reads the same copy of the free-list entries, and every dataset's
    outputs go behind its own survivors of the chunks before."""
    x, st = _spawn_and_update_case(max(n, 1), n + page_kb)
    limit = f32(1.0 - death)
    script = id_cases.spawn_and_update_with_events_script(events_first)
    consts = [id_cases.update_constants(limit)]
    dtypes = [f32, np.int32, np.int32]
    want, got = st.copy(), st.copy()
    ref_more, got_more = [], []
    ref, ref_n, _ = run_oracle(script, consts, [x], dtypes, n, ids=want, page_kb=page_kb, instance_offset=offset, more_datasets=[[f32, np.int32]], more_results=ref_more)
    out, out_n, _ = run_hip(script, consts, [x], dtypes, n, ids=got, page_kb=page_kb, instance_offset=offset, more_datasets=[[f32, np.int32]], more_results=got_more)
    assert out_n == ref_n == int((x[:n] < limit).sum())
    assert got_more[0][1] == ref_more[0][1] == n - out_n
    for a, b in zip(out, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))          # (survivors and the untouched fill around them)
    for a, b in zip(got_more[0][0], ref_more[0][0]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used, got.num_spawned) == (want.num_free, want.max_used, want.num_spawned)


def test_acquire_id_and_update_id_on_one_dataset_twice_in_a_row():
    """Two Execs in a row on one state: the second starts from the tables and counters the first left.  (As a member of a batch, with
    page sizes, offsets and other instructions around it: tools/gpu_fuzz_linker.py, one script in sixteen.)"""
    n, page_kb = 6000, 16
    rng = np.random.default_rng(5)
    x = rng.random(n).astype(f32)
    st = IdState(np.full(4 * n + 100, -1, np.int32), np.full(8 * n, -7, np.int32), num_free=2 * n + 40, max_used=3, num_spawned=2, tag=9)
    st.free_table[:2 * n + 40] = rng.permutation(4 * n + 100)[:2 * n + 40]      # (IDs for two Execs; the spawned-ID log of both fits behind them)
    script = id_cases.spawn_and_update_script()
    consts = [id_cases.update_constants(f32(0.5))]
    dtypes = [f32, np.int32, np.int32]
    want, got = st.copy(), st.copy()
    ref, ref_n, _ = run_oracle(script, consts, [x], dtypes, n, ids=want, page_kb=page_kb)
    ref2, ref2_n, _ = run_oracle(script, consts, [x], dtypes, n, ids=want, page_kb=page_kb)
    out, out_n, _ = run_hip(script, consts, [x], dtypes, n, ids=got, page_kb=page_kb, repeat=2)
    assert out_n == ref2_n
    for a, b in zip(out, ref2):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(got.id_table, want.id_table) and np.array_equal(got.free_table, want.free_table)
    assert (got.num_free, got.max_used, got.num_spawned) == (want.num_free, want.max_used, want.num_spawned)


def test_update_id_ahead_of_acquire_id_on_one_dataset_is_refused():
    """In that order a chunk would take back the IDs its own dead instances have just returned: not built."""
    from renderer_amd.assembler import FLOAT, INT32, Script
    s = Script()
    x = s.input(0, FLOAT, 0)
    pid = s.input(0, INT32, 0)
    keep = s.temp()
    s.op("cmplt", x, s.const(0), keep)
    s.acquireindex(keep, 0)
    s.update_id(pid, pid, 0)
    t0, t1 = s.temps(2)
    s.acquire_id(t0, t1, 0)
    s.output_data(INT32, 0, [(t0, s.output(0, INT32, 0))])
    with pytest.raises(abi.VVMError) as e:
        run_hip(s.build(), [bytes(16)], [np.zeros(8, f32), np.zeros(8, np.int32)], [np.int32], 8)
    assert "one dataset" in str(e.value)
