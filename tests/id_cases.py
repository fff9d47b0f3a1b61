"""Scripts with the persistent-ID instructions, and a second statement of what they do, in numpy
(VectorVMRuntime.cpp:1129-1440), shared by the oracle tests (CPU) and the parity tests (GPU).

Two results of these instructions depend on the reference's chunking (VectorVMRuntime.cpp:2218-2261), so the numpy
statement takes the chunk size:
  * update_id raises MaxUsedID to the maximum over the FIRST NumOutput entries of each chunk's ID register, NumOutput
    being the chunk's survivor count -- not over the survivors' IDs (:1141-1174);
  * acquire_id logs each chunk's IDs in front of the previous chunk's at the end of FreeIDTable (:1425-1433).
"""
import struct

import numpy as np

from renderer_amd.assembler import FLOAT, INT32, Script
from vm_harness import IdState


def update_script(extra_outputs: int = 0, id_from_temp: bool = False, live_values: int = 0):
    """keep = x < limit; acquireindex(keep); update_id(id); outputs x and id (and extra copies of x).
    live_values: that many more values are alive across the ID instruction (the kernels of many registers); their sum is one more
    output."""
    s = Script()
    x = s.input(0, FLOAT, 0)
    pid = s.input(0, INT32, 0)
    keep = s.temp()
    s.op("cmplt", x, s.const(0), keep)
    s.acquireindex(keep, 0)
    held = s.temps(live_values) if live_values else []
    for i, t in enumerate(held):
        s.op("mad", x if i == 0 else held[i - 1], s.const(2), s.const(2), t)
    id_reg = pid
    if id_from_temp:
        id_reg = s.temp()
        s.op("addi", pid, s.const(1), id_reg)          # constant 1 holds integer 0: the temp equals the input
    s.update_id(id_reg, pid, 0)
    pairs = [(x, s.output(0, FLOAT, 0))]
    for i in range(extra_outputs):
        t = s.temp()
        s.op("add", x, s.const(2), t)
        pairs.append((t, s.output(0, FLOAT, 1 + i)))
    if held:
        acc = held[0]
        for t in held[1:]:
            nxt = s.temp()
            s.op("add", acc, t, nxt)
            acc = nxt
        pairs.append((acc, s.output(0, FLOAT, 1 + extra_outputs)))
    s.output_data(FLOAT, 0, pairs)
    s.output_data(INT32, 0, [(pid, s.output(0, INT32, 0))])
    return s.build()


def update_constants(limit: float) -> bytes:
    return struct.pack("<fif", limit, 0, 1.0) + bytes(4)


def spawn_script():
    """acquire_id -> (id, tag); keep everything; outputs id and tag."""
    s = Script()
    new_id, tag = s.temps(2)
    s.acquire_id(new_id, tag, 0)
    s.acquireindex(s.const(0), 0)
    s.output_data(INT32, 0, [(new_id, s.output(0, INT32, 0)), (tag, s.output(0, INT32, 1))])
    return s.build()


def np_update_id(st: IdState, ids: np.ndarray, keep: np.ndarray, chunk: int, instance_offset: int = 0) -> IdState:
    st = st.copy()
    n = len(ids)
    keep = keep.astype(bool)
    rank = np.cumsum(keep) - keep                                   # survivors before each instance
    st.id_table[ids[keep]] = instance_offset + rank[keep]           # :1191-1193 (and every case of the switch)
    dead = ids[~keep]
    st.free_table[st.num_free:st.num_free + len(dead)] = dead       # chunks in order, instance order within: :1175, 1181
    st.num_free += len(dead)
    for start in range(0, n, chunk):
        nout = int(keep[start:start + chunk].sum())
        if nout:
            m = int(ids[start:start + nout].max())                  # :1141-1174, the uncompacted register
            if m != -1 and m > st.max_used:
                st.max_used = m
    return st


def np_acquire_id(st: IdState, n: int, chunk: int):
    st = st.copy()
    out = st.free_table[st.num_free - n:st.num_free][::-1].copy()   # :1417-1421, chunk after chunk from the end of the free list
    table_max = len(st.free_table)
    spawned = st.num_spawned
    for start in range(0, n, chunk):
        cnt = min(chunk, n - start)
        spawned += cnt
        st.free_table[table_max - spawned:table_max - spawned + cnt] = out[start:start + cnt]       # :1429-1432
    st.num_free -= n
    st.num_spawned += n
    return st, out, np.full(n, st.tag, np.int32)


# ---- ID instructions on more than one dataset of one script ----------------------------------------------------------------------
def update_and_spawn_script():
    """Dataset 0: keep = x < limit, acquireindex, update_id(id).  Dataset 1 (no outputs of its own): acquire_id -> (id, tag), which go
    out with dataset 0's survivors."""
    s = Script()
    x = s.input(0, FLOAT, 0)
    pid = s.input(0, INT32, 0)
    keep = s.temp()
    s.op("cmplt", x, s.const(0), keep)
    s.acquireindex(keep, 0)
    s.update_id(pid, pid, 0)
    new_id, tag = s.temps(2)
    s.acquire_id(new_id, tag, 1)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, 0))])
    s.output_data(INT32, 0, [(pid, s.output(0, INT32, 0)), (new_id, s.output(0, INT32, 1)), (tag, s.output(0, INT32, 2))])
    return s.build()


def two_spawns_script():
    """acquire_id on datasets 0 and 1; everything survives; outputs both IDs and both tags."""
    s = Script()
    id0, tag0, id1, tag1 = s.temps(4)
    s.acquire_id(id0, tag0, 0)
    s.acquire_id(id1, tag1, 1)
    s.acquireindex(s.const(0), 0)
    s.output_data(INT32, 0, [(id0, s.output(0, INT32, 0)), (tag0, s.output(0, INT32, 1)), (id1, s.output(0, INT32, 2)), (tag1, s.output(0, INT32, 3))])
    return s.build()


def two_updates_script():
    """Two acquired datasets with their own keep masks (x < limit0, x < limit1) and their own update_id, on two ID streams."""
    s = Script()
    x = s.input(0, FLOAT, 0)
    pid0, pid1 = s.input(0, INT32, 0), s.input(0, INT32, 1)
    keep0, keep1 = s.temps(2)
    s.op("cmplt", x, s.const(0), keep0)
    s.op("cmplt", x, s.const(1), keep1)
    s.acquireindex(keep0, 0)
    s.acquireindex(keep1, 1)
    s.update_id(pid0, pid0, 0)
    s.update_id(pid1, pid1, 1)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, 0))])
    s.output_data(INT32, 0, [(pid0, s.output(0, INT32, 0))])
    s.output_data(FLOAT, 1, [(x, s.output(1, FLOAT, 0))])
    s.output_data(INT32, 1, [(pid1, s.output(1, INT32, 0))])
    return s.build()


# ---- acquire_id and update_id on ONE dataset of one script ---------------------------------------------------------------------------
def spawn_and_update_script(id_from_temp: bool = False):
    """A spawn script that carries the new particles' first update: acquire_id -> (id, tag); keep = x < limit; acquireindex(keep);
    update_id(id, tag) on the same dataset; outputs x, id and tag of the survivors."""
    s = Script()
    x = s.input(0, FLOAT, 0)
    new_id, tag = s.temps(2)
    s.acquire_id(new_id, tag, 0)
    keep = s.temp()
    s.op("cmplt", x, s.const(0), keep)
    s.acquireindex(keep, 0)
    id_reg = new_id
    if id_from_temp:
        id_reg = s.temp()
        s.op("addi", new_id, s.const(1), id_reg)       # constant 1 holds integer 0: the temp equals the acquired ID
    s.update_id(id_reg, tag, 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, 0))])
    s.output_data(INT32, 0, [(new_id, s.output(0, INT32, 0)), (tag, s.output(0, INT32, 1))])
    return s.build()


def np_spawn_and_update(st: IdState, keep: np.ndarray, chunk: int, instance_offset: int = 0):
    """The reference runs both instructions chunk after chunk on the dataset's one free list (VectorVMRuntime.cpp:1175-1181,
    1411-1435): a chunk takes its IDs from the end of the list, which holds what the chunks before it have freed, logs them as spawned,
    and returns the IDs of its dead instances.  Returns (state, the ID of every instance, in instance order)."""
    st = st.copy()
    keep = keep.astype(bool)
    n = len(keep)
    table_max = len(st.free_table)
    all_ids = np.empty(n, np.int32)
    written = 0
    for start in range(0, n, chunk):
        cnt = min(chunk, n - start)
        ids = st.free_table[st.num_free - cnt:st.num_free][::-1].copy()                               # :1417-1421
        st.num_free -= cnt
        st.num_spawned += cnt
        st.free_table[table_max - st.num_spawned:table_max - st.num_spawned + cnt] = ids                # :1433-1435
        all_ids[start:start + cnt] = ids
        k = keep[start:start + cnt]
        rank = np.cumsum(k) - k
        st.id_table[ids[k]] = instance_offset + written + rank[k]                                      # :1191-1193
        dead = ids[~k]
        st.free_table[st.num_free:st.num_free + len(dead)] = dead                                      # :1175, 1181
        st.num_free += len(dead)
        nout = int(k.sum())
        if nout:
            m = int(ids[:nout].max())                                                                  # :1141-1174
            if m > st.max_used:
                st.max_used = m
        written += nout
    return st, all_ids


def spawn_and_update_with_events_script(events_first: bool = False):
    """spawn_and_update_script with a second acquired dataset that receives the instances that die (x and the ID they were given):
    acquire_id and update_id on dataset 0 of a script that acquires two datasets."""
    s = Script()
    x = s.input(0, FLOAT, 0)
    new_id, tag = s.temps(2)
    s.acquire_id(new_id, tag, 0)
    keep, dead = s.temps(2)
    s.op("cmplt", x, s.const(0), keep)
    s.op("cmpge", x, s.const(0), dead)

    def events():
        s.acquireindex(dead, 1)
        s.output_data(FLOAT, 1, [(x, s.output(1, FLOAT, 0))])
        s.output_data(INT32, 1, [(new_id, s.output(1, INT32, 0))])

    if events_first:
        events()
    s.acquireindex(keep, 0)
    s.update_id(new_id, tag, 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, 0))])
    s.output_data(INT32, 0, [(new_id, s.output(0, INT32, 0)), (tag, s.output(0, INT32, 1))])
    if not events_first:
        events()
    return s.build()
