"""Shared test helpers: run one script through the CPU oracle or through libvvm_hip.so on the same inputs."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from renderer_amd import abi
from renderer_amd.assembler import BuiltScript

PAD = 16   # extra elements past N in every stream (the reference needs align4(N) + 16 bytes of slack)


class IdState:
    """The persistent-ID tables of a dataset, as numpy arrays and ints; run_oracle / run_hip update it in place."""

    def __init__(self, id_table: np.ndarray, free_table: np.ndarray, num_free: int, max_used: int = -1, num_spawned: int = 0, tag: int = 0):
        self.id_table = np.ascontiguousarray(id_table, dtype=np.int32)
        self.free_table = np.ascontiguousarray(free_table, dtype=np.int32)      # its length is FreeIDTable->Max()
        self.num_free, self.max_used, self.num_spawned, self.tag = int(num_free), int(max_used), int(num_spawned), int(tag)

    def copy(self) -> "IdState":
        return IdState(self.id_table.copy(), self.free_table.copy(), self.num_free, self.max_used, self.num_spawned, self.tag)

    def attach(self, ds: abi.DataSet, id_ptr: int, free_ptr: int) -> None:
        def table(ptr, n):
            t = abi.IdTable()
            t.data = ptr or None
            t.num, t.max = n, n
            return t
        ds.set_id_tables(table(id_ptr, len(self.id_table)), table(free_ptr, len(self.free_table)), table(0, 0),
                         self.num_free, self.max_used, self.num_spawned, self.tag)

    def read_counters(self, ds: abi.DataSet) -> None:
        self.num_free, self.max_used, self.num_spawned = ds.id_tables[3].value, ds.id_tables[4].value, ds.id_tables[5].value


def _type_offsets(streams: Sequence[np.ndarray]) -> List[int]:
    """Streams are ordered float32 first, then int32, then float16 (half)."""
    nf = sum(1 for s in streams if s.dtype == np.float32)
    ni = sum(1 for s in streams if s.dtype in (np.int32, np.uint32))
    return [0, nf, nf + ni]


def run_oracle(script: BuiltScript, const_tables: Sequence[bytes], inputs: Sequence[np.ndarray],
               out_dtypes: Sequence, n: int, instance_offset: int = 0, ext_fns: Sequence = (),
               seed: int = 0, exec_index_base: int = 0, rand_instance_base: int = 0, strict: bool = False,
               page_kb: int = 64, out_fill: int = 0xCD, ids: Optional[IdState] = None,
               more_datasets: Sequence[Sequence] = (), more_results: Optional[list] = None,
               more_ids: Sequence[Optional[IdState]] = (), xorwow: Optional[Sequence[int]] = None) -> Tuple[List[np.ndarray], int, abi.CallFrame]:
    """xorwow: start state of the reference's random generator (24 words); None = the counter-based generator keyed by `seed`.
    more_datasets: output dtypes of datasets 1, 2, ... (outputs only); their (streams, survivors) are appended to more_results.
    more_ids: the ID tables of datasets 1, 2, ... (None: the dataset has none)."""
    from oracle_binding import load_oracle
    be = load_oracle()
    be.set_strict(strict)
    be.set_page_kb(page_kb)
    be.set_options(seed, exec_index_base, rand_instance_base)
    be.set_xorwow(xorwow)
    total = instance_offset + n + PAD
    ins = []
    for a in inputs:
        buf = np.zeros(total, dtype=a.dtype)
        buf[instance_offset:instance_offset + n] = a[:n]
        ins.append(buf)
    outs = [np.frombuffer(bytes([out_fill]) * (total * np.dtype(dt).itemsize), dtype=dt).copy() for dt in out_dtypes]
    ds = abi.DataSet([a.ctypes.data for a in ins], [a.ctypes.data for a in outs], instance_offset,
                     _type_offsets(ins), _type_offsets(outs))
    if ids is not None:
        ids.attach(ds, ids.id_table.ctypes.data, ids.free_table.ctypes.data)
    more_outs = [[np.frombuffer(bytes([out_fill]) * (total * np.dtype(dt).itemsize), dtype=dt).copy() for dt in dts] for dts in more_datasets]
    more_ds = [abi.DataSet([], [a.ctypes.data for a in o], instance_offset, [0, 0, 0], _type_offsets(o)) for o in more_outs]
    for d, st in zip(more_ds, more_ids):
        if st is not None:
            st.attach(d, st.id_table.ctypes.data, st.free_table.ctypes.data)
    state = be.alloc_state(script.blob)
    try:
        frame = abi.CallFrame(state, [ds] + more_ds, n, const_tables, ext_fns)
        rc = be.exec(frame)
        if rc != 0:
            raise abi.VVMError(rc, "vvm_oracle_exec")
        surv = frame.survivors(0)
        if more_results is not None:
            more_results.extend((o, frame.survivors(1 + i)) for i, o in enumerate(more_outs))
        if ids is not None:
            ids.read_counters(frame.datasets[0])
        for i, st in enumerate(more_ids):
            if st is not None:
                st.read_counters(frame.datasets[1 + i])
    finally:
        be.free_state(state)
        be.set_xorwow(None)
    return outs, surv, frame


def run_hip(script: BuiltScript, const_tables: Sequence[bytes], inputs: Sequence[np.ndarray],
            out_dtypes: Sequence, n: int, instance_offset: int = 0, ext_fns: Sequence = (),
            seed: int = 0, exec_index_base: int = 0, rand_instance_base: int = 0, flags: int = 0,
            out_fill: int = 0xCD, repeat: int = 1, ids: Optional[IdState] = None, page_kb: int = 64,
            more_datasets: Sequence[Sequence] = (), more_results: Optional[list] = None,
            batch_with: Optional[Sequence] = None, device_counts: Optional[list] = None,
            more_ids: Sequence[Optional[IdState]] = (), xorwow: Optional[Sequence[int]] = None,
            last_width: Optional[list] = None, no_alloc: bool = False, batch_handle: bool = False) -> Tuple[List[np.ndarray], int, abi.ProgramInfo]:
    """batch_handle (with batch_with): the batch runs through a vvm_batch handle, twice (the second call finds everything in place).
    no_alloc: the Exec goes through vvm_exec_noalloc (a refusal raises VVMError like any other).
    last_width: a list; the instances per lane of the kernel the Exec's last launch ran on is appended to it.
    Same contract as run_oracle, through the C ABI of libvvm_hip.so with streams in device memory.
    batch_with: call frames of other emitters; the Exec then runs as the middle member of one vvm_exec_batch call with them.
    device_counts: a list; the Exec gets a buffer of eight ints (filled with -1) as vvm_exec_options.DeviceCounts, whose contents
    after the Exec are appended to the list."""
    be = abi.load_hip_backend()
    L = be.lib
    total = instance_offset + n + PAD
    dev_in, dev_out, host_out, dev_ids = [], [], [], []
    be.set_page_size_kb(page_kb)
    try:
        for a in inputs:
            buf = np.zeros(total, dtype=a.dtype)
            buf[instance_offset:instance_offset + n] = a[:n]
            p = L.vvm_device_alloc(buf.nbytes)
            assert p, be.error_text()
            dev_in.append(p)
            be._check(L.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
        for dt in out_dtypes:
            nbytes = total * np.dtype(dt).itemsize
            p = L.vvm_device_alloc(nbytes)
            assert p, be.error_text()
            dev_out.append(p)
            be._check(L.vvm_device_memset(p, out_fill, nbytes), "memset")
        in_off = _type_offsets([np.empty(0, a.dtype) for a in inputs])
        out_off = _type_offsets([np.empty(0, dt) for dt in out_dtypes])
        ds = abi.DataSet(dev_in, dev_out, instance_offset, in_off, out_off)
        if ids is not None:
            for a in (ids.id_table, ids.free_table):
                p = L.vvm_device_alloc(max(a.nbytes, 4))
                assert p, be.error_text()
                dev_ids.append(p)
                if a.nbytes:
                    be._check(L.vvm_device_upload(p, a.ctypes.data, a.nbytes), "upload")
            ids.attach(ds, dev_ids[0], dev_ids[1])
        more_dev = []
        for dts in more_datasets:
            ptrs = []
            for dt in dts:
                nbytes = total * np.dtype(dt).itemsize
                p = L.vvm_device_alloc(nbytes)
                assert p, be.error_text()
                dev_ids.append(p)          # (freed with the other extra allocations)
                ptrs.append(p)
                be._check(L.vvm_device_memset(p, out_fill, nbytes), "memset")
            more_dev.append(ptrs)
        more_ds = [abi.DataSet([], ptrs, instance_offset, [0, 0, 0], _type_offsets([np.empty(0, dt) for dt in dts])) for ptrs, dts in zip(more_dev, more_datasets)]
        more_id_dev = []
        for d, st in zip(more_ds, more_ids):
            ptrs = []
            if st is not None:
                for a in (st.id_table, st.free_table):
                    p = L.vvm_device_alloc(max(a.nbytes, 4))
                    assert p, be.error_text()
                    dev_ids.append(p)
                    ptrs.append(p)
                    if a.nbytes:
                        be._check(L.vvm_device_upload(p, a.ctypes.data, a.nbytes), "upload")
                st.attach(d, ptrs[0], ptrs[1])
            more_id_dev.append(ptrs)
        state = be.alloc_state(script.blob)
        try:
            frame = abi.CallFrame(state, [ds] + more_ds, n, const_tables, ext_fns)
            opt = abi.ExecOptions(seed, exec_index_base, rand_instance_base, None, flags)
            if xorwow is not None:
                xorwow_words = (C.c_uint32 * 24)(*xorwow)      # (alive until the Exec has returned)
                opt.RandXorwowState = xorwow_words
            if device_counts is not None:
                p = L.vvm_device_alloc(32)
                assert p, be.error_text()
                dev_ids.append(p)
                be._check(L.vvm_device_memset(p, 0xFF, 32), "memset")
                opt.DeviceCounts = p
            for _ in range(repeat):
                if no_alloc:
                    be.exec_noalloc(frame, opt)
                elif batch_with is None:
                    be.exec_ex(frame, opt)
                else:
                    others = list(batch_with)
                    half = len(others) // 2
                    plain = abi.ExecOptions(0, 0, 0, None, 0)
                    members, their_opts = others[:half] + [frame] + others[half:], [plain] * half + [opt] + [plain] * (len(others) - half)
                    if batch_handle:
                        arr, oarr = be.pack_batch(members, their_opts)
                        handle = be.batch_create(arr, oarr)
                        try:
                            be.batch_exec(handle, 0)
                            be.batch_exec(handle, 0)
                        finally:
                            be.batch_destroy(handle)
                    else:
                        be.exec_batch(members, their_opts)
            surv = frame.survivors(0)
            info = be.info(state)
            if last_width is not None:
                last_width.append(int(L.vvm_state_last_exec_width(state)))
            if device_counts is not None:
                h = np.empty(8, dtype=np.int32)
                be._check(L.vvm_device_download(h.ctypes.data, opt.DeviceCounts, h.nbytes), "download")
                device_counts.extend(int(v) for v in h)
            if more_results is not None:
                for i, (ptrs, dts) in enumerate(zip(more_dev, more_datasets)):
                    hs = []
                    for p, dt in zip(ptrs, dts):
                        h = np.empty(total, dtype=dt)
                        be._check(L.vvm_device_download(h.ctypes.data, p, h.nbytes), "download")
                        hs.append(h)
                    more_results.append((hs, frame.survivors(1 + i)))
            if ids is not None:
                ids.read_counters(frame.datasets[0])
                for a, p in zip((ids.id_table, ids.free_table), dev_ids):
                    if a.nbytes:
                        be._check(L.vvm_device_download(a.ctypes.data, p, a.nbytes), "download")
            for i, st in enumerate(more_ids):
                if st is not None:
                    st.read_counters(frame.datasets[1 + i])
                    for a, p in zip((st.id_table, st.free_table), more_id_dev[i]):
                        if a.nbytes:
                            be._check(L.vvm_device_download(a.ctypes.data, p, a.nbytes), "download")
        finally:
            be.free_state(state)
        for p, dt in zip(dev_out, out_dtypes):
            h = np.empty(total, dtype=dt)
            be._check(L.vvm_device_download(h.ctypes.data, p, h.nbytes), "download")
            host_out.append(h)
    finally:
        for p in dev_in + dev_out + dev_ids:
            L.vvm_device_free(p)
        be.set_page_size_kb(64)
    return host_out, surv, info


def ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance in units in the last place between float32 arrays (NaN == NaN, +0 == -0 counts as 0)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)
