"""ctypes mirror of ``include/vvm_abi.h`` and the host-side call frame helpers.

The structs reproduce FVectorVMExecContext (Runtime/VectorVM/Public/VectorVMRuntime.h:10-28) and
FDataSetMeta (Runtime/VectorVM/Public/VectorVM.h:246-273) field for field.  ``Backend`` wraps the three
reference entry points (VectorVMRuntime.h:30-33) of a loaded library; the shipped backend is
libvvm_hip.so (``load_hip_backend``), which must exist -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

VVM_OK = 0
(VVM_E_INVALID_ARG, VVM_E_EXTFN_MISMATCH, VVM_E_NO_BYTECODE, VVM_E_UNSUPPORTED_OP, VVM_E_BAD_BLOB, VVM_E_HIP, VVM_E_RCCL,
 VVM_E_TOO_LARGE) = range(1, 9)      # include/vvm_abi.h status codes
VVM_EXTFN_HOST_CALLBACK, VVM_EXTFN_PERLIN3, VVM_EXTFN_CURL_NOISE3 = 0, 1, 2
VVM_EXEC_NO_VGPR_WIDE = 1
VVM_EXEC_NO_EIGHT_WIDE = 2
VVM_EXEC_FORCE_EIGHT_WIDE = 4
VVM_EXEC_PARAMS_IN_MEMORY = 8
VVM_EXEC_NO_LIGHT_KERNEL = 16
VVM_EXEC_SEPARATE_PASSES = 32
VVM_EXEC_NO_THROUGH_KERNEL = 64
VVM_BATCH_CHANGED_INSTANCES, VVM_BATCH_CHANGED_CONSTANTS, VVM_BATCH_CHANGED_STREAMS = 1, 2, 4

STATUS_NAMES = {0: "VVM_OK", 1: "VVM_E_INVALID_ARG", 2: "VVM_E_EXTFN_MISMATCH", 3: "VVM_E_NO_BYTECODE",
                4: "VVM_E_UNSUPPORTED_OP", 5: "VVM_E_BAD_BLOB", 6: "VVM_E_HIP", 7: "VVM_E_RCCL", 8: "VVM_E_TOO_LARGE"}


class IdTable(C.Structure):
    _fields_ = [("data", C.c_void_p), ("num", C.c_int32), ("max", C.c_int32)]


class RegisterView(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_void_p)), ("num", C.c_int32)]


class DataSetMeta(C.Structure):
    _fields_ = [
        ("InputRegisters", RegisterView),
        ("OutputRegisters", RegisterView),
        ("InputRegisterTypeOffsets", C.c_uint32 * 3),
        ("OutputRegisterTypeOffsets", C.c_uint32 * 3),
        ("DataSetAccessIndex", C.c_int32),
        ("InstanceOffset", C.c_int32),
        ("IDTable", C.POINTER(IdTable)),
        ("FreeIDTable", C.POINTER(IdTable)),
        ("SpawnedIDsTable", C.POINTER(IdTable)),
        ("NumFreeIDs", C.POINTER(C.c_int32)),
        ("MaxUsedID", C.POINTER(C.c_int32)),
        ("NumSpawnedIDs", C.POINTER(C.c_int32)),
        ("IDAcquireTag", C.c_int32),
    ]


class ExtFnContext(C.Structure):
    _fields_ = [
        ("RegisterData", C.POINTER(C.POINTER(C.c_uint32))),
        ("RegInc", C.POINTER(C.c_uint8)),
        ("RegReadCount", C.c_int32),
        ("NumRegisters", C.c_int32),
        ("StartInstance", C.c_int32),
        ("NumInstances", C.c_int32),
        ("NumLoops", C.c_int32),
        ("PerInstanceFnInstanceIdx", C.c_int32),
        ("UserPtrTable", C.POINTER(C.c_void_p)),
        ("NumUserPtrs", C.c_int32),
        ("RandStream", C.c_void_p),
        ("RandCounters", C.POINTER(C.POINTER(C.c_int32))),
        ("DataSets", C.POINTER(DataSetMeta)),
        ("NumDataSets", C.c_int32),
    ]


HOST_EXTFN = C.CFUNCTYPE(None, C.POINTER(ExtFnContext))


class ExtFunction(C.Structure):
    _fields_ = [("builtin_id", C.c_int32), ("host_fn", HOST_EXTFN), ("user", C.c_void_p)]


class _Internal(C.Structure):
    _fields_ = [("NumBytesRequiredPerBatch", C.c_uint32), ("PerBatchRegisterDataBytesRequired", C.c_uint32),
                ("MaxChunksPerBatch", C.c_uint32), ("MaxInstancesPerChunk", C.c_uint32)]


class _DataSetView(C.Structure):
    _fields_ = [("data", C.POINTER(DataSetMeta)), ("num", C.c_int32)]


class _ExtFnView(C.Structure):
    _fields_ = [("data", C.POINTER(C.POINTER(ExtFunction))), ("num", C.c_int32)]


class _UserPtrView(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_void_p)), ("num", C.c_int32)]


class ExecContext(C.Structure):
    _fields_ = [
        ("Internal", _Internal),
        ("VVMState", C.c_void_p),
        ("DataSets", _DataSetView),
        ("ExtFunctionTable", _ExtFnView),
        ("UserPtrTable", _UserPtrView),
        ("NumInstances", C.c_int32),
        ("ConstantTableData", C.POINTER(C.c_void_p)),
        ("ConstantTableNumBytes", C.POINTER(C.c_int)),
        ("ConstantTableCount", C.c_int32),
    ]


class ExecOptions(C.Structure):
    _fields_ = [("RandSeed", C.c_uint64), ("ExecIndexBase", C.c_int32), ("RandInstanceBase", C.c_int32),
                ("Stream", C.c_void_p), ("Flags", C.c_uint32), ("DeviceCounts", C.c_void_p),
                ("RandXorwowState", C.POINTER(C.c_uint32))]


MAX_SHARD_DATASETS = 8
VVM_COMM_BUFFERS_HOST, VVM_COMM_BUFFERS_DEVICE = 0, 1
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_void_p)


class ShardResult(C.Structure):
    _fields_ = [("NumDataSets", C.c_int32), ("LocalSurvivors", C.c_int32 * MAX_SHARD_DATASETS),
                ("GlobalOffset", C.c_int64 * MAX_SHARD_DATASETS), ("GlobalSurvivors", C.c_int64 * MAX_SHARD_DATASETS)]


class ProgramInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "NumTempRegisters", "NumConstBuffers", "NumInputBuffers", "NumOutputBuffers", "NumInstructions",
        "NumVirtualRegisters", "InstancesPerLane", "LdsSpillRegisters", "BytesInPerInstance", "BytesOutPerInstance", "Flags")]


class VVMError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__("%s failed: %s%s" % (where, STATUS_NAMES.get(code, str(code)), (": " + detail) if detail else ""))


# ---------------------------------------------------------------------------------------------------
# call-frame builders
# ---------------------------------------------------------------------------------------------------

class DataSet:
    """Owns the arrays a vvm_dataset_meta points to.  ``inputs`` / ``outputs`` are stream addresses (ints) ordered
    float streams first, then int32, then half; ``*_type_offsets`` give the first index of each type."""

    def __init__(self, inputs: Sequence[int], outputs: Sequence[int], instance_offset: int = 0,
                 input_type_offsets: Optional[Sequence[int]] = None, output_type_offsets: Optional[Sequence[int]] = None):
        self._in = (C.c_void_p * max(1, len(inputs)))(*inputs)
        self._out = (C.c_void_p * max(1, len(outputs)))(*outputs)
        self.num_in, self.num_out = len(inputs), len(outputs)
        self.instance_offset = instance_offset
        self.input_type_offsets = list(input_type_offsets or (0, len(inputs), len(inputs)))
        self.output_type_offsets = list(output_type_offsets or (0, len(outputs), len(outputs)))
        self.id_tables = None     # (IdTable, IdTable, IdTable, c_int32 numfree, c_int32 maxused, c_int32 numspawned, tag)

    def set_id_tables(self, id_table: IdTable, free_table: IdTable, spawned: IdTable, num_free: int, max_used: int,
                      num_spawned: int, tag: int) -> None:
        self.id_tables = (id_table, free_table, spawned, C.c_int32(num_free), C.c_int32(max_used), C.c_int32(num_spawned), tag)

    def fill(self, m: DataSetMeta) -> None:
        m.InputRegisters.data = C.cast(self._in, C.POINTER(C.c_void_p))
        m.InputRegisters.num = self.num_in
        m.OutputRegisters.data = C.cast(self._out, C.POINTER(C.c_void_p))
        m.OutputRegisters.num = self.num_out
        for i in range(3):
            m.InputRegisterTypeOffsets[i] = self.input_type_offsets[i]
            m.OutputRegisterTypeOffsets[i] = self.output_type_offsets[i]
        m.DataSetAccessIndex = -1
        m.InstanceOffset = self.instance_offset
        m.IDAcquireTag = -1
        if self.id_tables is not None:
            t = self.id_tables
            m.IDTable, m.FreeIDTable, m.SpawnedIDsTable = C.pointer(t[0]), C.pointer(t[1]), C.pointer(t[2])
            m.NumFreeIDs, m.MaxUsedID, m.NumSpawnedIDs = C.pointer(t[3]), C.pointer(t[4]), C.pointer(t[5])
            m.IDAcquireTag = t[6]


class CallFrame:
    """A vvm_exec_context plus everything it points to, kept alive together."""

    def __init__(self, state_ptr: int, datasets: Sequence[DataSet], num_instances: int,
                 const_tables: Sequence[bytes] = (), ext_fns: Sequence = (), user_ptrs: Sequence[int] = ()):
        self.datasets = list(datasets)
        self._metas = (DataSetMeta * max(1, len(datasets)))()
        for ds, m in zip(self.datasets, self._metas):
            ds.fill(m)
        self._const_bufs = [C.create_string_buffer(bytes(t), max(1, len(t))) for t in const_tables]
        self._const_ptrs = (C.c_void_p * max(1, len(const_tables)))(*[C.addressof(b) for b in self._const_bufs])
        self._const_sizes = (C.c_int * max(1, len(const_tables)))(*[len(t) for t in const_tables])
        self._ext = []
        for f in ext_fns:
            if isinstance(f, ExtFunction):
                self._ext.append(f)
            elif isinstance(f, int):
                self._ext.append(ExtFunction(f, HOST_EXTFN(), None))
            else:   # python callable taking POINTER(ExtFnContext)
                self._ext.append(ExtFunction(VVM_EXTFN_HOST_CALLBACK, HOST_EXTFN(f), None))
        self._ext_ptrs = (C.POINTER(ExtFunction) * max(1, len(self._ext)))(*[C.pointer(e) for e in self._ext])
        self._user = (C.c_void_p * max(1, len(user_ptrs)))(*user_ptrs)
        c = ExecContext()
        c.VVMState = state_ptr
        c.DataSets.data = C.cast(self._metas, C.POINTER(DataSetMeta))
        c.DataSets.num = len(datasets)
        c.ExtFunctionTable.data = C.cast(self._ext_ptrs, C.POINTER(C.POINTER(ExtFunction)))
        c.ExtFunctionTable.num = len(self._ext)
        c.UserPtrTable.data = C.cast(self._user, C.POINTER(C.c_void_p))
        c.UserPtrTable.num = len(user_ptrs)
        c.NumInstances = num_instances
        c.ConstantTableData = C.cast(self._const_ptrs, C.POINTER(C.c_void_p))
        c.ConstantTableNumBytes = C.cast(self._const_sizes, C.POINTER(C.c_int))
        c.ConstantTableCount = len(const_tables)
        self.ctx = c

    def survivors(self, dataset: int = 0) -> int:
        """DataSetAccessIndex + 1, i.e. the number of instances written (VectorVMRuntime.cpp:2305-2308)."""
        return self._metas[dataset].DataSetAccessIndex + 1

    def meta(self, dataset: int = 0) -> DataSetMeta:
        return self._metas[dataset]


class Backend:
    """The three reference entry points of one loaded library."""

    def __init__(self, lib: C.CDLL, alloc: str, free: str, exec_: str, last_error: Optional[str] = None):
        self.lib = lib
        self._alloc = getattr(lib, alloc)
        self._alloc.restype = C.c_void_p
        self._alloc.argtypes = [C.c_char_p, C.c_size_t]
        self._free = getattr(lib, free)
        self._free.restype = None
        self._free.argtypes = [C.c_void_p]
        self._exec = getattr(lib, exec_)
        self._exec.restype = C.c_int
        self._exec.argtypes = [C.POINTER(ExecContext)]
        self._last_error = None
        if last_error:
            self._last_error = getattr(lib, last_error)
            self._last_error.restype = C.c_char_p

    def error_text(self) -> str:
        if self._last_error is None:
            return ""
        t = self._last_error()
        return t.decode() if t else ""

    def alloc_state(self, blob: bytes) -> int:
        p = self._alloc(blob, len(blob))
        if not p:
            raise VVMError(5, "alloc_state", self.error_text())
        return p

    def free_state(self, state: int) -> None:
        self._free(state)

    def exec(self, frame: CallFrame) -> int:
        return self._exec(C.byref(frame.ctx))


_HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvvm_hip.so")
_hip_backend = None


class HipBackend(Backend):
    def __init__(self, lib: C.CDLL):
        super().__init__(lib, "vvm_alloc_state", "vvm_free_state", "vvm_exec", "vvm_last_error")
        L = lib
        L.vvm_exec_ex.restype = C.c_int
        L.vvm_exec_ex.argtypes = [C.POINTER(ExecContext), C.POINTER(ExecOptions)]
        L.vvm_debug_set_launch_epoch.restype = C.c_int
        L.vvm_debug_set_launch_epoch.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.vvm_exec_noalloc.restype = C.c_int
        L.vvm_exec_noalloc.argtypes = [C.POINTER(ExecContext), C.POINTER(ExecOptions)]
        L.vvm_exec_async.restype = C.c_int
        L.vvm_exec_async.argtypes = [C.POINTER(ExecContext), C.POINTER(ExecOptions)]
        L.vvm_exec_finish.restype = C.c_int
        L.vvm_exec_finish.argtypes = [C.POINTER(ExecContext)]
        L.vvm_exec_batch.restype = C.c_int
        L.vvm_exec_batch.argtypes = [C.POINTER(ExecContext), C.c_int32, C.POINTER(ExecOptions)]
        L.vvm_batch_create.restype = C.c_void_p
        L.vvm_batch_create.argtypes = [C.POINTER(ExecContext), C.c_int32, C.POINTER(ExecOptions)]
        L.vvm_batch_exec.restype = C.c_int
        L.vvm_batch_exec.argtypes = [C.c_void_p, C.c_uint32]
        L.vvm_batch_destroy.restype = None
        L.vvm_batch_destroy.argtypes = [C.c_void_p]
        L.vvm_comm_unique_id.restype = C.c_int
        L.vvm_comm_unique_id.argtypes = [C.c_char_p]
        L.vvm_comm_create_rccl.restype = C.c_void_p
        L.vvm_comm_create_rccl.argtypes = [C.c_int32, C.c_int32, C.c_char_p]
        L.vvm_comm_create_callback.restype = C.c_void_p
        L.vvm_comm_create_callback.argtypes = [C.c_int32, C.c_int32, ALL_GATHER_FN, C.c_void_p, C.c_int32]
        L.vvm_comm_destroy.restype = None
        L.vvm_comm_destroy.argtypes = [C.c_void_p]
        L.vvm_comm_world.restype = C.c_int32
        L.vvm_comm_world.argtypes = [C.c_void_p]
        L.vvm_exec_sharded.restype = C.c_int
        L.vvm_exec_sharded.argtypes = [C.POINTER(ExecContext), C.POINTER(ExecOptions), C.c_void_p, C.POINTER(ShardResult)]
        L.vvm_exec_sharded_async.restype = C.c_int
        L.vvm_exec_sharded_async.argtypes = [C.POINTER(ExecContext), C.POINTER(ExecOptions), C.c_void_p]
        L.vvm_exec_sharded_wait.restype = C.c_int
        L.vvm_exec_sharded_wait.argtypes = [C.c_void_p, C.POINTER(ShardResult)]
        L.vvm_shard_offsets.restype = C.c_int
        L.vvm_shard_offsets.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(ShardResult)]
        L.vvm_state_device_counts.restype = C.c_void_p
        L.vvm_state_device_counts.argtypes = [C.c_void_p]
        L.vvm_state_max_output_dataset.restype = C.c_int32
        L.vvm_state_max_output_dataset.argtypes = [C.c_void_p]
        L.vvm_state_info.restype = C.c_int
        L.vvm_state_info.argtypes = [C.c_void_p, C.POINTER(ProgramInfo)]
        L.vvm_inspect_blob.restype = C.c_int
        L.vvm_inspect_blob.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(ProgramInfo)]
        L.vvm_inspect_code.restype = C.c_int
        L.vvm_inspect_code.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.c_int32]
        L.vvm_debug_batch_plans_made.restype = C.c_uint64
        L.vvm_debug_batch_plans_made.argtypes = []
        L.vvm_inspect_device_code.restype = C.c_int
        L.vvm_inspect_device_code.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.c_int32]
        L.vvm_state_last_exec_width.restype = C.c_int32
        L.vvm_state_last_exec_width.argtypes = [C.c_void_p]
        L.vvm_xorwow_seed.restype = None
        L.vvm_xorwow_seed.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        L.vvm_set_page_size_kb.restype = None
        L.vvm_set_page_size_kb.argtypes = [C.c_int]
        L.vvm_device_alloc.restype = C.c_void_p
        L.vvm_device_alloc.argtypes = [C.c_size_t]
        L.vvm_device_free.restype = None
        L.vvm_device_free.argtypes = [C.c_void_p]
        for n in ("vvm_device_upload", "vvm_device_download"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vvm_device_memset.restype = C.c_int
        L.vvm_device_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.vvm_device_synchronize.restype = C.c_int
        L.vvm_device_count.restype = C.c_int
        L.vvm_set_device.restype = C.c_int
        L.vvm_set_device.argtypes = [C.c_int]
        L.vvm_state_enable_timing.restype = C.c_int
        L.vvm_state_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.vvm_state_collect_kernel_ms.restype = C.c_int
        L.vvm_state_collect_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.vvm_state_copy_counts.restype = C.c_int
        L.vvm_state_copy_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vvm_optimize_script.restype = C.c_int
        L.vvm_optimize_script.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        L.vvm_free_blob.restype = None
        L.vvm_free_blob.argtypes = [C.c_void_p]
        L.vvm_generate_human_readable_script.restype = C.c_int
        L.vvm_generate_human_readable_script.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint32,
                                                         C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        L.vvm_free_text.restype = None
        L.vvm_free_text.argtypes = [C.c_void_p]

    def _check(self, rc: int, where: str) -> None:
        if rc != VVM_OK:
            raise VVMError(rc, where, self.error_text())

    def exec_ex(self, frame: CallFrame, opt: Optional[ExecOptions] = None) -> None:
        self._check(self.lib.vvm_exec_ex(C.byref(frame.ctx), C.byref(opt) if opt is not None else None), "vvm_exec_ex")

    def exec_noalloc(self, frame: CallFrame, opt: Optional[ExecOptions] = None) -> None:
        """vvm_exec_ex without any allocation: the Exec works in the device's scratch array (limits in include/vvm_abi.h)."""
        self._check(self.lib.vvm_exec_noalloc(C.byref(frame.ctx), C.byref(opt) if opt is not None else None), "vvm_exec_noalloc")

    def exec_async(self, frame: CallFrame, opt: Optional[ExecOptions] = None) -> None:
        self._check(self.lib.vvm_exec_async(C.byref(frame.ctx), C.byref(opt) if opt is not None else None), "vvm_exec_async")

    def exec_finish(self, frame: CallFrame) -> None:
        self._check(self.lib.vvm_exec_finish(C.byref(frame.ctx)), "vvm_exec_finish")

    def exec_batch(self, frames: Sequence[CallFrame], opts: Optional[Sequence[ExecOptions]] = None) -> None:
        arr, oarr = self.pack_batch(frames, opts)
        self.exec_batch_packed(arr, oarr)

    @staticmethod
    def pack_batch(frames: Sequence[CallFrame], opts: Optional[Sequence[ExecOptions]] = None):
        """The contiguous vvm_exec_context / vvm_exec_options arrays vvm_exec_batch takes.  Results (DataSetAccessIndex) land in
        the frames' dataset metas, which the copies point to; a caller that runs the same batch every tick packs once."""
        n = len(frames)
        arr = (ExecContext * n)()
        for i, f in enumerate(frames):
            C.memmove(C.byref(arr[i]), C.byref(f.ctx), C.sizeof(ExecContext))
        oarr = (ExecOptions * n)(*opts) if opts is not None else None
        return arr, oarr

    def exec_batch_packed(self, arr, oarr=None) -> None:
        self._check(self.lib.vvm_exec_batch(arr, len(arr), oarr), "vvm_exec_batch")

    def batch_create(self, arr, oarr=None) -> int:
        """vvm_batch_create over the packed arrays of pack_batch (which must outlive the handle)."""
        h = self.lib.vvm_batch_create(arr, len(arr), oarr)
        if not h:
            raise VVMError(VVM_E_INVALID_ARG, "vvm_batch_create", self.error_text())
        return h

    def batch_exec(self, handle: int, changed: int = 0) -> None:
        self._check(self.lib.vvm_batch_exec(handle, changed), "vvm_batch_exec")

    def batch_destroy(self, handle: int) -> None:
        self.lib.vvm_batch_destroy(handle)

    # ---- sharded Execs (include/vvm_abi.h: "instance ranges sharded over several GPUs") ----
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self.lib.vvm_comm_unique_id(buf), "vvm_comm_unique_id")
        return buf.raw

    def comm_create_rccl(self, rank: int, world: int, unique_id: bytes) -> int:
        h = self.lib.vvm_comm_create_rccl(rank, world, unique_id)
        if not h:
            raise VVMError(VVM_E_RCCL, "vvm_comm_create_rccl", self.error_text())
        return h

    def comm_create_callback(self, rank: int, world: int, fn, where: int = VVM_COMM_BUFFERS_HOST):
        """fn(send, recv, count, stream) -> 0 on success; send / recv are ctypes int32 pointers.  Returns (handle, keepalive)."""
        cb = ALL_GATHER_FN(lambda user, send, recv, count, stream: fn(send, recv, count, stream))
        h = self.lib.vvm_comm_create_callback(rank, world, cb, None, where)
        if not h:
            raise VVMError(VVM_E_INVALID_ARG, "vvm_comm_create_callback", self.error_text())
        return h, cb

    def comm_destroy(self, comm: int) -> None:
        self.lib.vvm_comm_destroy(comm)

    def exec_sharded(self, frame: CallFrame, opt: Optional[ExecOptions], comm: int) -> ShardResult:
        out = ShardResult()
        self._check(self.lib.vvm_exec_sharded(C.byref(frame.ctx), C.byref(opt) if opt is not None else None, comm, C.byref(out)), "vvm_exec_sharded")
        return out

    def exec_sharded_async(self, frame: CallFrame, opt: Optional[ExecOptions], comm: int) -> None:
        self._check(self.lib.vvm_exec_sharded_async(C.byref(frame.ctx), C.byref(opt) if opt is not None else None, comm), "vvm_exec_sharded_async")

    def exec_sharded_wait(self, comm: int) -> ShardResult:
        out = ShardResult()
        self._check(self.lib.vvm_exec_sharded_wait(comm, C.byref(out)), "vvm_exec_sharded_wait")
        return out

    def shard_offsets(self, rank: int, world: int, num_datasets: int, all_counts: Sequence[int]) -> ShardResult:
        """The arithmetic of the count exchange (no device): all_counts[r * num_datasets + d]."""
        arr = (C.c_int32 * max(1, len(all_counts)))(*all_counts)
        out = ShardResult()
        self._check(self.lib.vvm_shard_offsets(rank, world, num_datasets, arr, C.byref(out)), "vvm_shard_offsets")
        return out

    def inspect_blob(self, blob: bytes) -> ProgramInfo:
        """Thaw + link only (no device needed); raises VVMError with the linker's message on a bad script."""
        pi = ProgramInfo()
        self._check(self.lib.vvm_inspect_blob(blob, len(blob), C.byref(pi)), "vvm_inspect_blob")
        return pi

    def xorwow_seed(self, cycles_first: int, cycles_second: int, batch_state_address: int):
        """The 24 words of vvm_exec_options.RandXorwowState as the reference's seeding derives them (VectorVMRuntime.cpp:445-464)."""
        out = (C.c_uint32 * 24)()
        self.lib.vvm_xorwow_seed(cycles_first, cycles_second, batch_state_address, out)
        return list(out)

    def set_page_size_kb(self, kb: int) -> None:
        """vm.PageSizeInKB of the reference (VectorVMRuntime.cpp:44-50); 0 restores the default of 64."""
        self.lib.vvm_set_page_size_kb(kb)

    def inspect_code(self, blob: bytes, narrow: bool = False):
        """Linked device program of a blob as a list of (w0, w1, w2, w3) tuples (no device needed)."""
        n = self.lib.vvm_inspect_code(blob, len(blob), int(narrow), None, 0)
        if n < 0:
            self._check(-n, "vvm_inspect_code")
        buf = (C.c_uint32 * (4 * n))()
        self.lib.vvm_inspect_code(blob, len(blob), int(narrow), buf, n)
        return [tuple(buf[4 * i:4 * i + 4]) for i in range(n)]

    def inspect_device_code(self, blob: bytes, narrow: int = 0):
        """The program as uploaded: a list of 8-word tuples (linker words, routine offset, operand bytes a|c|d|e, operand byte b,
        destination byte), one trailing `done` included (no device needed).  narrow = 2: the light eight-wide program."""
        n = self.lib.vvm_inspect_device_code(blob, len(blob), int(narrow), None, 0)
        if n < 0:
            self._check(-n, "vvm_inspect_device_code")
        buf = (C.c_uint32 * (8 * n))()
        self.lib.vvm_inspect_device_code(blob, len(blob), int(narrow), buf, n)
        return [tuple(buf[8 * i:8 * i + 8]) for i in range(n)]

    def info(self, state: int) -> ProgramInfo:
        pi = ProgramInfo()
        self._check(self.lib.vvm_state_info(state, C.byref(pi)), "vvm_state_info")
        return pi

    def optimize_script(self, bytecode: bytes, ext_fns: Sequence = (), hash_id: int = 0, flags: int = 0) -> bytes:
        """OptimizeVectorVMScript + FreezeOptimizerContext (VectorVMOptimizer.cpp:700, VectorVMBridge.cpp:151): bytecode in the
        compiler's format -> frozen blob.  ``ext_fns`` = (NumInputs, NumOutputs) per external function.  An empty script gives b"".
        Raises VVMError carrying the optimizer's error bits."""
        io = (C.c_int32 * (2 * len(ext_fns)))(*[v for pair in ext_fns for v in pair]) if ext_fns else None
        blob, size, err = C.c_void_p(), C.c_size_t(), C.c_uint32()
        rc = self.lib.vvm_optimize_script(bytecode, len(bytecode), C.cast(io, C.c_void_p) if io else None, len(ext_fns), hash_id, flags,
                                          C.byref(blob), C.byref(size), C.byref(err))
        if rc != VVM_OK:
            raise VVMError(rc, "vvm_optimize_script", "optimizer error flags 0x%x" % err.value)
        try:
            return C.string_at(blob, size.value) if blob.value else b""
        finally:
            self.lib.vvm_free_blob(blob)

    def human_readable_script(self, bytecode: bytes, ext_fns: Sequence = (), flags: int = 0) -> str:
        """GenerateHumanReadableScript (VectorVMOptimizer.cpp:220-253): the optimized instruction list of a compiler-format script."""
        io = (C.c_int32 * (2 * len(ext_fns)))(*[v for pair in ext_fns for v in pair]) if ext_fns else None
        text, size, err = C.c_void_p(), C.c_size_t(), C.c_uint32()
        rc = self.lib.vvm_generate_human_readable_script(bytecode, len(bytecode), C.cast(io, C.c_void_p) if io else None, len(ext_fns), flags,
                                                         C.byref(text), C.byref(size), C.byref(err))
        if rc != VVM_OK:
            raise VVMError(rc, "vvm_generate_human_readable_script", "optimizer error flags 0x%x" % err.value)
        try:
            return C.string_at(text, size.value).decode("ascii")
        finally:
            self.lib.vvm_free_text(text)


def hip_library_path() -> str:
    return _HIP_LIB_PATH


def load_hip_backend() -> HipBackend:
    """Load libvvm_hip.so.  Raises if it has not been built: the executor has no other path."""
    global _hip_backend
    if _hip_backend is None:
        if not os.path.exists(_HIP_LIB_PATH):
            raise RuntimeError("%s is missing: build it with `python __graft_entry__.py build` "
                               "(there is no CPU fallback for the VectorVM executor)" % _HIP_LIB_PATH)
        # PyTorch brings its own copy of the HIP runtime.  A process that has initialised one copy cannot open the GPU through
        # the other ("No HIP GPUs are available"), so when torch is installed its copy is loaded first and this library binds to
        # it (same soname); callers that never use torch lose nothing.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip_backend = HipBackend(C.CDLL(_HIP_LIB_PATH))
    return _hip_backend
