"""Host-side mirror of the reference runtime interface over the C ABI.

``VectorVMState`` stands for FVectorVMState (AllocVectorVMState / FreeVectorVMState,
Runtime/VectorVM/Public/VectorVMRuntime.h:30-31) and ``VectorVMState.exec`` for ExecVectorVMState (:33).
``ShardedExec`` adds what the single-process reference has no counterpart for: instance ranges split over
ranks, one Exec per rank, and an all-gather of the per-rank survivor counts that yields each rank's global
output offset (concatenating stable compactions of contiguous shards in rank order is the stable compaction
of the whole, so the global order is the reference's).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

from . import abi


class VectorVMState:
    def __init__(self, blob: bytes, backend: Optional[abi.Backend] = None):
        self.backend = backend if backend is not None else abi.load_hip_backend()
        self.ptr = self.backend.alloc_state(blob)

    def close(self) -> None:
        if self.ptr:
            self.backend.free_state(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frame(self, datasets: Sequence[abi.DataSet], num_instances: int, const_tables: Sequence[bytes] = (),
              ext_fns: Sequence = ()) -> abi.CallFrame:
        return abi.CallFrame(self.ptr, datasets, num_instances, const_tables, ext_fns)

    def exec(self, frame: abi.CallFrame, options: Optional[abi.ExecOptions] = None) -> int:
        """Synchronous Exec; returns the survivor count of dataset 0."""
        be = self.backend
        if isinstance(be, abi.HipBackend):
            be.exec_ex(frame, options)
        else:
            rc = be.exec(frame)
            if rc != abi.VVM_OK:
                raise abi.VVMError(rc, "exec")
        return frame.survivors(0)

    def exec_async(self, frame: abi.CallFrame, options: Optional[abi.ExecOptions] = None) -> None:
        self.backend.exec_async(frame, options)

    def finish(self, frame: abi.CallFrame) -> int:
        self.backend.exec_finish(frame)
        return frame.survivors(0)

    def last_exec_width(self) -> int:
        """Instances per lane of the kernel the last Exec ran on (8, 4 or 1)."""
        return int(self.backend.lib.vvm_state_last_exec_width(self.ptr))

    def enable_timing(self, on: bool = True, span: bool = False) -> None:
        """span: one event pair around all Execs up to the next collect_kernel_ms (the Execs carry no events of their own) instead of
        a pair per Exec."""
        self.backend._check(self.backend.lib.vvm_state_enable_timing(self.ptr, (2 if span else 1) if on else 0), "vvm_state_enable_timing")

    def collect_kernel_ms(self):
        """(sum of kernel durations in ms, number of launches) since timing was enabled or last collected."""
        ms, n = C.c_float(0.0), C.c_int32(0)
        self.backend._check(self.backend.lib.vvm_state_collect_kernel_ms(self.ptr, C.byref(ms), C.byref(n)), "vvm_state_collect_kernel_ms")
        return float(ms.value), int(n.value)

    def copy_counts(self, dst_device_ptr: int, stream: int = 0) -> None:
        self.backend._check(self.backend.lib.vvm_state_copy_counts(self.ptr, dst_device_ptr, stream or None), "vvm_state_copy_counts")

    def info(self) -> abi.ProgramInfo:
        return self.backend.info(self.ptr)


def shard_range(num_instances: int, rank: int, world: int):
    """Contiguous instance range of ``rank``: [rank * ceil(N / world), min(N, (rank + 1) * ceil(N / world)))."""
    per = (num_instances + world - 1) // world
    lo = min(num_instances, rank * per)
    hi = min(num_instances, lo + per)
    return lo, hi


def global_offsets(counts: Sequence[int]) -> List[int]:
    """Exclusive prefix of the per-rank survivor counts: where each rank's survivors start in the global order."""
    out, acc = [], 0
    for c in counts:
        out.append(acc)
        acc += int(c)
    return out


class ShardedExec:
    """One rank's part of an Exec over instances sharded across ``world`` ranks.

    ``local_exec(lo, hi)`` runs this rank's Exec over global instances [lo, hi) (with ExecIndexBase = lo and
    RandInstanceBase = lo so that exec_index and the random streams are shard-invariant) and returns the local
    survivor count; ``all_gather(count)`` returns every rank's count in rank order.  The survivors stay where
    the rank wrote them: the global array is the concatenation of the per-rank prefixes [0, S_r).
    """

    def __init__(self, rank: int, world: int, local_exec: Callable[[int, int], int],
                 all_gather: Callable[[int], Sequence[int]]):
        self.rank, self.world = rank, world
        self.local_exec = local_exec
        self.all_gather = all_gather

    def run(self, num_instances: int):
        lo, hi = shard_range(num_instances, self.rank, self.world)
        local = int(self.local_exec(lo, hi))
        counts = [int(c) for c in self.all_gather(local)]
        offs = global_offsets(counts)
        total = sum(counts)
        return {"range": (lo, hi), "local_survivors": local, "global_offset": offs[self.rank], "global_survivors": total,
                "dataset_access_index": total - 1, "counts": counts}
