#!/usr/bin/env python3
"""Diagnostic: which HIP resource, held in numbers by one process, keeps a NEW process from starting on the same device?
Stops at the first child that fails."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.diag_child_after_states import child  # noqa: E402


def main():
    from renderer_amd import abi
    be = abi.load_hip_backend()
    p = be.lib.vvm_device_alloc(16)
    hip = ctypes.CDLL("libamdhip64.so")
    vp = ctypes.c_void_p
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [vp]
    hip.hipHostMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostFree.argtypes = [vp]
    hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    hip.hipFree.argtypes = [vp]
    hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    hip.hipEventDestroy.argtypes = [vp]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500

    def many(make, free, tag):
        held = []
        for _ in range(n):
            h = vp()
            rc = make(ctypes.byref(h))
            assert rc == 0, (tag, rc)
            held.append(h)
        ok = child("%d %s held" % (n, tag))
        for h in held:
            free(h)
        if ok:
            ok = child("%d %s released again" % (n, tag))
        return ok

    if not child("parent holds one allocation"):
        return 1
    if not many(lambda h: hip.hipMalloc(h, 4096), hip.hipFree, "device allocations of 4 KiB"):
        return 1
    if not many(lambda h: hip.hipHostMalloc(h, 4096, 0), hip.hipHostFree, "pinned host allocations of 4 KiB"):
        return 1
    if not many(lambda h: hip.hipEventCreateWithFlags(h, 2), hip.hipEventDestroy, "events"):
        return 1
    if not many(lambda h: hip.hipStreamCreateWithFlags(h, 1), hip.hipStreamDestroy, "non-blocking streams"):
        return 1
    be.lib.vvm_device_free(p)
    return 0


if __name__ == "__main__":
    sys.exit(main())
