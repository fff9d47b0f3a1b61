#!/usr/bin/env python3
"""Diagnostic: does a GPU process that has held many vvm_states keep a NEW process from starting on the same device?
Parent: creates K states (optionally keeps them alive), then starts a child that makes its first HIP calls through the library.
Stops at the first child that fails."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHILD = r"""
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, %r)
from renderer_amd import abi
be = abi.load_hip_backend()
p = be.lib.vvm_device_alloc(4096)
assert p
be._check(be.lib.vvm_device_memset(p, 0, 4096), "memset")
be.lib.vvm_device_free(p)
print("child ok", flush=True)
""" % ROOT


def child(tag):
    r = subprocess.run([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    tail = r.stdout.decode(errors="replace").strip().splitlines()[-3:]
    print("%-44s child exit %d  %s" % (tag, r.returncode, " | ".join(tail)[:200]), flush=True)
    return r.returncode == 0


def main():
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    be = abi.load_hip_backend()
    if not child("parent has made no GPU call"):
        return 1
    w = scripts.fountain()
    for k, alive in ((1, True), (256, True), (256, False), (1024, True), (1024, False), (4096, True), (4096, False)):
        states = [VectorVMState(w.script.blob, be) for _ in range(k)]
        if not alive:
            for s in states:
                s.close()
        ok = child("%d states %s" % (k, "alive" if alive else "closed again"))
        if alive:
            for s in states:
                s.close()
        if not ok:
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
