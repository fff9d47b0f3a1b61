#!/usr/bin/env python3
"""usage: gpu_fuzz.py <first seed> <last seed>   (on a GPU box)  random compiler-format scripts (the generator of
tests/test_optimizer_fuzz.py) through the optimizer, then on libvvm_hip.so against the CPU oracle, bit for bit; every script runs
on the four-wide kernel and, where the linker allows it, on the eight-wide ones (light and general)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                                      # noqa: E402
from renderer_amd import abi                                            # noqa: E402
from test_optimizer import ALL_ONES, Blob, optimize                     # noqa: E402
from test_optimizer_fuzz import gen                                     # noqa: E402
from vm_harness import run_hip, run_oracle                              # noqa: E402

f32 = np.float32
bad = 0
widths = {4: 0, 8: 0, 1: 0}
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    o, nf, ni, fo, io = gen(seed)
    n = 2048 * 2 + 1 + seed % 2048
    rng = np.random.default_rng(seed)
    ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, n).astype(np.int32) for _ in range(ni)]
    consts = ALL_ONES + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
    dts = [f32] * fo + [np.int32] * io
    blob = Blob(optimize(o))
    ref = run_oracle(blob, [consts], ins, dts, n)
    for flags in (abi.VVM_EXEC_NO_EIGHT_WIDE, abi.VVM_EXEC_FORCE_EIGHT_WIDE, abi.VVM_EXEC_FORCE_EIGHT_WIDE | abi.VVM_EXEC_NO_LIGHT_KERNEL):
        got = run_hip(blob, [consts], ins, dts, n, flags=flags)
        ok = got[1] == ref[1]
        for k in range(len(dts)):
            if not ok:
                break
            x, y = got[0][k][:got[1]].view(np.uint32), ref[0][k][:ref[1]].view(np.uint32)
            both_nan = np.isnan(x.view(f32)) & np.isnan(y.view(f32)) if k < fo else np.zeros(got[1], bool)
            ok = np.array_equal(x[~both_nan], y[~both_nan])
        if not ok:
            bad += 1
            print("seed", seed, "flags", flags, "MISMATCH", flush=True)
    if seed % 100 == 0:
        print("seed", seed, "failures so far", bad, flush=True)
print("done: failures", bad)
sys.exit(1 if bad else 0)
