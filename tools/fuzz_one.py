#!/usr/bin/env python3
"""usage: fuzz_one.py <seed> [flags]   (on a GPU box)  one script of gpu_fuzz_linker.py against the oracle, with the place of the first difference
of every stream; the environment switches of the library (VVM_NO_CHAINS, VVM_NO_CONSTANT_FOLDING) apply as usual."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np                                                      # noqa: E402
import gpu_fuzz_linker as g                                             # noqa: E402
from renderer_amd import abi                                            # noqa: E402
from vm_harness import run_hip, run_oracle                              # noqa: E402

f32 = np.float32


def main():
    seed = int(sys.argv[1])
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else (0, abi.VVM_EXEC_FORCE_EIGHT_WIDE, abi.VVM_EXEC_SEPARATE_PASSES, abi.VVM_EXEC_FORCE_EIGHT_WIDE | abi.VVM_EXEC_NO_LIGHT_KERNEL)[seed % 4]
    built, nf, ni, nh, dts = g.gen(seed)
    gen = g.gen
    n = (seed // 7) % 10 if seed % 7 == 0 else 2048 + 1 + (seed * 37) % 3000
    if g.LARGE and seed % 17 == 0 and seed % 7 != 0:
        n = 200_000 + (seed * 7919) % 400_000
    if g.LARGE and seed % 97 == 0 and seed % 7 != 0:
        n = 1_200_000 + (seed * 7919) % 2_000_000
    offset = 1 + seed % 9 if seed % 3 == 0 else 0
    rng = np.random.default_rng(seed)
    ins = [rng.uniform(-3, 3, max(n, 1)).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, max(n, 1)).astype(np.int32) for _ in range(ni)] + \
          [rng.uniform(-2, 2, max(n, 1)).astype(np.float16) for _ in range(nh)]
    consts = bytes([255] * 4) + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
    more = gen.more_datasets
    assert not gen.with_ids, "this helper does not set up ID tables"
    ref_more, got_more = [], []
    ref = run_oracle(built, [consts], ins, dts, n, seed=seed, instance_offset=offset, more_datasets=more, more_results=ref_more, ext_fns=list(gen.ext))
    got = run_hip(built, [consts], ins, dts, n, seed=seed, instance_offset=offset, flags=flags, more_datasets=more, more_results=got_more, ext_fns=list(gen.ext))
    info = got[2]
    print("seed", seed, "flags", flags, "n", n, "offset", offset, "survivors", got[1], ref[1], "vregs", info.NumVirtualRegisters, "width", info.InstancesPerLane, "state flags", hex(info.Flags))
    for k, dt in enumerate(dts):
        bits = np.uint16 if dt == np.float16 else np.uint32
        x, y = got[0][k][offset:offset + got[1]].view(bits), ref[0][k][offset:offset + ref[1]].view(bits)
        m = min(len(x), len(y))
        bad = np.nonzero(x[:m] != y[:m])[0]
        print(" stream", k, np.dtype(dt).name, "differences", bad.size, "first", bad[:5], [hex(int(v)) for v in x[bad[:3]]], [hex(int(v)) for v in y[bad[:3]]])
    for (gs, gn), (rs, rn) in zip(got_more, ref_more):
        print(" second dataset survivors", gn, rn)
        for k in range(len(gs)):
            x, y = gs[k][offset:offset + gn].view(np.uint32), rs[k][offset:offset + rn].view(np.uint32)
            m = min(len(x), len(y))
            bad = np.nonzero(x[:m] != y[:m])[0]
            print("  stream", k, "differences", bad.size, "first", bad[:5])
    from renderer_amd import opcodes
    for w in abi.load_hip_backend().inspect_code(built.blob, 0):
        op = w[0] & 0xFF
        print("  ", opcodes.NAME.get(op, op), "nin", (w[0] >> 8) & 7, "aux", (w[0] >> 16) & 0xFF, "aux2", w[0] >> 24,
              [hex(v) for v in (w[1] & 0xFFFF, w[1] >> 16, w[2] & 0xFFFF, w[2] >> 16, w[3] & 0xFFFF, w[3] >> 16)])


if __name__ == "__main__":
    main()
