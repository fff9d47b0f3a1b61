#!/usr/bin/env python3
"""Instruction-cost probe of the interpreter kernel: runs a ladder of synthetic scripts, two Execs each, in a fixed order, so
that a `rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU ...` run of this program attributes scalar / vector instruction counts
to the fixed per-tile work (loads, acquireindex, look-back, outputs) and to one interpreted instruction of each operand shape.

  base      8 streams in, acquireindex(all-ones constant), 8 streams out, no arithmetic
  add_vv    base + K x add(vreg, vreg)
  add_vu    base + K x add(vreg, constant)
  mad_vuv   base + K x mad(vreg, constant, vreg)
  madmul    base + K x mad_mul(constant, constant, vreg, constant)
  in1/out1  one stream in / eight out, eight in / one out  (cost of a stream)

tools/salu_probe.sh prints the per-wave differences.  Prints the variant names in launch order (two launches each).
"""
import argparse
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

K = 8


def build(variant):
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    ones, c1, c2, c3 = s.const(0), s.const(1), s.const(2), s.const(3)
    n_in = 1 if variant == "in1" else 8
    n_out = 1 if variant == "out1" else 8
    ins = [s.input(0, FLOAT, i) for i in range(n_in)]
    s.acquireindex(ones, 0)
    cur = list(ins)
    if variant in ("add_vv", "add_vu", "mad_vuv", "madmul"):
        for k in range(K):
            t = s.temp()
            a = cur[k % n_in]
            b = cur[(k + 1) % n_in]
            if variant == "add_vv":
                s.op("add", a, b, t)
            elif variant == "add_vu":
                s.op("add", a, c1, t)
            elif variant == "mad_vuv":
                s.op("mad", a, c1, b, t)
            else:
                s.op("mad_mul", c1, c2, a, c3, t)
            cur[k % n_in] = t
    s.output_data(FLOAT, 0, [(cur[i % n_in], s.output(0, FLOAT, i)) for i in range(n_out)])
    return s.build(alias_outputs=True), n_in, n_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=16 * 1024 * 1024)
    ap.add_argument("--variants", default="base,add_vv,add_vu,mad_vuv,madmul,in1,out1")
    ap.add_argument("--width", type=int, default=4, choices=[4, 8], help="instances per lane of the kernel to probe")
    args = ap.parse_args()
    import torch
    from renderer_amd import abi
    from renderer_amd.runtime import VectorVMState
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    be = abi.load_hip_backend()
    n = args.particles
    src = [torch.rand(n + 64, dtype=torch.float32, device=dev) for _ in range(8)]
    dst = [torch.zeros(n + 64, dtype=torch.float32, device=dev) for _ in range(8)]
    consts = struct.pack("<I3f", 0xFFFFFFFF, 0.5, 0.25, 0.125)
    torch.cuda.synchronize()
    for v in args.variants.split(","):
        built, n_in, n_out = build(v)
        st = VectorVMState(built.blob, be)
        frame = st.frame([abi.DataSet([t.data_ptr() for t in src[:n_in]], [t.data_ptr() for t in dst[:n_out]], 0)], n, [consts])
        st.enable_timing(True)
        opt = abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_FORCE_EIGHT_WIDE if args.width == 8 else abi.VVM_EXEC_NO_EIGHT_WIDE)
        for _ in range(2):
            surv = st.exec(frame, opt)
        ms, launches = st.collect_kernel_ms()
        assert surv == n
        print("variant %s width %d launches 2 kernel_ms %.4f" % (v, args.width, ms / max(launches, 1)), flush=True)
        st.close()


if __name__ == "__main__":
    main()
