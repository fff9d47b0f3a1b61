#!/usr/bin/env python3
"""Which four-wide kernel a script with transcendental / noise instructions and at most sixteen registers is faster on: the one with
the compiled interpreter loop (three workgroups per CU) or the one that runs ordinary instructions through the hand-written
routines (two per CU).  One script per mix of ordinary instructions and heavy ones, 16 Mi instances, 4 streams in / 4 out.
Run twice, with VVM_FOUR_WIDE_ROUTINES=0 and =1 (the library reads it once), and compare the lines."""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def script(n_ordinary, heavy, n_heavy):
    from renderer_amd import abi
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    ins = [s.input(0, FLOAT, i) for i in range(4)]
    s.acquireindex(s.const(0), 0)
    ext = []
    fn = None
    if heavy == "perlin3":
        fn = s.ext_fn(3, 1); ext.append(abi.VVM_EXTFN_PERLIN3)
    if heavy == "curl_noise3":
        fn = s.ext_fn(3, 3); ext.append(abi.VVM_EXTFN_CURL_NOISE3)
    acc = []
    for i in range(8):
        t = s.temp()
        s.op("mul", ins[i % 4], s.const(1 + i % 2), t)
        acc.append(t)
    every = max(1, n_ordinary // max(1, n_heavy))
    done_heavy = 0
    for k in range(n_ordinary):
        t = s.temp()
        a, b = acc[k % 8], acc[(k + 1) % 8]
        if k % 3 == 0:
            s.op("mad", a, s.const(1), b, t)
        elif k % 3 == 1:
            s.op("mul", a, b, t)
        else:
            s.op("add", a, s.const(2), t)
        acc[k % 8] = t
        if k % every == every - 1 and done_heavy < n_heavy:
            done_heavy += 1
            u = s.temp()
            if heavy in ("sin", "exp", "log"):
                s.op(heavy, acc[k % 8], u)
                acc[k % 8] = u
            elif heavy == "pow":
                s.op("pow", acc[k % 8], s.const(1), u)
                acc[k % 8] = u
            elif heavy == "perlin3":
                s.call(fn, [acc[0], acc[1], acc[2]], [u])
                acc[k % 8] = u
            else:
                v, w = s.temps(2)
                s.call(fn, [acc[0], acc[1], acc[2]], [u, v, w])
                acc[3], acc[4], acc[5] = u, v, w
    outs = []
    for i in range(4):
        u = s.temp()
        s.op("add", acc[i], acc[i + 4], u)
        outs.append(u)
    s.output_data(FLOAT, 0, [(outs[i], s.output(0, FLOAT, i)) for i in range(4)])
    return s.build(alias_outputs=True), ext


def main():
    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    dev = torch.device("cuda", 0)
    be = abi.load_hip_backend()
    n = 16 * 1024 * 1024
    ins = scripts.fountain_streams_torch(0, n, dev, pad=64)[:4]
    outs = [torch.zeros(n + 64, dtype=torch.float32, device=dev) for _ in range(4)]
    consts = struct.pack("<I2f", 0xFFFFFFFF, 1.0009765625, 0.125) + bytes(4)
    for heavy in ("sin", "exp", "pow", "perlin3", "curl_noise3"):
        for n_ordinary, n_heavy in ((16, 1), (16, 4), (64, 1), (64, 4), (64, 16), (160, 4), (160, 16)):
            built, ext = script(n_ordinary, heavy, n_heavy)
            st = VectorVMState(built.blob, be)
            info = st.info()
            frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [consts], ext_fns=list(ext))
            for _ in range(2):
                st.exec(frame)
            st.enable_timing(True)
            for _ in range(5):
                st.exec_async(frame)
            st.finish(frame)
            ms, launches = st.collect_kernel_ms()
            print(json.dumps({"heavy": heavy, "ordinary": n_ordinary, "heavy_count": n_heavy, "registers": int(info.NumVirtualRegisters),
                              "routines": os.environ.get("VVM_FOUR_WIDE_ROUTINES", "default"), "kernel_ms": round(ms / launches, 4)}), flush=True)
            st.close()


if __name__ == "__main__":
    main()
