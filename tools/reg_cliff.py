#!/usr/bin/env python3
"""How the kernel shapes compare on one arithmetic script as the number of live registers grows: all registers in VGPRs (eight or
four instances per lane), some in LDS (four per lane), one instance per lane.  4 streams in / 4 out, 64 dependent instructions
over `live` accumulators.  --trig: two of the instructions are sines (the script then runs on the kernels that have the transcendental
functions).  --four-wide: no eight-wide kernels (scripts of at most sixteen registers run four instances per lane, as small Execs do).
--instances N."""
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def script(n_ops, live, trig=False):
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    ins = [s.input(0, FLOAT, i) for i in range(4)]
    s.acquireindex(s.const(0), 0)
    acc = []
    for i in range(live):                 # `live` values that all stay alive to the end
        t = s.temp()
        s.op("mul", ins[i % 4], s.const(1 + i % 2), t)
        acc.append(t)
    for k in range(n_ops):
        t = s.temp()
        a, b = acc[k % live], acc[(k + 1) % live]
        if trig and k in (20, 41):
            s.op("sin", a, t)
        elif k % 3 == 0:
            s.op("mad", a, s.const(1), b, t)
        elif k % 3 == 1:
            s.op("mul", a, b, t)
        else:
            s.op("add", a, s.const(2), t)
        acc[k % live] = t
    outs = []
    for i in range(4):                    # fold everything into four outputs
        t = acc[i]
        for j in range(i + 4, live, 4):
            u = s.temp()
            s.op("add", t, acc[j], u)
            t = u
        outs.append(t)
    s.output_data(FLOAT, 0, [(outs[i], s.output(0, FLOAT, i)) for i in range(4)])
    return s.build(alias_outputs=True)


def main():
    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    dev = torch.device("cuda", 0)
    be = abi.load_hip_backend()
    n = int(sys.argv[sys.argv.index("--instances") + 1]) if "--instances" in sys.argv else 16 * 1024 * 1024
    opt = abi.ExecOptions(Flags=abi.VVM_EXEC_NO_EIGHT_WIDE) if "--four-wide" in sys.argv else None
    ins = scripts.fountain_streams_torch(0, n, dev, pad=64)[:4]
    outs = [torch.zeros(n + 64, dtype=torch.float32, device=dev) for _ in range(4)]
    consts = struct.pack("<I2f", 0xFFFFFFFF, 1.0009765625, 0.125) + bytes(4)
    for live in (4, 8, 12, 16, 20, 28, 40, 44, 48, 55, 64):
        built = script(64, live, "--trig" in sys.argv)
        st = VectorVMState(built.blob, be)
        info = st.info()
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [consts])
        for _ in range(2):
            st.exec(frame, opt)
        st.enable_timing(True)
        for _ in range(5):
            st.exec_async(frame, opt)
        st.finish(frame)
        ms, launches = st.collect_kernel_ms()
        n_ins = 64 + live + sum(len(range(i + 4, live, 4)) for i in range(4))
        print(json.dumps({"live_registers": live, "virtual_registers": int(info.NumVirtualRegisters), "lds_registers": int(info.LdsSpillRegisters),
                          "width": st.last_exec_width(), "instructions": n_ins, "kernel_ms": ms / launches,
                          "us_per_instruction": ms / launches * 1e3 / n_ins}), flush=True)
        st.close()


if __name__ == "__main__":
    main()
