#!/usr/bin/env python3
"""What misaligned output positions cost: the fountain script at 64 Mi particles, nobody dying, with the output streams starting
0, 1, 2 or 3 elements past a 16-byte boundary (the situation of every tile behind a tile with a survivor count that is not a
multiple of four).  Prints the kernel time per offset."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    be = abi.load_hip_backend()
    n = 64 * 1024 * 1024
    w = scripts.fountain()
    ins = scripts.fountain_streams_torch(0, n, dev, pad=64)
    outs = [torch.zeros(n + 128, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    for shift in (0, 1, 2, 3):
        st = VectorVMState(w.script.blob, be)
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() + 4 * shift for t in outs], 0)], n, [w.constants])
        for _ in range(3):
            st.exec(frame)
        st.enable_timing(True)
        for _ in range(10):
            st.exec_async(frame)
        assert st.finish(frame) == n
        ms, launches = st.collect_kernel_ms()
        print("outputs %d element(s) past a 16-byte boundary: kernel %.4f ms" % (shift, ms / launches), flush=True)
        st.close()


if __name__ == "__main__":
    main()
