#!/usr/bin/env python3
"""usage: sync_latency.py   (on a GPU box)  wall time of one synchronous vvm_exec (the drop-in call) of the fountain script by size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = scripts.fountain()
    for n in (1024, 16384, 131072, 1048576):
        ins = scripts.fountain_streams_torch(0, n, dev, pad=64)
        outs = [torch.zeros(n + 64, dtype=torch.float32, device=dev) for _ in range(8)]
        torch.cuda.synchronize()
        st = VectorVMState(w.script.blob)
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants])
        for _ in range(20):
            st.exec(frame)
        reps = 500
        t0 = time.perf_counter()
        for _ in range(reps):
            st.exec(frame)
        el = time.perf_counter() - t0
        print("%8d particles: %.1f us per synchronous vvm_exec" % (n, el / reps * 1e6), flush=True)
        st.close()


if __name__ == "__main__":
    main()
