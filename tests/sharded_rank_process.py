"""One rank of a sharded Exec, as a process of its own (started by tests/test_gpu_sharded_processes.py):

    python sharded_rank_process.py RANK WORLD PORT N RESULT_FILE

The ranks share device 0.  Each uploads its contiguous range of the instances, creates a communicator of the C ABI whose
all-gather callback is a gloo exchange between the rank processes, and runs its shard through vvm_exec_sharded (once
synchronously, then three pipelined steps through vvm_exec_sharded_async / _wait).  A second script, which acquires dataset 1
only, then goes through the same communicator: the count word of dataset 0 in the communicator's buffer still holds what the
first script's Exec left there and must be reported as zero.  What the rank saw is written to RESULT_FILE as JSON plus raw
survivor bytes; the parent compares with one Exec over all instances on the oracle.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

f32 = np.float32


def secondary_only_script():
    """Acquires and writes dataset 1 only (dataset 0 exists in the call frame but the script never touches it)."""
    from renderer_amd.assembler import FLOAT, INT32, Script
    s = Script()
    age, life = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)
    keep = s.temp(); s.op("cmpge", age, life, keep)
    s.acquireindex(keep, 1)
    idx = s.temp(); s.op("exec_index", idx)
    s.output_data(FLOAT, 1, [(age, s.output(1, FLOAT, 0))])
    s.output_data(INT32, 1, [(idx, s.output(1, INT32, 0))])
    return s.build()


def main():
    import faulthandler
    faulthandler.enable()
    trace = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "libnative_trace.so")
    if os.path.exists(trace):      # (a crash inside a native library then prints its native call stack as well)
        import ctypes
        ctypes.CDLL(trace).native_trace_install()

    def stage(text):      # (what the parent prints when a rank dies)
        print("rank %s: %s" % (sys.argv[1], text), flush=True)
    rank, world, port, n = (int(a) for a in sys.argv[1:5])
    result_file = sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from renderer_amd import abi, scripts
        from renderer_amd.runtime import VectorVMState, shard_range
        from test_sharded_gloo import _script_with_index_and_random
        from vm_harness import PAD

        be = abi.load_hip_backend()
        be._check(be.lib.vvm_set_device(0), "vvm_set_device")
        lo, hi = shard_range(n, rank, world)
        m = hi - lo
        age = scripts.u01_np(lo, max(m, 1), 3).astype(f32)
        life = np.full(max(m, 1), 0.5, f32)
        gathers = []

        def gloo_all_gather(send, recv, count, stream):
            assert not stream
            mine = torch.tensor([send[i] for i in range(count)], dtype=torch.int32)
            parts = [torch.zeros(count, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(parts, mine)
            for r in range(world):
                for i in range(count):
                    recv[r * count + i] = int(parts[r][i])
            gathers.append([int(mine[i]) for i in range(count)])
            return 0

        comm, keepalive = be.comm_create_callback(rank, world, gloo_all_gather, abi.VVM_COMM_BUFFERS_HOST)
        assert int(be.lib.vvm_comm_world(comm)) == world
        allocations = []

        def device_buffer(host=None, nbytes=None):
            nbytes = host.nbytes if host is not None else nbytes
            p = be.lib.vvm_device_alloc(nbytes)
            assert p, be.error_text()
            allocations.append(p)
            if host is not None:
                be._check(be.lib.vvm_device_upload(p, host.ctypes.data, nbytes), "upload")
            else:
                be._check(be.lib.vvm_device_memset(p, 0xCD, nbytes), "memset")
            return p

        def download(p, count, dtype):
            h = np.empty(max(m, 1) + PAD, dtype)
            be._check(be.lib.vvm_device_download(h.ctypes.data, p, h.nbytes), "download")
            return h[:count]

        def padded(a):
            buf = np.zeros(max(m, 1) + PAD, f32)
            buf[:m] = a[:m]
            return buf

        ins = [device_buffer(padded(age)), device_buffer(padded(life))]
        out_bytes = (max(m, 1) + PAD) * 4
        opt = abi.ExecOptions(77, lo, lo, None, 0)
        report = {"rank": rank, "range": [lo, hi]}

        # --- the script with exec_index and a random draw, dataset 0 ---
        outs = [device_buffer(nbytes=out_bytes) for _ in range(3)]
        state = VectorVMState(_script_with_index_and_random().blob, be)
        frame = state.frame([abi.DataSet(ins, outs, 0, [0, 2, 2], [0, 2, 3])], m, [bytes(16)])
        stage("first script, synchronous")
        res = be.exec_sharded(frame, opt, comm)
        stage("first script, pipelined")
        report["sync"] = [res.NumDataSets, res.LocalSurvivors[0], res.GlobalOffset[0], res.GlobalSurvivors[0], frame.survivors(0)]
        for _ in range(3):
            be.exec_sharded_async(frame, opt, comm)
        report["pipelined"] = []
        for _ in range(3):
            res = be.exec_sharded_wait(comm)
            report["pipelined"].append([res.LocalSurvivors[0], res.GlobalOffset[0], res.GlobalSurvivors[0]])
        be.exec_finish(frame)
        local = frame.survivors(0)
        payload = b"".join(download(p, local, dt).tobytes() for p, dt in zip(outs, (f32, f32, np.int32)))
        report["payload_bytes_0"] = len(payload)
        state.close()

        # --- a script that acquires dataset 1 only, through the same communicator ---
        outs1 = [device_buffer(nbytes=out_bytes) for _ in range(2)]
        state1 = VectorVMState(secondary_only_script().blob, be)
        frame1 = state1.frame([abi.DataSet(ins, [], 0, [0, 2, 2], [0, 0, 0]), abi.DataSet([], outs1, 0, [0, 0, 0], [0, 1, 2])], m, [bytes(16)])
        stage("second script")
        res = be.exec_sharded(frame1, opt, comm)
        stage("second script done")
        report["secondary"] = [res.NumDataSets, res.LocalSurvivors[0], res.GlobalOffset[0], res.GlobalSurvivors[0],
                               res.LocalSurvivors[1], res.GlobalOffset[1], res.GlobalSurvivors[1], frame1.survivors(0), frame1.survivors(1)]
        local1 = frame1.survivors(1)
        payload1 = b"".join(download(p, local1, dt).tobytes() for p, dt in zip(outs1, (f32, np.int32)))
        report["payload_bytes_1"] = len(payload1)
        state1.close()
        report["gathers"] = gathers

        stage("releasing")
        be.comm_destroy(comm)
        for p in allocations:
            be.lib.vvm_device_free(p)
        with open(result_file + ".bin", "wb") as f:
            f.write(payload + payload1)
        with open(result_file, "w") as f:
            json.dump(report, f)
        stage("results written")
    finally:
        dist.destroy_process_group()
    stage("process group closed")


if __name__ == "__main__":
    main()
