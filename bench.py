#!/usr/bin/env python3
"""Headline benchmark: particle updates per second of the VectorVM executor on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one Exec of the fountain update script (8 float32 streams in, 8 out, kill by acquireindex) over one
batch of synthetic particles that is already resident in HBM.  At N = 1 the batch is 64 Mi particles
(BASELINE.json configs[1]); at N > 1 every rank runs the same-sized shard of a larger emitter (weak scaling), with
exec_index / random streams based at the shard's first global instance, and every step ends with the all-gather
of the per-rank survivor counts (RCCL) from which each rank's global acquireindex offset follows.  A step of rank r is
one vvm_exec_sharded_async call of the C ABI: the Exec leaves its count in the buffer the all-gather reads, and the
all-gather (ncclAllGather on the library's own communicator, created from an id that rank 0 broadcasts) runs on a stream
of its own, so the next step's Exec (which needs only the local count) does not wait for the other ranks.
(--comm torch keeps the exchange in torch.distributed instead; the library's communicator falls back to it by itself
when RCCL cannot be set up, and `config.parallelism` says which one ran.)

Rank 0 prints ONE JSON line.  `roofline` prices the interpreter kernel against HBM: algorithmic bytes per launch
(4 bytes x (input streams x instances + output streams x survivors)) over the kernel's mean duration, which is
measured here with HIP events recorded on the kernel's own stream: one pair around the K launches of the timed region,
divided by K (--timing per-exec: a pair around every launch).

--workload batch is BASELINE.json configs[4]: 4096 emitters x 16 Ki particles with 64 distinct bytecode bodies, one
vvm_exec_batch call per step; at N > 1 the emitters are dealt to the ranks in turn (strong scaling, no collective).
`cpu_baseline` is the CPU oracle (a restatement of the reference interpreter in its own structure: interpreted
bytecode, 4-wide SSE, 64 KB-page chunks, one thread) timed on a bounded sample on the host's cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E peak
PAD = 64


def cpu_baseline(workload_name, seconds=12.0, n=1 << 20):
    """Times the oracle (tests/oracle_binding.py) on `n` instances, repeating whole Execs for about `seconds`."""
    import numpy as np
    from renderer_amd import abi, scripts
    from oracle_binding import load_oracle
    be = load_oracle()
    be.set_strict(False)
    be.set_page_kb(64)
    be.set_options(0, 0, 0)
    w = scripts.fountain()
    kill = workload_name == "fountain_kill_half"
    ins = [np.concatenate([a, np.zeros(PAD, np.float32)]) for a in scripts.fountain_streams_np(0, n, kill_half=kill)]
    outs = [np.zeros(n + PAD, np.float32) for _ in range(8)]
    ds = abi.DataSet([a.ctypes.data for a in ins], [a.ctypes.data for a in outs], 0)
    state = be.alloc_state(w.script.blob)
    frame = abi.CallFrame(state, [ds], n, [w.constants], ())
    for _ in range(2):
        be.exec(frame)
    t0 = time.perf_counter()
    reps = 0
    while True:
        be.exec(frame)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    be.free_state(state)
    result = {"value": n * reps / el, "unit": "particle updates/s", "cores": 1, "kind": "port",
              "sample": "%d Execs of %d particles (fountain script, BASELINE.json configs[0] size) in %.1f s; CPU restatement in "
                        "reference structure (interpreted, 4-wide SSE, 64 KB-page chunks), not the reference's own binary" % (reps, n, el)}

    # The reference runs one Exec on one core (VectorVMRuntime.cpp:2276-2295); Niagara gets its parallelism from running many
    # emitters at once.  Second figure: P threads, each with its own state, each running Execs over its own n / P instances.
    import threading
    # P = the host cores this GPU's job may use: the box's per-GPU CPU share (OMP_NUM_THREADS is set to it there), capped by the
    # machine's physical cores as lscpu reports them
    physical = None
    try:
        import subprocess
        fields = dict(l.split(":", 1) for l in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines() if ":" in l)
        physical = int(fields["Core(s) per socket"].strip()) * int(fields["Socket(s)"].strip())
    except Exception:      # noqa: BLE001  (no lscpu: the logical count stands in)
        physical = None
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    threads = max(1, min(share, physical or share, 64))
    per = n // threads
    frames, states = [], []
    for t in range(threads):
        sl = slice(t * per, (t + 1) * per)
        tin = [np.concatenate([a[sl], np.zeros(PAD, np.float32)]) for a in ins]
        tout = [np.zeros(per + PAD, np.float32) for _ in range(8)]
        st = be.alloc_state(w.script.blob)
        frames.append((abi.CallFrame(st, [abi.DataSet([a.ctypes.data for a in tin], [a.ctypes.data for a in tout], 0)], per, [w.constants], ()), tin, tout))
        states.append(st)
    budget = max(2.0, seconds / 3.0)
    done = [0] * threads
    start = threading.Barrier(threads + 1)

    def worker(t):
        start.wait()
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end:
            be.exec(frames[t][0])
            done[t] += 1

    pool = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    start.wait()
    t1 = time.perf_counter()
    for th in pool:
        th.join()
    el2 = time.perf_counter() - t1
    for st in states:
        be.free_state(st)
    result["threads"] = {"value": per * sum(done) / el2, "cores": threads,
                         "physical_cores_of_the_machine": physical,
                         "sample": "%d threads (this GPU's share of the host's cores), each its own state and %d particles per Exec, %d Execs in %.1f s"
                                   % (threads, per, sum(done), el2)}
    return result


def recorded_traffic(workload, particles):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE).  Counters cannot be read from inside this process;
    tools/pmc_traffic.sh collects them with rocprofv3 in separate passes of this same command and profiles/ holds the result.
    Returned only when it was taken on this workload and size."""
    for name in ("r03_traffic.json", "r03_traffic_kill_half.json", "r02_traffic.json", "r02_traffic_kill_half.json", "r01_traffic.json", "r01_traffic_kill_half.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("workload") == workload and t.get("particles_per_gpu") == particles:
            return t.get("hbm_bytes_per_launch")
    return None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environment(rank, world, port, base=None):
    """The environment of one rank's process as the launcher below sets it up (what torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch_ranks(world, argv, timeout_s=1500.0, program=None):
    """`python bench.py --gpus N` without an outside launcher: this process (which has made no GPU call and makes none) starts one
    fresh process per rank, hands rank 0's result line through, and fails when any rank fails.  Returns the exit code."""
    import subprocess
    import threading
    port = _free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, program or os.path.abspath(__file__)] + list(argv), env=rank_environment(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().decode().splitlines()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with status %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = "no result within %.0f s" % timeout_s
            break
        time.sleep(0.05)
    if failed:
        # the ranks that are left would wait for the lost one in a collective: they are ended here, each by its own process id
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: %s\n" % failed)
        return 1
    reader.join(10)
    result = [l for l in lines if l.startswith("{")]
    if len(result) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d result lines\n" % len(result))
        return 1
    sys.stdout.write(result[0] + "\n")
    sys.stdout.flush()
    return 0


def run_batch(args, real_stdout):
    """configs[4]: every rank runs its share of the emitters (emitter e belongs to rank e mod N) as one vvm_exec_batch call per step."""
    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                sys.stderr.write("bench.py: rank %d has no GPU of its own (%d visible); --backend gloo rehearses the multi-rank path on one GPU\n"
                                 % (rank, torch.cuda.device_count()))
                sys.exit(2)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    be = abi.load_hip_backend()
    be._check(be.lib.vvm_set_device(dev.index), "vvm_set_device")
    emitters, per = args.batch_emitters, args.batch_particles
    mine = [e for e in range(emitters) if e % world == rank]
    total_local = len(mine) * per
    ins = scripts.fountain_streams_torch(rank * total_local, total_local, dev, kill_half=True, pad=PAD)
    outs = [torch.zeros(total_local + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    workloads = [scripts.fountain(variant=v) for v in range(64)]
    states, frames, opts = [], [], []
    for k, e in enumerate(mine):
        w = workloads[e % 64]
        st = VectorVMState(w.script.blob, be)
        states.append(st)
        off = k * per * 4
        frames.append(st.frame([abi.DataSet([t.data_ptr() + off for t in ins], [t.data_ptr() + off for t in outs], 0)], per, [w.constants]))
        opts.append(abi.ExecOptions(0, 0, 0, None, 0))
    arr, oarr = be.pack_batch(frames, opts)

    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(max(args.warmup, 3) + 20):      # (host-bound: the first calls also measure the host's clocks ramping up)
        be.exec_batch_packed(arr, oarr)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        be.exec_batch_packed(arr, oarr)      # (synchronous: every emitter's count is back in its context when it returns)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    survivors = sum(f.survivors(0) for f in frames)
    if dist is not None:
        t = torch.tensor([elapsed, float(survivors)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, survivors = float(tmax[0].item()), int(t[1].item())
    if rank == 0:
        total = emitters * per
        alg_bytes = 32.0 * total + 32.0 * survivors      # whole job, per step
        ms = elapsed / args.steps * 1e3
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        line = {"metric": "particle updates/sec", "value": float(total) * args.steps / elapsed, "unit": "particle updates/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "batch: %d emitters x %d particles, 64 distinct bytecode bodies, half of the particles dying, one vvm_exec_batch call per "
                                       "step and rank (BASELINE.json configs[4])" % (emitters, per),
                           "emitters_per_gpu": len(mine), "survivors": survivors,
                           "parallelism": "emitters dealt to %d rank(s) in turn, no collective in the step" % world if world > 1 else "single GPU"},
                "roofline": {"bound": "hbm", "achieved": achieved / world, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / world / HBM_PEAK_GBPS, "traffic": None,
                             "kernel": "vvm_exec_kernel (several launches per call)", "note": "per GPU, from the wall clock of the synchronous call "
                             "(upload of the parameter blocks, launches and read-back of the counts included): the call is host-bound"}}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline("fountain_kill_half", args.cpu_seconds)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    for st in states:
        st.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fountain", choices=["fountain", "fountain_kill_half", "batch"])
    ap.add_argument("--batch-emitters", type=int, default=4096, help="--workload batch: emitters of the whole job")
    ap.add_argument("--batch-particles", type=int, default=16384, help="--workload batch: particles per emitter")
    ap.add_argument("--particles", type=int, default=64 * 1024 * 1024, help="instances per GPU")
    ap.add_argument("--width", default="auto", choices=["auto", "4", "8"], help="instances per lane of the kernel (auto: the library's choice)")
    ap.add_argument("--no-light", action="store_true", help="keep the eight-wide Exec on the general kernel (comparison)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU: all ranks then share device 0)")
    ap.add_argument("--collective-at-one", action="store_true",
                    help="rehearsal on a one-GPU box: run the multi-rank step (count copy, all-gather on its own stream) with a single rank")
    ap.add_argument("--comm", default="c-abi", choices=["c-abi", "torch"],
                    help="who runs the count exchange at N > 1: the library's own RCCL communicator (vvm_exec_sharded_async) or torch.distributed")
    ap.add_argument("--exchange-depth", type=int, default=2, choices=[1, 2, 3, 4],
                    help="count exchanges that may be pending (c-abi communicator): the host collects step k - depth before it enqueues step k")
    ap.add_argument("--timing", default="span", choices=["span", "per-exec"],
                    help="how the kernel's mean duration is taken with HIP events on the Execs' stream: one event pair around the K launches of the "
                         "timed region (span / K: includes the gaps between kernels, adds nothing to the steps), or a pair around every launch "
                         "(two events per step, about 8 us of stream time each step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # Rank 0 prints ONE line on stdout.  Libraries write there too (RCCL prints a version banner when its first communicator
    # comes up), so everything else that lands on descriptor 1 goes to stderr and the result line to the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.workload == "batch":
        if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
            sys.stderr.write("bench.py: --gpus %d, but the launcher started %s rank(s) (WORLD_SIZE)\n" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
            sys.exit(2)
        return run_batch(args, real_stdout)

    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState, global_offsets

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d, but the launcher started %d rank(s) (WORLD_SIZE)\n" % (args.gpus, world))
        sys.exit(2)
    distributed = world > 1 or args.collective_at_one
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                sys.stderr.write("bench.py: rank %d has no GPU of its own (%d visible); --backend gloo rehearses the multi-rank path on one GPU\n"
                                 % (rank, torch.cuda.device_count()))
                sys.exit(2)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)
    be = abi.load_hip_backend()
    be._check(be.lib.vvm_set_device(dev.index), "vvm_set_device")

    n = args.particles
    kill = args.workload == "fountain_kill_half"
    first = rank * n                       # this rank's first global instance
    w = scripts.fountain()
    ins = scripts.fountain_streams_torch(first, n, dev, kill_half=kill, pad=PAD)
    outs = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()

    state = VectorVMState(w.script.blob)
    ds = abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)
    frame = state.frame([ds], n, [w.constants])
    # everything of a step is enqueued on ONE explicit stream: the Exec, the device-to-device copy of its survivor count, and
    # (through torch's current stream) the all-gather that reads that count.  (Stream 0 would mean "the state's own stream" to
    # the library, which the collective does not wait for.)
    torch_stream = torch.cuda.Stream(device=dev)
    stream = torch_stream.cuda_stream
    flags = {"auto": 0, "4": abi.VVM_EXEC_NO_EIGHT_WIDE, "8": abi.VVM_EXEC_FORCE_EIGHT_WIDE}[args.width] | (abi.VVM_EXEC_NO_LIGHT_KERNEL if args.no_light else 0)
    base = first if first < 2 ** 31 else 0
    opt = abi.ExecOptions(0, base, base, stream, flags)
    # The exchange of a step runs beside the next step's Exec: the Exec writes its count into the step's buffer, the all-gather waits
    # for the Exec on a stream of its own.  Two sets of buffers alternate; a set is written again only after the all-gather that read
    # it two steps earlier has finished (an event the Exec's stream waits for, long past by then).
    coll_stream = torch.cuda.Stream(device=dev)
    counts_dev = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [torch.zeros(world, dtype=torch.int32, device=dev) for _ in range(2)]
    counted = [torch.cuda.Event() for _ in range(2)]
    exchanged = [torch.cuda.Event() for _ in range(2)]
    steps_done = [0]

    # (the Exec leaves its survivor count in the buffer the all-gather reads: vvm_exec_options.DeviceCounts, no copy in between)
    opts = [abi.ExecOptions(0, base, base, stream, flags, counts_dev[k].data_ptr()) for k in range(2)]

    # the library's own communicator: rank 0 draws the id, torch.distributed (already up for the barriers) carries it to the others
    comm, last_result, pending = None, [None], [0]
    if distributed and args.comm == "c-abi" and args.backend == "nccl":
        # every rank takes part in both collectives below whatever happens on it, so that a rank without the library's
        # communicator cannot leave the others waiting in one of them
        id_t = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                id_t.copy_(torch.frombuffer(bytearray(be.comm_unique_id()), dtype=torch.uint8))
            except Exception as e:      # noqa: BLE001  (no RCCL to load: the id stays zero, which every rank reads as "no communicator")
                sys.stderr.write("rank 0: no RCCL id (%s); the exchange stays in torch.distributed\n" % e)
        dist.broadcast(id_t, 0)
        id_bytes = bytes(id_t.cpu().numpy().tobytes())
        if any(id_bytes):
            # the communicator is set up on a thread of its own: if that does not return in two minutes the bench goes on with
            # torch.distributed instead of hanging (the thread is left behind)
            import threading
            made, failure = [None], [None]

            def create():
                try:
                    torch.cuda.set_device(dev)
                    made[0] = be.comm_create_rccl(rank, world, id_bytes)
                except Exception as e:      # noqa: BLE001
                    failure[0] = e

            t = threading.Thread(target=create, daemon=True)
            t.start()
            t.join(120.0)
            if t.is_alive() or failure[0] is not None:
                sys.stderr.write("rank %d: library communicator unavailable (%s); the exchange stays in torch.distributed\n"
                                 % (rank, "no answer from ncclCommInitRank in 120 s" if t.is_alive() else failure[0]))
            else:
                comm = made[0]
        ok = torch.tensor([1 if comm else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm:
            be.comm_destroy(comm)
            comm = None

    comm_keepalive = None
    if distributed and args.comm == "c-abi" and args.backend == "gloo":
        # rehearsal of the same C-ABI path on one GPU: the library's callback communicator, its all-gather carried by gloo between
        # the ranks' processes (host buffers, so every step waits for its Exec: a functional rehearsal, not a measurement)
        def gloo_all_gather(send, recv, count, _stream):
            mine = torch.tensor([send[i] for i in range(count)], dtype=torch.int32)
            parts = [torch.zeros(count, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(parts, mine)
            for r in range(world):
                for i in range(count):
                    recv[r * count + i] = int(parts[r][i])
            return 0

        comm, comm_keepalive = be.comm_create_callback(rank, world, gloo_all_gather, abi.VVM_COMM_BUFFERS_HOST)
    comm_ranks = int(be.lib.vvm_comm_world(comm)) if comm else 0
    if comm and comm_ranks != world:
        sys.stderr.write("bench.py: the communicator reports %d rank(s), the job has %d\n" % (comm_ranks, world))
        sys.exit(2)

    def step():
        k = steps_done[0] & 1
        steps_done[0] += 1
        if comm:
            # at most `--exchange-depth` exchanges in flight: the oldest is collected before the next step is enqueued
            if pending[0] == args.exchange_depth:
                last_result[0] = be.exec_sharded_wait(comm)
                pending[0] -= 1
            be.exec_sharded_async(frame, opt, comm)
            pending[0] += 1
            return
        with torch.cuda.stream(torch_stream):
            if distributed:
                # the path's one exchange: every rank's survivor count, for the global acquireindex offsets
                torch_stream.wait_event(exchanged[k])
                state.exec_async(frame, opts[k])
                counted[k].record(torch_stream)
            else:
                state.exec_async(frame, opt)
        if distributed:
            if args.backend == "nccl":
                with torch.cuda.stream(coll_stream):
                    coll_stream.wait_event(counted[k])
                    dist.all_gather_into_tensor(gathered[k], counts_dev[k])
                    exchanged[k].record(coll_stream)
            else:      # rehearsal: gloo gathers host tensors
                torch_stream.synchronize()
                parts = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
                dist.all_gather(parts, counts_dev[k].cpu())
                gathered[k].copy_(torch.cat(parts))

    def drain():
        while comm and pending[0]:
            last_result[0] = be.exec_sharded_wait(comm)
            pending[0] -= 1

    for _ in range(args.warmup):
        step()
    survivors = state.finish(frame)
    drain()
    torch.cuda.synchronize()

    state.enable_timing(True, span=args.timing == "span")
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    survivors = state.finish(frame)
    drain()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms_sum, launches = state.collect_kernel_ms()
    state.enable_timing(False)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if comm:
            res = last_result[0]
            assert res is not None and res.LocalSurvivors[0] == survivors, (rank, survivors)
            # every rank's figures are consistent with the others': offsets are the running sum of the local counts in rank order
            mine = torch.tensor([res.LocalSurvivors[0], res.GlobalOffset[0], res.GlobalSurvivors[0]], dtype=torch.int64, device=dev)
            everyone = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(everyone, mine)
            rows = [e.tolist() for e in everyone]
            assert all(rows[r][1] == sum(q[0] for q in rows[:r]) and rows[r][2] == sum(q[0] for q in rows) for r in range(world)), rows
        else:
            last = gathered[(steps_done[0] - 1) & 1].tolist()
            offs = global_offsets(last)
            assert last[rank] == survivors, (last, rank, survivors)
            assert offs[rank] == sum(last[:rank])

    # what a plain device-to-device copy of the same sixteen streams sustains on this box, for scale (not a bound the kernel is
    # priced against: `peak` stays the HBM peak)
    copy_gbps = None
    if rank == 0:
        for _ in range(2):
            for a, b in zip(ins, outs):
                b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for a, b in zip(ins, outs):
                b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 8 * 2 * 4.0 * (n + PAD) / (e0.elapsed_time(e1) * 1e-3) / 1e9

    if rank == 0:
        info = state.info()
        total_updates = float(n) * world * args.steps
        value = total_updates / elapsed
        alg_bytes = float(info.BytesInPerInstance) * n + float(info.BytesOutPerInstance) * survivors
        kernel_ms = kernel_ms_sum / max(launches, 1)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        line = {
            "metric": "particle updates/sec", "value": value, "unit": "particle updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s script (8 f32 streams in, 8 out, kill via acquireindex), %d particles per GPU, "
                                   "%.1f%% survive" % (args.workload, n, 100.0 * survivors / max(n, 1)),
                       "particles_per_gpu": n, "survivors_rank0": survivors, "instances_per_lane": state.last_exec_width(),
                       "parallelism": "instance ranges sharded over %d rank(s)%s; all-gather of survivor counts by %s" % (
                           world, "" if args.backend == "nccl" else " that share GPU 0 (gloo rehearsal)",
                           ("the library's RCCL communicator of %d rank(s) (vvm_exec_sharded_async)" % comm_ranks if args.backend == "nccl" else
                            "the library's callback communicator of %d rank(s) over gloo (vvm_exec_sharded_async)" % comm_ranks) if comm
                           else "torch.distributed (%d ranks)" % dist.get_world_size()) if distributed
                       else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": recorded_traffic(args.workload, n), "kernel": "vvm_exec_kernel", "kernel_ms": kernel_ms, "launches": launches,
                         "kernel_ms_from": ("one HIP event pair around the %d launches on their stream, divided by %d" % (launches, launches)) if args.timing == "span"
                                           else "mean of one HIP event pair per launch",
                         "algorithmic_bytes_per_launch": alg_bytes, "device_copy_GBps": copy_gbps},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        elif world == 1:
            line["cpu_baseline"] = None
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()

    if comm:
        be.comm_destroy(comm)
    state.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
