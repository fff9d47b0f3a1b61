"""Several large launches on one GPU at the same time, from different processes.  Every launch is a grid of persistent workgroups
sized for the whole chip, so none of them is placed completely while the others run; a launch must still claim all of its tiles
with the workgroups it has (a tile's look-back waits for every earlier tile).  With the class of ticket counters that a workgroup
draws from tied to the XCD it runs on, two such launches could each hold the slots the other needed and sit until the look-back's
poll limit ended both ("look-back timed out"); this test failed about every second time on such a build."""
import os
import subprocess
import sys
import tempfile
import time

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("processes, n, execs", [(2, 8 * 1024 * 1024, 12), (3, 3 * 1024 * 1024, 16)])
def test_processes_sharing_the_gpu_all_finish(processes, n, execs):
    with tempfile.TemporaryDirectory() as d:
        go = os.path.join(d, "go")
        ready = [os.path.join(d, "ready%d" % i) for i in range(processes)]
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_process_case.py"), str(11 + i), str(n), str(execs), ready[i], go],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for i in range(processes)]
        try:
            t_end = time.time() + 150.0
            while not all(os.path.exists(r) for r in ready) and time.time() < t_end and all(p.poll() is None for p in procs):
                time.sleep(0.01)
            open(go, "w").close()
            outs = []
            for p in procs:
                out, _ = p.communicate(timeout=150)
                outs.append(out.decode(errors="replace"))
            assert [p.returncode for p in procs] == [0] * processes, "\n".join(o[-2000:] for o in outs)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()


def test_four_states_of_one_process_in_flight_together():
    """The same within one process: four states (the library gives them its four streams in turn), mid-sized Execs of different sizes
    queued on all of them before any is finished, several rounds; every state's last Exec equals the oracle."""
    from batch_epoch_case import Emitter
    from renderer_amd import abi
    be = abi.load_hip_backend()
    sizes = (700_001, 1_300_000, 2_100_003, 3_000_000)
    ems = [Emitter(be, i, n, i % 2 == 0, 21 + i, cap=max(sizes) + 64) for i, n in enumerate(sizes)]      # (one stream length for all: the helper keeps it in a global)
    try:
        opt = abi.ExecOptions(0, 0, 0, None, 0)
        for _ in range(10):
            for e in ems:
                be.exec_async(e.frame, opt)
        for e in ems:
            be.exec_finish(e.frame)
        for i, e in enumerate(ems):
            assert e.check("state %d" % i, tail=False)
    finally:
        for e in ems:
            e.free()


def test_twelve_states_on_twelve_caller_streams():
    """Twelve launches sized for the whole chip in flight together, each on a stream of the caller's (vvm_exec_options.Stream)."""
    from batch_epoch_case import Emitter
    from renderer_amd import abi, hip
    be = abi.load_hip_backend()
    n = 1_500_000
    ems = [Emitter(be, i % 8, n - 1000 * i, i % 3 != 0, 40 + i, cap=n + 64) for i in range(12)]
    streams = [hip.Stream() for _ in ems]
    try:
        opts = [abi.ExecOptions(0, 0, 0, s.value, 0) for s in streams]
        for _ in range(6):
            for e, o in zip(ems, opts):
                be.exec_async(e.frame, o)
        for e in ems:
            be.exec_finish(e.frame)
        for i, e in enumerate(ems):
            assert e.check("state %d" % i, tail=False)
    finally:
        for e in ems:
            e.free()
        for s in streams:
            s.destroy()


def test_host_threads_each_with_a_state_of_their_own():
    """Six host threads call the library at once, each on its own state (ctypes releases the interpreter lock inside a call): synchronous
    Execs of different sizes and kernel shapes, some through the entry point that allocates nothing (which takes turns on a lock), each
    result against the oracle."""
    import threading
    from batch_epoch_case import Emitter
    from renderer_amd import abi
    be = abi.load_hip_backend()
    sizes = (3_000, 70_001, 300_000, 9_001, 1_000_000, 150_000)
    cap = max(sizes) + 64
    ems = [Emitter(be, i, n, i % 2 == 1, 60 + i, cap=cap) for i, n in enumerate(sizes)]
    errors = []

    def work(i):
        try:
            e = ems[i]
            opt = abi.ExecOptions(0, 0, 0, None, 0)
            for k in range(40):
                if i % 3 == 0 and k % 4 == 0:
                    be.exec_noalloc(e.frame, opt)
                else:
                    be.exec_ex(e.frame, opt)
        except Exception as ex:      # noqa: BLE001
            errors.append((i, repr(ex)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(ems))]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        for i, e in enumerate(ems):
            assert e.check("thread %d" % i, tail=False)
    finally:
        for e in ems:
            e.free()
