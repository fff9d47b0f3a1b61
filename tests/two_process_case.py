"""One of several processes that run large Execs on ONE GPU at the same time (tests/test_gpu_concurrent_launches.py starts them).
Each Exec is compared with the oracle; exit status 0: all equal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from batch_epoch_case import Emitter  # noqa: E402
from renderer_amd import abi  # noqa: E402


def main():
    seed, n, execs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    be = abi.load_hip_backend()
    e = Emitter(be, seed % 8, n, True, seed, cap=n + 64)
    try:
        # wait for the others, so that the launches of all processes are in flight together
        ready, go = sys.argv[4], sys.argv[5]
        open(ready, "w").close()
        import time
        t_end = time.time() + 60.0
        while not os.path.exists(go) and time.time() < t_end:
            time.sleep(0.001)
        opt = abi.ExecOptions(0, 0, 0, None, 0)
        for _ in range(execs):
            be.exec_async(e.frame, opt)
        be.exec_finish(e.frame)
        ok = e.check("process %d" % seed, tail=False)
    finally:
        e.free()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
