"""Stale look-back words in vvm_exec_batch, constructed on purpose.  Run as a program in a fresh process: what the batch
workspace's words and launch words hold then is known.  (tests/test_gpu_batch_tags.py builds the same situation inside the
test process, on the fresh workspace of a vvm_batch handle; this program is what showed the defect on a library built with
the old tagging: VVM_TEST_LIB=<that library> python tests/batch_epoch_case.py.)

The launches of a batch share one array of look-back words.  Three calls:
  A  an eight-wide Exec and a four-wide one: two kernel shapes, so two launches;
  B  one four-wide Exec of 59 tiles with nobody dying: its words end up holding inclusive prefixes 1024, 2048, ...;
  C  an eight-wide Exec of two tiles, then a four-wide one of 40 tiles with half of the instances dying, whose words lie where
     B's were.
With VVM_DEBUG_SKIP=1024 the odd tiles publish their counts late, so the even tiles of C's second Exec poll words that only B has
written.  If such a word carried a tag that C's launch accepts, the tile would take B's prefix as its own and write its survivors
at the wrong place.  Every Exec of C is compared with the oracle.  Exit status 0: all equal; 1: a difference (printed).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if __name__ == "__main__":
    os.environ["VVM_DEBUG_SKIP"] = "1024"

import numpy as np  # noqa: E402

from renderer_amd import abi, scripts  # noqa: E402
from renderer_amd.runtime import VectorVMState  # noqa: E402
from vm_harness import run_oracle  # noqa: E402

if os.environ.get("VVM_TEST_LIB"):          # (a library built from other sources, to show what the case catches)
    abi._HIP_LIB_PATH = os.environ["VVM_TEST_LIB"]

f32 = np.float32
CAP = 80_000      # elements per stream: a misplaced tile must still land inside the allocation


class Emitter:
    def __init__(self, be, variant, n, kill_half, seed, cap=None):
        global CAP
        if cap is not None:
            CAP = cap
        self.be, self.n = be, n
        self.w = scripts.fountain(variant=variant)
        self.host_in = scripts.fountain_streams_np(seed, CAP if cap is not None else n, kill_half=kill_half)      # (cap given: the size may change later)
        self.ins, self.outs = [], []
        for a in self.host_in:
            buf = np.zeros(CAP, f32)
            buf[:len(a)] = a
            p = be.lib.vvm_device_alloc(buf.nbytes)
            assert p
            be._check(be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
            self.ins.append(p)
        for _ in range(8):
            p = be.lib.vvm_device_alloc(CAP * 4)
            assert p
            be._check(be.lib.vvm_device_memset(p, 0xCD, CAP * 4), "memset")
            self.outs.append(p)
        self.state = VectorVMState(self.w.script.blob, be)
        self.frame = self.state.frame([abi.DataSet(self.ins, self.outs, 0)], n, [self.w.constants])

    def free(self):
        self.state.close()
        for p in self.ins + self.outs:
            self.be.lib.vvm_device_free(p)

    def check(self, name, tail=True):
        ref, ref_n, _ = run_oracle(self.w.script, [self.w.constants], self.host_in, [f32] * 8, self.n)
        got_n = self.frame.survivors(0)
        if got_n != ref_n:
            print("MISMATCH %s: %d survivors, the oracle has %d" % (name, got_n, ref_n))
            return False
        for k in range(8):
            h = np.empty(CAP, f32)
            self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], h.nbytes), "download")
            if not np.array_equal(h[:ref_n].view(np.uint32), ref[k][:ref_n].view(np.uint32)):
                bad = int(np.nonzero(h[:ref_n].view(np.uint32) != ref[k][:ref_n].view(np.uint32))[0][0])
                print("MISMATCH %s: stream %d differs from element %d on" % (name, k, bad))
                return False
            if tail and not np.all(h[ref_n:].view(np.uint32) == 0xCDCDCDCD):      # (only meaningful for the first Exec into fresh streams)
                print("MISMATCH %s: stream %d written past its survivors" % (name, k))
                return False
        return True


def main():
    be = abi.load_hip_backend()
    wide = abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_FORCE_EIGHT_WIDE)
    plain = abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_NO_EIGHT_WIDE)
    a8, a4 = Emitter(be, 0, 20_000, False, 1), Emitter(be, 1, 20_000, False, 2)
    b4 = Emitter(be, 2, 60_000, False, 3)
    c8, c4 = Emitter(be, 3, 4_000, True, 4), Emitter(be, 4, 40_000, True, 5)
    be.exec_batch([a8.frame, a4.frame], [wide, plain])
    be.exec_batch([b4.frame], [plain])
    be.exec_batch([c8.frame, c4.frame], [wide, plain])
    assert (c8.state.last_exec_width(), c4.state.last_exec_width()) == (8, 4)
    ok = a8.check("A/eight-wide") & a4.check("A/four-wide") & b4.check("B") & c8.check("C/eight-wide") & c4.check("C/four-wide")
    # a second round on the same states, sizes swapped around
    be.exec_batch([c4.frame, a8.frame], [plain, wide])
    be.exec_batch([b4.frame, c8.frame, a4.frame], [plain, wide, plain])
    ok = ok & c4.check("D/four-wide") & a8.check("D/eight-wide") & b4.check("E") & c8.check("E/eight-wide") & a4.check("E/four-wide")
    print("batch tags: %s" % ("all Execs equal the oracle" if ok else "DIFFERENCES"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
