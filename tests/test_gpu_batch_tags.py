"""Look-back words left by earlier launches of a batch workspace are never taken for the current launch's.

The launches of a batch share one array of look-back words through one set of launch words per kernel shape.  A vvm_batch
handle owns a fresh workspace, so what its words hold is known; three ticks build the situation:
  1  an eight-wide Exec (300 k instances) and a four-wide one: two kernel shapes, two launches;
  2  the first Exec empty, the second with 250 k instances and nobody dying: ONE shape, whose words end up holding the inclusive
     prefixes 1024, 2048, ...;
  3  300 k instances again, and 40 k with half of them dying in the second Exec, whose words now lie where tick 2's were.
With VVM_DEBUG_SKIP=1024 (debug kernels) odd tiles publish their counts late, so the even tiles of tick 3's second Exec poll
words that only tick 2 has written.  If those carried a tag that tick 3's launch accepts (launch words counting their own
epochs would produce exactly that here), the tiles would take tick 2's prefixes and write their survivors elsewhere.
"""
import os

import pytest

from batch_epoch_case import Emitter
from renderer_amd import abi

pytestmark = pytest.mark.gpu


def test_stale_look_back_words_are_not_accepted():
    be = abi.load_hip_backend()
    cap = 600_000          # a misplaced tile would still land inside the allocation
    big = Emitter(be, 0, 300_000, True, 1, cap=cap)
    small = Emitter(be, 1, 20_000, False, 2, cap=cap)
    handle = None
    saved = os.environ.get("VVM_DEBUG_SKIP")
    os.environ["VVM_DEBUG_SKIP"] = "1024"
    try:
        arr, oarr = be.pack_batch([big.frame, small.frame], [abi.ExecOptions(0, 0, 0, None, 0), abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_NO_EIGHT_WIDE)])
        handle = be.batch_create(arr, oarr)

        def tick(n_big, n_small, tag):
            big.n, small.n = n_big, n_small
            arr[0].NumInstances, arr[1].NumInstances = n_big, n_small
            be.batch_exec(handle, abi.VVM_BATCH_CHANGED_INSTANCES)
            assert big.check(tag + "/first", tail=False) and small.check(tag + "/second", tail=False)

        tick(300_000, 20_000, "tick 1")
        assert (big.state.last_exec_width(), small.state.last_exec_width()) == (8, 4)
        tick(0, 250_000, "tick 2")
        tick(300_000, 40_000, "tick 3")
        assert (big.state.last_exec_width(), small.state.last_exec_width()) == (8, 4)
        tick(100_000, 250_000, "tick 4")
        tick(300_000, 3_000, "tick 5")
    finally:
        if saved is None:
            os.environ.pop("VVM_DEBUG_SKIP", None)
        else:
            os.environ["VVM_DEBUG_SKIP"] = saved
        if handle:
            be.batch_destroy(handle)
        big.free()
        small.free()
