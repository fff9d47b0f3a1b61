"""The epoch that tags a launch's look-back words counts 1 .. 2^30 - 1 and round again.  After the wrap, a word written 2^30
launches earlier by an Exec with more tiles than any since would carry the live tag again; the launch that follows the wrap
therefore clears the look-back words first.  Built here in three Execs on one state (vvm_debug_set_launch_epoch stands in for
the 2^30 launches in between):
  B  250 k instances four-wide at epoch 1: the words end up holding B's inclusive prefixes, tagged 1;
  W  2 k instances at epoch 2^30 - 1: the count goes round, the next epoch is 1 again;
  C  40 k instances of OTHER particles at epoch 1, with the debug kernels' publish delay (VVM_DEBUG_SKIP=1024: odd tiles
     publish late), so that even tiles poll words only B has written.  Without the clear they would take B's prefixes.
The same through vvm_exec_noalloc, whose launch words and look-back words are the device's scratch array."""
import os

import pytest

from batch_epoch_case import Emitter
from renderer_amd import abi

pytestmark = pytest.mark.gpu
LAST_EPOCH = (1 << 30) - 1


def _run(no_alloc, through_the_wrap=True):
    be = abi.load_hip_backend()
    cap = 600_000          # a misplaced tile would still land inside the allocation
    b = Emitter(be, 0, 250_000, True, 1, cap=cap)
    c = Emitter(be, 0, 40_000, True, 7, cap=cap)          # other particles, same script
    # C's streams through B's state
    c.frame = b.state.frame([abi.DataSet(c.ins, c.outs, 0)], 40_000, [c.w.constants])
    opt = abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_NO_EIGHT_WIDE)
    run = be.exec_noalloc if no_alloc else be.exec_ex
    scratch = 1 if no_alloc else 0
    saved = os.environ.get("VVM_DEBUG_SKIP")
    try:
        be._check(be.lib.vvm_debug_set_launch_epoch(b.state.ptr, 1, scratch), "set epoch")
        run(b.frame, opt)
        assert b.check("B", tail=False)
        if through_the_wrap:
            be._check(be.lib.vvm_debug_set_launch_epoch(b.state.ptr, LAST_EPOCH, scratch), "set epoch")
            b.n = b.frame.ctx.NumInstances = 2_000
            run(b.frame, opt)
            assert b.check("W", tail=False)
        else:      # what a library without the clear does: the epoch is 1 again and nothing has been cleared
            be._check(be.lib.vvm_debug_set_launch_epoch(b.state.ptr, 1, scratch), "set epoch")
        os.environ["VVM_DEBUG_SKIP"] = "1024"
        run(c.frame, opt)
        ok = c.check("C", tail=False)
        # and the Execs after it are ordinary ones
        os.environ.pop("VVM_DEBUG_SKIP", None)
        b.n = b.frame.ctx.NumInstances = 250_000
        run(b.frame, opt)
        assert b.check("B again", tail=False)
        return ok
    finally:
        if saved is None:
            os.environ.pop("VVM_DEBUG_SKIP", None)
        else:
            os.environ["VVM_DEBUG_SKIP"] = saved
        c.state.close()
        b.free()
        for p in c.ins + c.outs:
            be.lib.vvm_device_free(p)


@pytest.mark.parametrize("no_alloc", [False, True], ids=["own launch words", "scratch array"])
def test_the_launch_after_the_wrap_clears_the_look_back_words(no_alloc):
    assert _run(no_alloc)


if __name__ == "__main__":      # shows what the clear is for: the same Execs with the epoch set back to 1 and no clear in between
    import sys
    for no_alloc in (False, True):
        print("%s: through the wrap: %s" % ("scratch array" if no_alloc else "own launch words", "equal to the oracle" if _run(no_alloc) else "DIFFERS"))
        print("%s: epoch set back without a clear: %s" % ("scratch array" if no_alloc else "own launch words",
                                                            "equal to the oracle" if _run(no_alloc, through_the_wrap=False) else "DIFFERS (as expected)"))
    sys.exit(0)
