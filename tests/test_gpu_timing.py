"""vvm_state_enable_timing / vvm_state_collect_kernel_ms: an event pair per Exec (1), or one pair around all Execs between two
collects (2: the Execs carry no events; the span includes the gaps between kernels).  Both on the same queue of Execs."""
import pytest

from batch_epoch_case import Emitter
from renderer_amd import abi

pytestmark = pytest.mark.gpu


def test_timing_per_exec_and_over_a_span():
    be = abi.load_hip_backend()
    e = Emitter(be, 0, 2_000_000, True, 3, cap=2_000_064)
    try:
        opt = abi.ExecOptions(0, 0, 0, None, 0)
        be.exec_ex(e.frame, opt)
        st = e.state
        results = {}
        for name, span in (("per exec", False), ("span", True), ("per exec again", False)):
            st.enable_timing(True, span=span)
            for _ in range(10):
                st.exec_async(e.frame, opt)
            st.finish(e.frame)
            ms, launches = st.collect_kernel_ms()
            assert launches == 10 and ms > 0.0
            results[name] = ms / launches
            # a collect with nothing enqueued since the last one reports nothing
            assert st.collect_kernel_ms() == (0.0, 0)
        st.enable_timing(False)
        assert e.check("after the timed Execs", tail=False)
        # the span's mean holds the kernels and the gaps between them: the same order of magnitude as the kernels' own mean (wide bounds:
        # another process may be using the GPU while one of the three is taken)
        per = (results["per exec"], results["per exec again"])
        assert 0.25 * min(per) < results["span"] < 4.0 * max(per) + 0.1, results
    finally:
        e.free()
