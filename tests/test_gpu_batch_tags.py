"""vvm_exec_batch: look-back words left by earlier calls are never taken for this call's (see batch_epoch_case.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_stale_look_back_words_are_not_accepted():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.pop("VVM_TEST_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(here, "batch_epoch_case.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
