"""Bring-up helper (not collected by pytest): one tiny fountain Exec; prints what came back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import scripts
from vm_harness import run_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = scripts.fountain()
ins = scripts.fountain_streams_np(0, max(n, 1))
got, surv, info = run_hip(w.script, [w.constants], ins, [np.float32] * 8, n)
print("skip=%s n=%d survivors=%d out0[:4]=%s" % (os.environ.get("VVM_DEBUG_SKIP"), n, surv, got[0][:4]), flush=True)
