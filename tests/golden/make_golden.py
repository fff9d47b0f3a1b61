#!/usr/bin/env python3
"""Writes tests/golden/fountain_fixtures.json from the CPU oracle.

The reference holds no fixture for the VectorVM path, so these are regression fixtures of this project's own oracle (itself
pinned by tests/test_oracle.py): survivor counts and CRC-32 of every compacted output stream of the fountain script at the
edge sizes, and one small case spelled out bit by bit.  Inputs come from scripts.fountain_streams_np (a counter hash), so a
fixture is (n, kill_half) -> expected outputs.   Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np

from renderer_amd import scripts
from vm_harness import run_oracle

SIZES = [0, 1, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1000, 14289]


def main():
    f32 = np.float32
    w = scripts.fountain()
    cases = []
    for n in SIZES:
        for kill in (False, True):
            ins = scripts.fountain_streams_np(0, max(n, 1), kill_half=kill)
            out, surv, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
            cases.append({"n": n, "kill_half": kill, "survivors": surv, "crc32": [zlib.crc32(o[:surv].tobytes()) for o in out]})
    n = 12
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    out, surv, _ = run_oracle(w.script, [w.constants], ins, [f32] * 8, n)
    small = {"n": n, "survivors": surv, "outputs_hex": [[int(v) for v in o[:surv].view(np.uint32)] for o in out]}
    with open(os.path.join(HERE, "fountain_fixtures.json"), "w") as f:
        json.dump({"script": "fountain (renderer_amd.scripts.fountain())", "inputs": "scripts.fountain_streams_np(0, n, kill_half=...)",
                   "cases": cases, "small": small}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
