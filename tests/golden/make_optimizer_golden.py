#!/usr/bin/env python3
"""Writes tests/golden/optimizer_fixtures.json: compiler-format scripts (tests/original_cases.py) and the blobs
vvm_optimize_script turns them into today.  The reference holds no optimizer vectors and may not be run, so these pin this
project's own output against accidental change; they say nothing about byte-identity with the reference."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import original_cases as oc          # noqa: E402
from renderer_amd import abi         # noqa: E402


def main():
    be = abi.load_hip_backend()
    cases = {"fountain": oc.fountain()[0]}
    for first, second, position, merged in oc.MERGE_PAIRS[::7]:
        cases["%s-%s-%d" % (first, second, position)] = oc.merge_pair(first, second, position)[0]
    out = {}
    for name, o in cases.items():
        bc = o.done()
        out[name] = {"bytecode": bc.hex(), "ext_fns": list(o.ext), "blob": be.optimize_script(bc, o.ext).hex()}
    with open(os.path.join(HERE, "optimizer_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
