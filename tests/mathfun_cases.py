"""The "mathfun" sin / cos / exp / log (the third-party header behind the reference's VectorSin / VectorCos / VectorExp /
VectorLog, absent from the reference tree) pinned to the published algorithm: tests/golden/mathfun_bits.json holds bit patterns
computed from the AVX port of that header which ships with PyTorch, tests/golden/mathfun_constants.json its constants and the
order in which each function uses them (both written by tests/golden/make_mathfun_fixtures.py).  Shared by the oracle test (CPU)
and the parity test (GPU)."""
import base64
import json
import os
import re

import numpy as np

from op_cases import ALL_ONES, single_op_script

f32 = np.float32
_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BITS = json.load(open(os.path.join(_GOLDEN, "mathfun_bits.json")))["functions"]
CONSTANTS = json.load(open(os.path.join(_GOLDEN, "mathfun_constants.json")))
FUNCTIONS = ("sin", "cos", "exp", "log")
# constants every function shares with plain arithmetic (0.5, 1) and the integer masks are not looked for in the sources
_TRIVIAL = {"1", "0p5", "min_norm_pos", "mant_mask", "inv_mant_mask", "sign_mask", "inv_sign_mask"}


def _array(text):
    return np.frombuffer(base64.b64decode(text), dtype="<u4")


def check_function(fn, run):
    """run = run_oracle or run_hip.  The VM's op must give the header's bits on every input (NaN results: NaN, payload free)."""
    rec = BITS[fn]
    x, want = _array(rec["inputs"]).view(f32), _array(rec["outputs"]).copy()
    assert len(x) == rec["count"] >= 2000
    if fn == "log":
        # the op is `a > 0 ? log_ps(a) : 0` (VectorVMRuntime.cpp:1919)
        want[~(x > 0)] = 0
    script, n_in, n_out = single_op_script(fn)
    assert (n_in, n_out) == (1, 1)
    outs, surv, _ = run(script, [ALL_ONES], [x], [f32], len(x))
    assert surv == len(x)
    got = outs[0][:len(x)].view(np.uint32)
    nan = np.isnan(want.view(f32))
    assert np.array_equal(np.isnan(got.view(f32)), nan)
    bad = np.nonzero((got != want) & ~nan)[0]
    assert len(bad) == 0, "%s: %d of %d differ, first at x = %r: got %08x, header %08x" % (
        fn, len(bad), len(x), float(x[bad[0]]), got[bad[0]], want[bad[0]])


def expected_constant_bits(fn):
    """float32 bits of the non-trivial constants of `fn` in the order the header's function body uses them."""
    plain = {int(f32(v).view(np.uint32)) for v in (0.0, 0.5, 1.0)}      # (cephes_exp_p5 = 5.0000001201E-1 is 0.5 in float32)
    return [CONSTANTS["constants"][name]["bits"] for name in CONSTANTS["order_of_use"][fn]
            if name not in _TRIVIAL and CONSTANTS["constants"][name]["bits"] not in plain]


def _function_body(source, name):
    m = re.search(r"\b%s\s*\([^;{)]*\)\s*\{" % re.escape(name), source)
    assert m, name
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(source[i], 0)
        i += 1
    return source[m.end():i - 1]


def literal_bits(source, function_names):
    """float32 bits of the float literals (other than 0, 0.5, 1) in the bodies of the named functions, in source order."""
    out = []
    for name in function_names:
        body = re.sub(r"//[^\n]*|/\*.*?\*/", "", _function_body(source, name), flags=re.S)
        for lit in re.findall(r"(?<![\w.])(-?\d+\.\d*(?:[eE][+-]?\d+)?)[fF]\b", body):
            v = np.float32(float(lit))
            if float(v) not in (0.0, 0.5, 1.0):
                out.append(int(v.view(np.uint32)))
    return out
