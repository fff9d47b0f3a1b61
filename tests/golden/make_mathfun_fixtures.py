#!/usr/bin/env python3
"""Writes tests/golden/mathfun_constants.json and tests/golden/mathfun_bits.json: what pins this project's statement of the
"mathfun" sin / cos / exp / log (the absent third-party header behind the reference's VectorSin / VectorCos / VectorExp /
VectorLog, Core/Private/Math/UnrealMathSSE.cpp:8-38; call sites UnrealMathSSE.h:1947, 2940, 2995, 3066, 3091) to the
published algorithm.

The source of truth is a header that ships with this container's PyTorch, ATen/native/cpu/avx_mathfun.h: the AVX port of the
same sse_mathfun.h (zlib licence; same constants, same operation order, eight lanes instead of four).  It is neither part of
the reference nor of this repository.  Two things are taken from it:

  mathfun_constants.json  its _PS256_CONST(...) literals, read as TEXT, with the float32 each literal rounds to, and for each of
                          the four functions the order in which the function's body names its constants (the Horner steps);
  mathfun_bits.json       bit patterns of the four functions on the sample inputs of the reference's own math test
                          (tests/golden/mathtest_vectors.json), on sweeps of each function's domain and on edge values, computed
                          by mathfun_probe.cpp (beside this script), which this script compiles with g++ against that header
                          and runs here.  Arrays are base64 of little-endian uint32.

Tests read only the two JSON files.  Run again only where that header is present (this container)."""
import base64
import json
import os
import re
import subprocess
import sys
import tempfile
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_RELATIVE = "ATen/native/cpu/avx_mathfun.h"
FUNCTIONS = {"sin": "sin256_ps", "cos": "cos256_ps", "exp": "exp256_ps", "log": "log256_ps"}


def torch_include():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "include")


def nearest_float32(literal: str) -> int:
    """Bits of the float32 nearest to a decimal literal (what a C compiler stores for `float x = <literal>`, with or without an
    f suffix: a double literal that initialises a float is rounded from its exact decimal value once on every compiler this was
    checked against, and the candidates' neighbours are compared exactly here)."""
    text = literal.strip().rstrip("fF").replace(" ", "")
    exact = Fraction(text) if "e" not in text.lower() else Fraction(text.lower().split("e")[0]) * Fraction(10) ** int(text.lower().split("e")[1])
    guess = np.float32(float(exact))
    best = None
    for cand in (np.nextafter(guess, np.float32(-np.inf)), guess, np.nextafter(guess, np.float32(np.inf))):
        err = abs(Fraction(float(cand)) - exact)
        if best is None or err < best[0] or (err == best[0] and not (int(cand.view(np.uint32)) & 1)):
            best = (err, cand)
    # a literal without suffix is a double first: the two roundings agree unless the double lands exactly on a float32 tie
    assert np.float32(float(exact)) == best[1], literal
    return int(best[1].view(np.uint32))


def read_constants(header_text):
    consts = {}
    for m in re.finditer(r"^_PS256_CONST\(\s*(\w+)\s*,\s*([^;]+?)\s*\)\s*;", header_text, re.M):
        name, lit = m.group(1), m.group(2)
        consts[name] = {"literal": lit, "bits": nearest_float32(lit)}
    for m in re.finditer(r"^_PS256_CONST_TYPE\(\s*(\w+)\s*,\s*int\s*,\s*([^;]+?)\s*\)\s*;", header_text, re.M):
        name, lit = m.group(1), m.group(2)
        value = eval(lit.replace("(int)", ""), {"__builtins__": {}}) & 0xFFFFFFFF      # an integer expression such as ~0x7f800000
        consts[name] = {"literal": lit, "bits": value}
    order = {}
    for fn, cname in FUNCTIONS.items():
        m = re.search(r"inline\s+v8sf\s+%s\s*\(v8sf x\)\s*\{(.*?)^\}" % cname, header_text, re.S | re.M)
        body = re.sub(r"//[^\n]*", "", m.group(1))
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        order[fn] = re.findall(r"_ps256_(\w+)", body)
    return consts, order


def sample_inputs():
    """Per function: float32 inputs as uint32 bit patterns."""
    f32 = np.float32
    vectors = json.load(open(os.path.join(HERE, "mathtest_vectors.json")))["records"]
    ref = {r["op"]: np.array(r["inputs"][0], dtype=np.uint32) for r in vectors if r["op"] in ("sin", "cos", "exp", "log")}
    rng = np.random.default_rng(20240607)

    def bits(a):
        return np.asarray(a, dtype=f32).view(np.uint32)

    edge = bits([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 1e-45, 1.1754942e-38, 1.17549435e-38, 3.4028235e38, -3.4028235e38,
                 np.inf, -np.inf, np.nan, 0.70710677, 0.70710683, 0.78539816, 1.5707964, 3.1415927, 6.2831855, 8192.0, -8192.0,
                 88.37626, 88.376265, 88.3763, -88.37626, -88.3763, 89.0, -89.0, -103.0, 127.0, 128.0, 2147483520.0, 2147483648.0, 4e9, -4e9])
    trig = np.concatenate([ref["sin"], edge, bits(np.linspace(-8192.0, 8192.0, 1201)), bits(np.linspace(-7.0, 7.0, 701)),
                           bits(rng.uniform(-1e5, 1e5, 400)), bits(rng.uniform(-2e9, 2e9, 100))])
    expo = np.concatenate([ref["exp"], edge, bits(np.linspace(-100.0, 100.0, 1601)), bits(rng.uniform(-20, 20, 600))])
    loga = np.concatenate([ref["log"], edge, bits(np.logspace(-45, 38.5, 1601)), bits(rng.uniform(0.25, 4.0, 600)),
                           rng.integers(1, 0x7F800000, 300, dtype=np.uint32)])
    return {"sin": trig, "cos": trig, "exp": expo, "log": loga}


def b64(a):
    return base64.b64encode(np.asarray(a, dtype="<u4").tobytes()).decode()


def main():
    inc = torch_include()
    header = os.path.join(inc, HEADER_RELATIVE)
    text = open(header, encoding="utf-8", errors="replace").read()
    consts, order = read_constants(text)
    with open(os.path.join(HERE, "mathfun_constants.json"), "w") as f:
        json.dump({"about": "constants and their order of use in avx_mathfun.h (AVX port of sse_mathfun.h), read as text by make_mathfun_fixtures.py",
                   "source": HEADER_RELATIVE, "constants": consts, "order_of_use": order}, f, indent=1, sort_keys=True)
        f.write("\n")

    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "mathfun_probe")
        subprocess.check_call(["g++", "-O1", "-mavx2", "-mno-fma", "-ffp-contract=off", "-I", inc, os.path.join(HERE, "mathfun_probe.cpp"), "-o", exe])
        ins = sample_inputs()
        out = {"about": "bit patterns of avx_mathfun.h's sin256_ps / cos256_ps / exp256_ps / log256_ps, computed by mathfun_probe.cpp; "
                        "base64 of little-endian uint32 arrays; the first entries of each input array are the sample inputs of "
                        "UnrealMathTest.cpp:1366-1397, 1488-1502, 1510-1520", "functions": {}}
        for k, (fn, x) in enumerate(ins.items()):
            res = subprocess.run([exe], input=np.asarray(x, dtype="<u4").tobytes(), stdout=subprocess.PIPE, check=True).stdout
            all_four = np.frombuffer(res, dtype="<u4").reshape(4, -1)
            y = all_four[list(FUNCTIONS).index(fn)]
            out["functions"][fn] = {"count": int(len(x)), "inputs": b64(x), "outputs": b64(y)}
    with open(os.path.join(HERE, "mathfun_bits.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote mathfun_constants.json (%d constants) and mathfun_bits.json (%s)" % (len(consts), ", ".join("%s: %d" % (k, v["count"]) for k, v in out["functions"].items())))


if __name__ == "__main__":
    sys.exit(main())
