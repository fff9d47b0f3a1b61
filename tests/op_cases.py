"""One-instruction scripts and operand generators shared by the oracle tests (CPU) and the parity tests (GPU).

Every op of the optimized format that takes plain register operands gets a script of the form

    acquireindex(all-ones constant)          keep every instance
    out... = op(in0, in1, ...)               inputs are float-typed input streams carrying raw 32-bit patterns
    outputdata_float(out...)

so that a test can feed any bit pattern to any op and read the raw result back.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

from renderer_amd.assembler import FLOAT, HALF, INT32, Script
from renderer_amd.opcodes import INT_OPS, OPS

ALL_ONES = struct.pack("<I", 0xFFFFFFFF) + bytes(12)

# ops whose result depends on (seed, instance): checked for determinism and range, not against edge values
RANDOM_OPS = ("random", "randomi", "random_add", "random_2x")


def single_op_script(name: str, alias_outputs: bool = False):
    """Returns (BuiltScript, n_in, n_out)."""
    _, n_in, n_out = OPS[name]
    s = Script()
    ones = s.const(0)
    ins = [s.input(0, FLOAT, i) for i in range(n_in)]
    s.acquireindex(ones, 0)
    outs = s.temps(n_out)
    s.op(name, *ins, *outs)
    s.output_data(FLOAT, 0, [(outs[i], s.output(0, FLOAT, i)) for i in range(n_out)])
    return s.build(alias_outputs=alias_outputs), n_in, n_out


F32_EDGES = np.array([
    0x00000000, 0x80000000, 0x3F800000, 0xBF800000, 0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00000, 0x7FA00001,
    0x00000001, 0x80000001, 0x007FFFFF, 0x00800000, 0x7F7FFFFF, 0xFF7FFFFF,
    0x322BCC77, 0xB22BCC77, 0x322BCC76, 0x322BCC78,          # +-1e-8 (the guards of div/sqrt/rcp/rsq) and its neighbours
    0x4F000000, 0xCF000000, 0x4EFFFFFF, 0xCF000001, 0x4F800000,   # +-2^31 and neighbours, 2^32
    0x3F000000, 0xBF000000, 0x3FC00000, 0xBFC00000, 0x40200000, 0xC0200000, 0x3F7FFFFF, 0x3F800001,   # .5 1.5 2.5 ties
    0x40490FDB, 0xC0490FDB, 0x3FC90FDB, 0x40C90FDB, 0x42B17218, 0xC2B17218, 0x42B0C0A5,             # pi, pi/2, 2pi, +-88.72, 88.376
    0x3F3504F3, 0x3F3504F4, 0x41200000, 0x447A0000, 0x4B000000, 0x4B800000, 0xCB000000,             # sqrt(.5), 10, 1000, 2^23, 2^24
], dtype=np.uint32)

I32_EDGES = np.array([0, 1, 2, 3, 31, 32, 33, 64, 0xFFFFFFFF, 0xFFFFFFFE, 0x7FFFFFFF, 0x80000000, 0x80000001, 0x0000FFFF,
                      0xFFFF0000, 0x55555555, 0xAAAAAAAA, 7, 1000, 0xFFFFFC18], dtype=np.uint32)


def operand_streams(name: str, n_in: int, n: int = 4096, seed: int = 1234) -> List[np.ndarray]:
    """n_in float32-typed arrays of n raw bit patterns: all combinations of edge values first, then random material."""
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    edges = I32_EDGES if name in INT_OPS else F32_EDGES
    m = len(edges)
    cols = []
    for k in range(n_in):
        col = np.empty(n, np.uint32)
        # operand k walks the edge list with period m**k, so the first m**n_in rows hold every combination (as far as n allows)
        idx = (np.arange(n) // (m ** k)) % m
        col[:] = edges[idx]
        cols.append(col)
    cut = min(n, m ** min(n_in, 2) if n_in else 0)
    cut = min(cut, n // 2)
    for k in range(n_in):
        r = rng.integers(0, 2 ** 32, n - cut, dtype=np.uint64).astype(np.uint32)
        if name not in INT_OPS:
            # half of the random rows are moderate floats (so sums and products stay finite), half raw bits
            mod = (rng.standard_normal(n - cut) * (10.0 ** rng.integers(-3, 4, n - cut))).astype(np.float32).view(np.uint32)
            pick = rng.random(n - cut) < 0.5
            r = np.where(pick, mod, r)
        else:
            small = rng.integers(-50, 50, n - cut).astype(np.int32).view(np.uint32)
            pick = rng.random(n - cut) < 0.5
            r = np.where(pick, small, r)
        cols[k][cut:] = r
    return [c.view(np.float32) for c in cols]


# ---- tolerance classes of the GPU kernel against the oracle -------------------------------------------------------
# EXACT (default): bit-identical.  Everything that is IEEE add/sub/mul/div/sqrt, compares, selects, conversions, integer
# work, and the polynomial functions the kernel evaluates with the oracle's operation order.
# The rest call a math library whose GPU build differs from glibc, or a hardware estimate; the bound is stated in ULP of the
# oracle's result (None = relative bound, see REL_TOLERANCE).
ULP_TOLERANCE: Dict[str, int] = {
    "tan": 2, "atan": 2, "exp2": 2, "log2": 2,           # libm-class functions: glibc on the CPU, the device math library on the GPU
}
REL_TOLERANCE: Dict[str, float] = {
    "rcp": 2.0 ** -11,      # both sides are reciprocal ESTIMATES (rcpps: 1.5 * 2^-12 relative; the GPU's estimate is 1 ulp)
}


def all_plain_ops() -> List[str]:
    """half_to_float is left to the half-stream tests: its operand is a 16-bit stream, not a 32-bit register."""
    return sorted(n for n in OPS if n not in RANDOM_OPS and n != "half_to_float")
