"""The reference's random generator (VectorVMRuntime.cpp:445-483, 953-956) stated a second time, in numpy, and the script the
xorwow tests run.  The expected values below come from this statement alone: it pins the oracle, and through the oracle the GPU."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from renderer_amd.assembler import FLOAT, INT32, Script

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def py_seed(cycles_first: int, cycles_second: int, batch_state_address: int) -> List[int]:
    """SetupRandStateForBatch, :445-464."""
    st = cycles_first & M64
    inc = ((((batch_state_address << 32) & M64) ^ 0xCAFEF00DD15EA5E5) | 1) & M64
    st ^= (cycles_second << 32) & M64
    words = []
    for _ in range(20):
        old = st
        st = (old * 6364136223846793005 + inc) & M64
        xs = (((old >> 18) ^ old) >> 27) & M32
        rot = old >> 59
        words.append(((xs >> rot) | (xs << ((-rot) & 31))) & M32)
    return words + [inc & M32, inc >> 32, st & M32, st >> 32]


class PyXorwow:
    """Four generators side by side (one per lane), :468-483."""

    def __init__(self, state24: Sequence[int]):
        a = np.array(state24, dtype=np.uint32)
        self.s = [a[4 * i:4 * i + 4].copy() for i in range(5)]
        self.ctr = a[20:24].copy()

    def step(self) -> np.ndarray:
        t, s = self.s[4], self.s[0]
        self.s[4], self.s[3], self.s[2], self.s[1] = self.s[3], self.s[2], self.s[1], s
        t = t ^ (t >> np.uint32(2))
        t = t ^ (t << np.uint32(1))
        t = t ^ (s ^ (s ^ (s << np.uint32(4))))
        self.s[0] = t
        self.ctr = self.ctr + np.uint32(362437)
        return t + self.ctr


def py_draw_words(state24: Sequence[int], n: int, widths: Sequence[int], chunk_instances: int) -> np.ndarray:
    """[draw][instance] the 32 bits of every draw of a script whose random instructions make `widths` draws per loop, in the
    reference's order: chunk by chunk (:2276-2295), instruction by instruction, loop by loop."""
    gen = PyXorwow(state24)
    draws = sum(widths)
    loops_total = (n + 3) // 4
    out = np.zeros((draws, loops_total * 4), np.uint32)
    start = 0
    while start < n:
        count = min(chunk_instances, n - start)
        loops = (count + 3) // 4
        first = 0
        for w in widths:
            for l in range(loops):
                for j in range(w):
                    out[first + j, start + 4 * l:start + 4 * l + 4] = gen.step()
            first += w
        start += chunk_instances
    return out


def unit_float(words: np.ndarray) -> np.ndarray:
    """asfloat((x >> 9) | 0x3F800000) - 1.0, :955."""
    return ((words >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


WIDTHS = (1, 2, 1, 1)      # of random_script's random instructions, in program order


def random_script(kill_some: bool = False):
    """random, random_2x, random_add, randomi on a float stream `scale` and an int stream `mod`; outputs: four floats, one int.
    kill_some: instances whose `scale` is negative are dropped."""
    s = Script()
    x, m = s.input(0, FLOAT, 0), s.input(0, INT32, 0)
    if kill_some:
        keep = s.temp()
        s.op("cmpge", x, s.const(1), keep)      # constant 1 = 0.0 in the tests' table (the all-ones mask, then zeros)
        s.acquireindex(keep, 0)
    else:
        s.acquireindex(s.const(0), 0)
    r0, r1, r2, r3, r4 = s.temps(5)
    s.op("random", x, r0); s.op("random_2x", x, r1, r2); s.op("random_add", x, x, r3); s.op("randomi", m, r4)
    s.output_data(FLOAT, 0, [(r, s.output(0, FLOAT, i)) for i, r in enumerate((r0, r1, r2, r3))])
    s.output_data(INT32, 0, [(r4, s.output(0, INT32, 0))])
    return s.build()


def expected_outputs(words: np.ndarray, scale: np.ndarray, mod: np.ndarray, n: int):
    """What random_script writes for instances [0, n) (none dropped), from the draws' words."""
    u = [unit_float(words[d, :n]) for d in range(5)]
    f = np.float32
    as_float = mod[:n].astype(f)
    return [u[0] * scale[:n], u[1] * scale[:n], u[2] * scale[:n], u[3] * scale[:n] + scale[:n], (u[4] * as_float).astype(np.int32)]
