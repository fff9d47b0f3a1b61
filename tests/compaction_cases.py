"""Compaction checked against the reference's own tables (tests/golden/compaction_tables.json: the byte selectors and the
destination advances of VectorVMRuntime.cpp:132-186), shared by the oracle test (CPU) and the parity test (GPU).

A script copies one stream of every kind through the four kinds of output instruction under an acquireindex whose predicate is
an input stream, so the keep mask of every group of four is whatever the test says.  What the outputs must hold follows from the
tables alone (`by_the_tables`): per group, the selector row of the group's mask picks the source bytes, the destination moves
on by the advance of the mask (VVM_Output32, VectorVMRuntime.cpp:744-822; VVM_Output16 and _from_16, :579-742; the mask byte is
acquireindex's, one bit per instance, instances past the end never survive, :838-857).
"""
import json
import os

import numpy as np

from renderer_amd.assembler import FLOAT, HALF, INT32, Script

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compaction_tables.json")


def load_tables():
    with open(FIXTURE) as f:
        t = json.load(f)["tables"]
    return {k: np.array(v["values"], dtype=np.uint8) for k, v in t.items()}


def compaction_script():
    """inputs: keep (int32: the sign bit keeps), x (float32), i (int32), h (half) -> float, int32, half (from x), half (from h)"""
    s = Script()
    x = s.input(0, FLOAT, 0)
    keep, i = s.input(0, INT32, 0), s.input(0, INT32, 1)
    h = s.input(0, HALF, 0)
    s.acquireindex(keep, 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, 0))])
    s.output_data(INT32, 0, [(i, s.output(0, INT32, 0))])
    s.output_data(HALF, 0, [(x, s.output(0, HALF, 0))])
    s.output_data(HALF, 0, [(h, s.output(0, HALF, 1))], from_half=True)
    return s.build()


OUT_DTYPES = (np.float32, np.int32, np.float16, np.float16)


def inputs_for(keep: np.ndarray, seed: int = 7):
    """Streams in the harness's order (float32, then int32, then half).  x takes values a half holds exactly, so that the
    float -> half output is a matter of order alone."""
    n = len(keep)
    rng = np.random.default_rng(seed)
    x = (rng.integers(-1024, 1024, n).astype(np.float32)) * np.float32(0.25)
    i = rng.integers(-2 ** 31, 2 ** 31 - 1, n, dtype=np.int64).astype(np.int32)
    h = (rng.integers(-2048, 2048, n).astype(np.float32) * np.float32(0.5)).astype(np.float16)
    k = np.where(keep, np.int32(-1), np.int32(rng.integers(0, 2 ** 31 - 1))).astype(np.int32)
    return [x, k, i, h]


def by_the_tables(values: np.ndarray, keep: np.ndarray, tables) -> np.ndarray:
    """The survivors of `values` as the reference's loop leaves them: group by group through the selector and advance tables."""
    n = len(values)
    size = values.dtype.itemsize
    shuffle = tables["shuffle32" if size == 4 else "shuffle16"].reshape(16, 16)
    advance = tables["advance32" if size == 4 else "advance16"]
    loops = (n + 3) // 4
    src = np.zeros(loops * 4, dtype=values.dtype)
    src[:n] = values
    src = src.view(np.uint8).reshape(loops, 4 * size)
    kept = np.zeros(loops * 4, dtype=bool)
    kept[:n] = keep                                       # (instances past the end never survive)
    masks = (kept.reshape(loops, 4) * np.array([1, 2, 4, 8])).sum(axis=1)
    dst = np.zeros(loops * 4 * size + 16, dtype=np.uint8)
    at = 0
    for g in range(loops):
        row = shuffle[masks[g]][:4 * size]
        picked = np.where(row == 0xFF, 0, src[g][np.minimum(row, 4 * size - 1)])      # pshufb: a selector with the top bit set yields zero
        dst[at:at + 4 * size] = picked
        at += int(advance[masks[g]])
    return dst[:at].view(values.dtype)


def every_pair_of_masks() -> np.ndarray:
    """2048 instances: every pair of 4-bit masks in two neighbouring groups, once."""
    keep = np.zeros(2048, dtype=bool)
    for a in range(16):
        for b in range(16):
            base = (a * 16 + b) * 8
            for l in range(4):
                keep[base + l] = bool(a & (1 << l))
                keep[base + 4 + l] = bool(b & (1 << l))
    return keep
