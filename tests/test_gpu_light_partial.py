"""Tiles of the light eight-wide kernel that lose some of their instances: its seventh and eighth output leave a tile in one of
three ways (as they lie when nothing died; compacted within the lane when more than three quarters of the tile survive; through the
unused ends of the staging slots otherwise, 512 survivors per slot end).  Survivor counts on both sides of every boundary of that
scheme, every keep pattern of a lane's four instances, and death rates from 0.1 % to 99.9 %, against the CPU oracle."""
import numpy as np
import pytest

from renderer_amd import abi, scripts
from test_gpu_eight_wide import _check8

pytestmark = pytest.mark.gpu
f32 = np.float32
TILE = 2048


def _run(dead, instance_offset=0):
    w = scripts.fountain()
    n = dead.size
    ins = scripts.fountain_streams_np(0, n)
    ins[6] = np.where(dead, ins[7], ins[6] * f32(0.0)).astype(f32)
    got_n = _check8(w.script, [w.constants], ins, [f32] * 8, n, instance_offset=instance_offset)
    assert got_n == int((~dead).sum())


@pytest.mark.parametrize("survivors", [0, 1, 2, 3, 4, 5, 511, 512, 513, 1023, 1024, 1025, 1535, 1536, 1537, 1538, 1540, 2044, 2046, 2047])
@pytest.mark.parametrize("where", ["front", "back", "spread"])
def test_survivor_counts_around_every_boundary(survivors, where):
    """Five tiles; the second and the fourth keep exactly `survivors` instances, the others all of theirs or a random half."""
    rng = np.random.default_rng(survivors * 3 + len(where))
    dead = np.zeros(5 * TILE - 77, bool)
    dead[2 * TILE:3 * TILE] = rng.random(TILE) < 0.5
    for t in (1, 3):
        d = np.ones(TILE, bool)
        if where == "front":
            d[:survivors] = False
        elif where == "back":
            d[TILE - survivors:] = False
        else:
            d[rng.permutation(TILE)[:survivors]] = False
        dead[t * TILE:(t + 1) * TILE] = d
    _run(dead, instance_offset=3)


def test_every_keep_pattern_of_a_lane():
    """Lane l of a tile holds instances 4l .. 4l+3 and 1024 + 4l .. 1024 + 4l+3: all sixteen keep patterns in both halves, in tiles
    that keep more than three quarters (patterns thinned out with fully alive lanes) and in tiles that keep fewer."""
    for alive_lanes in (0, 3):          # of every four lanes, that many keep everything
        dead = np.zeros(4 * TILE, bool)
        for i in range(dead.size // 4):
            lane = i % 256
            if lane % 4 < alive_lanes:
                continue
            pattern = (i * 7 + i // 256) % 16
            for k in range(4):
                dead[4 * i + k] = not ((pattern >> k) & 1)
        _run(dead)


@pytest.mark.parametrize("rate", [0.001, 0.01, 0.1, 0.24, 0.26, 0.5, 0.9, 0.999])
def test_death_rates(rate):
    rng = np.random.default_rng(int(rate * 1000))
    _run(rng.random(150_001) < rate, instance_offset=1)
