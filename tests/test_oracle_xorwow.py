"""The oracle's statement of the reference's own random generator (vvm_oracle_set_xorwow) against a second statement in numpy
(tests/xorwow_cases.py), and the seeding helper of libvvm_hip.so against both.  The reference seeds the generator from the cycle
counter (VectorVMRuntime.cpp:447-449), so it holds no vectors for it; what is pinned here is the step, the order of the steps
and the float mapping as the source states them."""
import ctypes as C

import numpy as np
import pytest

from op_cases import ALL_ONES
from oracle_binding import load_oracle
from renderer_amd import abi
from vm_harness import run_oracle
from xorwow_cases import WIDTHS, PyXorwow, expected_outputs, py_draw_words, py_seed, random_script, unit_float

f32 = np.float32
SEED_ARGS = (0x0123456789ABCDEF, 0x00000000FEDCBA98, 0x00007F12345678C0)
CONSTS = ALL_ONES + np.array([0.0], f32).tobytes()


def test_seeding_matches_the_second_statement():
    want = py_seed(*SEED_ARGS)
    assert load_oracle().xorwow_seed(*SEED_ARGS) == want
    assert abi.load_hip_backend().xorwow_seed(*SEED_ARGS) == want
    # the counters are the two 64-bit numbers of the seeding as four 32-bit lanes (:464); the increment is odd
    assert want[20] & 1 == 1 and len(set(want)) == 24
    assert py_seed(1, 2, 3) != py_seed(1, 2, 4) and py_seed(1, 2, 3) != py_seed(2, 2, 3)


def test_first_steps_by_hand():
    """One lane, all state words zero but State[4] = 1: t = 1 ^ (1 >> 2) = 1, t ^= t << 1 -> 3, s = 0 -> State[0] = 3; the counter
    advances by 362437 before it is added (:480-481)."""
    st = [0] * 24
    st[16] = 1                      # State[4], lane 0
    st[20] = 10                     # counter, lane 0
    g = PyXorwow(st)
    assert int(g.step()[0]) == 3 + 10 + 362437
    # second step: t = old State[3] = 0, s = State[0] = 3: only s << 4 is mixed in (s ^ s cancels, :478)
    assert int(g.step()[0]) == (3 << 4) + 10 + 2 * 362437
    assert unit_float(np.array([0xFFFFFFFF], np.uint32))[0] < 1.0 and unit_float(np.array([0], np.uint32))[0] == 0.0


@pytest.mark.parametrize("n", [1, 3, 4, 5, 37, 1000])
def test_one_chunk(n):
    state = py_seed(*SEED_ARGS)
    rng = np.random.default_rng(n)
    scale = rng.uniform(0.5, 9.0, n).astype(f32)
    mod = rng.integers(1, 5000, n).astype(np.int32)
    out, surv, frame = run_oracle(random_script(), [CONSTS], [scale, mod], [f32] * 4 + [np.int32], n, xorwow=state)
    assert surv == n and frame.ctx.Internal.MaxChunksPerBatch == 1
    words = py_draw_words(state, n, WIDTHS, frame.ctx.Internal.MaxInstancesPerChunk)
    for got, want in zip(out, expected_outputs(words, scale, mod, n)):
        np.testing.assert_array_equal(got[:n].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("n,page_kb", [(4001, 4), (9973, 4), (30000, 16)])
def test_several_chunks(n, page_kb):
    """A chunk runs every random instruction over all its loops before the next chunk starts, so the draws of one instruction
    are not consecutive steps across a chunk boundary."""
    state = py_seed(7, 9, 0x5000)
    rng = np.random.default_rng(n)
    scale = rng.uniform(0.5, 9.0, n).astype(f32)
    mod = rng.integers(1, 5000, n).astype(np.int32)
    out, surv, frame = run_oracle(random_script(), [CONSTS], [scale, mod], [f32] * 4 + [np.int32], n, xorwow=state, page_kb=page_kb)
    chunk = frame.ctx.Internal.MaxInstancesPerChunk
    assert surv == n and (n + chunk - 1) // chunk >= 3
    words = py_draw_words(state, n, WIDTHS, chunk)
    for got, want in zip(out, expected_outputs(words, scale, mod, n)):
        np.testing.assert_array_equal(got[:n].view(np.uint32), want.view(np.uint32))
    # and the chunking matters: taken as one chunk the same state gives other values from the second chunk on
    one = py_draw_words(state, n, WIDTHS, 1 << 30)
    assert np.array_equal(one[0, :chunk], words[0, :chunk]) and not np.array_equal(one[0, chunk:2 * chunk], words[0, chunk:2 * chunk])


def test_every_exec_starts_from_the_injected_state_and_none_restores_the_other_generator():
    state = py_seed(*SEED_ARGS)
    n = 64
    scale = np.ones(n, f32)
    mod = np.full(n, 100, np.int32)
    a, _, _ = run_oracle(random_script(), [CONSTS], [scale, mod], [f32] * 4 + [np.int32], n, xorwow=state)
    b, _, _ = run_oracle(random_script(), [CONSTS], [scale, mod], [f32] * 4 + [np.int32], n, xorwow=state)
    c, _, _ = run_oracle(random_script(), [CONSTS], [scale, mod], [f32] * 4 + [np.int32], n, seed=5)
    d, _, _ = run_oracle(random_script(), [CONSTS], [scale, mod], [f32] * 4 + [np.int32], n, seed=5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(c[0], d[0]) and not np.array_equal(a[0][:n], c[0][:n])
