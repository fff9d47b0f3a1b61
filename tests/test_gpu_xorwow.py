"""vvm_exec_options.RandXorwowState on the GPU: the random ops draw from the reference's generator (VectorVMRuntime.cpp:445-483) in
its execution order, bit for bit with the oracle, which tests/test_oracle_xorwow.py pins against a second statement."""
import ctypes as C

import numpy as np
import pytest

from op_cases import ALL_ONES
from renderer_amd import abi, scripts
from renderer_amd.assembler import FLOAT, INT32, Script
from vm_harness import run_hip, run_oracle
from xorwow_cases import WIDTHS, expected_outputs, py_draw_words, py_seed, random_script

pytestmark = pytest.mark.gpu
f32 = np.float32
CONSTS = ALL_ONES + np.array([0.0], f32).tobytes()
STATE = py_seed(0x0123456789ABCDEF, 0x00000000FEDCBA98, 0x00007F12345678C0)
DTS = [f32] * 4 + [np.int32]


def _streams(n, negative_share=0.0):
    rng = np.random.default_rng(n)
    scale = rng.uniform(0.5, 9.0, n).astype(f32)
    if negative_share:
        scale = np.where(rng.random(n) < negative_share, -scale, scale).astype(f32)
    return scale, rng.integers(1, 5000, n).astype(np.int32)


def _same(got, ref, count):
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g[:count].view(np.uint32), r[:count].view(np.uint32))


@pytest.mark.parametrize("n", [1, 5, 255, 1024, 1025, 14289, 100003])
@pytest.mark.parametrize("flags", [0, abi.VVM_EXEC_NO_VGPR_WIDE, abi.VVM_EXEC_FORCE_EIGHT_WIDE])
def test_matches_the_oracle(n, flags):
    scale, mod = _streams(n)
    ref, ref_n, _ = run_oracle(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE)
    got, got_n, info = run_hip(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE, flags=flags)
    assert got_n == ref_n == n
    _same(got, ref, n)


def test_matches_the_second_statement_directly():
    n = 20011
    scale, mod = _streams(n)
    got, got_n, _ = run_hip(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE, page_kb=8)
    ref, _, frame = run_oracle(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE, page_kb=8)
    chunk = frame.ctx.Internal.MaxInstancesPerChunk
    assert (n + chunk - 1) // chunk >= 3
    _same(got, expected_outputs(py_draw_words(STATE, n, WIDTHS, chunk), scale, mod, n), n)


@pytest.mark.parametrize("page_kb", [4, 16, 64])
def test_chunk_boundaries_and_dropped_instances(page_kb):
    """Dropped instances draw too (the generator steps per loop, not per survivor), and the survivors keep their own draws."""
    n = 50021
    scale, mod = _streams(n, negative_share=0.4)
    ref, ref_n, _ = run_oracle(random_script(True), [CONSTS], [scale, mod], DTS, n, xorwow=STATE, page_kb=page_kb)
    got, got_n, _ = run_hip(random_script(True), [CONSTS], [scale, mod], DTS, n, xorwow=STATE, page_kb=page_kb)
    assert got_n == ref_n and 0 < ref_n < n
    _same(got, ref, ref_n)


def test_two_million_instances():
    n = 2_000_003
    scale, mod = _streams(n)
    ref, ref_n, _ = run_oracle(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE)
    got, got_n, _ = run_hip(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE)
    assert got_n == ref_n == n
    _same(got, ref, n)


def test_a_script_of_the_full_kernels_and_one_with_many_registers():
    # transcendental ops beside the draws (the script's own kernel is a full one)
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    r, t, u = s.temps(3)
    s.op("random", x, r); s.op("sin", r, t); s.op("random_add", t, x, u)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0)), (u, s.output(0, FLOAT, 1))])
    built = s.build()
    n = 30001
    scale, _ = _streams(n)
    ref, ref_n, _ = run_oracle(built, [CONSTS], [scale], [f32, f32], n, xorwow=STATE)
    got, got_n, _ = run_hip(built, [CONSTS], [scale], [f32, f32], n, xorwow=STATE)
    assert got_n == ref_n == n
    _same(got, ref, n)
    # forty live values: registers in LDS slots
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    regs = s.temps(40)
    for k, reg in enumerate(regs):
        s.op("random" if k % 3 == 0 else "random_add", *((x, reg) if k % 3 == 0 else (x, regs[k - 1], reg)))
    acc = s.temp()
    s.op("add", regs[0], regs[1], acc)
    for reg in regs[2:]:
        s.op("add", acc, reg, acc)
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0)), (regs[39], s.output(0, FLOAT, 1))])
    built = s.build()
    ref, ref_n, _ = run_oracle(built, [CONSTS], [scale], [f32, f32], n, xorwow=STATE)
    got, got_n, info = run_hip(built, [CONSTS], [scale], [f32, f32], n, xorwow=STATE)
    assert got_n == ref_n == n and info.NumVirtualRegisters > 16
    _same(got, ref, n)


def test_scripts_without_draws_ignore_the_state_and_repeated_execs_agree():
    w = scripts.fountain()
    n = 5000
    ins = scripts.fountain_streams_np(0, n, 0.5)
    a, an, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n)
    b, bn, _ = run_hip(w.script, [w.constants], ins, [f32] * 8, n, xorwow=STATE)
    assert an == bn
    _same(a, b, an)
    scale, mod = _streams(n)
    c, _, _ = run_hip(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE, repeat=3)
    d, _, _ = run_hip(random_script(), [CONSTS], [scale, mod], DTS, n, xorwow=STATE)
    _same(c, d, n)


def test_where_the_state_is_refused():
    be = abi.load_hip_backend()
    L = be.lib
    n = 4096
    built = random_script()
    bufs = []
    try:
        def dev(nbytes):
            p = L.vvm_device_alloc(nbytes); assert p
            be._check(L.vvm_device_memset(p, 0, nbytes), "memset")
            bufs.append(p)
            return p
        ds = abi.DataSet([dev(4 * n + 64), dev(4 * n + 64)], [dev(4 * n + 64) for _ in range(5)], 0, [0, 1, 2], [0, 4, 5])
        state = be.alloc_state(built.blob)
        try:
            frame = abi.CallFrame(state, [ds], n, [CONSTS], ())
            words = (C.c_uint32 * 24)(*STATE)
            opt = abi.ExecOptions(0, 0, 0, None, 0)
            opt.RandXorwowState = words
            be.exec_ex(frame, opt)                                   # the plain entry point takes it
            assert L.vvm_exec_noalloc(C.byref(frame.ctx), C.byref(opt)) == abi.VVM_E_UNSUPPORTED_OP
            with pytest.raises(abi.VVMError) as e:
                be.exec_batch([frame], [opt])
            assert e.value.code == abi.VVM_E_UNSUPPORTED_OP
        finally:
            be.free_state(state)
    finally:
        for p in bufs:
            L.vvm_device_free(p)


def _two_dataset_random_script():
    """Dataset 0 keeps instances with x < 0.5 and writes a draw scaled by x; dataset 1 keeps those with y < 0.7 and writes another."""
    s = Script()
    x, y = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)
    k0, k1, r0, r1 = s.temps(4)
    s.op("cmplt", x, s.const(4), k0)      # (constants 0..3: the all-ones mask and three zeros)
    s.op("cmplt", y, s.const(5), k1)
    s.acquireindex(k0, 0)
    s.op("random", x, r0)
    s.acquireindex(k1, 1)
    s.op("random_add", y, x, r1)
    s.output_data(FLOAT, 0, [(r0, s.output(0, FLOAT, 0))])
    s.output_data(FLOAT, 1, [(r1, s.output(1, FLOAT, 0))])
    return s.build()


@pytest.mark.parametrize("mode", ["fused", "separate", "chains"])
def test_two_acquired_datasets(mode, monkeypatch):
    """One program per dataset: every pass numbers the script's draws alike and reads the same words.  The kernels that read the
    words have one chain: a state whose script runs both datasets in one pass (chains) takes the per-dataset programs for such an
    Exec."""
    if mode != "chains":
        monkeypatch.setenv("VVM_NO_CHAINS", "1")      # (read when the state is allocated)
    flags = abi.VVM_EXEC_SEPARATE_PASSES if mode == "separate" else 0
    n = 30011
    rng = np.random.default_rng(3)
    x, y = rng.random(n).astype(f32), rng.random(n).astype(f32)
    consts = [ALL_ONES + np.array([0.5, 0.7], f32).tobytes()]
    built = _two_dataset_random_script()
    ref_more, got_more = [], []
    ref, ref_n, _ = run_oracle(built, consts, [x, y], [f32], n, more_datasets=[[f32]], more_results=ref_more, xorwow=STATE, page_kb=8)
    got, got_n, info = run_hip(built, consts, [x, y], [f32], n, more_datasets=[[f32]], more_results=got_more, xorwow=STATE, page_kb=8, flags=flags)
    assert bool(info.Flags & 0x40000) == (mode == "chains")
    assert got_n == ref_n and got_more[0][1] == ref_more[0][1] and 0 < got_n < n
    _same(got, ref, got_n)
    _same(got_more[0][0], ref_more[0][0], got_more[0][1])
