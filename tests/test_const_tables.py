"""Several constant tables per Exec (Niagara passes engine, emitter and system tables): oracle and HIP library map a constant to
(table, dword) by the reference's walk (VectorVMRuntime.cpp:490-508) and get the result of the single-table call."""
import numpy as np
import pytest

from const_table_cases import many_constant_script, splits
from renderer_amd import abi, scripts
from vm_harness import run_hip, run_oracle

f32 = np.float32


def _cases():
    w = scripts.fountain()
    n = 3001
    yield "fountain", w.script, w.constants, scripts.fountain_streams_np(3, n, kill_half=True), [f32] * 8, n
    script, consts = many_constant_script()
    yield "mad-chain", script, consts, [np.linspace(-2, 2, n).astype(f32)], [f32], n


def _check(run):
    for name, script, consts, ins, dts, n in _cases():
        want, want_n, _ = run_oracle(script, [consts], ins, dts, n)
        for split_name, tables in splits(consts):
            got, got_n, _ = run(script, tables, ins, dts, n)
            assert got_n == want_n, (name, split_name)
            for a, b in zip(got, want):
                assert np.array_equal(a[:got_n].view(np.uint32), b[:want_n].view(np.uint32)), (name, split_name)


def test_oracle_with_several_constant_tables():
    _check(run_oracle)


@pytest.mark.gpu
def test_hip_with_several_constant_tables():
    _check(run_hip)


@pytest.mark.gpu
def test_hip_refuses_a_constant_past_the_tables():
    script, consts = many_constant_script()
    x = [np.ones(8, f32)]
    with pytest.raises(abi.VVMError) as e:
        run_hip(script, [consts[:8], consts[8:-8]], x, [f32], 8)      # (the script reads dwords 0..8 of the ten)
    assert e.value.code == abi.VVM_E_INVALID_ARG      # (the reference leaves such a constant's map entry unset: no defined result to match)
