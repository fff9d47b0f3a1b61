"""The executor called from C++ through the reference's interface names (include/vvm_runtime.hpp: AllocVectorVMState /
ExecVectorVMState / FVectorVMExecContext / FDataSetMeta, VectorVMRuntime.h:10-33), in a process of its own with no Python and no
PyTorch in it: tests/cpp/runtime_caller.cpp (built by __graft_entry__.build()) runs the case files written here, and what it
leaves is compared with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from renderer_amd import scripts
from vm_harness import run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLER = os.path.join(ROOT, "tests", "cpp", "runtime_caller")
f32 = np.float32


def test_the_mirror_header_compiles_as_plain_cpp(tmp_path):
    """Layout checks of include/vvm_runtime.hpp against vvm_abi.h are static_asserts: compiling it is the test (no GPU)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "vvm_runtime.hpp"\nint main() { VectorVM::Runtime::FVectorVMExecContext c; return c.NumInstances; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


@pytest.mark.gpu
@pytest.mark.parametrize("n,offset,tables", [(100_003, 0, 1), (5_000, 37, 3), (300_000, 8, 2)])
def test_cpp_caller_matches_the_oracle(tmp_path, n, offset, tables):
    assert os.path.exists(CALLER), "tests/cpp/runtime_caller is built by __graft_entry__.build()"
    w = scripts.fountain()
    ins = scripts.fountain_streams_np(0, n, kill_half=True)
    ref, ref_n, ref_frame = run_oracle(w.script, [w.constants], ins, [f32] * 8, n, instance_offset=offset)
    d = str(tmp_path)
    open(os.path.join(d, "blob.bin"), "wb").write(w.script.blob)
    open(os.path.join(d, "consts.bin"), "wb").write(w.constants)
    # the constants cut into `tables` tables at dword boundaries (VectorVMRuntime.cpp:490-508 walks them as one index space)
    dwords = len(w.constants) // 4
    cuts = [dwords * k // tables for k in range(tables + 1)]
    sizes = [4 * (b - a) for a, b in zip(cuts, cuts[1:])]
    open(os.path.join(d, "case.txt"), "w").write("%d %d 8 8\n%s\n" % (n, offset, " ".join(map(str, sizes))))
    for k, a in enumerate(ins):
        buf = np.zeros(offset + n, f32)
        buf[offset:] = a[:n]
        buf.tofile(os.path.join(d, "in_%d.f32" % k))
    subprocess.run([CALLER, d], check=True, timeout=100)
    survivors, status, chunk = map(int, open(os.path.join(d, "result.txt")).read().split())
    assert status == 0 and survivors == ref_n and 0 < ref_n < n
    assert chunk == ref_frame.ctx.Internal.MaxInstancesPerChunk
    for k in range(8):
        got = np.fromfile(os.path.join(d, "out_%d.f32" % k), f32)
        assert np.array_equal(got[offset:offset + ref_n].view(np.uint32), ref[k][offset:offset + ref_n].view(np.uint32)), k
        assert np.all(got[:offset].view(np.uint32) == 0xCDCDCDCD) and np.all(got[offset + ref_n:].view(np.uint32) == 0xCDCDCDCD)
