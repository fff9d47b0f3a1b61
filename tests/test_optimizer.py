"""CPU tests of vvm_optimize_script (the restatement of OptimizeVectorVMScript, VectorVMOptimizer.cpp:700-3177, plus
FreezeOptimizerContext, VectorVMBridge.cpp:151-178).

The reference's tests hold no optimizer vectors, so byte-identity with its output is unpinned.  Pinned here: (1) the optimized
script computes, bit for bit, what the same script computes when it is translated one instruction at a time (tests/original_cases.py:
no reordering, no merged ops), both run by the CPU oracle; (2) the structure each pass is documented to produce.
"""
import numpy as np
import pytest

import original_cases as oc
from renderer_amd import abi
from renderer_amd.assembler import FLOAT, HALF, INT32
from renderer_amd.disasm import thaw
from renderer_amd.original import K, UNUSED, OriginalScript
from vm_harness import run_oracle

f32 = np.float32
ALL_ONES = b"\xff\xff\xff\xff"


class Blob:
    def __init__(self, blob):
        self.blob = blob


def be():
    return abi.load_hip_backend()


def optimize(o: OriginalScript):
    return be().optimize_script(o.done(), o.ext)


def both(o: OriginalScript, consts, inputs, out_dtypes, n, runner=run_oracle, **kw):
    """(outputs, survivors) of the optimized blob and of the instruction-by-instruction translation."""
    blob = optimize(o)
    direct = oc.translate(o).build()
    a = runner(Blob(blob), [consts], inputs, out_dtypes, n, **kw)
    b = runner(direct, [consts], inputs, out_dtypes, n, **kw)
    return a, b, thaw(blob)


def assert_same(a, b, n_out):
    assert a[1] == b[1]
    for k in range(n_out):
        np.testing.assert_array_equal(a[0][k][:a[1]].view(np.uint32), b[0][k][:b[1]].view(np.uint32))


def streams(n, seed=1):
    rng = np.random.default_rng(seed)
    fl = [rng.uniform(-4, 4, n).astype(f32) for _ in range(3)]
    it = [rng.integers(-50, 50, n).astype(np.int32) for _ in range(3)]
    return fl, it


# ---- whole scripts ----------------------------------------------------------------------------------------------------------
def test_fountain_in_compiler_format():
    o, meta = oc.fountain()
    n = 1000
    rng = np.random.default_rng(3)
    ins = [rng.uniform(-100, 100, n).astype(f32) for _ in range(6)] + [rng.uniform(0, 2, n).astype(f32), rng.uniform(1, 2, n).astype(f32)]
    a, b, t = both(o, meta["consts"], ins, [f32] * 8, n)
    assert_same(a, b, 8)
    assert 0 < a[1] < n
    names = [c[0] for c in t.code]
    # the kill test and acquireindex come first; mad + mul became mad_mul; every output sits right below the last instruction
    # that mentions its source, grouped where several meet
    assert names == ["add", "cmplt", "acquireindex"] + ["mad_mul", "mad", "outputdata_float"] * 3
    assert [len(c[2]) for c in t.code if c[0] == "outputdata_float"] == [2, 2, 4]
    # inputs are operands, not instructions, and every value lives in an output buffer: the predicate borrows output 0's, which
    # the first output instruction then overwrites; no temp register is left
    assert t.code[1] == ("cmplt", "o6", "i7", "o0") and t.code[2] == ("acquireindex", "o0", 0)
    assert t.code[-1][2] == [("o2", "o2"), ("o5", "o5"), ("o6", "o6"), ("i7", "o7")]
    assert t.num_temps == 0
    assert t.input_remap == list(range(8)) and t.input_dataset_offsets == [0, 8, 8, 8, 8, 8, 8, 0]
    assert t.const_remap == [0, 1, 4, 2, 3]               # dt, gx, drag, gy, gz: order of first use in the compiler's stream
    assert t.bytecode[-9:] == bytes(9) and t.out_dst == list(range(8)) and t.out_type == [0] * 8 and t.max_output_dataset == 1


@pytest.mark.parametrize("first,second,position,merged", oc.MERGE_PAIRS, ids=["%s-%s-%d" % p[:3] for p in oc.MERGE_PAIRS])
def test_merged_pairs(first, second, position, merged):
    o, _ = oc.merge_pair(first, second, position)
    n = 257
    fl, it = streams(n)
    if first in ("bit_lshift", "bit_rshift") or second in ("bit_lshift", "bit_rshift"):
        it = [np.abs(a) % 9 for a in it]
    if (first, second) == ("mad", "sqrt"):          # sqrt returns 0 below 1e-8, mad_sqrt is a bare square root (RT.cpp:1914, 2033)
        fl = [np.abs(a) + f32(0.5) for a in fl]
    a, b, t = both(o, ALL_ONES, fl + it, [f32], n)
    names = [c[0] for c in t.code]
    assert merged in names, names
    if (first, second, position) == ("neg", "cmplt", 1):
        # cmplt(x, -a) comes out as neg_cmplt(a, x) = cmplt(-a, x): the reference's merge keeps no record of which side the
        # negated value was on (VectorVMOptimizer.cpp:2435-2449, VectorVMRuntime.cpp:2067).  Restated as it is.
        want = np.where(-fl[0] < fl[1], 0xFFFFFFFF, 0).astype(np.uint32)
        np.testing.assert_array_equal(a[0][0][:n].view(np.uint32), want)
        return
    assert_same(a, b, 1)
    assert first not in names and (second not in names or second == "cmplt" and first == "select")


def test_a_result_read_twice_is_not_merged():
    o = OriginalScript()
    x, y = o.input(0, FLOAT, 0, o.temp()), o.input(0, FLOAT, 1, o.temp())
    idx = o.temp(); o.acquireindex(K(0), idx, 0)
    m = o.temp(); o.op("mul", x, y, m)
    s = o.temp(); o.op("add", m, x, s)
    o.output(FLOAT, 0, idx, s, 0)
    o.output(FLOAT, 0, idx, m, 1)          # second reader of the product
    n = 100
    fl, _ = streams(n)
    a, b, t = both(o, ALL_ONES, fl[:2], [f32, f32], n)
    assert_same(a, b, 2)
    assert [c[0] for c in t.code if c[0] in ("mul", "add", "mul_add")] == ["mul", "add"]


def test_unused_ops_and_casts_go_away():
    o = OriginalScript()
    x, y = o.input(0, FLOAT, 0, o.temp()), o.input(0, INT32, 0, o.temp())
    idx = o.temp(); o.acquireindex(K(0), idx, 0)
    dead1 = o.temp(); o.op("mul", x, x, dead1)
    dead2 = o.temp(); o.op("add", dead1, x, dead2)         # only reader of dead1, itself unread
    bits = o.temp(); o.op("fasi", x, bits)                 # float bits as int
    masked = o.temp(); o.op("bit_and", bits, K(1), masked)
    back = o.temp(); o.op("iasf", masked, back)
    r = o.temp(); o.op("add", back, K(2), r)
    o.output(FLOAT, 0, idx, r, 0)
    n = 64
    fl, it = streams(n)
    consts = ALL_ONES + np.array([0x7FFFFFFF], np.uint32).tobytes() + np.array([1.5], f32).tobytes()
    a, b, t = both(o, consts, [fl[0], it[0]], [f32], n)
    assert_same(a, b, 1)
    np.testing.assert_array_equal(a[0][0][:n], np.abs(fl[0]) + f32(1.5))
    names = [c[0] for c in t.code]
    assert "mul" not in names and "fasi" not in names and "iasf" not in names and names.count("add") == 1
    assert len(t.input_remap) == 2          # the int input is still listed: input instructions are never removed


def test_registers_reused_by_the_compiler_are_renamed():
    """One temp register written three times: each value gets its own name (and the first is dead)."""
    o = OriginalScript()
    x, y = o.input(0, FLOAT, 0, o.temp()), o.input(0, FLOAT, 1, o.temp())
    idx = o.temp(); o.acquireindex(K(0), idx, 0)
    r = o.temp()
    o.op("add", x, y, r)
    o.output(FLOAT, 0, idx, r, 0)
    o.op("mul", x, y, r)
    o.output(FLOAT, 0, idx, r, 1)
    o.op("sub", x, y, r)
    o.output(FLOAT, 0, idx, r, 2)
    n = 50
    fl, _ = streams(n)
    a, b, t = both(o, ALL_ONES, fl[:2], [f32] * 3, n)
    assert_same(a, b, 3)
    np.testing.assert_array_equal(a[0][1][:n], fl[0] * fl[1])
    assert t.num_temps == 0 and t.out_dst == [0, 1, 2]


def test_half_and_noadvance_inputs_and_constant_outputs():
    o = OriginalScript()
    h = o.input(0, HALF, 0, o.temp())
    x = o.input(0, FLOAT, 0, o.temp())
    na = o.input(0, FLOAT, 1, o.temp(), noadvance=True)
    idx = o.temp(); o.acquireindex(K(0), idx, 0)
    a1 = o.temp(); o.op("mul", h, x, a1)              # half read by an op: goes through half_to_float
    a2 = o.temp(); o.op("add", h, na, a2)             # second reader: the conversion is not repeated
    o.output(FLOAT, 0, idx, a1, 0)
    o.output(FLOAT, 0, idx, a2, 1)
    o.output(FLOAT, 0, idx, K(1), 2)                  # constant source
    o.output(HALF, 0, idx, h, 0)                      # half copied to a half output
    n = 77
    rng = np.random.default_rng(5)
    hs = rng.uniform(-2, 2, n).astype(np.float16)
    xs = rng.uniform(-2, 2, n).astype(f32)
    nas = np.full(n, 0.25, f32)
    consts = ALL_ONES + np.array([7.5], f32).tobytes()
    a, b, t = both(o, consts, [xs, nas, hs], [f32, f32, f32, np.float16], n)
    assert_same(a, b, 3)
    np.testing.assert_array_equal(a[0][3][:n].view(np.uint16), b[0][3][:n].view(np.uint16))
    np.testing.assert_array_equal(a[0][0][:n], hs.astype(f32) * xs)
    np.testing.assert_array_equal(a[0][2][:n], np.full(n, 7.5, f32))
    names = [c[0] for c in t.code]
    assert names.count("half_to_float") == 1 and "outputdata_half_from_half" in names
    # inputs sorted by kind: float, half, then no-advance float; offsets per kind; one no-advance input
    assert t.input_dataset_offsets == [0, 1, 1, 2, 3, 3, 3, 1] and t.num_noadvance == 1 and t.input_remap == [0, 0, 1]
    assert t.out_type == [0, 0, 0, 2]


def test_half_input_copied_to_a_float_output():
    """A half input that an output writes as float becomes outputdata_float_from_half (VectorVMOptimizer.cpp:1280-1398).  The
    reference's runtime handler for that opcode is an empty stub (VectorVMRuntime.cpp:573-577); here the instruction converts."""
    o = OriginalScript()
    h = o.input(0, HALF, 0, o.temp())
    x = o.input(0, FLOAT, 0, o.temp())
    idx = o.temp(); o.acquireindex(x, idx, 0)
    a1 = o.temp(); o.op("add", h, x, a1)
    o.output(FLOAT, 0, idx, h, 0)                     # half copied to a float output
    o.output(FLOAT, 0, idx, a1, 1)
    o.output(HALF, 0, idx, h, 0)
    n = 203
    rng = np.random.default_rng(11)
    hs = rng.integers(0, 0x7C00, n, dtype=np.uint32).astype(np.uint16).view(np.float16)      # finite halves, subnormals included
    xs = np.where(rng.random(n) < 0.5, f32(-2.0), f32(2.0)).astype(f32)                      # sign bit = keep
    a, b, t = both(o, ALL_ONES, [xs, hs], [f32, f32, np.float16], n)
    assert_same(a, b, 2)
    keep = xs < 0
    assert a[1] == int(keep.sum())
    np.testing.assert_array_equal(a[0][0][:a[1]].view(np.uint32), hs[keep].astype(f32).view(np.uint32))
    np.testing.assert_array_equal(a[0][2][:a[1]].view(np.uint16), hs[keep].view(np.uint16))
    names = [c[0] for c in t.code]
    assert "outputdata_float_from_half" in names and "outputdata_half_from_half" in names and names.count("half_to_float") == 1
    assert t.out_type == [0, 0, 2]


def test_exec_index_merges():
    o = OriginalScript()
    x = o.input(0, INT32, 0, o.temp())
    y = o.input(0, FLOAT, 0, o.temp())
    idx = o.temp(); o.acquireindex(K(0), idx, 0)
    e1 = o.temp(); o.exec_index(e1)
    f = o.temp(); o.op("i2f", e1, f)
    e2 = o.temp(); o.exec_index(e2)
    g = o.temp(); o.op("addi", e2, x, g)
    e3 = o.temp(); o.exec_index(e3)                    # read twice: stays
    h = o.temp(); o.op("addi", e3, e3, h)
    o.output(FLOAT, 0, idx, f, 0)
    o.output(INT32, 0, idx, g, 0)
    o.output(INT32, 0, idx, h, 1)
    n = 300
    fl, it = streams(n)
    a, b, t = both(o, ALL_ONES, [fl[0], it[0]], [f32, np.int32, np.int32], n)
    assert_same(a, b, 3)
    np.testing.assert_array_equal(a[0][0][:n], np.arange(n, dtype=f32))
    np.testing.assert_array_equal(a[0][1][:n], np.arange(n, dtype=np.int32) + it[0])
    names = [c[0] for c in t.code]
    assert names.count("exec_indexf") == 1 and names.count("exec_index_addi") == 1 and names.count("exec_index") == 1


def test_kill_predicate_chain_is_hoisted_and_outputs_follow_their_sources():
    """The instructions acquireindex depends on move up in front of everything else; outputs sit right below the instruction
    that computes their source and are grouped where they meet."""
    o = OriginalScript()
    x, y, z = [o.input(0, FLOAT, i, o.temp()) for i in range(3)]
    a = o.temp(); o.op("mul", x, y, a)
    b = o.temp(); o.op("sin", a, b)
    c = o.temp(); o.op("add", z, K(1), c)
    keep = o.temp(); o.op("cmplt", c, K(2), keep)
    idx = o.temp(); o.acquireindex(keep, idx, 0)
    o.output(FLOAT, 0, idx, b, 0)
    o.output(FLOAT, 0, idx, c, 1)
    n = 500
    fl, _ = streams(n, seed=9)
    consts = ALL_ONES + np.array([0.5, 1.0], f32).tobytes()
    r1, r2, t = both(o, consts, fl, [f32, f32], n)
    assert_same(r1, r2, 2)
    assert 0 < r1[1] < n
    names = [c_[0] for c_ in t.code]
    # c is computed above acquireindex, and the search for an output's place stops there: its output stays with b's at the end
    assert names == ["add", "cmplt", "acquireindex", "mul", "sin", "outputdata_float"] and len(t.code[-1][2]) == 2


def test_two_datasets_and_an_external_function():
    o = OriginalScript()
    x, y, z = [o.input(0, FLOAT, i, o.temp()) for i in range(3)]
    fn = o.ext_fn(3, 3)
    i0 = o.temp(); o.acquireindex(K(0), i0, 0)
    keep = o.temp(); o.op("cmplt", x, K(1), keep)
    i1 = o.temp(); o.acquireindex(keep, i1, 1)
    u, v = o.temp(), o.temp()
    o.call(fn, [x, y, z], [u, UNUSED, v])
    o.output(FLOAT, 0, i0, u, 0)
    o.output(FLOAT, 1, i1, v, 0)
    o.output(FLOAT, 1, i1, x, 1)
    n = 200
    fl, _ = streams(n, seed=4)
    consts = ALL_ONES + np.array([0.0], f32).tobytes()
    ext = [abi.VVM_EXTFN_CURL_NOISE3]
    more_a, more_b = [], []
    blob = optimize(o)
    direct = oc.translate(o).build()
    a = run_oracle(Blob(blob), [consts], fl, [f32], n, ext_fns=ext, more_datasets=[[f32, f32]], more_results=more_a)
    b = run_oracle(direct, [consts], fl, [f32], n, ext_fns=ext, more_datasets=[[f32, f32]], more_results=more_b)
    assert_same(a, b, 1)
    assert more_a[0][1] == more_b[0][1] and 0 < more_a[0][1] < n
    for k in range(2):
        np.testing.assert_array_equal(more_a[0][0][k][:more_a[0][1]].view(np.uint32), more_b[0][0][k][:more_b[0][1]].view(np.uint32))
    t = thaw(blob)
    call = [c for c in t.code if c[0] == "external_func_call"][0]
    assert call[2][4] == "-" and t.num_dummy_regs == 1 and t.ext_fns == [(3, 3)] and t.max_ext_fn_registers == 6
    assert t.max_output_dataset == 2 and t.out_dataset == [0, 1, 1]


def test_persistent_id_instructions_pass_through():
    o = OriginalScript()
    ids = o.input(0, INT32, 0, o.temp())
    age = o.input(0, FLOAT, 0, o.temp())
    keep = o.temp(); o.op("cmplt", age, K(1), keep)
    idx = o.temp(); o.acquireindex(keep, idx, 0)
    o.update_id(0, ids, idx)
    o.output(INT32, 0, idx, ids, 0)
    t = thaw(optimize(o))
    names = [c[0] for c in t.code]
    assert names == ["cmplt", "acquireindex", "update_id", "outputdata_int32"]
    upd = t.code[2]
    assert upd[1] == "i1" and upd[3] == 0               # the ID stream as an input operand (int32 inputs sort after float)
    assert be().inspect_blob(optimize(o)).NumInputBuffers == 2


def test_constants_are_remapped_in_order_of_first_use():
    o = OriginalScript()
    x = o.input(0, FLOAT, 0, o.temp())
    idx = o.temp(); o.acquireindex(K(9), idx, 0)
    a = o.temp(); o.op("add", x, K(5), a)
    b = o.temp(); o.op("mul", a, K(9), b)
    c = o.temp(); o.op("sub", b, K(2), c)
    o.output(FLOAT, 0, idx, c, 0)
    t = thaw(optimize(o))
    assert t.const_remap == [9, 5, 2] and t.num_consts_alloced == 512


def test_output_runs_are_sorted_and_batched():
    o = OriginalScript()
    x = o.input(0, FLOAT, 0, o.temp())
    k = o.input(0, INT32, 0, o.temp())
    idx = o.temp(); o.acquireindex(K(0), idx, 0)
    o.output(INT32, 0, idx, k, 1)
    o.output(FLOAT, 0, idx, x, 3)
    o.output(INT32, 0, idx, k, 0)
    o.output(FLOAT, 0, idx, x, 1)
    t = thaw(optimize(o))
    outs = [c for c in t.code if c[0].startswith("outputdata")]
    assert [c[0] for c in outs] == ["outputdata_float", "outputdata_int32"]
    # sorted by destination within a kind; serial numbers (order of appearance) name the output registers
    assert [d for _, d in outs[0][2]] == ["o3", "o1"] and [d for _, d in outs[1][2]] == ["o2", "o0"]
    assert t.out_dst == [1, 3, 0, 1] and t.out_type == [1, 0, 1, 0]


# ---- boundary ---------------------------------------------------------------------------------------------------------------
def test_empty_scripts_and_bad_bytecode():
    assert be().optimize_script(b"") == b"" and be().optimize_script(bytes([0])) == b""
    with pytest.raises(abi.VVMError):
        be().optimize_script(bytes([1, 0, 0]))                  # truncated add
    with pytest.raises(abi.VVMError):
        be().optimize_script(bytes([36, 0, 0, 0, 0, 0]))        # noise: not part of the format
    o = OriginalScript()
    x = o.input(0, FLOAT, 0, o.temp())
    with pytest.raises(abi.VVMError) as e:
        o2 = OriginalScript()
        o2.bc += bytes([1, 1]) + (6).to_bytes(2, "little") + (0).to_bytes(2, "little") + (1).to_bytes(2, "little")   # constant at byte 6
        be().optimize_script(o2.done())
    assert "0x10" in str(e.value)                               # the constant-remap error bit


def test_optimized_blob_links_for_the_gpu():
    o, _ = oc.fountain()
    info = be().inspect_blob(optimize(o))
    assert (info.NumInputBuffers, info.NumOutputBuffers, info.BytesInPerInstance, info.BytesOutPerInstance) == (8, 8, 32, 32)
    assert info.InstancesPerLane == 4 and info.NumVirtualRegisters <= 16


def test_malformed_bytecode_is_rejected_or_optimized_never_fatal():
    """Random bytes and mutated / truncated copies of a real script: a status comes back every time (tools/fuzz_optimizer.cpp runs
    the same under the host sanitizers)."""
    import random
    r = random.Random(1)
    base = oc.fountain()[0].done()
    outcomes = {"ok": 0, "rejected": 0}
    for _ in range(3000):
        if r.random() < 0.3:
            b = bytes(r.randrange(256) for _ in range(r.randrange(2, 80)))
        else:
            m = bytearray(base)
            for _ in range(r.randrange(1, 6)):
                m[r.randrange(len(m))] = r.randrange(256)
            b = bytes(m[:r.randrange(2, len(m))] if r.random() < 0.2 else m)
        try:
            blob = be().optimize_script(b, [(3, 3), (3, 1)])
            outcomes["ok"] += 1
            if blob:
                try:
                    be().inspect_blob(blob)
                except abi.VVMError:
                    pass
        except abi.VVMError:
            outcomes["rejected"] += 1
    assert outcomes["ok"] > 100 and outcomes["rejected"] > 100


def test_golden_blobs():
    """Today's output for a set of scripts (tests/golden/make_optimizer_golden.py wrote them): a change of the optimizer's output
    must be deliberate."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optimizer_fixtures.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 10
    for name, c in cases.items():
        got = be().optimize_script(bytes.fromhex(c["bytecode"]), [tuple(e) for e in c["ext_fns"]])
        assert got.hex() == c["blob"], name


def test_listing_reads_like_the_script():
    from renderer_amd.disasm import listing
    text = listing(optimize(oc.fountain()[0]))
    assert "  1 | cmplt                  o0 = (o6, i7)" in text and "acquireindex           o0, dataset 0" in text
    assert "c2   = constant dword 4" in text and "i7   = dataset 0 float input 7" in text and text.count("outputdata_float") == 3


def test_human_readable_script_of_the_fountain():
    """GenerateHumanReadableScript's layout (VectorVMOptimizer.cpp:104-253): summary, empty constant-table section, one numbered
    line per instruction that was not folded into another (the positions of the folded ones are skipped)."""
    o, meta = oc.fountain()
    text = be().human_readable_script(o.done(), o.ext)
    lines = text.split("\n")
    assert lines[1:4] == ["-------------------------------", "Summary", "-------------------------------"]
    assert "Num Constants: 5" in lines and any(l.startswith("Byte Code (") for l in lines)
    body = [l for l in lines if "\t| " in l]
    assert body[0].endswith("= inputdata_float();") and body[-1].endswith("| done();")
    ops = [l.split("| ")[1] for l in body]
    assert "O[6] = add(I[6], C[0]);" in ops and "O[0] = acquireindex(O[0]);" in ops
    assert sum("mad_mul(" in l for l in ops) == 3 and not any(" mul(" in l for l in ops)      # mad + mul are listed as the merged op only
    numbers = [int(l.split("\t")[0]) for l in body]
    assert numbers == sorted(numbers) and len(numbers) < numbers[-1] + 1      # positions of merged instructions are missing
    assert sum(l.startswith("outputdata_float(0, ") for l in ops) == 8
    assert be().human_readable_script(b"", ()).count("Num Byte Code Ops: 0") == 1
