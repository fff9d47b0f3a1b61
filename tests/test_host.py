"""CPU tests of the host side: blob format, the linker behind vvm_inspect_blob / vvm_inspect_code, the exported C ABI.

No compute call is made: libvvm_hip.so is loaded (it links against the HIP runtime, which loads without a GPU) and only its
device-free entry points are used.
"""
import ctypes as C
import os
import re
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from op_cases import ALL_ONES, single_op_script
from renderer_amd import abi, opcodes, scripts
from renderer_amd.assembler import FLOAT, HALF, INT32, Script

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# device instruction fields (renderer_amd/csrc/vvm_device.h)
OPC_VREG, OPC_UNIF, OPC_CONST, OPC_NOADV, OPC_NOADV_HALF, OPC_NONE = 0, 1, 2, 4, 6, 7
DOP_LOADN16, DOP_LOADN, DOP_ACQUIRE, DOP_OUT32, DOP_OUT16, DOP_OUT16F16, DOP_OUT32F16 = 190, 191, 192, 193, 194, 195, 196


def be():
    return abi.load_hip_backend()


def decode(ins):
    w0, w1, w2, w3 = ins
    return {"op": w0 & 255, "nin": (w0 >> 8) & 7, "aux": (w0 >> 16) & 255, "aux2": w0 >> 24,
            "o": [w1 & 0xFFFF, w1 >> 16, w2 & 0xFFFF, w2 >> 16, w3 & 0xFFFF, w3 >> 16]}


def cls(o):
    return o >> 13


def idx(o):
    return o & 0x1FFF


# ---- exported ABI -----------------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_entry_point():
    header = open(os.path.join(ROOT, "include", "vvm_abi.h")).read()
    names = sorted(set(re.findall(r"VVM_API\s+[\w\s\*]+?\b(vvm_\w+)\s*\(", header)))
    assert len(names) == header.count("\nVVM_API ") and len(names) >= 20
    lib = be().lib
    for n in names:
        getattr(lib, n)


def test_ctypes_mirrors_match_the_c_layout():
    """sizeof / offsetof of the C structs, printed by a program compiled from the header, against the ctypes mirrors."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "vvm_abi.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu\n", sizeof(vvm_dataset_meta), sizeof(vvm_exec_context), sizeof(vvm_exec_options),
           sizeof(vvm_ext_function), sizeof(vvm_extfn_context), sizeof(vvm_program_info));
    printf("%zu %zu %zu %zu\n", offsetof(vvm_dataset_meta, InstanceOffset), offsetof(vvm_dataset_meta, DataSetAccessIndex),
           offsetof(vvm_exec_context, NumInstances), offsetof(vvm_exec_context, ConstantTableCount));
    return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    sizes = [C.sizeof(t) for t in (abi.DataSetMeta, abi.ExecContext, abi.ExecOptions, abi.ExtFunction, abi.ExtFnContext, abi.ProgramInfo)]
    assert [int(x) for x in out[:6]] == sizes
    offs = [abi.DataSetMeta.InstanceOffset.offset, abi.DataSetMeta.DataSetAccessIndex.offset,
            abi.ExecContext.NumInstances.offset, abi.ExecContext.ConstantTableCount.offset]
    assert [int(x) for x in out[6:]] == offs


def test_opcode_numbers_match_the_header():
    header = open(os.path.join(ROOT, "include", "vvm_abi.h")).read()
    found = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"VVM_OP_(\w+)\s*=\s*(\d+)", header))
    assert len(found) == opcodes.NUM_OPCODES == 179
    for name, code in opcodes.CODE.items():
        assert found[name] == code, name
    # spot values from the reference's list (VectorVM.h:50-233): the list starts done, add, sub, mul and ends with the two
    # *_from_half outputs
    assert (opcodes.CODE["done"], opcodes.CODE["add"], opcodes.CODE["acquireindex"]) == (0, 1, 86)
    assert opcodes.CODE["outputdata_half_from_half"] == 178


# ---- blob ---------------------------------------------------------------------------------------------------------------------
def test_blob_layout():
    """54 packed header bytes, sections 16-aligned from byte 64 (VectorVMBridge.cpp:81-148)."""
    w = scripts.fountain()
    b = w.script.blob
    num_bytecode, max_out_ds = struct.unpack_from("<II", b, 0)
    num_temps, = struct.unpack_from("<I", b, 10)
    num_consts, num_inputs = struct.unpack_from("<HH", b, 14)
    assert num_bytecode == len(w.script.bytecode) and max_out_ds == 1
    assert num_temps == w.script.num_temps and num_consts == 5 and num_inputs == 8
    assert b[64:64 + num_bytecode] == w.script.bytecode
    assert w.script.bytecode[-9:] == bytes(9)                  # eight `done` + one pad byte (VectorVMOptimizer.cpp:3124-3155)
    assert len(b) % 16 == 0
    info = be().inspect_blob(b)
    assert (info.NumTempRegisters, info.NumConstBuffers, info.NumInputBuffers) == (num_temps, 5, 8)
    assert (info.BytesInPerInstance, info.BytesOutPerInstance) == (32, 32)


def test_operand_sizes_of_the_optimized_encoding():
    """acquireindex 1+3 bytes, outputs 1+2+4n, ops 1+2(in+out) (VectorVMRuntime.cpp:826-828, 555-569, 1558-1904)."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    t = s.temp()
    s.op("mad", s.input(0, FLOAT, 0), s.const(1), s.const(2), t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
    built = s.build()
    assert len(built.bytecode) == (1 + 3) + (1 + 8) + (1 + 2 + 4) + 9
    assert built.bytecode[0] == opcodes.CODE["acquireindex"] and built.bytecode[4] == opcodes.CODE["mad"]


@pytest.mark.parametrize("mutate,code", [
    (lambda b: b"", abi.VVM_E_BAD_BLOB),
    (lambda b: b[:40], abi.VVM_E_BAD_BLOB),
    (lambda b: b[:len(b) - 32], abi.VVM_E_BAD_BLOB),
    (lambda b: b[:64] + bytes([250]) + b[65:], abi.VVM_E_UNSUPPORTED_OP),              # not an opcode
    (lambda b: b[:64] + bytes([opcodes.CODE["inputdata_float"]]) + b[65:], abi.VVM_E_UNSUPPORTED_OP),   # never emitted in this format
])
def test_bad_blobs_are_rejected(mutate, code):
    w = scripts.fountain()
    with pytest.raises(abi.VVMError) as e:
        be().inspect_blob(mutate(w.script.blob))
    assert e.value.code == code


def test_register_index_out_of_range_is_rejected():
    w = scripts.fountain()
    b = bytearray(w.script.blob)
    # first instruction is `add in, const, temp`: point its first operand far past the register space
    assert b[64] == opcodes.CODE["add"]
    b[65:67] = struct.pack("<H", 0x7000)
    with pytest.raises(abi.VVMError) as e:
        be().inspect_blob(bytes(b))
    assert e.value.code == abi.VVM_E_BAD_BLOB


# ---- linker ---------------------------------------------------------------------------------------------------------------------
def test_fountain_links_for_the_light_kernel_with_two_outputs_in_registers():
    """Eight outputs: six are staged, two stay in registers 8 and 9, which the instructions that make their values write directly
    (both come after the acquireindex), so the program has no copy and two output instructions."""
    w = scripts.fountain()
    code = [decode(i) for i in be().inspect_code(w.script.blob, 2)]
    ops = [c["op"] for c in code]
    assert ops[0] == DOP_LOADN and code[0]["aux"] == 8 and ops.count(DOP_OUT32) == 2 and 189 not in ops and 197 not in ops
    at = ops.index(DOP_ACQUIRE)
    writers = {idx(c["o"][5]): k for k, c in enumerate(code) if 0 < c["op"] < 190}
    assert writers[8] > at and writers[9] > at and max(writers) == 9
    assert sorted(r for r in writers if r < 8) == sorted(set(r for r in writers if r < 8))
    staged = [o for c in code if c["op"] == DOP_OUT32 for o in c["o"][:c["aux"]]]
    assert len(staged) == 6 and all(idx(o) < 8 for o in staged)


def test_light_link_copies_values_that_nothing_after_the_acquireindex_makes():
    s = Script()
    xs = [s.input(0, FLOAT, i) for i in range(8)]
    keep = s.temp(); s.op("cmplt", xs[0], xs[1], keep)
    s.acquireindex(keep, 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, i)) for i, x in enumerate(xs)])
    code = [decode(i) for i in be().inspect_code(s.build().blob, 2)]
    copies = [c for c in code if c["op"] == 189]
    assert [idx(c["o"][5]) for c in copies] == [8, 9] and [idx(c["o"][0]) for c in copies] == [6, 7]
    # nine outputs need a second group of staging slots: no light program
    s = Script()
    x = s.input(0, FLOAT, 0)
    s.acquireindex(s.const(0), 0)
    s.output_data(FLOAT, 0, [(x, s.output(0, FLOAT, i)) for i in range(9)])
    with pytest.raises(abi.VVMError):
        be().inspect_code(s.build().blob, 2)


def test_fountain_links_to_the_expected_program():
    w = scripts.fountain()
    code = [decode(i) for i in be().inspect_code(w.script.blob)]
    ops = [c["op"] for c in code]
    assert ops[0] == DOP_LOADN and code[0]["aux"] == 8                     # all eight input streams in one load instruction
    assert ops.count(DOP_ACQUIRE) == 1 and ops.count(DOP_OUT32) == 3       # 8 outputs as 3 + 3 + 2 pairs
    assert ops[-2:] == [0, 0]                                              # done plus the instruction fetched ahead of it
    body = [c for c in code if 0 < c["op"] < 190]
    # (the cmplt that makes the keep mask has no other reader: it is folded into the acquireindex, aux2 = the compare)
    assert [opcodes.NAME[c["op"]] for c in body] == ["add"] + ["mad_mul", "mad"] * 3
    acq = [c for c in code if c["op"] == DOP_ACQUIRE][0]
    assert acq["aux2"] == 1 and cls(acq["o"][0]) == OPC_VREG and cls(acq["o"][1]) == OPC_VREG          # VVM_ACQ_LT over two per-lane registers
    info = be().inspect_blob(w.script.blob)
    assert info.InstancesPerLane == 4 and info.LdsSpillRegisters == 0
    assert info.NumVirtualRegisters == 8           # the eight streams; every result takes over the register of a stream read for the last time
    # constants are uniform-register operands, streams and temps per-lane registers
    for c in body:
        for o in c["o"][:c["nin"]]:
            assert cls(o) in (OPC_VREG, OPC_UNIF)
        assert cls(c["o"][5]) == OPC_VREG


def test_two_output_ops_are_split():
    for name, first_aux in (("sin_cos", None), ("mul_2x", None), ("b2i_2x", None), ("random_2x", None)):
        script, n_in, n_out = single_op_script(name)
        code = [decode(i) for i in be().inspect_code(script.blob)]
        body = [c for c in code if 0 < c["op"] < 190]
        assert len(body) == 2, name
        assert body[0]["o"][5] != body[1]["o"][5]
        single = {"sin_cos": "sin_cos", "mul_2x": "mul", "b2i_2x": "b2i", "random_2x": "random"}[name]
        assert all(opcodes.NAME[c["op"]] == single for c in body)
        if name == "sin_cos":
            assert sorted(c["aux"] for c in body) == [0, 1]                # sine / cosine selector
        if name == "random_2x":
            assert sorted(c["aux2"] for c in body) == [0, 1]               # consecutive draw ordinals


def test_split_outputs_that_alias_inputs_go_through_scratch():
    """curl_noise3(x, y, z) -> (x, y, z): every output register is also an input; the reference reads all inputs first."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    p = s.temps(3)
    for i in range(3):
        s.op("mul", s.input(0, FLOAT, i), s.const(1), p[i])
    f = s.ext_fn(3, 3)
    s.call(f, p, p)
    s.output_data(FLOAT, 0, [(p[i], s.output(0, FLOAT, i)) for i in range(3)])
    built = s.build()
    code = [decode(i) for i in be().inspect_code(built.blob)]
    calls = [c for c in code if c["op"] == opcodes.CODE["external_func_call"]]
    assert len(calls) == 3 and sorted(c["aux2"] for c in calls) == [0, 1, 2]
    srcs = set(calls[0]["o"][:3])
    written = [c["o"][5] for c in calls]
    assert written[0] not in srcs and written[1] not in srcs and written[2] in srcs     # two scratch registers, then the last in place
    copies = [c for c in code if c["op"] == opcodes.CODE["bit_or"]]
    assert len(copies) == 2 and {c["o"][5] for c in copies} | {written[2]} == srcs


def test_unused_ext_fn_outputs_are_dropped():
    s = Script()
    s.acquireindex(s.const(0), 0)
    from renderer_amd.assembler import INVALID
    x = [s.input(0, FLOAT, i) for i in range(3)]
    t = s.temp()
    f = s.ext_fn(3, 3)
    s.call(f, x, [INVALID, t, INVALID])
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
    code = [decode(i) for i in be().inspect_code(s.build().blob)]
    calls = [c for c in code if c["op"] == opcodes.CODE["external_func_call"]]
    assert len(calls) == 1 and calls[0]["aux2"] == 1


def test_many_registers_spill_to_lds_then_narrow_kernel():
    def chain(n_temps):
        s = Script()
        s.acquireindex(s.const(0), 0)
        x = s.input(0, FLOAT, 0)
        ts = s.temps(n_temps)
        for i, t in enumerate(ts):
            s.op("add", x if i == 0 else ts[i - 1], s.const(1), t)
        acc = s.temp()
        s.op("add", ts[0], ts[1], acc)
        for t in ts[2:]:                                   # keeps every temp alive to the end
            nxt = s.temp()
            s.op("add", acc, t, nxt)
            acc = nxt
        s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0))])
        return s.build()
    small = be().inspect_blob(chain(10).blob)
    assert small.InstancesPerLane == 4 and small.LdsSpillRegisters == 0
    mid = be().inspect_blob(chain(22).blob)                 # more than 16: the four-wide kernel with 44 resident registers
    assert mid.InstancesPerLane == 4 and mid.LdsSpillRegisters == 0 and mid.NumVirtualRegisters == 22
    more = be().inspect_blob(chain(36).blob)
    assert more.InstancesPerLane == 4 and more.LdsSpillRegisters == 0
    most = be().inspect_blob(chain(48).blob)                # more than 44: the rest in LDS slots
    assert most.InstancesPerLane == 4 and most.LdsSpillRegisters == 4 and most.NumVirtualRegisters == 48
    assert be().inspect_blob(chain(55).blob).InstancesPerLane == 4
    assert be().inspect_blob(chain(56).blob).InstancesPerLane == 1
    big = be().inspect_blob(chain(60).blob)
    assert big.InstancesPerLane == 1
    with pytest.raises(abi.VVMError) as e:
        be().inspect_blob(chain(400).blob)
    assert e.value.code == abi.VVM_E_TOO_LARGE


def test_registers_are_packed_by_live_range():
    """12 streams summed through 11 fresh temps: every sum takes over the register of a stream or sum read for the last time."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    xs = [s.input(0, FLOAT, i) for i in range(12)]
    acc = xs[0]
    for x in xs[1:]:
        t = s.temp()
        s.op("add", acc, x, t)
        acc = t
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0))])
    built = s.build()
    info = be().inspect_blob(built.blob)
    assert info.NumVirtualRegisters == 12 and info.InstancesPerLane == 4 and info.LdsSpillRegisters == 0
    code = [decode(i) for i in be().inspect_code(built.blob)]
    adds = [c for c in code if c["op"] == opcodes.CODE["add"]]
    assert all(c["o"][5] in c["o"][:2] for c in adds)                            # in place
    # a value that is still needed keeps its register: b = a + 1; c = b + 1; out = b + c  (a's register is free for b, b's not for c)
    s = Script()
    s.acquireindex(s.const(0), 0)
    a = s.input(0, FLOAT, 0)
    b, c, d = s.temps(3)
    s.op("add", a, s.const(1), b)
    s.op("add", b, s.const(1), c)
    s.op("add", b, c, d)
    s.output_data(FLOAT, 0, [(d, s.output(0, FLOAT, 0))])
    code = [decode(i) for i in be().inspect_code(s.build().blob)]
    adds = [c for c in code if c["op"] == opcodes.CODE["add"]]
    assert adds[0]["o"][5] == adds[0]["o"][0] and adds[1]["o"][5] != adds[1]["o"][0] and adds[2]["o"][5] in adds[2]["o"][:2]


def test_instructions_nothing_reads_are_left_out():
    """A chain whose result reaches no output, acquireindex or ID table is not linked, and the stream only it reads is not loaded."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    a, b, unused = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1), s.input(0, FLOAT, 2)
    d1, d2, r = s.temps(3)
    s.op("sin", unused, d1)
    s.op("mul", d1, a, d2)              # dead: nothing reads d2
    s.op("add", a, b, r)
    s.output_data(FLOAT, 0, [(r, s.output(0, FLOAT, 0))])
    built = s.build()
    code = [decode(i) for i in be().inspect_code(built.blob)]
    names = [opcodes.NAME[c["op"]] for c in code if 0 < c["op"] < 190]
    assert names == ["add"]
    assert code[0]["op"] == DOP_LOADN and code[0]["aux"] == 2          # two streams loaded, not three
    info = be().inspect_blob(built.blob)
    assert info.NumVirtualRegisters == 2 and info.InstancesPerLane == 4


def test_a_random_draw_keeps_its_number_when_earlier_draws_are_left_out():
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    dead, kept = s.temps(2)
    s.op("random", x, dead)             # nothing reads it
    s.op("random", x, kept)
    s.output_data(FLOAT, 0, [(kept, s.output(0, FLOAT, 0))])
    code = [decode(i) for i in be().inspect_code(s.build().blob)]
    draws = [c for c in code if c["op"] == opcodes.CODE["random"]]
    assert len(draws) == 1 and draws[0]["aux2"] == 1


def test_half_and_noadvance_inputs():
    s = Script()
    s.acquireindex(s.const(0), 0)
    h = s.input(0, HALF, 0)
    na = s.input(0, FLOAT, 0, noadvance=True)
    f = s.temp(); s.op("half_to_float", h, f)
    g = s.temp(); s.op("mul", f, na, g)
    s.output_data(FLOAT, 0, [(g, s.output(0, FLOAT, 0))])
    code = [decode(i) for i in be().inspect_code(s.build().blob)]
    assert code[0]["op"] == DOP_LOADN16 and code[0]["aux"] == 1
    mul = [c for c in code if c["op"] == opcodes.CODE["mul"]][0]
    assert cls(mul["o"][1]) == OPC_NOADV


def test_more_than_64_constants_fall_back_to_the_table():
    s = Script()
    s.acquireindex(s.const(0), 0)
    acc = s.input(0, FLOAT, 0)
    for i in range(1, 70):
        t = s.temp()
        s.op("add", acc, s.const(i), t)
        acc = t
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0))])
    code = [decode(i) for i in be().inspect_code(s.build().blob)]
    adds = [c for c in code if c["op"] == opcodes.CODE["add"]]
    classes = [cls(c["o"][1]) for c in adds]
    assert classes.count(OPC_UNIF) == 63 and classes.count(OPC_CONST) == 6        # remapped constants 1..63 and 64..69


def test_outputs_are_grouped_by_staging_slots():
    """20 outputs: no output instruction straddles a group of eight list entries, and a flush separates the groups."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    ts = s.temps(20)
    for i, t in enumerate(ts):
        s.op("add", x if i == 0 else ts[i - 1], s.const(1), t)
    s.output_data(FLOAT, 0, [(ts[i], s.output(0, FLOAT, i)) for i in range(20)])
    code = [decode(i) for i in be().inspect_code(s.build().blob)]
    tail = [c for c in code if c["op"] >= DOP_OUT32]
    DOP_FLUSH = 197
    assert [c["op"] for c in tail] == [DOP_OUT32] * 3 + [DOP_FLUSH] + [DOP_OUT32] * 3 + [DOP_FLUSH] + [DOP_OUT32] * 2
    outs = [c for c in tail if c["op"] == DOP_OUT32]
    assert [c["aux"] for c in outs] == [3, 3, 2, 3, 3, 2, 3, 1]                    # pairs per instruction
    first = [c["o"][4] | (c["o"][5] << 16) for c in outs]
    assert first == [0, 3, 6, 8, 11, 14, 16, 19]                                  # list index of each instruction's first pair
    assert all(f // 8 == (f + c["aux"] - 1) // 8 for f, c in zip(first, outs))


def test_id_instructions_link():
    """update_id takes the last entry of its group of the output list; acquire_id becomes one instruction per destination."""
    import id_cases
    code = [decode(i) for i in be().inspect_code(id_cases.update_script(extra_outputs=9).blob)]
    DOP_FLUSH, DOP_UPDATE_ID = 197, 198
    tail = [c for c in code if c["op"] >= DOP_OUT32]
    assert [c["op"] for c in tail] == [DOP_UPDATE_ID] + [DOP_OUT32] * 3 + [DOP_FLUSH] + [DOP_OUT32] * 2
    assert tail[0]["o"][4] == 7                                                   # list entry 7 = last of group 0
    assert [c["o"][4] for c in tail if c["op"] == DOP_OUT32] == [0, 3, 6, 8, 11]      # seven outputs in group 0, the rest in group 1
    assert [c["aux"] for c in tail if c["op"] == DOP_OUT32] == [3, 3, 1, 3, 1]
    info = be().inspect_blob(id_cases.update_script().blob)
    assert info.InstancesPerLane == 4
    code = [decode(i) for i in be().inspect_code(id_cases.spawn_script().blob)]
    acq = [c for c in code if c["op"] == opcodes.CODE["acquire_id"]]
    assert sorted(c["aux"] for c in acq) == [0, 1] and all(c["nin"] == 0 for c in acq)


def test_two_acquire_datasets_link_to_one_pass_with_two_chains():
    """Both acquireindex in one program (aux = chain), every output instruction names the chain of its dataset."""
    import dataset_cases
    blob = dataset_cases.two_dataset_script().blob
    info = be().inspect_blob(blob)
    code = [decode(i) for i in be().inspect_code(blob)]
    acq = [c for c in code if c["op"] == DOP_ACQUIRE]
    outs = [c for c in code if c["op"] == DOP_OUT32]
    assert [c["aux"] for c in acq] == [0, 1]
    assert sorted(c["aux2"] for c in outs) == [0, 1, 1] and sum(c["aux"] for c in outs) == 4      # two outputs of dataset 0, a float and an int of dataset 1
    assert info.InstancesPerLane == 4


def test_two_acquire_datasets_link_to_one_program_per_dataset_without_chains(monkeypatch):
    """VVM_NO_CHAINS (read at every link): one program per dataset, as for scripts whose kernels have no second chain."""
    import dataset_cases
    monkeypatch.setenv("VVM_NO_CHAINS", "1")
    blob = dataset_cases.two_dataset_script().blob
    code = [decode(i) for i in be().inspect_code(blob)]                           # the first pass: dataset 0 only
    acq = [c for c in code if c["op"] == DOP_ACQUIRE]
    outs = [c for c in code if c["op"] == DOP_OUT32]
    assert len(acq) == 1 and acq[0]["aux"] == 0
    assert len(outs) == 1 and outs[0]["aux"] == 2 and outs[0]["aux2"] == 0


def test_unsupported_features_are_named():
    s = Script()
    t0, t1 = s.temps(2)
    s.acquire_id(t0, t1, 0)
    s.acquireindex(s.const(0), 0)
    s.update_id(t0, t1, 0)
    s.output_data(INT32, 0, [(t0, s.output(0, INT32, 0))])
    be().inspect_blob(s.build().blob)          # acquire_id, then update_id on one dataset: accepted (such an Exec runs chunk after chunk)
    s = Script()
    pid = s.input(0, INT32, 0)
    s.acquireindex(s.const(0), 0)
    s.update_id(pid, pid, 0)
    t0, t1 = s.temps(2)
    s.acquire_id(t0, t1, 0)                   # the other way round
    s.output_data(INT32, 0, [(t0, s.output(0, INT32, 0))])
    with pytest.raises(abi.VVMError) as e:
        be().inspect_blob(s.build().blob)
    assert e.value.code == abi.VVM_E_UNSUPPORTED_OP and "acquire_id" in str(e.value)
    s = Script()
    pid = s.input(0, INT32, 0)
    s.update_id(pid, pid, 0)                                                      # no acquireindex for the dataset
    with pytest.raises(abi.VVMError) as e:
        be().inspect_blob(s.build().blob)
    assert e.value.code == abi.VVM_E_UNSUPPORTED_OP and "acquireindex" in str(e.value)
    s = Script()
    s.acquireindex(s.const(0), 0)
    f = s.ext_fn(2, 1)
    t = s.temp()
    s.call(f, [s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)], [t])
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))])
    with pytest.raises(abi.VVMError) as e:
        be().inspect_blob(s.build().blob)
    assert e.value.code == abi.VVM_E_UNSUPPORTED_OP


def test_every_plain_op_links():
    for name in sorted(opcodes.OPS):
        script, _, _ = single_op_script(name)
        code = be().inspect_code(script.blob)
        assert len(code) >= 5, name


# ---- sharding arithmetic -----------------------------------------------------------------------------------------------------
def test_shard_ranges_and_offsets():
    from renderer_amd.runtime import global_offsets, shard_range
    for n in (0, 1, 7, 8, 9, 1000, 2_000_000_000):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges if hi > lo or n == 0) <= (n + world - 1) // world
    assert global_offsets([5, 0, 7, 1]) == [0, 5, 5, 12]


def test_product_path_has_no_cpu_fallback():
    """The package never imports the oracle, and loading the backend fails loudly when the library is absent."""
    pkg = os.path.join(ROOT, "renderer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in text.replace("oracle_binding", "").lower() or fn == "abi.py" and "vvm_oracle" not in text, fn
    real_path, real_backend = abi._HIP_LIB_PATH, abi._hip_backend
    try:
        abi._HIP_LIB_PATH, abi._hip_backend = "/nonexistent/libvvm_hip.so", None
        with pytest.raises(RuntimeError) as e:
            abi.load_hip_backend()
        assert "no CPU fallback" in str(e.value)
    finally:
        abi._HIP_LIB_PATH, abi._hip_backend = real_path, real_backend


def test_shard_offsets_arithmetic():
    """vvm_shard_offsets: exclusive prefix over ranks and totals, per dataset (no device involved)."""
    be = abi.load_hip_backend()
    counts = [5, 0, 7,   0, 2, 1,   9, 9, 0]          # three ranks, three datasets
    for rank in range(3):
        r = be.shard_offsets(rank, 3, 3, counts)
        assert r.NumDataSets == 3
        for d in range(3):
            assert r.LocalSurvivors[d] == counts[rank * 3 + d]
            assert r.GlobalOffset[d] == sum(counts[q * 3 + d] for q in range(rank))
            assert r.GlobalSurvivors[d] == sum(counts[q * 3 + d] for q in range(3))
    with pytest.raises(abi.VVMError):
        be.shard_offsets(3, 3, 3, counts)
    with pytest.raises(abi.VVMError):
        be.shard_offsets(0, 1, 9, [0] * 9)
    with pytest.raises(abi.VVMError):
        be.shard_offsets(0, 2, 1, [1, -1])
