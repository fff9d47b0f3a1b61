"""The opcode numbering and operand counts of this project against the reference's own list, held as data in
tests/golden/opcode_table.json (the rows of VVM_OP_XM_LIST, VectorVM.h:51-233, read from the source text by
tests/golden/make_opcode_table.py): EVectorVMOp is generated from that list in order, and the decoders take the operand counts
from it (VectorVMRuntime.cpp:1558-1904)."""
import json
import os
import re

from renderer_amd import opcodes

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "opcode_table.json")) as f:
    TABLE = json.load(f)["ops"]


# ordinary ops this project's operand table leaves out: noise, fasi and iasf never reach the optimized format
# (VectorVMOptimizer.cpp:3044-3046, 3112-3113); exec_index and exec_indexf take no inputs and are assembled by their own helpers
NOT_IN_OPS = {"noise", "fasi", "iasf", "exec_index", "exec_indexf"}


def test_the_list_is_complete_and_in_order():
    assert len(TABLE) == 179 == opcodes.NUM_OPCODES
    assert [o["value"] for o in TABLE] == list(range(179))


def test_numeric_values_of_the_python_table():
    assert {o["name"]: o["value"] for o in TABLE} == opcodes.CODE


def test_numeric_values_in_the_c_header():
    text = open(os.path.join(HERE, "..", "include", "vvm_abi.h")).read()
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"\bVVM_OP_(\w+)\s*=\s*(\d+)", text)}
    assert enum == {o["name"]: o["value"] for o in TABLE}
    assert int(re.search(r"VVM_NUM_OPCODES\s*=\s*(\d+)", text).group(1)) == len(TABLE)


def test_operand_counts_of_the_ordinary_ops():
    """Every op of category Op decodes as num_inputs + num_outputs register indices; this project's table says the same."""
    for o in TABLE:
        if o["category"] != "Op":
            assert o["name"] not in opcodes.OPS, o["name"]
            continue
        if o["name"] in NOT_IN_OPS:
            continue
        code, nin, nout = opcodes.OPS[o["name"]]
        assert (code, nin, nout) == (o["value"], o["num_inputs"], o["num_outputs"]), o["name"]


def test_the_linker_decodes_by_the_same_counts():
    """One instruction of every ordinary op, assembled with the reference's operand counts, thaws, links and reports that one
    instruction (the linker's own table, vvm_program.cpp, drives its decoder: a wrong count would misread the stream)."""
    from renderer_amd import abi
    from renderer_amd.assembler import FLOAT, Script
    be = abi.load_hip_backend()
    for o in TABLE:
        if o["category"] != "Op" or o["name"] in NOT_IN_OPS:
            continue
        s = Script()
        ins = [s.input(0, FLOAT, k) for k in range(o["num_inputs"])]
        outs = s.temps(o["num_outputs"])
        s.acquireindex(s.const(0), 0)
        s.op(o["name"], *ins, *outs)
        s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, k)) for k, t in enumerate(outs)])
        info = be.inspect_blob(s.build().blob)
        assert info.NumInputBuffers == o["num_inputs"] and info.NumOutputBuffers >= 1, o["name"]
