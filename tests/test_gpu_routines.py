"""Every hand-written routine of the kernels that have them against the oracle: each op that has routines, with every combination
of operand kinds (per-lane register / uniform register) a routine exists for, executed from both instruction buffers, on the general
eight-wide kernel and, where the script fits it, on the light one; and the routines for four instance slots per register, on the
four-wide kernel that scripts of more than sixteen registers run on.  Bit for bit (NaN payloads aside)."""
import importlib.util
import os
import struct

import numpy as np
import pytest

from op_cases import operand_streams
from renderer_amd import abi, opcodes
from renderer_amd.assembler import FLOAT, Script
from renderer_amd.opcodes import INT_OPS, OPS
from test_gpu_eight_wide import _check8, _compare8
from vm_harness import run_hip, run_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def routine_ops():
    spec = importlib.util.spec_from_file_location("gen_threaded", os.path.join(ROOT, "renderer_amd", "csrc", "gen_threaded.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    return sorted((row[0][len("VVM_OP_"):], row[1]) for row in g.OPS if row[0].startswith("VVM_OP_"))      # (not the linker's own copy op)


FLOAT_CONSTS = [[1.5, -0.25, 3.0, 0.5, -2.0], [float("nan"), float("inf"), -0.0, 1e-8, 2147483648.0]]
INT_CONSTS = [[3, 0xFFFFFFFB, 7, 0x80000000, 1], [0, 31, 32, 0xFFFFFFFF, 0x7FFFFFFF]]


def constants(name, which):
    if name in INT_OPS:
        vals = struct.pack("<5I", *INT_CONSTS[which])
    else:
        vals = struct.pack("<5f", *FLOAT_CONSTS[which])
    # constant 0: the all-ones predicate of the acquireindex; constants 1..5: operand k as a uniform; 6: an all-ones / zero mask
    return struct.pack("<I", 0xFFFFFFFF) + vals + struct.pack("<I", 0xFFFFFFFF if which == 0 else 0) + bytes(4)


PADS = 18      # values kept alive across the tested instructions: more than the sixteen registers of the eight-wide file


def script_for(name, nin, mask, many=False, trig=False):
    s = Script()
    ones = s.const(0)
    operands, streams = [], 0
    for k in range(nin):
        if (mask >> k) & 1:
            # the first operand of the selects is a mask: a uniform one is all ones or zero
            operands.append(s.const(6 if (k == 0 and "select" in name and not name.startswith("cmp")) else 1 + k))
        else:
            operands.append(s.input(0, FLOAT, streams))
            streams += 1
    s.acquireindex(ones, 0)
    pads = []
    if many:
        first_stream = next(o for k, o in enumerate(operands) if not (mask >> k) & 1)
        pads = s.temps(PADS)
        for i, p in enumerate(pads):
            s.op("bit_xor", first_stream if i == 0 else pads[i - 1], s.const(1 + i % 5), p)
        if trig:      # (an instruction of the kernels with the transcendental functions: the whole script runs there)
            s.op("sin", s.const(1), pads[0])
    t = s.temps(2)
    s.op(name, *operands, t[0])
    s.op(name, *operands, t[1])      # (the second one executes from the other instruction buffer)
    pairs = [(t[0], s.output(0, FLOAT, 0)), (t[1], s.output(0, FLOAT, 1))]
    if many:
        acc = pads[0]
        for p in pads[1:]:
            nxt = s.temp()
            s.op("bit_or", acc, p, nxt)
            acc = nxt
        pairs.append((acc, s.output(0, FLOAT, 2)))
    s.output_data(FLOAT, 0, pairs)
    return s.build(), streams


@pytest.mark.parametrize("name,nin", routine_ops(), ids=[n for n, _ in routine_ops()])
def test_every_routine_of_the_op(name, nin):
    assert OPS[name][1] == nin and OPS[name][2] == 1
    be = abi.load_hip_backend()
    code_of_op = opcodes.CODE[name]
    n = 2300      # two tiles of the eight-wide kernels, the second one partial
    for mask in range((1 << nin) - 1):
        built, streams = script_for(name, nin, mask)
        dev = be.inspect_device_code(built.blob, 0)
        with_routine = [d for d in dev if (d[0] & 0xFF) == code_of_op and ((d[0] >> 8) & 7) == nin and d[4] != 0]
        assert len(with_routine) == 2, "%s mask %d: the general program does not run the op in a routine twice" % (name, mask)
        assert len({d[4] for d in with_routine}) == 1
        ins = operand_streams(name, streams, n, seed=11 + mask)
        for which in (0, 1):
            _check8(built, [constants(name, which)], ins, [f32, f32], n)


@pytest.mark.parametrize("trig", [False, True], ids=["plain", "with_sin"])
@pytest.mark.parametrize("name,nin", routine_ops(), ids=[n for n, _ in routine_ops()])
def test_every_routine_of_the_op_four_slots(name, nin, trig):
    """the same with eighteen more values alive across the tested instructions: the script runs four instances per lane on the
    kernel of many registers, whose routines work on four VGPRs per register; with a sine in the script, on the kernel of many
    registers that has the transcendental functions"""
    be = abi.load_hip_backend()
    code_of_op = opcodes.CODE[name]
    n = 1300      # two tiles of the four-wide kernels, the second one partial
    for mask in range((1 << nin) - 1):
        built, streams = script_for(name, nin, mask, many=True, trig=trig)
        dev = be.inspect_device_code(built.blob, 0)
        with_routine = [d for d in dev if (d[0] & 0xFF) == code_of_op and ((d[0] >> 8) & 7) == nin and d[4] != 0]
        assert len(with_routine) >= 2, "%s mask %d: the program does not run the op in a routine twice" % (name, mask)      # (the fill uses two of the ops itself)
        assert max(d[7] for d in with_routine) >= 4 * 16, "the results lie in the registers the eight-wide file has as well"
        ins = operand_streams(name, streams, n, seed=11 + mask)
        for which in (0, 1):
            consts = [constants(name, which)]
            ref, ref_n, _ = run_oracle(built, consts, ins, [f32] * 3, n)
            got, got_n, info = run_hip(built, consts, ins, [f32] * 3, n)
            assert info.InstancesPerLane == 4 and info.NumVirtualRegisters > 16 and info.LdsSpillRegisters == 0
            assert bool(info.Flags & 0x10000) == trig      # (the script needs the kernels with the transcendental functions)
            _compare8(got, got_n, ref, ref_n, n, {})
