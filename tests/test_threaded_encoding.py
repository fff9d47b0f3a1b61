"""The device form of a program (32 bytes per instruction) against the routine table of the eight-wide kernels' hand-written core:
the offsets the host writes (renderer_amd/csrc/vvm_threaded.h, compiled into the library) must be the ones gen_threaded.py places
the routines at, and the operand bytes must name the linker's operands.  A wrong offset is a jump into the middle of other code."""
import importlib.util
import os
import subprocess
import sys

import pytest

from renderer_amd import abi, opcodes, scripts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "renderer_amd", "csrc")


def generator():
    spec = importlib.util.spec_from_file_location("gen_threaded", os.path.join(CSRC, "gen_threaded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def routine_offsets(g):
    """{(opcode name, operand count): offset of the routine for uniform mask 0}, slot sizes, table size"""
    base, per, total = g.layout()
    first = {}
    for n in range(1, 6):
        for r, row in enumerate(per[n]):
            first[(row[0], n)] = base[n] + r * ((1 << n) - 1) * g.SLOT[n]
    return first, g.SLOT, total


def test_generated_header_is_current():
    assert subprocess.run([sys.executable, os.path.join(CSRC, "gen_threaded.py"), "--check"], stdout=subprocess.PIPE).returncode == 0


def test_routines_fit_their_slots_and_do_not_overlap():
    g = generator()
    first, slot, total = routine_offsets(g)
    spans = sorted((off, off + ((1 << n) - 1) * slot[n]) for (_, n), off in first.items())
    assert spans[0][0] >= 256
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0
    assert spans[-1][1] == total
    assert all(off % 4 == 0 for off, _ in spans)


def many_register_script(n_temps, trig=False):
    """more values alive at once than the sixteen registers of the eight-wide file: such a program runs four instances per lane on
    44 registers of four VGPRs each"""
    from renderer_amd.assembler import FLOAT, Script
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    ts = s.temps(n_temps)
    for i, t in enumerate(ts):
        s.op("mad", x if i == 0 else ts[i - 1], s.const(1), s.const(2), t)
    acc = s.temp()
    s.op("sin" if trig else "abs", ts[0], acc)      # (a sine: the script runs on the kernels with the transcendental functions, 32 registers in the file)
    s.op("add", acc, ts[1], acc)
    for t in ts[2:]:
        nxt = s.temp()
        s.op("mul_add", acc, s.const(3), t, nxt)
        acc = nxt
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0)), (ts[5], s.output(0, FLOAT, 1))])
    return s.build()


def programs():
    out = [("fountain", scripts.fountain().script), ("fountain_unmerged", scripts.fountain(merged=False).script),
           ("transcendental", scripts.transcendental().script)]
    for v in (1, 7, 22, 63):
        out.append(("fountain_v%d" % v, scripts.fountain(variant=v).script))
    out.append(("registers_24", many_register_script(24)))
    out.append(("registers_40", many_register_script(40)))
    out.append(("registers_50", many_register_script(50)))      # (some registers in LDS slots: instructions that touch one have no routine)
    out.append(("registers_24_sin", many_register_script(24, True)))
    out.append(("registers_40_sin", many_register_script(40, True)))
    return out


@pytest.mark.parametrize("narrow", [0, 2])
@pytest.mark.parametrize("name,script", programs(), ids=[p[0] for p in programs()])
def test_device_program_names_the_right_routines(name, script, narrow):
    be = abi.load_hip_backend()
    blob = script.blob
    try:
        dev = be.inspect_device_code(blob, narrow)
    except Exception:
        if narrow == 2:
            pytest.skip("the script does not fit the light kernel")
        raise
    plain = be.inspect_code(blob, narrow)
    g = generator()
    first, slot, total = routine_offsets(g)
    many = name.startswith("registers_")
    file_regs = 10 if narrow == 2 else (32 if name.endswith("_sin") else 44) if many else 16
    vgprs = 4 if many else 8      # VGPRs per register: the instance slots of the kernel the program runs on
    assert len(dev) == len(plain) + 1
    assert [d[:4] for d in dev[:-1]] == [tuple(p) for p in plain]
    assert dev[-1][0] & 0xFF == opcodes.CODE["done"] and (dev[-1][0] >> 8) & 7 == 7 and dev[-1][4] == 0
    threaded = 0
    for d in dev[:-1]:
        w0, w1, w2, w3, routine, opnds, opnd_b, dst = d
        op, nin = w0 & 0xFF, (w0 >> 8) & 7
        o = [w1 & 0xFFFF, w1 >> 16, w2 & 0xFFFF, w2 >> 16, w3 & 0xFFFF, w3 >> 16]
        key = ("VVM_OP_" + opcodes.NAME.get(op, "?"), nin)
        kinds = [(x >> 13, x & 0x1FFF) for x in o]
        eligible = (1 <= nin <= 5 and key in first and kinds[5][0] == 0 and kinds[5][1] < file_regs and
                    all((c == 0 and i < file_regs) or (c == 1 and i < 64) for c, i in kinds[:nin]) and
                    not all(c == 1 for c, _ in kinds[:nin]))
        if not eligible:
            assert routine == 0 and opnds == 0 and opnd_b == 0 and dst == 0
            continue
        threaded += 1
        mask = sum(1 << k for k in range(nin) if kinds[k][0] == 1)
        assert routine == first[key] + mask * slot[nin], (key, mask)
        assert 256 <= routine < total
        byte = [(i if c == 1 else vgprs * i) for c, i in kinds[:nin]] + [0] * (5 - nin)
        assert opnds == byte[0] | byte[2] << 8 | byte[3] << 16 | byte[4] << 24
        assert opnd_b == byte[1]
        assert dst == vgprs * kinds[5][1]
    if name.startswith("fountain"):
        assert threaded >= 4
    if many:
        assert threaded >= 30
