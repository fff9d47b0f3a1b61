"""Reads a frozen VectorVM blob back into Python values: header, side tables and the optimized instruction stream.

Layout: Runtime/VectorVM/Private/VectorVMBridge.cpp:81-148 (54 packed header bytes, 16-byte aligned sections from offset 64);
instruction encodings: the decoders of Runtime/VectorVM/Private/VectorVMRuntime.cpp cited in ``opcodes.SPECIAL``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Tuple

from .opcodes import NAME, OPS, SPECIAL


@dataclass
class Thawed:
    bytecode: bytes
    max_output_dataset: int
    num_consts_alloced: int
    num_temps: int
    num_noadvance: int
    num_output_instructions: int
    max_ext_fn_registers: int
    num_dummy_regs: int
    max_ext_fn_used: int
    flags: int
    hash_id: int
    const_remap: List[int]
    input_remap: List[int]
    input_dataset_offsets: List[int]
    out_dataset: List[int]
    out_type: List[int]
    out_dst: List[int]
    ext_fns: List[Tuple[int, int]]
    code: List[Tuple] = field(default_factory=list)

    # register index space of the stream: [temps | constants | inputs | input copies | outputs]  (VectorVMRuntime.cpp:345-349)
    def reg(self, r: int) -> str:
        T, C, I = self.num_temps, len(self.const_remap), len(self.input_remap)
        if r == 0xFFFF:
            return "-"
        if r < T:
            return "t%d" % r
        if r < T + C:
            return "c%d" % (r - T)
        if r < T + C + I:
            return "i%d" % (r - T - C)
        if r < T + C + 2 * I:
            return "ibase%d" % (r - T - C - I)
        return "o%d" % (r - T - C - 2 * I)


def _al16(n: int) -> int:
    return (n + 15) & ~15


def thaw(blob: bytes) -> Thawed:
    (nbc, max_ods, nca, T, C, I, nna, nids, O, noi, E, mer, ndr, meu, flags, hash_id) = struct.unpack_from("<IIHIHHHHHHIIIiIQ", blob, 0)
    off = 64
    bc = blob[off:off + nbc]
    off = _al16(off + nbc)
    const_remap = list(struct.unpack_from("<%dH" % C, blob, off)); off = _al16(off + 2 * C)
    input_remap = list(struct.unpack_from("<%dH" % I, blob, off)); off = _al16(off + 2 * I)
    ido = list(struct.unpack_from("<%dH" % (8 * nids), blob, off)); off = _al16(off + 16 * nids)
    ods = list(blob[off:off + O]); off = _al16(off + O)
    ot = list(struct.unpack_from("<%dH" % O, blob, off)); off = _al16(off + 2 * O)
    od = list(struct.unpack_from("<%dH" % O, blob, off)); off = _al16(off + 2 * O)
    ext = [struct.unpack_from("<ii", blob, off + 16 * i + 8) for i in range(E)]
    t = Thawed(bc, max_ods, nca, T, nna, noi, mer, ndr, meu, flags, hash_id, const_remap, input_remap, ido, ods, ot, od, ext)
    t.code = disassemble(t)
    return t


def disassemble(t: Thawed) -> List[Tuple]:
    """[(name, operands...)]: ops -> register names, inputs first; outputs -> (name, dataset, [(src, dst), ...]);
    acquireindex -> (name, predicate, dataset); external call -> (name, fn, [registers]); ids -> (name, r0, r1, dataset)."""
    bc, p, out = t.bytecode, 0, []
    u16 = lambda at: struct.unpack_from("<H", bc, at)[0]
    while p < len(bc):
        name = NAME[bc[p]]
        p += 1
        if name == "done":
            break
        kind = SPECIAL.get(name)
        if kind == "dst":
            out.append((name, t.reg(u16(p)))); p += 2
        elif kind == "pred,ds":
            out.append((name, t.reg(u16(p)), bc[p + 2])); p += 3
        elif kind == "output":
            n, ds = bc[p], bc[p + 1]
            srcs = [t.reg(u16(p + 2 + 2 * i)) for i in range(n)]
            dsts = [t.reg(u16(p + 2 + 2 * n + 2 * i)) for i in range(n)]
            out.append((name, ds, list(zip(srcs, dsts)))); p += 2 + 4 * n
        elif kind == "extfn":
            fn = u16(p)
            n = sum(t.ext_fns[fn])
            out.append((name, fn, [t.reg(u16(p + 2 + 2 * i)) for i in range(n)])); p += 2 + 2 * n
        elif kind == "r0,r1,ds":
            out.append((name, t.reg(u16(p)), t.reg(u16(p + 2)), bc[p + 4])); p += 5
        else:
            _, ni, no = OPS[name]
            out.append((name,) + tuple(t.reg(u16(p + 2 * i)) for i in range(ni + no))); p += 2 * (ni + no)
    return out


def listing(blob: bytes) -> str:
    """The script as text, one instruction per line, with the side tables first: what the reference's GenerateHumanReadableScript
    (VectorVMOptimizer.cpp:220-253) is for."""
    t = thaw(blob)
    kinds = ("float", "int32", "half")
    lines = ["temps %d  constants %d  inputs %d  outputs %d  external functions %d  flags 0x%x  hash 0x%016x"
             % (t.num_temps, len(t.const_remap), len(t.input_remap), len(t.out_dst), len(t.ext_fns), t.flags, t.hash_id)]
    lines += ["c%-3d = constant dword %d" % (i, d) for i, d in enumerate(t.const_remap)]
    for ds in range(len(t.input_dataset_offsets) // 8):
        off = t.input_dataset_offsets[8 * ds:8 * ds + 8]
        for k in range(6):
            for i in range(off[k], off[k + 1]):
                lines.append("i%-3d = dataset %d %s%s input %d" % (i, ds, "no-advance " if k >= 3 else "", kinds[k % 3], t.input_remap[i]))
    for i, d in enumerate(t.out_dst):
        lines.append("o%-3d = dataset %d %s output %s" % (i, t.out_dataset[i], kinds[t.out_type[i]], "-" if d == 0xFFFF else d))
    for i, (a, b) in enumerate(t.ext_fns):
        lines.append("fn%-2d = external function, %d inputs, %d outputs" % (i, a, b))
    for n, c in enumerate(t.code):
        name = c[0]
        if name.startswith("outputdata"):
            body = "dataset %d: %s" % (c[1], ", ".join("%s -> %s" % p for p in c[2]))
        elif name == "acquireindex":
            body = "%s, dataset %d" % (c[1], c[2])
        elif name == "external_func_call":
            body = "fn%d(%s)" % (c[1], ", ".join(c[2]))
        elif name in ("update_id", "acquire_id"):
            body = "%s, %s, dataset %d" % (c[1], c[2], c[3])
        else:
            _, ni, no = OPS.get(name, (0, 0, len(c) - 1))
            body = "%s = (%s)" % (", ".join(c[1 + ni:]), ", ".join(c[1:1 + ni])) if ni else ", ".join(c[1:])
        lines.append("%3d | %-22s %s" % (n, name, body))
    return "\n".join(lines)
