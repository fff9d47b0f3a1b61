"""Script assembler: builds *optimized-format* VectorVM bytecode and the frozen runtime blob.

The reference produces this format with its 18-step optimizer
(Runtime/VectorVM/Private/VectorVMOptimizer.cpp:700-3160) and serialises it with
FreezeOptimizerContext (Runtime/VectorVM/Private/VectorVMBridge.cpp:151-178).  Nothing in the
reference tree emits scripts (the Niagara compiler is not part of it), so tests and benchmarks
assemble them here.  What is reproduced from the reference is the *format* the runtime consumes:

* register index space  [temps | consts | inputs | input base copies | outputs]
  (VectorVMRuntime.cpp:345-349, writer VectorVMOptimizer.cpp:534-564);
* instruction encodings (decoders cited per op in ``opcodes.SPECIAL``);
* side tables ConstRemap / InputRemapTable / InputDataSetOffsets / OutputRemap*
  (VectorVMOptimizer.cpp:2525-2774, consumed by VectorVMRuntime.cpp:485-553 and :418-441);
* stream terminator of 8 ``done`` bytes plus one pad byte (VectorVMOptimizer.cpp:3124-3155);
* blob layout: 54 packed header bytes, sections aligned to 16 from offset 64
  (VectorVMBridge.cpp:81-148).

Temps are virtual while a script is being written; ``build()`` assigns physical temp registers by
linear scan over live ranges and can alias a temp onto the output buffer it is finally written to
(the reference's steps 16-17, VectorVMOptimizer.cpp:2776-2978), which produces output
instructions with ``src == dst`` (VectorVMRuntime.cpp:756-757).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import opcodes
from .opcodes import CODE, OPS

FLOAT, INT32, HALF = 0, 1, 2
_TYPE_NAMES = {FLOAT: "float", INT32: "int32", HALF: "half"}


@dataclass(frozen=True)
class Operand:
    kind: str          # 't' temp, 'c' const, 'i' input, 'o' output, 'x' invalid (0xFFFF)
    a: int = 0         # temp id | const dword index | dataset | dataset
    b: int = 0         # input/output: type
    c: int = 0         # input/output: register index within its type
    noadvance: bool = False


INVALID = Operand("x")


@dataclass
class _Ins:
    name: str
    ins: Tuple[Operand, ...] = ()
    outs: Tuple[Operand, ...] = ()
    dataset: int = 0
    pairs: Tuple[Tuple[Operand, Operand], ...] = ()   # output instructions
    fn_idx: int = 0


@dataclass
class BuiltScript:
    blob: bytes
    bytecode: bytes
    num_temps: int
    const_remap: List[int]
    input_remap: List[int]
    input_dataset_offsets: List[int]
    output_dataset: List[int]
    output_type: List[int]
    output_dst: List[int]
    num_ext_fns: int
    ext_fn_io: List[Tuple[int, int]]
    flags: int
    listing: List[str] = field(default_factory=list)
    bytes_in_per_instance: int = 0
    bytes_out_per_instance: int = 0


class Script:
    """Builder for one script.  Typical use::

        s = Script()
        age = s.input(0, FLOAT, 6); dt = s.const(0)
        t = s.temp(); s.op("add", age, dt, t)
        ...
        built = s.build()
    """

    def __init__(self):
        self._ins: List[_Ins] = []
        self._ntemps = 0
        self._ext: List[Tuple[int, int]] = []

    # ---- operands -------------------------------------------------------------------------------
    def temp(self) -> Operand:
        self._ntemps += 1
        return Operand("t", self._ntemps - 1)

    def temps(self, n: int) -> List[Operand]:
        return [self.temp() for _ in range(n)]

    @staticmethod
    def const(dword_index: int) -> Operand:
        """Constant at 32-bit index ``dword_index`` of the concatenated constant tables."""
        return Operand("c", dword_index)

    @staticmethod
    def input(dataset: int, typ: int, index: int, noadvance: bool = False) -> Operand:
        return Operand("i", dataset, typ, index, noadvance)

    @staticmethod
    def output(dataset: int, typ: int, index: int) -> Operand:
        return Operand("o", dataset, typ, index)

    # ---- instructions ---------------------------------------------------------------------------
    def op(self, name: str, *operands: Operand) -> None:
        if name in ("exec_index", "exec_indexf"):
            assert len(operands) == 1
            self._ins.append(_Ins(name, (), tuple(operands)))
            return
        code, n_in, n_out = OPS[name]
        assert len(operands) == n_in + n_out, "%s takes %d+%d operands" % (name, n_in, n_out)
        self._ins.append(_Ins(name, tuple(operands[:n_in]), tuple(operands[n_in:])))

    def acquireindex(self, predicate: Operand, dataset: int = 0) -> None:
        self._ins.append(_Ins("acquireindex", (predicate,), (), dataset))

    def output_data(self, typ: int, dataset: int, pairs: Sequence[Tuple[Operand, Operand]], from_half: bool = False) -> None:
        """One grouped output instruction: each pair is (source register, destination output register)."""
        name = {FLOAT: "outputdata_float", INT32: "outputdata_int32", HALF: "outputdata_half"}[typ]
        if from_half:
            name = {FLOAT: "outputdata_float_from_half", HALF: "outputdata_half_from_half"}[typ]
        assert 0 < len(pairs) <= 255
        for _, d in pairs:
            assert d.kind == "o" and d.a == dataset and d.b == typ
        self._ins.append(_Ins(name, (), (), dataset, tuple(pairs)))

    def ext_fn(self, num_inputs: int, num_outputs: int) -> int:
        self._ext.append((num_inputs, num_outputs))
        return len(self._ext) - 1

    def call(self, fn_idx: int, ins: Sequence[Operand], outs: Sequence[Operand]) -> None:
        ni, no = self._ext[fn_idx]
        assert len(ins) == ni and len(outs) == no
        self._ins.append(_Ins("external_func_call", tuple(ins), tuple(outs), fn_idx=fn_idx))

    def update_id(self, id_index: Operand, id_tag: Operand, dataset: int = 0) -> None:
        self._ins.append(_Ins("update_id", (id_index, id_tag), (), dataset))

    def acquire_id(self, id_index_out: Operand, id_tag_out: Operand, dataset: int = 0) -> None:
        self._ins.append(_Ins("acquire_id", (), (id_index_out, id_tag_out), dataset))

    # ---- build ----------------------------------------------------------------------------------
    def build(self, alias_outputs: bool = False, hash_id: int = 0, extra_temps: int = 0) -> BuiltScript:
        ins_list = self._ins

        # -- constants: remapped in first-use order
        const_remap: List[int] = []
        const_slot: Dict[int, int] = {}

        # -- inputs: sorted by (dataset, advancing before no-advance, type float/int/half, index)
        input_keys = set()
        out_keys: List[Tuple[int, int, int]] = []
        out_slot: Dict[Tuple[int, int, int], int] = {}

        def visit(o: Operand):
            if o.kind == "c" and o.a not in const_slot:
                const_slot[o.a] = len(const_remap)
                const_remap.append(o.a)
            elif o.kind == "i":
                input_keys.add((o.a, 1 if o.noadvance else 0, o.b, o.c))
            elif o.kind == "o":
                k = (o.a, o.b, o.c)
                if k not in out_slot:
                    out_slot[k] = len(out_keys)
                    out_keys.append(k)

        for ins in ins_list:
            for o in ins.ins + ins.outs:
                visit(o)
            for s, d in ins.pairs:
                visit(s)
                visit(d)

        inputs_sorted = sorted(input_keys)
        input_slot = {k: i for i, k in enumerate(inputs_sorted)}
        num_input_datasets = (max(k[0] for k in inputs_sorted) + 1) if inputs_sorted else 0
        input_remap = [k[3] for k in inputs_sorted]
        ids_offsets = [0] * (8 * num_input_datasets)
        for ds in range(num_input_datasets):
            counts = [sum(1 for k in inputs_sorted if k[0] == ds and k[1] == na and k[2] == t)
                      for na in (0, 1) for t in (FLOAT, INT32, HALF)]
            base = sum(1 for k in inputs_sorted if k[0] < ds)
            acc = base
            for j in range(6):
                ids_offsets[ds * 8 + j] = acc
                acc += counts[j]
            ids_offsets[ds * 8 + 6] = acc
            ids_offsets[ds * 8 + 7] = counts[3] + counts[4] + counts[5]
        num_noadvance = sum(1 for k in inputs_sorted if k[1] == 1)

        # -- temp liveness (instruction granularity)
        first_def: Dict[int, int] = {}
        last_use: Dict[int, int] = {}
        for idx, ins in enumerate(ins_list):
            reads = list(ins.ins) + [s for s, _ in ins.pairs]
            for o in reads:
                if o.kind == "t":
                    assert o.a in first_def, "temp %d read before written (instruction %d: %s)" % (o.a, idx, ins.name)
                    last_use[o.a] = idx
            for o in ins.outs:
                if o.kind == "t":
                    first_def.setdefault(o.a, idx)
                    last_use[o.a] = max(last_use.get(o.a, idx), idx)

        # -- optional aliasing of a temp onto the output buffer it is stored to.  Legal here when the temp is
        #    the source of exactly one output pair, never a source of anything after that output
        #    instruction, and that output register is not otherwise written before it.
        alias: Dict[int, Operand] = {}
        if alias_outputs:
            pair_uses: Dict[int, List[Tuple[int, Operand, int]]] = {}
            for idx, ins in enumerate(ins_list):
                for s, d in ins.pairs:
                    if s.kind == "t":
                        pair_uses.setdefault(s.a, []).append((idx, d, ins.name.count("half")))
            taken = set()
            for t, uses in pair_uses.items():
                if len(uses) != 1:
                    continue
                idx, d, is_half = uses[0]
                if is_half or last_use[t] != idx or (d.a, d.b, d.c) in taken:
                    continue
                taken.add((d.a, d.b, d.c))
                alias[t] = d

        # -- linear scan over the remaining temps
        phys: Dict[int, int] = {}
        free: List[int] = []
        active: List[Tuple[int, int]] = []   # (last_use, phys)
        num_phys = 0
        order = sorted((t for t in first_def if t not in alias), key=lambda t: first_def[t])
        for t in order:
            start = first_def[t]
            still = []
            for lu, p in active:
                # a register may be reused by the op that reads it for the last time (reads precede the write per lane)
                if lu <= start:
                    free.append(p)
                else:
                    still.append((lu, p))
            active = still
            if free:
                free.sort()
                p = free.pop(0)
            else:
                p = num_phys
                num_phys += 1
            phys[t] = p
            active.append((last_use[t], p))
        num_temps = num_phys + extra_temps

        T, C, I = num_temps, len(const_remap), len(inputs_sorted)

        def reg(o: Operand) -> int:
            if o.kind == "x":
                return 0xFFFF
            if o.kind == "t":
                if o.a in alias:
                    return reg(alias[o.a])
                return phys[o.a]
            if o.kind == "c":
                return T + const_slot[o.a]
            if o.kind == "i":
                return T + C + input_slot[(o.a, 1 if o.noadvance else 0, o.b, o.c)]
            if o.kind == "o":
                return T + C + 2 * I + out_slot[(o.a, o.b, o.c)]
            raise ValueError(o)

        # -- encode
        bc = bytearray()
        listing: List[str] = []
        flags = 0
        for ins in ins_list:
            code = CODE[ins.name]
            bc.append(code)
            if ins.name in opcodes.RANDOM_OPS:
                flags |= 1 << 3   # VVMFlag_HasRandInstruction
            if ins.name == "acquireindex":
                bc += struct.pack("<HB", reg(ins.ins[0]), ins.dataset)
                listing.append("acquireindex r%d ds%d" % (reg(ins.ins[0]), ins.dataset))
            elif ins.pairs:
                bc += struct.pack("<BB", len(ins.pairs), ins.dataset)
                bc += b"".join(struct.pack("<H", reg(s)) for s, _ in ins.pairs)
                bc += b"".join(struct.pack("<H", reg(d)) for _, d in ins.pairs)
                listing.append("%s ds%d %s" % (ins.name, ins.dataset,
                                               " ".join("r%d->r%d" % (reg(s), reg(d)) for s, d in ins.pairs)))
            elif ins.name == "external_func_call":
                bc += struct.pack("<H", ins.fn_idx)
                bc += b"".join(struct.pack("<H", reg(o)) for o in ins.ins + ins.outs)
                listing.append("call f%d %s" % (ins.fn_idx, " ".join("r%d" % reg(o) for o in ins.ins + ins.outs)))
            elif ins.name in ("update_id", "acquire_id"):
                ops_ = ins.ins + ins.outs
                bc += struct.pack("<HHB", reg(ops_[0]), reg(ops_[1]), ins.dataset)
                listing.append("%s r%d r%d ds%d" % (ins.name, reg(ops_[0]), reg(ops_[1]), ins.dataset))
            else:
                regs = [reg(o) for o in ins.ins + ins.outs]
                bc += b"".join(struct.pack("<H", r) for r in regs)
                listing.append("%s %s" % (ins.name, " ".join("r%d" % r for r in regs)))
        bc += bytes(8)   # 8 x done
        bc += bytes(1)   # pad

        out_dataset = [k[0] for k in out_keys]
        out_type = [k[1] for k in out_keys]
        out_dst = [k[2] for k in out_keys]
        max_output_dataset = (max(out_dataset) + 1) if out_keys else 0
        for ins in ins_list:
            if ins.name == "acquireindex":
                max_output_dataset = max(max_output_dataset, ins.dataset + 1)

        n_ext = len(self._ext)
        max_ext_regs = max([a + b for a, b in self._ext], default=0)
        dummy_regs = 0
        for ins in ins_list:
            if ins.name == "external_func_call":
                dummy_regs = max(dummy_regs, sum(1 for o in ins.ins + ins.outs if o.kind == "x"))

        header = struct.pack(
            "<IIHIHHHHHHIIIiIQ",
            len(bc), max_output_dataset, C, num_temps, C, I, num_noadvance, num_input_datasets,
            len(out_keys), len(out_keys), n_ext, max_ext_regs, dummy_regs, n_ext - 1 if n_ext else 0, flags, hash_id)
        assert len(header) == 54

        def align16(n):
            return (n + 15) & ~15

        blob = bytearray(header)
        blob += bytes(64 - len(blob))
        sections = [
            bytes(bc),
            struct.pack("<%dH" % C, *const_remap),
            struct.pack("<%dH" % I, *input_remap),
            struct.pack("<%dH" % len(ids_offsets), *ids_offsets),
            bytes(out_dataset),
            struct.pack("<%dH" % len(out_type), *out_type),
            struct.pack("<%dH" % len(out_dst), *out_dst),
            b"".join(struct.pack("<Qii", 0, a, b) for a, b in self._ext),
        ]
        for sec in sections:
            assert len(blob) % 16 == 0
            blob += sec
            blob += bytes(align16(len(blob)) - len(blob))

        stride = {FLOAT: 4, INT32: 4, HALF: 2}
        return BuiltScript(
            blob=bytes(blob), bytecode=bytes(bc), num_temps=num_temps, const_remap=const_remap,
            input_remap=input_remap, input_dataset_offsets=ids_offsets, output_dataset=out_dataset,
            output_type=out_type, output_dst=out_dst, num_ext_fns=n_ext, ext_fn_io=list(self._ext), flags=flags,
            listing=listing,
            bytes_in_per_instance=sum(stride[k[2]] for k in inputs_sorted if k[1] == 0),
            bytes_out_per_instance=sum(stride[t] for t in out_type))
