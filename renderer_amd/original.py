"""Writer for VectorVM bytecode in the format the Niagara compiler emits, i.e. the input of the reference's
OptimizeVectorVMScript (decoder: Runtime/VectorVM/Private/VectorVMOptimizer.cpp:744-1025).

Nothing in the reference tree produces this format (the compiler is not part of it), so tests and tools write it here and
hand it to ``HipBackend.optimize_script``.  Layouts, all little-endian:

* op with N inputs       ``[op][constFlags:u8, bit k = input k is a constant][in0..inN-1:u16][out:u16]``; a constant operand is a
  BYTE offset into the concatenated constant tables (a multiple of 4)                              (:986-1023, :329-330)
* inputdata_*            ``[op][dataset:u16][input:u16][dst temp:u16]``                            (:845-866)
* outputdata_*           ``[op][srcIsConst:u8][dataset:u16][index temp:u16][src:u16][dst:u16]``    (:871-891)
* acquireindex           ``[op][srcIsConst:u8][dataset:u16][predicate:u16][index temp out:u16]``   (:894-911)
* external_func_call     ``[op][fn:u8][inputs:u16...][outputs:u16...]``; an input with bit 15 set is a temp register, without it a
  constant byte offset; 0xFFFF = unused                                                            (:914-950)
* exec_index             ``[op][dst temp:u8][pad]``                                                (:952-956)
* enter_stat_scope       ``[op][id:u8][pad]``; exit_stat_scope ``[op]``                            (:959-961)
* update_id / acquire_id ``[op][dataset:u16][id temp:u16][tag temp:u16]``                          (:962-981)
* done                   ``[op]``

Every instruction is also kept as a tuple in ``ir`` so that a test can restate the script by other means.
"""
from __future__ import annotations

import struct
from typing import List, Sequence, Tuple, Union

from .opcodes import CODE

FLOAT, INT32, HALF = 0, 1, 2


class K:
    """Constant operand: 32-bit index into the concatenated constant tables."""

    def __init__(self, dword_index: int):
        self.dword = dword_index

    def __repr__(self):
        return "K(%d)" % self.dword


Operand = Union[int, K]          # an int is a temp register number
UNUSED = 0xFFFF

# inputs of the ops that exist in this format (VectorVMOptimizer.cpp:761-838)
_THREE = {"mad", "lerp", "clamp", "select", "clampi"}
_TWO = {"add", "sub", "mul", "div", "atan2", "fmod", "min", "max", "pow", "step", "cmplt", "cmple", "cmpgt", "cmpge", "cmpeq", "cmpneq",
        "addi", "subi", "muli", "divi", "mini", "maxi", "cmplti", "cmplei", "cmpgti", "cmpgei", "cmpeqi", "cmpneqi", "bit_and", "bit_or",
        "bit_xor", "bit_lshift", "bit_rshift", "logic_and", "logic_or", "logic_xor"}
_ONE = {"rcp", "rsq", "sqrt", "neg", "abs", "exp", "exp2", "log", "log2", "sin", "cos", "tan", "asin", "acos", "atan", "ceil", "floor",
        "frac", "trunc", "round", "sign", "random", "absi", "negi", "signi", "randomi", "bit_not", "logic_not", "f2i", "i2f", "f2b", "b2f",
        "i2b", "b2i", "fasi", "iasf"}


def op_inputs(name: str) -> int:
    return 3 if name in _THREE else 2 if name in _TWO else 1 if name in _ONE else 0


class OriginalScript:
    def __init__(self):
        self.bc = bytearray()
        self.ir: List[Tuple] = []
        self.ext: List[Tuple[int, int]] = []
        self._next_temp = 0
        self._done = False

    def temp(self) -> int:
        self._next_temp += 1
        return self._next_temp - 1

    def temps(self, n: int) -> List[int]:
        return [self.temp() for _ in range(n)]

    @staticmethod
    def _enc(o: Operand) -> int:
        return o.dword * 4 if isinstance(o, K) else o

    def input(self, dataset: int, typ: int, index: int, dst: int, noadvance: bool = False) -> int:
        name = "inputdata_" + ("noadvance_" if noadvance else "") + ("float", "int32", "half")[typ]
        self.bc += struct.pack("<BHHH", CODE[name], dataset, index, dst)
        self.ir.append(("input", dataset, typ, index, noadvance, dst))
        return dst

    def op(self, name: str, *operands: Operand) -> None:
        n = op_inputs(name)
        assert n and len(operands) == n + 1 and not isinstance(operands[-1], K), name
        flags = sum(1 << k for k in range(n) if isinstance(operands[k], K))
        self.bc += struct.pack("<BB", CODE[name], flags) + b"".join(struct.pack("<H", self._enc(o)) for o in operands)
        self.ir.append(("op", name, tuple(operands[:n]), operands[-1]))

    def acquireindex(self, predicate: Operand, index_temp: int, dataset: int = 0) -> None:
        self.bc += struct.pack("<BBHHH", CODE["acquireindex"], 1 if isinstance(predicate, K) else 0, dataset, self._enc(predicate), index_temp)
        self.ir.append(("acquireindex", predicate, index_temp, dataset))

    def output(self, typ: int, dataset: int, index_temp: int, src: Operand, dst_reg: int) -> None:
        name = "outputdata_" + ("float", "int32", "half")[typ]
        self.bc += struct.pack("<BBHHHH", CODE[name], 1 if isinstance(src, K) else 0, dataset, index_temp, self._enc(src), dst_reg)
        self.ir.append(("output", typ, dataset, index_temp, src, dst_reg))

    def ext_fn(self, num_inputs: int, num_outputs: int) -> int:
        self.ext.append((num_inputs, num_outputs))
        return len(self.ext) - 1

    def call(self, fn: int, ins: Sequence[Operand], outs: Sequence[int]) -> None:
        ni, no = self.ext[fn]
        assert len(ins) == ni and len(outs) == no
        words = []
        for o in ins:
            words.append(UNUSED if o == UNUSED else o.dword * 4 if isinstance(o, K) else (o | 0x8000))
        words += list(outs)
        self.bc += struct.pack("<BB", CODE["external_func_call"], fn) + b"".join(struct.pack("<H", w) for w in words)
        self.ir.append(("call", fn, tuple(ins), tuple(outs)))

    def exec_index(self, dst: int) -> None:
        assert dst < 256
        self.bc += struct.pack("<BBB", CODE["exec_index"], dst, 0)
        self.ir.append(("exec_index", dst))

    def enter_stat_scope(self, scope_id: int = 0) -> None:
        self.bc += struct.pack("<BBB", CODE["enter_stat_scope"], scope_id, 0)

    def exit_stat_scope(self) -> None:
        self.bc += struct.pack("<B", CODE["exit_stat_scope"])

    def update_id(self, dataset: int, id_temp: int, tag_temp: int) -> None:
        self.bc += struct.pack("<BHHH", CODE["update_id"], dataset, id_temp, tag_temp)
        self.ir.append(("update_id", dataset, id_temp, tag_temp))

    def acquire_id(self, dataset: int, id_temp: int, tag_temp: int) -> None:
        self.bc += struct.pack("<BHHH", CODE["acquire_id"], dataset, id_temp, tag_temp)
        self.ir.append(("acquire_id", dataset, id_temp, tag_temp))

    def done(self) -> bytes:
        """Ends the stream with `done` (once) and returns it."""
        if not self._done:
            self.bc += struct.pack("<B", CODE["done"])
            self._done = True
        return bytes(self.bc)
