#!/usr/bin/env python3
"""Writes vvm_threaded_gen.h: the routines of the eight-wide interpreter kernels' core as gfx950 assembly text, and the table the
linker picks an instruction's routine from.  (The design is described in vvm_threaded.h, which includes the generated file.)

A routine = the code for one op with one combination of operand kinds (R: per-lane register, read from the register file under the
GPR-index mode; U: uniform register, a scalar).  All routines of one arity have slots of one size; routine r of arity n with uniform
mask m (bit k: operand k is uniform; never all of them) starts at byte BASE[n] + (r * (2^n - 1) + m) * SLOT[n] of the table.  The
generated text places every routine with `.org`, which the assembler refuses to move backwards: a routine that outgrows its slot
fails the build.

Register conventions (assembler symbols set by the kernel before the text: VT_FB first VGPR of the file, VT_T / VT_Q first VGPRs of
two groups of eight temporaries, VT_C first SGPR of the routine's own instruction buffer, VT_N of the other buffer, VT_NB first
SGPR of the pair holding the address of the other buffer's routines):
  instruction words (vvm_device.h, VvmDevIns): s[VT_C+4] routine offset, s[VT_C+5] = a | c << 8 | d << 16 | e << 24, s[VT_C+6] = b,
  s[VT_C+7] = destination; values of uniform operands a..e in s[82:83] .. s[90:91] (both halves: a 64-bit scalar operand of a packed
  instruction), scratch s92, compare results vcc, s[94:95], s[96:97], s[98:99], jump target s[80:81].

Rules every routine keeps:
  * a register operand is read straight from the file by an instruction whose FIRST source is the file's first VGPRs under mode
    SRC0 (index = the operand byte); temporaries and scalars sit in the other source positions, so a mode left on by an earlier step
    never touches them; a step with a temporary as its first source switches the mode off first.  One exception: the last source
    of a three-source instruction is read under mode SRC2, right before the store;
  * the values of uniform operands are fetched first (v_readlane with the mode off), each followed by s_mov + s_nop 0: the wait states
    a VALU read of a VALU-written SGPR needs on this target;
  * a compare's result is read by the fourth instruction after it at the earliest (four result registers in turn);
  * the result goes to the file under mode DST, the mode is switched off, and the routine jumps to the next instruction's routine.
Arithmetic: the instructions the compiler emits for the C++ cases of the same ops (packed add / mul, nothing fused: the reference's
build has no FMA, UnrealMathSSE.h:1411-1418), so no result changes by a bit.
"""
import os
import re
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
SLOT = {1: 320, 2: 320, 3: 480, 4: 448, 5: 448}
LETTERS = "abcde"
SPAIR = {"a": 82, "b": 84, "c": 86, "d": 88, "e": 90}
MASKS = ["vcc", "s[94:95]", "s[96:97]", "s[98:99]"]


# ---- operand accessors: fam in {"F", "T", "Q", "Sa".."Se"} ----
def r32(fam, k):
    if fam == "F":
        return "v[VT_FB+%d]" % k
    if fam in ("T", "Q"):
        return "v[VT_%s+%d]" % (fam, k)
    return "s%d" % SPAIR[fam[1]]


def r64(fam, k):
    if fam == "F":
        return "v[VT_FB+%d:VT_FB+%d]" % (k, k + 1)
    if fam in ("T", "Q"):
        return "v[VT_%s+%d:VT_%s+%d]" % (fam, k, fam, k + 1)
    p = SPAIR[fam[1]]
    return "s[%d:%d]" % (p, p + 1)


def opreg(letter):
    """(instructions to run first, SGPR holding the operand's byte in its low eight bits)"""
    if letter == "a":
        return [], "s[VT_C+5]"
    if letter == "b":
        return [], "s[VT_C+6]"
    shift = {"c": 8, "d": 16, "e": 24}[letter]
    return ["s_lshr_b32 s92, s[VT_C+5], %d" % shift], "s92"


def idx(letter, mode="SRC0"):
    pre, reg = opreg(letter)
    return pre + ["s_set_gpr_idx_on %s, gpr_idx(%s)" % (reg, mode)]


OFF = ["s_set_gpr_idx_off"]


def pre(kind, letter):
    if kind == "R":
        return []
    p, reg = opreg(letter)
    s = SPAIR[letter]
    return p + ["v_readlane_b32 s%d, %%[uni], %s" % (s, reg), "s_mov_b32 s%d, s%d" % (s + 1, s), "s_nop 0"]


# Instance slots per lane that a routine works on: 8 (the eight-wide kernels: a register is eight consecutive VGPRs) or 4 (the
# four-wide kernel of 44 registers: four).  Set by main() around each of the two sets of routines; the routines of the two sets lie
# at the same offsets, so an instruction's routine offset does not depend on the kernel it runs on (its operand bytes do).
W = 8


def mov4(d, s):
    return ["v_mov_b64 %s, %s" % (r64(d, k), r64(s, k)) for k in range(0, W, 2)]


def pk4(op, d, p, x, mods=""):
    return ["%s %s, %s, %s%s" % (op, r64(d, k), r64(p, k), r64(x, k), mods) for k in range(0, W, 2)]


def l8(op, d, p, x):
    return ["%s %s, %s, %s" % (op, r32(d, k), r32(p, k), r32(x, k)) for k in range(W)]


def l8_3(op, d, p, x, y):
    return ["%s %s, %s, %s, %s" % (op, r32(d, k), r32(p, k), r32(x, k), r32(y, k)) for k in range(W)]


def l8_1(op, d, p):
    return ["%s %s, %s" % (op, r32(d, k), r32(p, k)) for k in range(W)]


def l8_pk(op, d, p, const):
    return ["%s %s, %s, %s" % (op, r32(d, k), r32(p, k), const) for k in range(W)]


def l8_kp(op, d, const, p):
    return ["%s %s, %s, %s" % (op, r32(d, k), const, r32(p, k)) for k in range(W)]


def cmp_use8(cmp, use):
    """four compares into the four result registers, then the four instructions that read them; once per four slots"""
    out = []
    for base in range(0, W, 4):
        for j in range(4):
            out += cmp(base + j, MASKS[j])
        for j in range(4):
            out += use(base + j, MASKS[j])
    return out


# ---- ops that are one instruction per slot (or pair): SI[name](form, d, p, x): form 0: d = p op x; form 1: d = x op p.  p is the
# instruction's first source either way (the file under the index mode, or a scalar), x a temporary or a scalar ----
def _pk(op, mods0="", mods1=""):
    return lambda form, d, p, x: pk4(op, d, p, x, mods1 if form else mods0)


def _l8(op0, op1=None):
    return lambda form, d, p, x: l8(op1 if (form and op1) else op0, d, p, x)


SI = {
    "PKADD": _pk("v_pk_add_f32"),
    "PKMUL": _pk("v_pk_mul_f32"),
    "PKSUB": _pk("v_pk_add_f32", " neg_lo:[0,1] neg_hi:[0,1]", " neg_lo:[1,0] neg_hi:[1,0]"),
    "ADDI": _l8("v_add_u32_e64"),
    "SUBI": _l8("v_sub_u32_e64", "v_subrev_u32_e64"),
    "MULI": _l8("v_mul_lo_u32"),
    "ANDB": _l8("v_and_b32_e64"),
    "ORB": _l8("v_or_b32_e64"),
    "XORB": _l8("v_xor_b32_e64"),
    "MINI": _l8("v_min_i32_e64"),
    "MAXI": _l8("v_max_i32_e64"),
}


def S(letter):
    return "S" + letter


def load(kind, letter, d):
    """d = operand"""
    return idx(letter) + mov4(d, "F") if kind == "R" else mov4(d, S(letter))


def first2(ka, kb, la, lb, op, d):
    """d = (operand la) op (operand lb)"""
    f = SI[op]
    if ka == "R" and kb == "R":
        return load("R", la, d) + idx(lb) + f(1, d, "F", d)
    if ka == "R":
        return idx(la) + f(0, d, "F", S(lb))
    if kb == "R":
        return idx(lb) + f(1, d, "F", S(la))
    return load("U", la, d) + f(1, d, S(lb), d)


def step(kind, fwd, letter, op, d):
    """fwd: d = d op operand; else d = operand op d"""
    f = SI[op]
    form = 1 if fwd else 0
    return idx(letter) + f(form, d, "F", d) if kind == "R" else f(form, d, S(letter), d)


def last3(kind, letter, op, d):
    """d = op(d, Q, operand): the operand as the instruction's last source"""
    return idx(letter, "SRC2") + l8_3(op, d, d, "Q", "F") if kind == "R" else OFF + l8_3(op, d, d, "Q", S(letter))


def store():
    return ["s_set_gpr_idx_on s[VT_C+7], gpr_idx(DST)"] + mov4("F", "T") + OFF


NEXT = ["s_waitcnt lgkmcnt(0)", "s_load_dwordx8 s[VT_C:VT_C+7], %[prog], %[pc]", "s_add_u32 %[pc], %[pc], 32",
        "s_add_u32 s80, s[VT_NB], s[VT_N+4]", "s_addc_u32 s81, s[VT_NB+1], 0", "s_setpc_b64 s[80:81]"]


# ---- ops on temporaries (mode off): operands in T (first) and Q (second), result in T ----
def mask_use(k, m):
    return ["v_cndmask_b32_e64 %s, 0, -1, %s" % (r32("T", k), m)]


def cmp_tq(name):
    return lambda k, m: ["%s %s, %s, %s" % (name, m, r32("T", k), r32("Q", k))]


def sel_tq(k, m):      # m ? T : Q
    return ["v_cndmask_b32_e64 %s, %s, %s, %s" % (r32("T", k), r32("Q", k), r32("T", k), m)]


CMP = {  # all ones where the compare holds (VectorVMRuntime.cpp:1942-1947, 1960-1965): ordered compares, cmpneq true on NaN
    "CMPLT": "v_cmp_lt_f32_e64", "CMPLE": "v_cmp_le_f32_e64", "CMPGT": "v_cmp_gt_f32_e64", "CMPGE": "v_cmp_ge_f32_e64",
    "CMPEQ": "v_cmp_eq_f32_e64", "CMPNEQ": "v_cmp_neq_f32_e64",
    "CMPLTI": "v_cmp_lt_i32_e64", "CMPLEI": "v_cmp_le_i32_e64", "CMPGTI": "v_cmp_gt_i32_e64", "CMPGEI": "v_cmp_ge_i32_e64",
    "CMPEQI": "v_cmp_eq_i32_e64", "CMPNEQI": "v_cmp_ne_i32_e64",
}


def mi(name):
    if name in CMP:
        return cmp_use8(cmp_tq(CMP[name]), mask_use)
    if name == "MIN":      # _mm_min_ps: (a < b) ? a : b, the second operand when unordered or equal (UnrealMathSSE.h:2380-2383)
        return cmp_use8(cmp_tq("v_cmp_lt_f32_e64"), sel_tq)
    if name == "MAX":      # (a > b) ? a : b (:2404-2407)
        return cmp_use8(cmp_tq("v_cmp_gt_f32_e64"), sel_tq)
    if name == "SHL":      # a << (b & 31): the hardware uses the low five bits of the count (VectorVMPlatformGeneric.h:61-89)
        return l8("v_lshlrev_b32_e64", "T", "Q", "T")
    if name == "SHR":
        return l8("v_lshrrev_b32_e64", "T", "Q", "T")
    if name == "STEP":     # step(a, b) = (b - a) >= 0 ? 1 : 0 (VectorVMRuntime.cpp:1939)
        return l8("v_sub_f32_e64", "T", "Q", "T") + cmp_use8(lambda k, m: ["v_cmp_ge_f32_e64 %s, %s, 0" % (m, r32("T", k))],
                                                              lambda k, m: ["v_cndmask_b32_e64 %s, 0, 1.0, %s" % (r32("T", k), m)])
    if name == "SUBNEG":   # 0 - (a - b)
        return l8("v_sub_f32_e64", "T", "T", "Q") + l8_kp("v_sub_f32_e64", "T", "0", "T")
    raise KeyError(name)


# ---- unary bodies: the operand is a register; result in T ----
def unary(name):
    a = idx("a")
    if name == "COPY":
        return a + mov4("T", "F")
    if name == "NEG":      # 0.0 - a (UnrealMathSSE.h:1311-1314): v_subrev D, a, 0
        return a + l8_pk("v_subrev_f32_e64", "T", "F", "0")
    if name == "ABS":
        return ["s_mov_b32 s92, 0x7fffffff"] + a + l8_pk("v_and_b32_e64", "T", "F", "s92")
    if name in ("FLOOR", "CEIL", "TRUNC"):
        return a + l8_1("v_%s_f32_e32" % name.lower(), "T", "F")
    if name == "ROUND":    # to nearest even (UnrealMathSSE.h:2819-2822)
        return a + l8_1("v_rndne_f32_e32", "T", "F")
    if name == "FRAC":     # a - trunc(a) (UnrealMathVectorCommon.h.inl:883-886)
        return a + l8_1("v_trunc_f32_e32", "T", "F") + l8("v_sub_f32_e64", "T", "F", "T")
    if name == "I2F":
        return a + l8_1("v_cvt_f32_i32_e32", "T", "F")
    if name == "NOT":
        return a + l8_1("v_not_b32_e32", "T", "F")
    if name == "NEGI":
        return a + l8_pk("v_subrev_u32_e64", "T", "F", "0")
    if name == "B2F":      # mask & bits(1.0f) (VectorVMRuntime.cpp:1978-1981)
        return a + l8_pk("v_and_b32_e64", "T", "F", "1.0")
    if name == "B2I":
        return a + l8_pk("v_and_b32_e64", "T", "F", "1")
    if name in ("F2B", "I2B"):      # a > 0 (float / signed integer)
        cmp = "v_cmp_gt_f32_e64" if name == "F2B" else "v_cmp_gt_i32_e64"
        return a + cmp_use8(lambda k, m: ["%s %s, %s, 0" % (cmp, m, r32("F", k))], mask_use)
    if name == "SIGN":     # a >= 0 ? 1 : -1 (UnrealMathSSE.h:2911-2915: NaN gives -1)
        return a + cmp_use8(lambda k, m: ["v_cmp_ge_f32_e64 %s, %s, 0" % (m, r32("F", k))],
                            lambda k, m: ["v_cndmask_b32_e64 %s, -1.0, 1.0, %s" % (r32("T", k), m)])
    if name == "ABSI":     # _mm_abs_epi32 = max(a, 0 - a) as signed integers (INT_MIN stays)
        return a + l8_pk("v_subrev_u32_e64", "T", "F", "0") + l8("v_max_i32_e64", "T", "F", "T")
    if name == "F2I":
        # _mm_cvttps_epi32: truncate; 0x80000000 for NaN and whatever does not fit (UnrealMathSSE.h:3409-3412).  The hardware conversion
        # saturates and gives 0 for NaN, so its value is kept only where |a| < 2^31 (a = -2^31 converts to 0x80000000 either way).
        return ["s_mov_b32 s92, 0x4f000000"] + a + mov4("Q", "F") + OFF + l8_1("v_cvt_i32_f32_e32", "T", "Q") + cmp_use8(
            lambda k, m: ["v_cmp_gt_f32_e64 %s, s92, |%s|" % (m, r32("Q", k))],
            lambda k, m: ["v_bfrev_b32_e32 %s, 1" % r32("Q", k), "v_cndmask_b32_e64 %s, %s, %s, %s" % (r32("T", k), r32("Q", k), r32("T", k), m)])
    raise KeyError(name)


# ---- routine bodies by shape; kinds = tuple of "R" / "U" per operand ----
def pres(kinds):
    out = []
    for k, letter in zip(kinds, LETTERS):
        out += pre(k, letter)
    return out


def body(shape, kinds, args):
    K = kinds
    if shape == "UN":
        return unary(args[0])
    if shape == "BIN_SI":
        return pres(K) + first2(K[0], K[1], "a", "b", args[0], "T")
    if shape == "BIN_MI":
        return pres(K) + load(K[0], "a", "T") + load(K[1], "b", "Q") + OFF + mi(args[0])
    if shape == "TER_ACC":      # (a OP1 b) OP2 c, or c OP2 (a OP1 b) when fwd = 0
        op1, op2, fwd = args
        return pres(K) + first2(K[0], K[1], "a", "b", op1, "T") + step(K[2], fwd, "c", op2, "T")
    if shape == "TER_CMPL":     # (a CMP b) AND / OR c
        return pres(K) + load(K[0], "a", "T") + load(K[1], "b", "Q") + OFF + mi(args[0]) + step(K[2], 1, "c", args[1], "T")
    if shape == "TER_SELECT":   # (mask & x) | (~mask & y) (UnrealMathSSE.h:1805-1810) = v_bfi_b32 D, mask, x, y
        return pres(K) + load(K[0], "a", "T") + load(K[1], "b", "Q") + last3(K[2], "c", "v_bfi_b32", "T")
    if shape == "TER_CLAMP":    # min(max(a, b), c) (UnrealMathVectorCommon.h.inl:495-498)
        return pres(K) + load(K[0], "a", "T") + load(K[1], "b", "Q") + OFF + mi("MAX") + load(K[2], "c", "Q") + OFF + mi("MIN")
    if shape == "TER_MULMAX":   # max(a * b, c)
        return pres(K) + first2(K[0], K[1], "a", "b", "PKMUL", "T") + load(K[2], "c", "Q") + OFF + mi("MAX")
    if shape == "QUAD_ACC":     # ((a OP1 b) OP2 c) OP3 d
        op1, op2, f2, op3, f3 = args
        return pres(K) + first2(K[0], K[1], "a", "b", op1, "T") + step(K[2], f2, "c", op2, "T") + step(K[3], f3, "d", op3, "T")
    if shape == "QUAD_PAIR":    # (c OP2 d) OP3 (a OP1 b)
        op1, op2, op3 = args
        return pres(K) + first2(K[0], K[1], "a", "b", op1, "T") + first2(K[2], K[3], "c", "d", op2, "Q") + OFF + SI[op3](0, "T", "Q", "T")
    if shape == "QUAD_CMPSEL":  # (a CMP b) ? c : d, bit by bit like the reference's select
        return pres(K) + load(K[0], "a", "T") + load(K[1], "b", "Q") + OFF + mi(args[0]) + load(K[2], "c", "Q") + last3(K[3], "d", "v_bfi_b32", "T")
    if shape == "QUAD_SELOP":   # select(a, b, c) OP d
        # (a scalar d is read with no mode switch of its own: mode SRC2, left on by a register c, goes off first)
        return (pres(K) + load(K[0], "a", "T") + load(K[1], "b", "Q") + last3(K[2], "c", "v_bfi_b32", "T") + (OFF if K[2] == "R" and K[3] == "U" else []) +
                step(K[3], 1, "d", args[0], "T"))
    if shape == "MADMAD0":      # d * e + (a * b + c) (VectorVMRuntime.cpp:2040)
        return (pres(K) + first2(K[0], K[1], "a", "b", "PKMUL", "T") + step(K[2], 1, "c", "PKADD", "T") + first2(K[3], K[4], "d", "e", "PKMUL", "Q") +
                OFF + SI["PKADD"](0, "T", "Q", "T"))
    if shape == "MADMAD1":      # ((a * b + c) * d) + e (:2041)
        return (pres(K) + first2(K[0], K[1], "a", "b", "PKMUL", "T") + step(K[2], 1, "c", "PKADD", "T") + step(K[3], 1, "d", "PKMUL", "T") +
                step(K[4], 1, "e", "PKADD", "T"))
    raise KeyError(shape)


# ---- the ops: (opcode name in vvm_abi.h / vvm_device.h, arity, shape, args) ----
OPS = [("VVM_DOP_COPY", 1, "UN", ("COPY",)), ("VVM_OP_neg", 1, "UN", ("NEG",)), ("VVM_OP_abs", 1, "UN", ("ABS",)), ("VVM_OP_floor", 1, "UN", ("FLOOR",)),
       ("VVM_OP_ceil", 1, "UN", ("CEIL",)), ("VVM_OP_trunc", 1, "UN", ("TRUNC",)), ("VVM_OP_round", 1, "UN", ("ROUND",)), ("VVM_OP_frac", 1, "UN", ("FRAC",)),
       ("VVM_OP_i2f", 1, "UN", ("I2F",)), ("VVM_OP_bit_not", 1, "UN", ("NOT",)), ("VVM_OP_logic_not", 1, "UN", ("NOT",)), ("VVM_OP_negi", 1, "UN", ("NEGI",)),
       ("VVM_OP_b2f", 1, "UN", ("B2F",)), ("VVM_OP_b2i", 1, "UN", ("B2I",)), ("VVM_OP_f2b", 1, "UN", ("F2B",)), ("VVM_OP_i2b", 1, "UN", ("I2B",)),
       ("VVM_OP_sign", 1, "UN", ("SIGN",)), ("VVM_OP_absi", 1, "UN", ("ABSI",)), ("VVM_OP_f2i", 1, "UN", ("F2I",))]
for _n, _si in (("add", "PKADD"), ("sub", "PKSUB"), ("mul", "PKMUL"), ("addi", "ADDI"), ("subi", "SUBI"), ("muli", "MULI"), ("bit_and", "ANDB"),
                ("logic_and", "ANDB"), ("bit_or", "ORB"), ("logic_or", "ORB"), ("bit_xor", "XORB"), ("logic_xor", "XORB"), ("mini", "MINI"), ("maxi", "MAXI")):
    OPS.append(("VVM_OP_" + _n, 2, "BIN_SI", (_si,)))
for _n, _m in (("min", "MIN"), ("max", "MAX"), ("cmplt", "CMPLT"), ("cmple", "CMPLE"), ("cmpgt", "CMPGT"), ("cmpge", "CMPGE"), ("cmpeq", "CMPEQ"),
               ("cmpneq", "CMPNEQ"), ("cmplti", "CMPLTI"), ("cmplei", "CMPLEI"), ("cmpgti", "CMPGTI"), ("cmpgei", "CMPGEI"), ("cmpeqi", "CMPEQI"),
               ("cmpneqi", "CMPNEQI"), ("bit_lshift", "SHL"), ("bit_rshift", "SHR"), ("step", "STEP"), ("sub_neg", "SUBNEG")):
    OPS.append(("VVM_OP_" + _n, 2, "BIN_MI", (_m,)))
OPS += [("VVM_OP_mad", 3, "TER_ACC", ("PKMUL", "PKADD", 1)), ("VVM_OP_mul_add", 3, "TER_ACC", ("PKMUL", "PKADD", 1)),
        ("VVM_OP_mul_sub0", 3, "TER_ACC", ("PKMUL", "PKSUB", 1)), ("VVM_OP_mul_sub1", 3, "TER_ACC", ("PKMUL", "PKSUB", 0)),
        ("VVM_OP_mul_mul", 3, "TER_ACC", ("PKMUL", "PKMUL", 1)), ("VVM_OP_add_add", 3, "TER_ACC", ("PKADD", "PKADD", 1)),
        ("VVM_OP_sub_mul", 3, "TER_ACC", ("PKSUB", "PKMUL", 1)), ("VVM_OP_muli_addi", 3, "TER_ACC", ("MULI", "ADDI", 1)),
        ("VVM_OP_addi_muli", 3, "TER_ACC", ("ADDI", "MULI", 1)), ("VVM_OP_clampi", 3, "TER_ACC", ("MAXI", "MINI", 1)),
        ("VVM_OP_select", 3, "TER_SELECT", ()), ("VVM_OP_clamp", 3, "TER_CLAMP", ()), ("VVM_OP_mul_max", 3, "TER_MULMAX", ())]
for _logic, _si in (("and", "ANDB"), ("or", "ORB")):
    for _c, _m in (("cmplt", "CMPLT"), ("cmple", "CMPLE"), ("cmpgt", "CMPGT"), ("cmpge", "CMPGE"), ("cmpeq", "CMPEQ"), ("cmpne", "CMPNEQ"),
                   ("cmplti", "CMPLTI"), ("cmplei", "CMPLEI"), ("cmpgti", "CMPGTI"), ("cmpgei", "CMPGEI"), ("cmpeqi", "CMPEQI"), ("cmpnei", "CMPNEQI")):
        OPS.append(("VVM_OP_%s_logic_%s" % (_c, _logic), 3, "TER_CMPL", (_m, _si)))
OPS += [("VVM_OP_mad_mul", 4, "QUAD_ACC", ("PKMUL", "PKADD", 1, "PKMUL", 1)), ("VVM_OP_mad_add", 4, "QUAD_ACC", ("PKMUL", "PKADD", 1, "PKADD", 1)),
        ("VVM_OP_mad_sub0", 4, "QUAD_ACC", ("PKMUL", "PKADD", 1, "PKSUB", 1)), ("VVM_OP_mad_sub1", 4, "QUAD_ACC", ("PKMUL", "PKADD", 1, "PKSUB", 0)),
        ("VVM_OP_mul_mad0", 4, "QUAD_ACC", ("PKMUL", "PKMUL", 1, "PKADD", 1)),
        ("VVM_OP_mul_mad1", 4, "QUAD_PAIR", ("PKMUL", "PKMUL", "PKADD")), ("VVM_OP_add_mad1", 4, "QUAD_PAIR", ("PKADD", "PKMUL", "PKADD")),
        ("VVM_OP_cmplt_select", 4, "QUAD_CMPSEL", ("CMPLT",)), ("VVM_OP_cmple_select", 4, "QUAD_CMPSEL", ("CMPLE",)), ("VVM_OP_cmpeq_select", 4, "QUAD_CMPSEL", ("CMPEQ",)),
        ("VVM_OP_cmplti_select", 4, "QUAD_CMPSEL", ("CMPLTI",)), ("VVM_OP_cmplei_select", 4, "QUAD_CMPSEL", ("CMPLEI",)), ("VVM_OP_cmpeqi_select", 4, "QUAD_CMPSEL", ("CMPEQI",)),
        ("VVM_OP_select_mul", 4, "QUAD_SELOP", ("PKMUL",)), ("VVM_OP_select_add", 4, "QUAD_SELOP", ("PKADD",)),
        ("VVM_OP_mad_mad0", 5, "MADMAD0", ()), ("VVM_OP_mad_mad1", 5, "MADMAD1", ())]


def layout():
    """per arity: (base offset, [(opcode, index)]); total bytes"""
    base, at = {}, 256      # (offset 0: the routine that leaves)
    per = {n: [o for o in OPS if o[1] == n] for n in range(1, 6)}
    for n in range(1, 6):
        base[n] = at
        at += len(per[n]) * ((1 << n) - 1) * SLOT[n]
    return base, per, at


def c_string(lines):
    return "\n".join('\t"%s\\n" \\' % l for l in lines)


def main():
    base, per, total = layout()
    out = ["// vvm_threaded_gen.h -- written by gen_threaded.py (see there and vvm_threaded.h); do not edit.", "#pragma once", ""]
    for n in range(1, 6):
        out.append("#define VVM_T_SLOT%d %du" % (n, SLOT[n]))
        out.append("#define VVM_T_BASE%d %du" % (n, base[n]))
    out.append("#define VVM_T_BYTES %du" % total)
    out.append("@LAYOUT@")
    out.append("")
    out.append("// the routine of an ordinary instruction: (opcode, operands, first routine's offset); routine for uniform mask m at offset + m * slot")
    out.append("#define VVM_T_OPS(X) \\")
    rows = []
    for n in range(1, 6):
        for r, (opcode, _, shape, args) in enumerate(per[n]):
            rows.append("\tX(%s, %d, %du)" % (opcode, n, base[n] + r * ((1 << n) - 1) * SLOT[n]))
    out.append(" \\\n".join(rows))
    # (a number that changes whenever a routine moves: the kernels' object file and the host's must have been built from the same table)
    out[out.index("@LAYOUT@")] = "#define VVM_T_LAYOUT 0x%08Xu" % zlib.crc32(("\n".join(rows) + repr(sorted(SLOT.items()))).encode())
    out.append("")
    global W
    count = 0
    for W, macro in ((8, "VVM_T_ROUTINES"), (4, "VVM_T4_ROUTINES")):
      out.append("// the routines for %d instance slots per lane, placed relative to the label TBL (a string, e.g. \"7b\")" % W)
      out.append("#define %s(TBL) \\" % macro)
      chunks = []
      for n in range(1, 6):
        for r, (opcode, _, shape, args) in enumerate(per[n]):
            for m in range((1 << n) - 1):
                kinds = tuple("U" if (m >> k) & 1 else "R" for k in range(n))
                if n == 1:
                    kinds = ("R",)
                lines = body(shape, kinds, args) + store() + NEXT
                off = base[n] + (r * ((1 << n) - 1) + m) * SLOT[n]
                chunks.append('\t".org " TBL " + %d\\n" /* %s %s */ \\\n%s' % (off, opcode, "".join(kinds), c_string(lines)))
                count += 1
      out.append("\n".join(chunks))
      out.append('\t".org " TBL " + %d\\n"' % total)
      out.append("")
    W = 8
    path = os.path.join(HERE, "vvm_threaded_gen.h")
    text = "\n".join(out) + "\n"
    if "--check" in sys.argv:
        ok = os.path.exists(path) and open(path).read() == text
        print("vvm_threaded_gen.h is %s" % ("up to date" if ok else "STALE: run gen_threaded.py"))
        return 0 if ok else 1
    with open(path, "w") as f:
        f.write(text)
    print("wrote %s: %d routines (two sets: eight and four instance slots per lane), table of %d bytes per instruction buffer" % (path, count, total))
    return 0


if __name__ == "__main__":
    sys.exit(main())
