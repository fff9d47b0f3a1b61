"""numpy restatement of the per-op semantics, used to pin the C oracle (a second, independent statement of the same rules).

Every function takes and returns uint32 bit patterns.  Citations: VectorVMRuntime.cpp (RT) op table :1906-2074 and the SSE
math layer it calls (Core/Public/Math/UnrealMathSSE.h = SSE).  Only ops whose result is exactly defined by IEEE arithmetic or
integer arithmetic are restated here; the polynomial and libm-class functions are checked by tolerance tests instead.
"""
import numpy as np

f32, u32, i32 = np.float32, np.uint32, np.int32


def F(x):
    return np.asarray(x, u32).view(f32)


def U(x):
    return np.asarray(x, f32).view(u32)


def I(x):
    return np.asarray(x, u32).view(i32)


def UI(x):
    return np.asarray(x, i32).view(u32)


def mask(b):
    return np.where(b, u32(0xFFFFFFFF), u32(0))


def sel(m, a, b):                       # SSE:1805-1810 bitwise select
    return (m & a) | (~m & b)


def _err():
    return np.errstate(all="ignore")


def add(a, b):
    with _err():
        return U(F(a) + F(b))


def sub(a, b):
    with _err():
        return U(F(a) - F(b))


def mul(a, b):
    with _err():
        return U(F(a) * F(b))


def mad(a, b, c):                       # SSE:1411-1418 with FMA3 off: two roundings
    return add(mul(a, b), c)


def fabs(a):
    return a & u32(0x7FFFFFFF)


def div(a, b):                          # RT:1909: |b| > 1e-8 ? a / b : 0
    with _err():
        q = U(F(a) / F(b))
        return np.where(F(fabs(b)) > f32(1e-8), q, u32(0))


def sqrt(a):                            # RT:1914
    with _err():
        return np.where(F(a) > f32(1e-8), U(np.sqrt(F(a))), u32(0))


def rsq(a):                             # RT:1913, SSE:2014-2022 (1 / sqrt, both exactly rounded)
    with _err():
        return np.where(F(a) > f32(1e-8), U(f32(1.0) / np.sqrt(F(a))), u32(0))


def neg(a):                             # SSE:1311-1314: 0 - a
    return sub(np.zeros_like(a), a)


def fmin(a, b):                         # _mm_min_ps: the second operand when unordered or equal
    with _err():
        return np.where(F(a) < F(b), a, b)


def fmax(a, b):
    with _err():
        return np.where(F(a) > F(b), a, b)


def cvtt(a):                            # _mm_cvttps_epi32: 0x80000000 for NaN and out of range
    with _err():
        x = F(a)
        ok = (x >= f32(-2147483648.0)) & (x < f32(2147483648.0))
        t = np.where(ok, x, f32(0)).astype(np.int64)
        return np.where(ok, t.astype(i32).view(u32), u32(0x80000000))


def i2f(a):
    return U(I(a).astype(f32))


def trunc(a):                           # SSE:2797-2800
    with _err():
        return U(np.trunc(F(a)))


def lerp(a, b, c):                      # VCOMMON.inl:532-536: b*c + a*(1-c)
    one = np.full_like(a, 0x3F800000)
    return mad(b, c, mul(a, sub(one, c)))


def cmp(op, a, b, integer=False):
    with _err():
        x, y = (I(a), I(b)) if integer else (F(a), F(b))
        return mask({"lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y, "eq": x == y, "ne": x != y}[op])


def idiv(a, b):                         # RT:958-973
    n, d = I(a).astype(np.int64), I(b).astype(np.int64)
    safe = np.where(d == 0, 1, d)
    q = np.trunc(n / safe) if False else (np.sign(n) * np.sign(safe) * (np.abs(n) // np.abs(safe)))
    q = np.where(d == 0, 0, q)
    q = np.where((d == -1) & (n == -2 ** 31), 2 ** 31 - 1, q)
    return q.astype(np.int64).astype(i32).view(u32)


def shl(a, b):                          # VectorVMPlatformGeneric.h:76-89: count & 31
    return (a.astype(np.uint64) << (b & u32(31)).astype(np.uint64)).astype(u32)


def shr(a, b):                          # logical, :61-74
    return a >> (b & u32(31))


ONE, MONE, ZERO = u32(0x3F800000), u32(0xBF800000), u32(0)

OPS = {
    "add": add, "sub": sub, "mul": mul, "div": div, "mad": mad, "lerp": lerp, "rsq": rsq, "sqrt": sqrt, "neg": neg,
    "abs": fabs,
    "ceil": lambda a: U(np.ceil(F(a))), "floor": lambda a: U(np.floor(F(a))), "trunc": trunc,
    "frac": lambda a: sub(a, trunc(a)),                                   # VCOMMON.inl:883-886
    "round": lambda a: U(np.rint(F(a))),                                   # SSE:2819-2822, nearest even
    "clamp": lambda a, b, c: fmin(fmax(a, b), c),                           # VCOMMON.inl:495-498
    "min": fmin, "max": fmax,
    "sign": lambda a: np.where(F(a) >= f32(0), ONE, MONE),                  # SSE:2911-2915
    "step": lambda a, b: np.where(F(sub(b, a)) >= f32(0), ONE, ZERO),       # RT:1939
    "cmplt": lambda a, b: cmp("lt", a, b), "cmple": lambda a, b: cmp("le", a, b), "cmpgt": lambda a, b: cmp("gt", a, b),
    "cmpge": lambda a, b: cmp("ge", a, b), "cmpeq": lambda a, b: cmp("eq", a, b), "cmpneq": lambda a, b: cmp("ne", a, b),
    "select": sel,
    "addi": lambda a, b: a + b, "subi": lambda a, b: a - b, "muli": lambda a, b: a * b, "divi": idiv,
    "clampi": lambda a, b, c: UI(np.minimum(np.maximum(I(a), I(b)), I(c))),
    "mini": lambda a, b: UI(np.minimum(I(a), I(b))), "maxi": lambda a, b: UI(np.maximum(I(a), I(b))),
    "absi": lambda a: np.where(I(a) < 0, u32(0) - a, a), "negi": lambda a: u32(0) - a,
    "signi": lambda a: np.where(I(a) >= 0, u32(1), u32(0xFFFFFFFF)),
    "cmplti": lambda a, b: cmp("lt", a, b, True), "cmplei": lambda a, b: cmp("le", a, b, True),
    "cmpgti": lambda a, b: cmp("gt", a, b, True), "cmpgei": lambda a, b: cmp("ge", a, b, True),
    "cmpeqi": lambda a, b: cmp("eq", a, b, True), "cmpneqi": lambda a, b: cmp("ne", a, b, True),
    "bit_and": lambda a, b: a & b, "bit_or": lambda a, b: a | b, "bit_xor": lambda a, b: a ^ b, "bit_not": lambda a: ~a,
    "logic_and": lambda a, b: a & b, "logic_or": lambda a, b: a | b, "logic_xor": lambda a, b: a ^ b, "logic_not": lambda a: ~a,
    "bit_lshift": shl, "bit_rshift": shr,
    "f2i": cvtt, "i2f": i2f,
    "f2b": lambda a: cmp("gt", a, np.zeros_like(a)),
    "b2f": lambda a: sel(a, np.full_like(a, ONE), np.zeros_like(a)),
    "i2b": lambda a: mask(I(a) > 0), "b2i": lambda a: a & u32(1),
    # merged ops, RT:1999-2074
    "cmplt_select": lambda a, b, c, d: sel(cmp("lt", a, b), c, d), "cmple_select": lambda a, b, c, d: sel(cmp("le", a, b), c, d),
    "cmpeq_select": lambda a, b, c, d: sel(cmp("eq", a, b), c, d),
    "cmplti_select": lambda a, b, c, d: sel(cmp("lt", a, b, True), c, d),
    "cmplei_select": lambda a, b, c, d: sel(cmp("le", a, b, True), c, d),
    "cmpeqi_select": lambda a, b, c, d: sel(cmp("eq", a, b, True), c, d),
    "mad_add": lambda a, b, c, d: add(mad(a, b, c), d), "mad_sub0": lambda a, b, c, d: sub(mad(a, b, c), d),
    "mad_sub1": lambda a, b, c, d: sub(d, mad(a, b, c)), "mad_mul": lambda a, b, c, d: mul(mad(a, b, c), d),
    "mad_mad0": lambda a, b, c, d, e: mad(d, e, mad(a, b, c)), "mad_mad1": lambda a, b, c, d, e: mad(mad(a, b, c), d, e),
    "mul_mad0": lambda a, b, c, d: mad(mul(a, b), c, d), "mul_mad1": lambda a, b, c, d: mad(c, d, mul(a, b)),
    "mul_add": lambda a, b, c: add(mul(a, b), c), "mul_sub0": lambda a, b, c: sub(mul(a, b), c),
    "mul_sub1": lambda a, b, c: sub(c, mul(a, b)), "mul_mul": lambda a, b, c: mul(mul(a, b), c),
    "mul_max": lambda a, b, c: fmax(mul(a, b), c),
    "add_mad1": lambda a, b, c, d: mad(c, d, add(a, b)), "add_add": lambda a, b, c: add(add(a, b), c),
    "sub_cmplt1": lambda a, b, c: cmp("lt", c, sub(a, b)), "sub_neg": lambda a, b: neg(sub(a, b)),
    "sub_mul": lambda a, b, c: mul(sub(a, b), c),
    "div_mad0": lambda a, b, c, d: mad(div(a, b), c, d), "div_f2i": lambda a, b: cvtt(div(a, b)),
    "div_mul": lambda a, b, c: mul(div(a, b), c),
    "muli_addi": lambda a, b, c: a * b + c, "addi_bit_rshift": lambda a, b, c: shr(a + b, c),
    "addi_muli": lambda a, b, c: (a + b) * c,
    "i2f_div0": lambda a, b: div(i2f(a), b), "i2f_div1": lambda a, b: div(b, i2f(a)), "i2f_mul": lambda a, b: mul(i2f(a), b),
    "i2f_mad0": lambda a, b, c: mad(i2f(a), b, c), "i2f_mad1": lambda a, b, c: mad(a, b, i2f(c)),
    "f2i_select1": lambda a, b, c: sel(a, cvtt(b), c), "f2i_maxi": lambda a, b: UI(np.maximum(I(cvtt(a)), I(b))),
    "f2i_addi": lambda a, b: cvtt(a) + b,
    "bit_and_i2f": lambda a, b: i2f(a & b), "bit_rshift_bit_and": lambda a, b, c: shr(a, b) & c,
    "neg_cmplt": lambda a, b: cmp("lt", neg(a), b), "bit_or_muli": lambda a, b, c: (a | b) * c,
    "bit_lshift_bit_or": lambda a, b, c: shl(a, b) | c, "max_f2i": lambda a, b: cvtt(fmax(a, b)),
    "select_mul": lambda a, b, c, d: mul(sel(a, b, c), d), "select_add": lambda a, b, c, d: add(sel(a, b, c), d),
    "mul_2x": lambda a, b: (mul(a, b), mul(a, b)), "b2i_2x": lambda a: (a & u32(1), a & u32(1)),
}
for _c in ("lt", "le", "gt", "ge", "eq", "ne"):
    for _i in (False, True):
        OPS["cmp%s%s_logic_and" % (_c, "i" if _i else "")] = (lambda c_, i_: lambda a, b, c: cmp(c_, a, b, i_) & c)(_c, _i)
        OPS["cmp%s%s_logic_or" % (_c, "i" if _i else "")] = (lambda c_, i_: lambda a, b, c: cmp(c_, a, b, i_) | c)(_c, _i)

# ops whose float result may be a NaN: payload bits are not compared (the reference pins none; x86 and the GPU pick the
# propagated payload by different rules)
def canon_nan(bits):
    bits = np.asarray(bits, u32)
    is_nan = ((bits & u32(0x7F800000)) == u32(0x7F800000)) & ((bits & u32(0x007FFFFF)) != 0)
    return np.where(is_nan, u32(0x7FC00000), bits)


# results that are integers or masks: a NaN-looking pattern is data there and must match exactly
INTEGER_RESULT = set("""cmplt cmple cmpgt cmpge cmpeq cmpneq addi subi muli divi clampi mini maxi absi negi signi cmplti cmplei
cmpgti cmpgei cmpeqi cmpneqi bit_and bit_or bit_xor bit_not bit_lshift bit_rshift logic_and logic_or logic_xor logic_not f2i f2b
i2b b2i sub_cmplt1 div_f2i muli_addi addi_bit_rshift addi_muli b2i_2x f2i_select1 f2i_maxi f2i_addi bit_rshift_bit_and neg_cmplt
bit_or_muli bit_lshift_bit_or max_f2i select cmplt_select cmple_select cmpeq_select cmplti_select cmplei_select cmpeqi_select
abs b2f""".split()) | {n for n in OPS if "_logic_" in n}
