"""Opcode table of the VectorVM bytecode.

Numeric values and operand counts follow the reference's opcode list
(Runtime/VectorVM/Public/VectorVM.h:50-233).  ``OPS[name] = (code, n_in, n_out)``
describes the ops whose optimized encoding is ``n_in + n_out`` little-endian
``u16`` register indices, inputs first (decoders: VectorVMRuntime.cpp:1558-1904).
Ops with their own encoding are listed in ``SPECIAL``.
"""

_NAMES = """
done add sub mul div mad lerp rcp rsq sqrt neg abs exp exp2 log log2 sin cos tan asin acos atan atan2
ceil floor fmod frac trunc clamp min max pow round sign step random noise cmplt cmple cmpgt cmpge cmpeq
cmpneq select addi subi muli divi clampi mini maxi absi negi signi randomi cmplti cmplei cmpgti cmpgei
cmpeqi cmpneqi bit_and bit_or bit_xor bit_not bit_lshift bit_rshift logic_and logic_or logic_xor logic_not
f2i i2f f2b b2f i2b b2i inputdata_float inputdata_int32 inputdata_half inputdata_noadvance_float
inputdata_noadvance_int32 inputdata_noadvance_half outputdata_float outputdata_int32 outputdata_half
acquireindex external_func_call exec_index noise2D noise3D enter_stat_scope exit_stat_scope update_id
acquire_id half_to_float fasi iasf exec_indexf exec_index_addi cmplt_select cmple_select cmpeq_select
cmplti_select cmplei_select cmpeqi_select cmplt_logic_and cmple_logic_and cmpgt_logic_and cmpge_logic_and
cmpeq_logic_and cmpne_logic_and cmplti_logic_and cmplei_logic_and cmpgti_logic_and cmpgei_logic_and
cmpeqi_logic_and cmpnei_logic_and cmplt_logic_or cmple_logic_or cmpgt_logic_or cmpge_logic_or
cmpeq_logic_or cmpne_logic_or cmplti_logic_or cmplei_logic_or cmpgti_logic_or cmpgei_logic_or
cmpeqi_logic_or cmpnei_logic_or mad_add mad_sub0 mad_sub1 mad_mul mad_sqrt mad_mad0 mad_mad1 mul_mad0
mul_mad1 mul_add mul_sub0 mul_sub1 mul_mul mul_max mul_2x add_mad1 add_add sub_cmplt1 sub_neg sub_mul
div_mad0 div_f2i div_mul muli_addi addi_bit_rshift addi_muli b2i_2x i2f_div0 i2f_div1 i2f_mul i2f_mad0
i2f_mad1 f2i_select1 f2i_maxi f2i_addi fmod_add bit_and_i2f bit_rshift_bit_and neg_cmplt bit_or_muli
bit_lshift_bit_or random_add random_2x max_f2i select_mul select_add sin_cos outputdata_float_from_half
outputdata_half_from_half
""".split()

assert len(_NAMES) == 179
CODE = {n: i for i, n in enumerate(_NAMES)}
NAME = {i: n for i, n in enumerate(_NAMES)}
NUM_OPCODES = len(_NAMES)


def _arity():
    a = {}
    one_one = ("rcp rsq sqrt neg abs exp exp2 log log2 sin cos tan asin acos atan ceil floor frac trunc round sign "
               "absi negi signi bit_not logic_not f2i i2f f2b b2f i2b b2i random randomi half_to_float "
               "exec_index_addi").split()
    two_one = ("add sub mul div atan2 fmod min max pow step cmplt cmple cmpgt cmpge cmpeq cmpneq addi subi muli divi "
               "mini maxi cmplti cmplei cmpgti cmpgei cmpeqi cmpneqi bit_and bit_or bit_xor bit_lshift bit_rshift "
               "logic_and logic_or logic_xor sub_neg div_f2i i2f_div0 i2f_div1 i2f_mul f2i_maxi f2i_addi bit_and_i2f "
               "neg_cmplt max_f2i random_add").split()
    three_one = ("mad lerp clamp select clampi mad_sqrt mul_add mul_sub0 mul_sub1 mul_mul mul_max add_add sub_cmplt1 "
                 "sub_mul div_mul muli_addi addi_bit_rshift addi_muli i2f_mad0 i2f_mad1 f2i_select1 fmod_add "
                 "bit_rshift_bit_and bit_or_muli bit_lshift_bit_or").split()
    four_one = ("cmplt_select cmple_select cmpeq_select cmplti_select cmplei_select cmpeqi_select mad_add mad_sub0 "
                "mad_sub1 mad_mul mul_mad0 mul_mad1 add_mad1 div_mad0 select_mul select_add").split()
    five_one = "mad_mad0 mad_mad1".split()
    for n in one_one:
        a[n] = (1, 1)
    for n in two_one:
        a[n] = (2, 1)
    for n in three_one:
        a[n] = (3, 1)
    for n in four_one:
        a[n] = (4, 1)
    for n in five_one:
        a[n] = (5, 1)
    for cmp_ in "lt le gt ge eq ne".split():
        for suffix in ("", "i"):
            for logic in ("and", "or"):
                a["cmp%s%s_logic_%s" % (cmp_, suffix, logic)] = (3, 1)
    a["mul_2x"] = (2, 2)
    a["b2i_2x"] = (1, 2)
    a["random_2x"] = (1, 2)
    a["sin_cos"] = (1, 2)
    return a


OPS = {n: (CODE[n],) + io for n, io in _arity().items()}

# ops that read their operands as integers / produce integer results: only used by tests to pick edge values
INT_OPS = set("addi subi muli divi clampi mini maxi absi negi signi cmplti cmplei cmpgti cmpgei cmpeqi cmpneqi bit_and "
              "bit_or bit_xor bit_not bit_lshift bit_rshift logic_and logic_or logic_xor logic_not i2b b2i muli_addi "
              "addi_bit_rshift addi_muli b2i_2x bit_rshift_bit_and bit_or_muli bit_lshift_bit_or".split())

SPECIAL = {
    "exec_index": "dst",            # VectorVMRuntime.cpp:880-900
    "exec_indexf": "dst",           # :902-923
    "acquireindex": "pred,ds",      # :824-878
    "outputdata_float": "output",   # :744-822 (header macro :555-569)
    "outputdata_int32": "output",
    "outputdata_half": "output",    # :659-742
    "outputdata_float_from_half": "output",  # :573-577
    "outputdata_half_from_half": "output",   # :579-657
    "external_func_call": "extfn",  # :1442-1489
    "update_id": "r0,r1,ds",        # :1129-1402
    "acquire_id": "r0,r1,ds",       # :1404-1440
}

RANDOM_OPS = ("random", "randomi", "random_add", "random_2x")
