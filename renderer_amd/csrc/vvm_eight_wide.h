// vvm_eight_wide.h -- the eight-instances-per-lane kernels' tile program: part of vvm_kernels.hip (included there, after the helpers
// it shares with the other kernels: Ctx, Pending, acquire_rank_and_publish, resolve_tile, store_tile, fetch_slow, the opcode case lists).
//
// These kernels run plain scripts (no transcendental / noise / ID instructions, every register resident): the light one, for scripts
// of eight registers and eight outputs, three workgroups per CU; and the general one, sixteen registers, two datasets, two per CU.
// The four-wide kernel for plain scripts of 17 to 43 registers has the same tile program with V = 4: its file is 176 VGPRs, 44
// registers of four instance slots (the compiled four-wide kernels keep 32 and the rest in LDS slots).
// Their register file lies in FIXED VGPRs, register-major (the V instance slots of register i are V consecutive VGPRs), and
// runs of ordinary instructions are executed by the hand-written routines of vvm_threaded.h; everything else of a tile (stream
// loads, acquireindex, output staging, the instructions without a routine) is the C++ below, which reaches the file through three
// small assembly helpers.  The compiler sees the file as three or four opaque vector values that every such statement takes and
// returns in the same physical registers, so it keeps them there.
#pragma once

// LIGHT:   registers 0..7 and the two riding ones (VVM_LIGHT_RIDE_REG, +1) = v72..v151, temporaries v152..v167 (budget: 168 VGPRs)
// general: registers 0..15 = v112..v239, temporaries v240..v255 (budget: 256 VGPRs)
// V = 4:   registers 0..43 = v64..v239, temporaries v240..v255 (budget: 256 VGPRs; the C++ side has a tile's stream loads of four
//          values per lane in flight where the eight-wide kernels have eight, and gets by with the 64 registers below the file)
// V = 4 with the transcendental / noise / ID instructions (FULL): registers 0..31 = v112..v239 (the general file, four VGPRs per
//          register): those functions are C++ and need the registers below
template <bool LIGHT> struct File8;
template <> struct File8<true> { vec32f f0, f1; vec16f f2; };
template <> struct File8<false> { vec32f f0, f1, f2, f3; };
struct File4 { vec32f f0, f1, f2, f3, f4; vec16f f5; };
template <int V, bool LIGHT, bool FULL> struct FileFor { typedef File8<LIGHT> type; };      // (V = 4, FULL: the general file as 32 registers of four slots)
template <> struct FileFor<4, false, false> { typedef File4 type; };
template <int V, bool LIGHT, bool FULL> using FileOf = typename FileFor<V, LIGHT, FULL>::type;
#define F4_SET ".set VT_FB, 64\n .set VT_T, 240\n .set VT_Q, 248\n"
#define F4_RW(F) "+{v[64:95]}"((F).f0), "+{v[96:127]}"((F).f1), "+{v[128:159]}"((F).f2), "+{v[160:191]}"((F).f3), "+{v[192:223]}"((F).f4), "+{v[224:239]}"((F).f5)
#define F4_RO(F) "{v[64:95]}"((F).f0), "{v[96:127]}"((F).f1), "{v[128:159]}"((F).f2), "{v[160:191]}"((F).f3), "{v[192:223]}"((F).f4), "{v[224:239]}"((F).f5)
#define F8L_SET ".set VT_FB, 72\n .set VT_T, 152\n .set VT_Q, 160\n"
#define F8G_SET ".set VT_FB, 112\n .set VT_T, 240\n .set VT_Q, 248\n"
#define F8L_RW(F) "+{v[72:103]}"((F).f0), "+{v[104:135]}"((F).f1), "+{v[136:151]}"((F).f2)
#define F8L_RO(F) "{v[72:103]}"((F).f0), "{v[104:135]}"((F).f1), "{v[136:151]}"((F).f2)
#define F8L_WO(F) "={v[72:103]}"((F).f0), "={v[104:135]}"((F).f1), "={v[136:151]}"((F).f2)
#define F8G_RW(F) "+{v[112:143]}"((F).f0), "+{v[144:175]}"((F).f1), "+{v[176:207]}"((F).f2), "+{v[208:239]}"((F).f3)
#define F8G_RO(F) "{v[112:143]}"((F).f0), "{v[144:175]}"((F).f1), "{v[176:207]}"((F).f2), "{v[208:239]}"((F).f3)
#define F8G_WO(F) "={v[112:143]}"((F).f0), "={v[144:175]}"((F).f1), "={v[176:207]}"((F).f2), "={v[208:239]}"((F).f3)
#define F8L_TEMPS "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
#define F8G_TEMPS "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define F8_SGPRS "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", \
	"s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc"

// the file as zeros (once per workgroup: a program reads only what it or the tile's loads have written, so nothing depends on it)
template <bool LIGHT> DEV void file_clear(File8<LIGHT>& F)
{
	F.f0 = 0.0f; F.f1 = 0.0f; F.f2 = 0.0f;
	if constexpr (!LIGHT) F.f3 = 0.0f;
}
DEV void file_clear(File4& F) { F.f0 = 0.0f; F.f1 = 0.0f; F.f2 = 0.0f; F.f3 = 0.0f; F.f4 = 0.0f; F.f5 = 0.0f; }
// the V instance slots of the register whose first VGPR is FILE + byte (byte = V x register index, wave-uniform)
template <int V, bool LIGHT, bool FULL> DEV void file_read(const FileOf<V, LIGHT, FULL>& F, uint32_t byte, float (&out)[V])
{
#define F8_READ_TEXT4 \
	"s_set_gpr_idx_on %[i], gpr_idx(SRC0)\n v_mov_b32 %0, v[VT_FB+0]\n v_mov_b32 %1, v[VT_FB+1]\n v_mov_b32 %2, v[VT_FB+2]\n v_mov_b32 %3, v[VT_FB+3]\n"
#define F8_READ_TEXT8 F8_READ_TEXT4 "v_mov_b32 %4, v[VT_FB+4]\n v_mov_b32 %5, v[VT_FB+5]\n v_mov_b32 %6, v[VT_FB+6]\n v_mov_b32 %7, v[VT_FB+7]\n"
	if constexpr (V == 4) {
		if constexpr (FULL) asm volatile(F8G_SET F8_READ_TEXT4 "s_set_gpr_idx_off\n" : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]) : [i] "s"(byte), F8G_RO(F) : "scc");
		else asm volatile(F4_SET F8_READ_TEXT4 "s_set_gpr_idx_off\n" : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]) : [i] "s"(byte), F4_RO(F) : "scc");
	} else if constexpr (LIGHT)
		asm volatile(F8L_SET F8_READ_TEXT8 "s_set_gpr_idx_off\n" : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
		             : [i] "s"(byte), F8L_RO(F) : "scc");
	else
		asm volatile(F8G_SET F8_READ_TEXT8 "s_set_gpr_idx_off\n" : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
		             : [i] "s"(byte), F8G_RO(F) : "scc");
#undef F8_READ_TEXT4
#undef F8_READ_TEXT8
}
template <int V, bool LIGHT, bool FULL> DEV void file_write(FileOf<V, LIGHT, FULL>& F, uint32_t byte, const float (&in)[V])
{
#define F8_WRITE_TEXT4 \
	"s_set_gpr_idx_on %[i], gpr_idx(DST)\n v_mov_b32 v[VT_FB+0], %[a0]\n v_mov_b32 v[VT_FB+1], %[a1]\n v_mov_b32 v[VT_FB+2], %[a2]\n v_mov_b32 v[VT_FB+3], %[a3]\n"
#define F8_WRITE_TEXT8 F8_WRITE_TEXT4 "v_mov_b32 v[VT_FB+4], %[a4]\n v_mov_b32 v[VT_FB+5], %[a5]\n v_mov_b32 v[VT_FB+6], %[a6]\n v_mov_b32 v[VT_FB+7], %[a7]\n"
#define F8_WRITE_IN4 [i] "s"(byte), [a0] "v"(in[0]), [a1] "v"(in[1]), [a2] "v"(in[2]), [a3] "v"(in[3])
	if constexpr (V == 4) {
		if constexpr (FULL) asm volatile(F8G_SET F8_WRITE_TEXT4 "s_set_gpr_idx_off\n" : F8G_RW(F) : F8_WRITE_IN4 : "scc");
		else asm volatile(F4_SET F8_WRITE_TEXT4 "s_set_gpr_idx_off\n" : F4_RW(F) : F8_WRITE_IN4 : "scc");
	} else if constexpr (LIGHT)
		asm volatile(F8L_SET F8_WRITE_TEXT8 "s_set_gpr_idx_off\n" : F8L_RW(F) : F8_WRITE_IN4, [a4] "v"(in[4]), [a5] "v"(in[5]), [a6] "v"(in[6]), [a7] "v"(in[7]) : "scc");
	else
		asm volatile(F8G_SET F8_WRITE_TEXT8 "s_set_gpr_idx_off\n" : F8G_RW(F) : F8_WRITE_IN4, [a4] "v"(in[4]), [a5] "v"(in[5]), [a6] "v"(in[6]), [a7] "v"(in[7]) : "scc");
#undef F8_WRITE_TEXT4
#undef F8_WRITE_TEXT8
#undef F8_WRITE_IN4
}

// Runs the program from byte offset `pc` on: every ordinary instruction that has a routine (vvm_threaded.h), until one that has
// none.  Returns that instruction's four linker words in x; pc = the byte offset of the instruction after it.
// The text: entry (the addresses of the two routine tables, the first two instructions into the buffers X = s[60:67] and Y =
// s[68:75], jump to the first one's routine), then the two tables.  Offset 0 of a table is the routine of an instruction that has
// none: it hands the instruction in its buffer back and jumps to the end (by address: the end is farther than a branch reaches).
// On the way out the load that is still in flight into the other buffer is waited for (its registers are the compiler's again).
#define F8_LEAVE(C) \
	"s_mov_b32 %[x0], s[" C "+0]\n s_mov_b32 %[x1], s[" C "+1]\n s_mov_b32 %[x2], s[" C "+2]\n s_mov_b32 %[x3], s[" C "+3]\n" \
	"s_getpc_b64 s[80:81]\n 2:\n s_add_u32 s80, s80, 9f - 2b\n s_addc_u32 s81, s81, 0\n s_setpc_b64 s[80:81]\n"
#define F8_RUN_TEXT(ROUTINES) \
	"s_getpc_b64 s[76:77]\n 1:\n" \
	"s_add_u32 s78, s76, 8f - 1b\n s_addc_u32 s79, s77, 0\n" \
	"s_add_u32 s76, s76, 7f - 1b\n s_addc_u32 s77, s77, 0\n" \
	"s_load_dwordx8 s[60:67], %[prog], %[pc]\n s_add_u32 %[pc], %[pc], 32\n" \
	"s_load_dwordx8 s[68:75], %[prog], %[pc]\n s_add_u32 %[pc], %[pc], 32\n" \
	"s_waitcnt lgkmcnt(0)\n" \
	"s_add_u32 s80, s76, s64\n s_addc_u32 s81, s77, 0\n s_setpc_b64 s[80:81]\n" \
	".p2align 8\n 7:\n" F8_LEAVE("60") \
	".set VT_C, 60\n .set VT_N, 68\n .set VT_NB, 78\n" ROUTINES("7b") \
	".p2align 8\n 8:\n" F8_LEAVE("68") \
	".set VT_C, 68\n .set VT_N, 60\n .set VT_NB, 76\n" ROUTINES("8b") \
	"9:\n s_waitcnt lgkmcnt(0)\n s_sub_u32 %[pc], %[pc], 32\n"
template <int V, bool LIGHT, bool FULL> DEV void run_ordinary8(FileOf<V, LIGHT, FULL>& F, const void* prog, uint32_t& pc, uint32_t uni, uint32_t (&x)[4])
{
	if constexpr (V == 4 && FULL)
		asm volatile(F8G_SET F8_RUN_TEXT(VVM_T4_ROUTINES) : [pc] "+s"(pc), [x0] "=&s"(x[0]), [x1] "=&s"(x[1]), [x2] "=&s"(x[2]), [x3] "=&s"(x[3]), F8G_RW(F)
		             : [prog] "s"(prog), [uni] "v"(uni) : F8G_TEMPS, F8_SGPRS, "memory");
	else if constexpr (V == 4)
		asm volatile(F4_SET F8_RUN_TEXT(VVM_T4_ROUTINES) : [pc] "+s"(pc), [x0] "=&s"(x[0]), [x1] "=&s"(x[1]), [x2] "=&s"(x[2]), [x3] "=&s"(x[3]), F4_RW(F)
		             : [prog] "s"(prog), [uni] "v"(uni) : F8G_TEMPS, F8_SGPRS, "memory");
	else if constexpr (LIGHT)
		asm volatile(F8L_SET F8_RUN_TEXT(VVM_T_ROUTINES) : [pc] "+s"(pc), [x0] "=&s"(x[0]), [x1] "=&s"(x[1]), [x2] "=&s"(x[2]), [x3] "=&s"(x[3]), F8L_RW(F)
		             : [prog] "s"(prog), [uni] "v"(uni) : F8L_TEMPS, F8_SGPRS, "memory");
	else
		asm volatile(F8G_SET F8_RUN_TEXT(VVM_T_ROUTINES) : [pc] "+s"(pc), [x0] "=&s"(x[0]), [x1] "=&s"(x[1]), [x2] "=&s"(x[2]), [x3] "=&s"(x[3]), F8G_RW(F)
		             : [prog] "s"(prog), [uni] "v"(uni) : F8G_TEMPS, F8_SGPRS, "memory");
}

// Registers of a four-wide program beyond the 44 of the file (FULL: 32) live in LDS slots, [slot][thread] (as the compiled kernels'
// reg_read / reg_spill have them), and instructions that touch one have no routine.  (A program of at most sixteen registers that
// runs four-wide, for its second chain, keeps the eight VGPRs per register it is encoded with and uses four of them: Ctx::reg_shift.)
#define VVM_FILE4_REGS(FULL) ((FULL) ? 32u : 44u)
// operand fetch for the C++ side (the class and the index are wave-uniform)
template <int V, bool LIGHT, bool FULL> DEV void fetch8(const FileOf<V, LIGHT, FULL>& F, uint32_t uni, const Ctx& c, uint32_t o, float (&out)[V])
{
	if (o < (1u << 13)) {                       // VVM_OPC_VREG
		if constexpr (V == 8) {
			file_read<8, LIGHT, FULL>(F, (o & 31u) * 8u, out);
		} else {
			if (o < VVM_FILE4_REGS(FULL)) {
				file_read<4, LIGHT, FULL>(F, o << c.reg_shift, out);
			} else {
				const f4a v = *(const f4a*)(c.spill + ((o - VVM_FILE4_REGS(FULL)) * VVM_BLOCK_THREADS + c.tid) * 4u);
				out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
			}
		}
	} else if (o < (2u << 13)) {                // VVM_OPC_UNIF
		const float v = bitsf((uint32_t)__builtin_amdgcn_readlane((int)uni, (int)(o & 63u)));
#pragma unroll
		for (int k = 0; k < V; ++k) out[k] = v;
	} else {
		fetch_slow<V>(c, o, out);
	}
}
// the V values of a result to register `idx` of the program
template <int V, bool LIGHT, bool FULL> DEV void store8(FileOf<V, LIGHT, FULL>& F, const Ctx& c, uint32_t idx, const float (&in)[V])
{
	if constexpr (V == 8) {
		file_write<8, LIGHT, FULL>(F, (idx & 31u) * 8u, in);
	} else {
		if (idx < VVM_FILE4_REGS(FULL)) {
			file_write<4, LIGHT, FULL>(F, idx << c.reg_shift, in);
		} else {
			f4a v; v.x = in[0]; v.y = in[1]; v.z = in[2]; v.w = in[3];
			*(f4a*)(c.spill + ((idx - VVM_FILE4_REGS(FULL)) * VVM_BLOCK_THREADS + c.tid) * 4u) = v;
		}
	}
}

// One ordinary instruction that has no routine (an op without one, or an operand that is neither a register of the file nor a uniform
// register), by the same cases as the other kernels' interpreter loop.
template <int V, bool LIGHT, bool FULL> DEV void exec_ordinary8(FileOf<V, LIGHT, FULL>& F, uint32_t uni, const Ctx& c, const uint32_t (&w)[4])
{
	const uint32_t op = w[0] & 0xFFu, nin = (w[0] >> 8) & 7u, aux = (w[0] >> 16) & 0xFFu, aux2 = w[0] >> 24;
	const uint32_t o[5] = {w[1] & 0xFFFFu, w[1] >> 16, w[2] & 0xFFFFu, w[2] >> 16, w[3] & 0xFFFFu};
	const uint32_t dst = w[3] >> 16;
	(void)aux;
	float a[V], b[V], cc[V], dd[V], ee[V], r[V];
#pragma unroll
	for (int k = 0; k < V; ++k) { a[k] = b[k] = cc[k] = dd[k] = ee[k] = 0.0f; r[k] = 0.0f; }
	if (nin >= 1) fetch8<V, LIGHT, FULL>(F, uni, c, o[0], a);
	if (nin >= 2) fetch8<V, LIGHT, FULL>(F, uni, c, o[1], b);
	if (nin >= 3) fetch8<V, LIGHT, FULL>(F, uni, c, o[2], cc);
	if (nin >= 4) fetch8<V, LIGHT, FULL>(F, uni, c, o[3], dd);
	if (nin >= 5) fetch8<V, LIGHT, FULL>(F, uni, c, o[4], ee);
	uint32_t inst_i = c.inst0;
	asm volatile("" : "+v"(inst_i));
#define LANES for (int k = 0; k < V; ++k)
#define A a[k]
#define B b[k]
#define C cc[k]
#define D dd[k]
#define E ee[k]
#define IA ((int32_t)fbits(a[k]))
#define IB ((int32_t)fbits(b[k]))
#define IC ((int32_t)fbits(cc[k]))
#define UA fbits(a[k])
#define UB fbits(b[k])
#define UC fbits(cc[k])
#define SETF(expr) { _Pragma("unroll") LANES r[k] = (expr); } break
#define SETI(expr) { _Pragma("unroll") LANES r[k] = bitsf((uint32_t)(expr)); } break
#define SETM(expr) { _Pragma("unroll") LANES r[k] = mask(expr); } break
	if (nin == 2) { switch (op) { VVM_CASES_NIN2 } }
	else if (nin == 3) { switch (op) { VVM_CASES_NIN3 } }
	else if (nin == 1) { switch (op) { VVM_CASES_NIN1 } }
	else if (nin == 4) { switch (op) { VVM_CASES_NIN4 } }
	else if (nin == 0) { switch (op) { VVM_CASES_NIN0 } }
	else { switch (op) { VVM_CASES_NIN5 } }
#undef LANES
#undef A
#undef B
#undef C
#undef D
#undef E
#undef IA
#undef IB
#undef IC
#undef UA
#undef UB
#undef UC
#undef SETF
#undef SETI
#undef SETM
	store8<V, LIGHT, FULL>(F, c, VVM_OPND_IDX(dst), r);
}

// One tile of eight instances per lane: two halves of 1024 instances, lane t holding instances [4 t, 4 t + 4) of each (slots 0..3 and
// 4..7), so that every stream access is a 128-bit access of four consecutive instances.  Same steps as run_tile.  (V = 4: one such
// half, the tile of the four-wide kernels.)
template <int V, bool LIGHT, bool FULL, bool RIDE = false>
DEV void run_tile8(const void* prog_in, gu64* status, bool last_tile, const Ctx& c, unsigned char* lds, uint32_t tile_local, uint32_t tag, VvmLaunchState* ls, uint32_t dbg, Pending& pend,
                   uint32_t& resolves, const u64 (&pre)[VVM_LB_WORDS_PER_LANE], bool have_pre, FileOf<V, LIGHT, FULL>& F)
{
	static_assert((V == 8 && !FULL) || (V == 4 && !LIGHT), "eight instance slots, or four (there also with the transcendental / noise / ID instructions)");
	constexpr bool TWO = !LIGHT && !FULL;      // the general kernels of plain scripts run two acquired datasets in one pass
	constexpr int HALVES = V == 8 ? 2 : 1;
	// (the program's address as a value of unknown origin: otherwise the compiler, short of scalar registers, loads it from the Exec's
	// descriptor again in front of every entry into the routines and waits for that load)
	const void* prog = prog_in;
	asm volatile("" : "+s"(prog));
	uint32_t* lds_ctrl = (uint32_t*)lds;
	uint32_t t_keep = 0, t_rank = 0, t_rank_b = 0;      // (general kernel; the light one keeps them in LDS: light_ride_word)
	uint32_t kw_new = 0;
	bool first_special = true;
	uint32_t t_total = 0, t_base = 0;
	bool t_pending = false;
	uint32_t u_keep = 0, u_rank = 0, u_rank_b = 0, u_total = 0, u_base = 0;
	bool u_pending = false;
	const uint32_t exec_tiles = TWO ? DESC_U32(c.dp, num_tiles) : 0u;
	uint32_t stage_end = 0;
	const uint32_t uni = fill_uniforms<0>(0u, c.consts, DESC_U32(c.dp, num_unif));
	const u64 tl0 = (dbg & 128u) ? __builtin_amdgcn_s_memtime() : 0;

	// Prologue: the stream loads.  A light program starts with exactly one load instruction for up to eight 32-bit streams into
	// registers 0..7 (the linker sees to it): every stream goes straight to its register; the slots past the count read through an
	// empty descriptor (zeros, no memory access).  Other programs: any number of load instructions, each into consecutive registers.
	const void* load_ptrs = (const void*)DESC_TABLE(c, load_ptrs);
	uint32_t pc0 = 0;
	if constexpr (LIGHT) {
		const uint32_t w0 = sload32(prog, 0);
		const uint32_t cnt = (w0 >> 16) & 0xFFu;
		uint64_t ptr[8];
		sload64x8(load_ptrs, sload32(prog, 1), ptr);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const bool used = (uint32_t)j < cnt && !(dbg & 2u);
			const rsrc_t rs = make_rsrc(ptr[j] + 4ull * c.tile_first, used ? 4u * c.valid : 0u);
			const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(c.tid * 16u), 0, AUX_STREAM_LOAD);
			const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(VVM_HALF_TILE * 4u + c.tid * 16u), 0, AUX_STREAM_LOAD);
			const float v[8] = {bitsf(lo.x), bitsf(lo.y), bitsf(lo.z), bitsf(lo.w), bitsf(hi.x), bitsf(hi.y), bitsf(hi.z), bitsf(hi.w)};
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				if (j < 4) F.f0[j * 8 + k] = v[k];
				else F.f1[(j - 4) * 8 + k] = v[k];
			}
		}
		pc0 = 1;
	} else {
		for (;; ++pc0) {
			const uint32_t w0 = sload32(prog, pc0 * (VVM_INS_STRIDE / 4u) + 0);
			const uint32_t lop = w0 & 0xFFu;
			if (lop != VVM_DOP_LOADN && lop != VVM_DOP_LOADN16) break;
			const uint32_t w1 = sload32(prog, pc0 * (VVM_INS_STRIDE / 4u) + 1);
			const uint32_t cnt = (w0 >> 16) & 0xFFu, first_reg = w0 >> 24;
			uint64_t ptr[8];
			sload64x8(load_ptrs, w1, ptr);
			float st[8][V];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
#pragma unroll
				for (int k = 0; k < V; ++k) st[j][k] = 0.0f;
				if ((uint32_t)j < cnt && !(dbg & 2u)) {
					if (lop == VVM_DOP_LOADN) load_stream32<V>(c, ptr[j], st[j]);
					else load_stream16<V>(c, ptr[j], st[j]);
				}
			}
			// (slots past the count write zeros to the last used slot's register, from the last one down: plain writes, no branch)
#pragma unroll
			for (int j = 7; j >= 0; --j) store8<V, LIGHT, FULL>(F, c, first_reg + min((uint32_t)j, cnt - 1u), st[j]);
		}
	}

	// Light kernel: the look-back words that wave 0 polled ahead of time (have_pre) wait in LDS while the program runs, not in eight
	// registers per lane: the kernel has 168, 96 of them the routines', and what does not fit goes to scratch memory.  (The words were
	// requested before the stream loads and arrive before them.  Half of them lie in the sink words, which are written only while
	// outputs are staged: after the previous tile's and before this one's.)
	static_assert(VVM_LB_WORDS_PER_LANE == 4, "two words per lane in the sink words, two in the kilobyte behind the keep words");
	u64* const pre_park[4] = {(u64*)(lds + LDS_STAGE_SINK_OFF(V, FULL, true)) + (c.tid & 63u), (u64*)(lds + LDS_STAGE_SINK_OFF(V, FULL, true)) + 64u + (c.tid & 63u),
	                          (u64*)(lds + LDS_LIGHT_PRE_OFF) + (c.tid & 63u), (u64*)(lds + LDS_LIGHT_PRE_OFF) + 64u + (c.tid & 63u)};
	if constexpr (LIGHT) {
		if (have_pre && c.wave == 0) {
#pragma unroll
			for (int j = 0; j < VVM_LB_WORDS_PER_LANE; ++j) *pre_park[j] = pre[j];
		}
	}
	const u64 tl1 = (dbg & 128u) ? __builtin_amdgcn_s_memtime() : 0;
	uint32_t pc = pc0 * VVM_INS_STRIDE;
	for (;;) {
		uint32_t x[4];
		run_ordinary8<V, LIGHT, FULL>(F, prog, pc, uni, x);
		const uint32_t op = x[0] & 0xFFu, aux = (x[0] >> 16) & 0xFFu, aux2 = x[0] >> 24;
		if (((x[0] >> 8) & 7u) != VVM_NIN_SPECIAL) {
			exec_ordinary8<V, LIGHT, FULL>(F, uni, c, x);
			continue;
		}
		const uint32_t o0 = x[1] & 0xFFFFu, o1 = x[1] >> 16, o2 = x[2] & 0xFFFFu, o4 = x[3] & 0xFFFFu, o5 = x[3] >> 16;

		if (op == VVM_DOP_ACQUIRE) {
			// sign bit of the predicate = keep (VectorVMRuntime.cpp:838-841); instances past the end never survive (:846-857).  aux2 != 0: the
			// compare that made the predicate is folded in: keep = o0 <cmp> o1
			float p[V];
			fetch8<V, LIGHT, FULL>(F, uni, c, o0, p);
			if (aux2 != VVM_ACQ_SIGN) {
				float q[V];
				fetch8<V, LIGHT, FULL>(F, uni, c, o1, q);
#define ACQ_CMP(expr) { _Pragma("unroll") for (int k = 0; k < V; ++k) { const float A = p[k], B = q[k]; const int32_t IA = (int32_t)fbits(A), IB = (int32_t)fbits(B); (void)IA; (void)IB; p[k] = mask(expr); } } break
				switch (aux2) {
				case VVM_ACQ_LT: ACQ_CMP(A < B);
				case VVM_ACQ_LE: ACQ_CMP(A <= B);
				case VVM_ACQ_GT: ACQ_CMP(A > B);
				case VVM_ACQ_GE: ACQ_CMP(A >= B);
				case VVM_ACQ_EQ: ACQ_CMP(A == B);
				case VVM_ACQ_NE: ACQ_CMP(A != B);
				case VVM_ACQ_LTI: ACQ_CMP(IA < IB);
				case VVM_ACQ_LEI: ACQ_CMP(IA <= IB);
				case VVM_ACQ_GTI: ACQ_CMP(IA > IB);
				case VVM_ACQ_GEI: ACQ_CMP(IA >= IB);
				case VVM_ACQ_EQI: ACQ_CMP(IA == IB);
				default: ACQ_CMP(IA != IB);
				}
#undef ACQ_CMP
			}
			uint32_t kb = 0;
#pragma unroll
			for (int k = 0; k < V; ++k)
				kb |= ((fbits(p[k]) >> 31) & ((c.inst0 + slot_offset<V>(k) < c.num_instances) ? 1u : 0u)) << k;
			const uint32_t chain = TWO ? aux : 0u;
			const uint3 rt = acquire_rank_and_publish<HALVES>(kb, c, lds_ctrl, status + (size_t)chain * exec_tiles, tile_local, tag, dbg, chain);
			if (TWO && chain) { u_keep = kb; u_rank = rt.x; u_rank_b = rt.y; u_total = SCALAR(rt.z); u_pending = true; }
			else if constexpr (LIGHT) { kw_new = kb | (rt.x << 8) | (rt.y << 20); t_total = SCALAR(rt.z); t_pending = true; }
			else { t_keep = kb; t_rank = rt.x; t_rank_b = rt.y; t_total = SCALAR(rt.z); t_pending = true; }
		}
		// The previous tile is finished here: after this tile has published its own count (every later tile's look-back waits for
		// that word, so it goes out first), and before this tile needs the staging slots.  (A light program writes its two riding
		// registers only after its acquireindex: the previous tile's values are still in them here.)
		if (pend.active) {
			u64 pre_here[VVM_LB_WORDS_PER_LANE] = {};
#pragma unroll
			for (int j = 0; j < VVM_LB_WORDS_PER_LANE; ++j) {
				if constexpr (LIGHT) { if (have_pre && c.wave == 0) pre_here[j] = *pre_park[j]; }
				else pre_here[j] = pre[j];
			}
			resolve_tile<V, TWO>(pend, c, lds_ctrl, resolves, tag, ls, dbg, pre_here, have_pre);
			float ride[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) { if constexpr (LIGHT) ride[k] = F.f2[k]; else ride[k] = 0.0f; }
			store_tile<V, FULL, LIGHT, TWO>(pend, c, lds, dbg, ride);
			__syncthreads();      // every thread has read its staged values: the slots may be written again
		}
		if constexpr (LIGHT) {
			// The lane's acquireindex result of THIS tile (keep bits, ranks) waits in LDS from here on, not in registers that would have to
			// survive the routines: the word of the previous tile has just been used (store_tile above), so its place is free.  A tile
			// starts without a result (0: nothing survives) and gets one at its acquireindex.
			if (op == VVM_DOP_ACQUIRE) *light_ride_word(lds, c.tid) = kw_new;
			else if (first_special) *light_ride_word(lds, c.tid) = 0u;
			first_special = false;
		}
		if (op == VVM_DOP_ACQUIRE) continue;
		if (op == VVM_OP_done) break;

		if (op == VVM_DOP_FLUSH) {
			Pending cur;
			cur.active = true; cur.unresolved = t_pending; cur.total = t_total; cur.base = t_base;
			cur.stage_end = stage_end; cur.last = last_tile;
			cur.tile_local = tile_local; cur.status = status; cur.out_list = c.out_list; cur.ride = 0;
			cur.unresolved1 = u_pending; cur.total1 = u_total; cur.base1 = u_base; cur.tiles = exec_tiles;
			resolve_tile<V, TWO>(cur, c, lds_ctrl, resolves, tag, ls, dbg, pre, false);
			const float none[16] = {};
			store_tile<V, FULL, false, TWO>(cur, c, lds, dbg, none);
			__syncthreads();
			t_base = cur.base; t_pending = false; stage_end = 0;
			u_base = cur.base1; u_pending = false;
			continue;
		}

		if (FULL && op == VVM_DOP_UPDATE_ID) {
			// the ID register of every instance of the tile goes to a staging slot as it is, with the lane's acquireindex result next to it;
			// apply_update_id does the rest once the tile's output position is known
			const uint32_t slot_index = o4 | (o5 << 16);
			float v[V];
			fetch8<V, LIGHT, FULL>(F, uni, c, o0, v);
			float* slot = (float*)(lds + LDS_STAGE_OFF(FULL)) + (slot_index & (VVM_STAGE_SLOTS - 1u)) * (VVM_BLOCK_THREADS * V) + c.tid * V;
			{ f4a q; q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3]; *(f4a*)slot = q; }
			uint32_t exists = 0;
#pragma unroll
			for (int k = 0; k < V; ++k) exists |= ((c.inst0 + slot_offset<V>(k) < c.num_instances) ? 1u : 0u) << k;
			((uint32_t*)(lds + LDS_STAGE_KEEP_OFF(V, FULL, LIGHT)))[c.tid] = t_rank | (t_keep << 16) | (exists << 24);
			stage_end = max(stage_end, slot_index + 1u);      // (the group's last slot: outputs staged after this one lie below it)
			continue;
		}

		// Output instruction: aux = pair count (1..3), sources o0..o2, w3 = output-list index of the first pair; the pair's staging slot is
		// that index modulo VVM_STAGE_SLOTS.  Without an acquireindex before it nothing survives (VectorVMRuntime.cpp:565, 769).
		{
			const bool second = TWO && aux2 != 0;
			uint32_t o_keep, o_rank, o_rank_b;
			if constexpr (LIGHT) {
				const uint32_t kw = *light_ride_word(lds, c.tid);
				o_keep = kw & 0xFFu; o_rank = (kw >> 8) & 0xFFFu; o_rank_b = kw >> 20;
			} else {
				o_keep = second ? u_keep : t_keep; o_rank = second ? u_rank : t_rank; o_rank_b = second ? u_rank_b : t_rank_b;
			}
			const uint32_t o_total = second ? u_total : t_total;
			const bool all_alive = (o_total == VVM_BLOCK_THREADS * V);
			const uint32_t kind = op - VVM_DOP_OUTPUT32;          // 0: 32-bit value  1: float -> half  2: low 16 bits as they are  3: half bits -> float
			const uint32_t first_slot = o4 | (o5 << 16);
			float* slot = (float*)(lds + LDS_STAGE_OFF(FULL)) + (first_slot & (VVM_STAGE_SLOTS - 1u)) * (VVM_BLOCK_THREADS * V);
			float* sink = (float*)(lds + LDS_STAGE_SINK_OFF(V, FULL, LIGHT)) + c.tid;
			uint64_t srcs = (uint64_t)o0 | ((uint64_t)o1 << 16) | ((uint64_t)o2 << 32);
			for (uint32_t j = 0; j < aux; ++j, srcs >>= 16, slot += VVM_BLOCK_THREADS * V) {
				float v[V];
				fetch8<V, LIGHT, FULL>(F, uni, c, (uint32_t)srcs & 0xFFFFu, v);
				if (kind & 1u) {
					if (kind == 1) {
#pragma unroll
						for (int k = 0; k < V; ++k) v[k] = bitsf(store_half(v[k]));
					} else {      // outputdata_float_from_half, whose handler in the reference is an empty stub (VectorVMRuntime.cpp:573-577)
#pragma unroll
						for (int k = 0; k < V; ++k) v[k] = load_half(fbits(v[k]));
					}
				}
				if (all_alive) {
					{ f4a q; q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3]; *(f4a*)(slot + c.tid * 4) = q; }
					if constexpr (V == 8) { f4a q; q.x = v[4]; q.y = v[5]; q.z = v[6]; q.w = v[7]; *(f4a*)(slot + VVM_HALF_TILE + c.tid * 4) = q; }
				} else {
					// survivor j of the lane goes to rank + j; a dead instance's value goes to the thread's sink dword (no per-lane branch)
					uint32_t r = o_rank;
#pragma unroll
					for (int k = 0; k < V; ++k) {
						if (V == 8 && k == 4) r = o_rank_b;
						const bool keep = (o_keep >> k) & 1u;
						float* q = keep ? (slot + r) : sink;
						*q = v[k];
						r += keep ? 1u : 0u;
					}
				}
			}
			stage_end = max(stage_end, first_slot + aux);
		}
	}
	if ((dbg & 128u) && c.wave == 0) {
		uint32_t* st = lds_ctrl + LC_STATS;
		const uint32_t s5 = st[5] + (uint32_t)((tl1 - tl0) >> 4), s6 = st[6] + (uint32_t)((__builtin_amdgcn_s_memtime() - tl1) >> 4);
		st[5] = s5; st[6] = s6;
	}
	// hand the rest of the tile's work to the next tile's load phase
	pend.active = (stage_end != 0) || t_pending || u_pending;
	pend.unresolved1 = u_pending; pend.total1 = u_total; pend.base1 = u_base; pend.tiles = exec_tiles;
	pend.unresolved = t_pending; pend.total = t_total; pend.base = t_base; pend.stage_end = stage_end; pend.last = last_tile;
	pend.tile_local = tile_local; pend.status = status; pend.out_list = c.out_list;
	pend.ride = LIGHT ? 0u : (t_keep | (t_rank << 8) | (t_rank_b << 20));
	if constexpr ((!LIGHT && V == 8) || RIDE) {
		// the lane's acquireindex result for vvm_through_kernel, straight from the tile that made it (the light kernel keeps the word for
		// its riding outputs and stores it in store_tile; four-wide: the instantiation that lean programs with transcendental functions run on)
		const uint64_t ride_base = sload64(c.out_list - 3, 0);
		if (ride_base != 0) {
			const rsrc_t rr = make_rsrc(ride_base + (uint64_t)tile_local * (VVM_BLOCK_THREADS * 4u), VVM_BLOCK_THREADS * 4u);
			__builtin_amdgcn_raw_buffer_store_b32(t_keep | (t_rank << 8) | (t_rank_b << 20), rr, (int)(c.tid * 4u), 0, 0);
		}
	}
	if constexpr (LIGHT) {
		// A tile that keeps at most three quarters of its instances leaves the last quarter of every staging slot unused (the staged
		// outputs are compacted: entries 0 .. total-1).  The two outputs that otherwise wait in registers go there, compacted as well
		// (512 survivors per slot end, three slot ends per output), so that store_tile writes them like the staged ones: 128 bits per lane
		// instead of one 32-bit store per survivor at its rank.  (The barrier inside resolve_tile lies between these writes and the reads.)
		if (stage_end != 0 && t_total <= VVM_LIGHT_STASH_MAX) {
			const uint64_t e6 = sload64(c.out_list, VVM_LIGHT_GROUP_SLOTS), e7 = sload64(c.out_list, VVM_LIGHT_GROUP_SLOTS + 1);
			if (((e6 | e7) & VVM_OUT_ADDR_MASK) != 0) {
				float* const stage = (float*)(lds + LDS_STAGE_OFF(FULL)) + VVM_LIGHT_STASH_MAX;
				float* const sink = (float*)(lds + LDS_STAGE_SINK_OFF(V, FULL, true)) + c.tid;
				const uint32_t kw = *light_ride_word(lds, c.tid);
				const uint32_t s_keep = kw & 0xFFu, s_rank = (kw >> 8) & 0xFFFu, s_rank_b = kw >> 20;
#pragma unroll
				for (uint32_t h = 0; h < 2; ++h) {
					uint32_t r = s_rank;
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						if (k == 4) r = s_rank_b;
						const bool keep = (s_keep >> k) & 1u;
						float* q = keep ? stage + (VVM_LIGHT_STASH_SLOTS * h + (r >> 9)) * (VVM_BLOCK_THREADS * V) + (r & 511u) : sink;
						*q = F.f2[h * 8 + k];
						r += keep ? 1u : 0u;
					}
				}
			}
		}
	}
}
