/*
 * vvm_oracle.c -- CPU restatement of the VectorVM runtime interpreter.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * shipped path (libvvm_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference tree holds no golden vector, known-answer test or fixture for this
 * path, and building or running the reference's own sources is not available to this project.  What
 * pins this file is (a) the hand-derived known answers in tests/ (fountain script, per-op edge values)
 * and (b) the fact that it uses the very SSE instructions the reference's math layer maps to
 * (_mm_min_ps, _mm_cvttps_epi32, _mm_round_ps, _mm_shuffle_epi8 ...), so their corner-case behaviour is
 * the hardware's, not a paraphrase.
 *
 * It follows the reference's structure, so that it can also serve as the timed CPU baseline:
 *   blob thaw                      VectorVMBridge.cpp:14-148, 181-199
 *   state allocation               VectorVMRuntime.cpp:2311-2417
 *   map-table caches               VectorVMRuntime.cpp:485-553
 *   page-sized chunking            VectorVMRuntime.cpp:2210-2261
 *   register pointer table         VectorVMRuntime.cpp:315-443, 2103-2133
 *   bytecode dispatch, 4-wide ops  VectorVMRuntime.cpp:1533-2175
 *   acquireindex / outputs         VectorVMRuntime.cpp:744-878 (half: 579-742)
 *   exec_index*, half_to_float     VectorVMRuntime.cpp:880-951, 1113-1127
 *   update_id / acquire_id         VectorVMRuntime.cpp:1129-1440
 *   external function call         VectorVMRuntime.cpp:1442-1489
 * Arithmetic follows Core's SSE layer (UnrealMathSSE.h, UnrealMathVectorCommon.h.inl,
 * GenericPlatformMath.{h,cpp}); each helper cites its lines.
 *
 * THIRD-PARTY ARITHMETIC NOT IN THE REFERENCE TREE: sin/cos/exp/log (and pow) call
 * ThirdParty/SSEMathFun/sse_mathfun_extension.h (included at Core/Private/Math/UnrealMathSSE.cpp:8-9, no
 * version pinned).  Its published algorithm (Julien Pommier's sse_mathfun, Cephes single-precision
 * polynomials) is restated below; the only reference-side pins are the tolerance tests vs libm in
 * Core/Tests/Math/UnrealMathTest.cpp:1395-1397, 1501-1502, 1519-1520, 2560-2562.
 *
 * Deliberate deviations (each switchable back with vvm_oracle_set_strict(1) where a reference
 * behaviour exists):
 *   - random ops: the reference seeds a 4-lane xorwow from the CPU cycle counter
 *     (VectorVMRuntime.cpp:445-483), so it has no reproducible output.  Here they use a counter-based
 *     generator keyed by (seed, instance index, ordinal of the random op in the script); the
 *     bits -> float mapping is the reference's (VectorVMRuntime.cpp:955).  vvm_oracle_set_xorwow() switches to
 *     the reference's own generator (the step of :468-483, advanced once per loop and draw in execution order:
 *     chunk by chunk, instruction by instruction) started from a state the caller injects in place of the
 *     cycle-counter seed; vvm_oracle_xorwow_seed() restates the seeding of :445-464 for three given numbers.
 *   - randomi writes its OUTPUT register with f2i(random(i2f(a))); the reference stores to the input
 *     register and never writes the output (VectorVMRuntime.cpp:1032, 1040).
 *   - exec_index* in a one-loop chunk returns StartInstance + lane; the reference returns 0..3 there
 *     (VectorVMRuntime.cpp:895-898, 918-921, 944-949).  strict reproduces the reference.
 *   - outputdata_float_from_half converts and compacts; the reference's handler is an empty stub
 *     (VectorVMRuntime.cpp:573-577).  strict reproduces the stub.
 *   - all outputs advance per chunk by the survivor count of their OWN dataset; the reference uses
 *     dataset 0's count for every output (VectorVMRuntime.cpp:2124).  strict reproduces the reference.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <immintrin.h>

#include "../include/vvm_abi.h"

#define ORACLE_API __attribute__((visibility("default")))

typedef struct oracle_extfn {
	int32_t num_inputs, num_outputs;
	const vvm_ext_function* fn;
} oracle_extfn;

typedef struct oracle_state {
	/* copied from the blob */
	uint8_t*  bytecode;
	uint32_t  num_bytecode_bytes;
	uint16_t* const_remap;
	uint16_t* input_remap;
	uint16_t* input_dataset_offsets;
	uint8_t*  out_dataset;
	uint16_t* out_type;
	uint16_t* out_dst;
	oracle_extfn* ext;
	uint32_t num_temps, num_consts, num_inputs, num_noadvance, num_input_datasets;
	uint32_t num_outputs_remapped, num_output_buffers, max_output_dataset;
	uint32_t num_ext, max_ext_regs, num_dummy_regs, flags;
	/* filled at exec */
	__m128i*  const_buffers;            /* consts, then the no-advance inputs */
	uint8_t*  const_map_idx;  uint16_t* const_map_src;
	uint8_t*  input_map_idx;  uint16_t* input_map_src;
	int32_t*  num_output_per_dataset;
	uint32_t  batch_overhead_size;
	int       map_cache_ready;
} oracle_state;

static int      g_strict = 0;
static int      g_page_kb = 64;           /* cvar vm.PageSizeInKB, VectorVMRuntime.cpp:44-50 */
static uint64_t g_rand_seed = 0;
static int32_t  g_exec_index_base = 0;
static int32_t  g_rand_instance_base = 0;
static int      g_xorwow = 0;             /* random ops draw from the reference's xorwow, started from g_xorwow_start at every Exec */
static uint32_t g_xorwow_start[24];       /* State[0..4] (four lanes each), then Counters (four lanes): FVectorVMBatchState::RandState */

ORACLE_API void vvm_oracle_set_strict(int s) { g_strict = s; }
ORACLE_API void vvm_oracle_set_page_kb(int kb) { g_page_kb = kb > 0 ? kb : 64; }
ORACLE_API void vvm_oracle_set_options(uint64_t seed, int32_t exec_index_base, int32_t rand_instance_base)
{
	g_rand_seed = seed; g_exec_index_base = exec_index_base; g_rand_instance_base = rand_instance_base;
}

/* state24 = State[0][0..3] ... State[4][0..3], Counters[0..3]; NULL switches back to the counter-based generator */
ORACLE_API void vvm_oracle_set_xorwow(const uint32_t* state24)
{
	g_xorwow = state24 != NULL;
	if (state24) memcpy(g_xorwow_start, state24, sizeof g_xorwow_start);
}

/* SetupRandStateForBatch, VectorVMRuntime.cpp:445-464, with its three sources of entropy as arguments: the two readings of the
 * cycle counter (:447, :449) and the address of the batch state (:448). */
ORACLE_API void vvm_oracle_xorwow_seed(uint64_t cycles_first, uint64_t cycles_second, uint64_t batch_state_address, uint32_t* state24)
{
	uint64_t pcg_state = cycles_first;
	uint64_t pcg_inc = ((batch_state_address << 32) ^ 0xCAFEF00DD15EA5E5ull) | 1;
	pcg_state ^= cycles_second << 32;
	for (int i = 0; i < 5; ++i)
		for (int j = 0; j < 4; ++j) {
			uint64_t old_state = pcg_state;
			pcg_state = old_state * 6364136223846793005ull + pcg_inc;
			uint32_t xor_shifted = (uint32_t)(((old_state >> 18) ^ old_state) >> 27);
			uint32_t rot = (uint32_t)(old_state >> 59);
			state24[4 * i + j] = (xor_shifted >> rot) | (xor_shifted << ((0u - rot) & 31));
		}
	/* MakeVectorRegisterInt64(pcg_inc, pcg_state), :464: two 64-bit lanes = four 32-bit counters, low half first */
	state24[20] = (uint32_t)pcg_inc;   state24[21] = (uint32_t)(pcg_inc >> 32);
	state24[22] = (uint32_t)pcg_state; state24[23] = (uint32_t)(pcg_state >> 32);
}

/* ------------------------------------------------------------------------------------------------
 * blob thaw -- VectorVMBridge.cpp:81-148
 * ---------------------------------------------------------------------------------------------- */
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static size_t al16(size_t n) { return (n + 15) & ~(size_t)15; }
static void* dup_bytes(const uint8_t* p, size_t n) { void* d = malloc(n ? n : 1); if (n) memcpy(d, p, n); return d; }

ORACLE_API oracle_state* vvm_oracle_alloc_state(const uint8_t* blob, size_t n)
{
	if (!blob || n < VVM_BLOB_FIRST_SECTION) return NULL;                 /* RT.cpp:2313-2316 */
	const uint8_t* h = blob;
	uint32_t num_bytecode = rd32(h + 0), max_out_ds = rd32(h + 4);
	uint32_t num_temps = rd32(h + 10);
	uint16_t num_consts = rd16(h + 14), num_inputs = rd16(h + 16), num_noadv = rd16(h + 18), num_in_ds = rd16(h + 20);
	uint16_t num_out_remapped = rd16(h + 22), num_out_ins = rd16(h + 24);
	uint32_t num_ext = rd32(h + 26), max_ext_regs = rd32(h + 30), num_dummy = rd32(h + 34);
	uint32_t flags = rd32(h + 42);
	size_t off_bc   = VVM_BLOB_FIRST_SECTION;
	size_t off_cr   = al16(off_bc + num_bytecode);
	size_t off_ir   = al16(off_cr + 2u * num_consts);
	size_t off_ido  = al16(off_ir + 2u * num_inputs);
	size_t off_ods  = al16(off_ido + 16u * num_in_ds);
	size_t off_ot   = al16(off_ods + num_out_remapped);
	size_t off_od   = al16(off_ot + 2u * num_out_remapped);
	size_t off_ext  = al16(off_od + 2u * num_out_remapped);
	size_t total    = al16(off_ext + (size_t)VVM_BLOB_EXTFN_ENTRY_BYTES * num_ext);
	if (total > n) return NULL;

	oracle_state* s = (oracle_state*)calloc(1, sizeof(*s));
	s->bytecode = (uint8_t*)dup_bytes(blob + off_bc, num_bytecode);
	s->num_bytecode_bytes = num_bytecode;
	s->const_remap = (uint16_t*)dup_bytes(blob + off_cr, 2u * num_consts);
	s->input_remap = (uint16_t*)dup_bytes(blob + off_ir, 2u * num_inputs);
	s->input_dataset_offsets = (uint16_t*)dup_bytes(blob + off_ido, 16u * num_in_ds);
	s->out_dataset = (uint8_t*)dup_bytes(blob + off_ods, num_out_remapped);
	s->out_type = (uint16_t*)dup_bytes(blob + off_ot, 2u * num_out_remapped);
	s->out_dst = (uint16_t*)dup_bytes(blob + off_od, 2u * num_out_remapped);
	s->ext = (oracle_extfn*)calloc(num_ext ? num_ext : 1, sizeof(oracle_extfn));
	for (uint32_t i = 0; i < num_ext; ++i) {
		const uint8_t* e = blob + off_ext + (size_t)VVM_BLOB_EXTFN_ENTRY_BYTES * i;
		s->ext[i].num_inputs = (int32_t)rd32(e + 8);
		s->ext[i].num_outputs = (int32_t)rd32(e + 12);
	}
	s->num_temps = num_temps; s->num_consts = num_consts; s->num_inputs = num_inputs; s->num_noadvance = num_noadv;
	s->num_input_datasets = num_in_ds; s->num_outputs_remapped = num_out_remapped; s->num_output_buffers = num_out_ins;
	s->max_output_dataset = max_out_ds; s->num_ext = num_ext; s->max_ext_regs = max_ext_regs; s->num_dummy_regs = num_dummy;
	s->flags = flags & ~(uint32_t)VVM_FLAG_DATA_MAP_CACHE_SETUP;

	s->const_buffers = (__m128i*)aligned_alloc(16, 16u * (num_consts + num_noadv + 1u));
	s->const_map_idx = (uint8_t*)calloc(num_consts + 1u, 1);
	s->const_map_src = (uint16_t*)calloc(num_consts + 1u, 2);
	s->input_map_idx = (uint8_t*)calloc(num_inputs + 1u, 1);
	s->input_map_src = (uint16_t*)calloc(num_inputs + 1u, 2);
	s->num_output_per_dataset = (int32_t*)calloc(max_out_ds + 1u, 4);

	/* fixed per-batch overhead, VectorVMRuntime.cpp:2391-2414; sizeof(FVectorVMBatchState) is 224 on a
	 * 64-bit target (VectorVMRuntime.cpp:52-86 laid out by hand), rounded up to 256. */
	size_t num_ptr_regs = num_temps + num_consts + 2u * num_inputs + num_out_ins;
	s->batch_overhead_size = (uint32_t)(256 + 4u * max_out_ds + 4u * max_out_ds + 8u * max_out_ds +
		9u * max_ext_regs + 16u * num_dummy + 9u * num_ptr_regs + 512u);
	return s;
}

ORACLE_API void vvm_oracle_free_state(oracle_state* s)
{
	if (!s) return;
	free(s->bytecode); free(s->const_remap); free(s->input_remap); free(s->input_dataset_offsets);
	free(s->out_dataset); free(s->out_type); free(s->out_dst); free(s->ext); free(s->const_buffers);
	free(s->const_map_idx); free(s->const_map_src); free(s->input_map_idx); free(s->input_map_src);
	free(s->num_output_per_dataset); free(s);
}

/* ------------------------------------------------------------------------------------------------
 * map-table caches -- VectorVMRuntime.cpp:485-553
 * ---------------------------------------------------------------------------------------------- */
static int build_map_caches(oracle_state* s, vvm_exec_context* ctx)
{
	for (uint32_t i = 0; i < s->num_consts; ++i) {
		uint32_t remapped = s->const_remap[i], acc = 0;
		int found = 0;
		for (int j = 0; j < ctx->ConstantTableCount; ++j) {
			uint32_t ndw = (uint32_t)ctx->ConstantTableNumBytes[j] >> 2;
			if (acc + ndw > remapped) {
				s->const_map_idx[i] = (uint8_t)j;
				s->const_map_src[i] = (uint16_t)(remapped - acc);
				found = 1;
				break;
			}
			acc += ndw;
		}
		if (!found) return VVM_E_INVALID_ARG;
	}
	if (s->num_input_datasets > 0) {
		int counter = 0;
		int nds = (int)s->num_input_datasets < ctx->DataSets.num ? (int)s->num_input_datasets : ctx->DataSets.num;
		const uint16_t* off = s->input_dataset_offsets;
		for (int i = 0; i < nds; ++i) {
			for (int j = 0; j < 3; ++j) {
				int cnt = off[(i << 3) + j + 1] - off[(i << 3) + j];
				int type_off = (int)ctx->DataSets.data[i].InputRegisterTypeOffsets[j];
				for (int k = 0; k < cnt; ++k) {
					int ridx = off[(i << 3) + j] + k;
					s->input_map_idx[counter] = (uint8_t)i;
					s->input_map_src[counter] = (uint16_t)((type_off + s->input_remap[ridx]) | ((j & 2) << 13));
					++counter;
				}
			}
			if (off[(i << 3) + 7] > 0) {
				for (int j = 0; j < 3; ++j) {
					int cnt = off[(i << 3) + j + 4] - off[(i << 3) + j + 3];
					int type_off = (int)ctx->DataSets.data[i].InputRegisterTypeOffsets[j];
					for (int k = 0; k < cnt; ++k) {
						int ridx = off[(i << 3) + j + 3] + k;
						s->input_map_idx[counter] = (uint8_t)i;
						s->input_map_src[counter] = (uint16_t)((type_off + s->input_remap[ridx]) | 0x8000 | ((j & 2) << 13));
						++counter;
					}
				}
			}
		}
		if ((uint32_t)counter != s->num_inputs) return VVM_E_INVALID_ARG;
	}
	return VVM_OK;
}

/* ------------------------------------------------------------------------------------------------
 * half conversion -- GenericPlatformMath.h:34-61 (load), :68-123 (store, RTNE)
 * ---------------------------------------------------------------------------------------------- */
static float load_half(uint16_t h)
{
	union { uint32_t u; float f; } v, magic;
	const uint32_t shifted_exp = 0x7c00u << 13;
	magic.u = 113u << 23;
	v.u = (uint32_t)(h & 0x7fff) << 13;
	uint32_t e = shifted_exp & v.u;
	v.u += (127u - 15u) << 23;
	if (e == shifted_exp) v.u += (128u - 16u) << 23;
	else if (e == 0) { v.u += 1u << 23; v.f -= magic.f; }
	v.u |= (uint32_t)(h & 0x8000) << 16;
	return v.f;
}

static uint16_t store_half(float x)
{
	union { uint32_t u; float f; } v, dm;
	v.f = x;
	const uint32_t f32infty = 255u << 23, f16max = (127u + 16u) << 23;
	dm.u = ((127u - 15u) + (23u - 10u) + 1u) << 23;
	uint32_t sign = v.u & 0x80000000u;
	uint16_t o;
	v.u ^= sign;
	if (v.u >= f16max) o = (v.u > f32infty) ? 0x7e00 : 0x7c00;
	else if (v.u < (113u << 23)) { v.f += dm.f; o = (uint16_t)(v.u - dm.u); }
	else { uint32_t odd = (v.u >> 13) & 1; v.u += ((uint32_t)(15 - 127) << 23) + 0xfff; v.u += odd; o = (uint16_t)(v.u >> 13); }
	return (uint16_t)(o | (sign >> 16));
}

/* ------------------------------------------------------------------------------------------------
 * 4-wide arithmetic.  cf = cast int->float bits, ci = float->int bits.
 * ---------------------------------------------------------------------------------------------- */
#define cf(x) _mm_castsi128_ps(x)
#define ci(x) _mm_castps_si128(x)
static inline __m128 set1(float v) { return _mm_set1_ps(v); }
static inline __m128 sel(__m128 m, __m128 a, __m128 b) { return _mm_or_ps(_mm_and_ps(m, a), _mm_andnot_ps(m, b)); }      /* SSE.h:1805-1810 */
static inline __m128i seli(__m128i m, __m128i a, __m128i b) { return _mm_or_si128(_mm_and_si128(m, a), _mm_andnot_si128(m, b)); } /* SSE.h:3368-3372 */
static inline __m128 vabs(__m128 a) { return _mm_and_ps(a, cf(_mm_set1_epi32(0x7fffffff))); }                             /* SSE.h:1287-1290 */
static inline __m128 vneg(__m128 a) { return _mm_sub_ps(_mm_setzero_ps(), a); }                                          /* SSE.h:1311-1314 */
static inline __m128 vmad(__m128 a, __m128 b, __m128 c) { return _mm_add_ps(_mm_mul_ps(a, b), c); }                      /* SSE.h:1411-1418, FMA3 off */
#define EPS set1(1.e-8f)                                                                                                   /* RT.cpp:109 */
static inline __m128 vdiv(__m128 a, __m128 b) { return sel(_mm_cmpgt_ps(vabs(b), EPS), _mm_div_ps(a, b), _mm_setzero_ps()); }   /* RT.cpp:1909 */
static inline __m128 vsqrtg(__m128 a) { return sel(_mm_cmpgt_ps(a, EPS), _mm_sqrt_ps(a), _mm_setzero_ps()); }                    /* RT.cpp:1914 */
static inline __m128 vrsq(__m128 a) { return sel(_mm_cmpgt_ps(a, EPS), _mm_div_ps(set1(1.f), _mm_sqrt_ps(a)), _mm_setzero_ps()); } /* RT.cpp:1913, SSE.h:2014-2022 */
static inline __m128 vrcp(__m128 a) { return sel(_mm_cmpgt_ps(vabs(a), EPS), _mm_rcp_ps(a), _mm_setzero_ps()); }                 /* RT.cpp:1912, SSE.h:2121-2125 */
static inline __m128 vlerp(__m128 a, __m128 b, __m128 c) { return vmad(b, c, _mm_mul_ps(a, _mm_sub_ps(set1(1.f), c))); }         /* VCOMMON.inl:532-536 */
static inline __m128 vclamp(__m128 a, __m128 b, __m128 c) { return _mm_min_ps(_mm_max_ps(a, b), c); }                            /* VCOMMON.inl:495-498 */
static inline __m128 vsign(__m128 a) { return sel(_mm_cmpge_ps(a, _mm_setzero_ps()), set1(1.f), set1(-1.f)); }                   /* SSE.h:2911-2915 */
static inline __m128 vstep(__m128 a) { return sel(_mm_cmpge_ps(a, _mm_setzero_ps()), set1(1.f), _mm_setzero_ps()); }             /* SSE.h:2923-2927 */
static inline __m128 vfrac(__m128 a) { return _mm_sub_ps(a, _mm_round_ps(a, _MM_FROUND_TRUNC)); }                                /* VCOMMON.inl:883-886 */

#define PER_LANE1(name, expr) static inline __m128 name(__m128 a) { float x[4], r[4]; _mm_storeu_ps(x, a); \
	for (int i = 0; i < 4; ++i) { float X = x[i]; r[i] = (expr); } return _mm_loadu_ps(r); }
#define PER_LANE2(name, expr) static inline __m128 name(__m128 a, __m128 b) { float x[4], y[4], r[4]; _mm_storeu_ps(x, a); _mm_storeu_ps(y, b); \
	for (int i = 0; i < 4; ++i) { float X = x[i], Y = y[i]; r[i] = (expr); } return _mm_loadu_ps(r); }

/* FMath::Atan2, GenericPlatformMath.cpp:33-76 */
/* atan2 of GenericPlatformMath.cpp:33-76: reduction to the first octant (ratio of the smaller to the larger magnitude), the
 * reference's minimax polynomial in the ratio's square by Horner's rule, then the three reflections that undo the reduction. */
static float first_octant_angle(float ratio)
{
	static const float coeff[7] = { +7.2128853633444123e-03f, -3.5059680836411644e-02f, +8.1675882859940430e-02f,
		-1.3374657325451267e-01f, +1.9856563505717162e-01f, -3.3324998579202170e-01f, +1.0f };
	const float square = ratio * ratio;
	float poly = coeff[0];
	for (int i = 1; i < 7; ++i) poly = poly * square + coeff[i];
	return poly * ratio;
}
static float ue_atan2(float y, float x)
{
	const float pi = 3.1415926535897932f;
	const float mag_x = fabsf(x), mag_y = fabsf(y);
	const int steep = mag_y > mag_x;
	const float larger = steep ? mag_y : mag_x, smaller = steep ? mag_x : mag_y;
	if (larger == 0.f) return 0.f;                      /* :47-50 */
	float angle = first_octant_angle(smaller / larger);
	if (steep) angle = (0.5f * pi) - angle;             /* about the diagonal */
	if (x < 0.0f) angle = pi - angle;                   /* about the vertical axis */
	if (y < 0.0f) angle = -angle;                       /* about the horizontal axis */
	return angle;
}
PER_LANE2(vatan2, ue_atan2(X, Y))                         /* SSE.h:3310-3324 */
PER_LANE1(vtan, tanf(X))                                  /* SSE.h:3188-3201, GPM.h:538 */
PER_LANE1(vatan, atanf(X))                                /* SSE.h:3279-3292, GPM.h:541 */
PER_LANE1(vexp2, powf(2.f, X))                            /* SSE.h:2960-2972, GPM.h:491 */
PER_LANE1(vlog2, logf(X) * 1.4426950f)                    /* SSE.h:3015-3027, GPM.h:502 */
PER_LANE2(vfmod_raw, fmodf(X, Y))
static inline __m128 vfmod(__m128 a, __m128 b)            /* SSE.h:2867-2886 */
{
	__m128 invalid = _mm_cmple_ps(vabs(b), set1(1.e-8f));
	return sel(invalid, _mm_setzero_ps(), vfmod_raw(a, b));
}
static inline __m128 vacos(__m128 a)                      /* RT.cpp:1924 */
{
	return vatan2(vsqrtg(_mm_mul_ps(_mm_sub_ps(set1(1.f), a), _mm_add_ps(set1(1.f), a))), a);
}
static inline __m128 vasin(__m128 a) { return _mm_sub_ps(set1(3.14159265359f * 0.25f), vacos(a)); }   /* RT.cpp:1925 with :111 (quarter pi, sic) */

/* --- sse_mathfun (third party, restated from its published algorithm) --- */
static inline __m128 mf_log(__m128 x)
{
	const __m128 one = set1(1.f);
	__m128 invalid = _mm_cmple_ps(x, _mm_setzero_ps());
	x = _mm_max_ps(x, cf(_mm_set1_epi32(0x00800000)));
	__m128i e0 = _mm_srli_epi32(ci(x), 23);
	x = _mm_and_ps(x, cf(_mm_set1_epi32(~0x7f800000)));
	x = _mm_or_ps(x, set1(0.5f));
	e0 = _mm_sub_epi32(e0, _mm_set1_epi32(0x7f));
	__m128 e = _mm_add_ps(_mm_cvtepi32_ps(e0), one);
	__m128 mask = _mm_cmplt_ps(x, set1(0.707106781186547524f));
	__m128 tmp = _mm_and_ps(x, mask);
	x = _mm_sub_ps(x, one);
	e = _mm_sub_ps(e, _mm_and_ps(one, mask));
	x = _mm_add_ps(x, tmp);
	__m128 z = _mm_mul_ps(x, x);
	static const float p[9] = { 7.0376836292E-2f, -1.1514610310E-1f, 1.1676998740E-1f, -1.2420140846E-1f, 1.4249322787E-1f,
		-1.6668057665E-1f, 2.0000714765E-1f, -2.4999993993E-1f, 3.3333331174E-1f };
	__m128 y = set1(p[0]);
	for (int i = 1; i < 9; ++i) y = _mm_add_ps(_mm_mul_ps(y, x), set1(p[i]));
	y = _mm_mul_ps(y, x);
	y = _mm_mul_ps(y, z);
	y = _mm_add_ps(y, _mm_mul_ps(e, set1(-2.12194440e-4f)));
	y = _mm_sub_ps(y, _mm_mul_ps(z, set1(0.5f)));
	tmp = _mm_mul_ps(e, set1(0.693359375f));
	x = _mm_add_ps(x, y);
	x = _mm_add_ps(x, tmp);
	return _mm_or_ps(x, invalid);
}

static inline __m128 mf_exp(__m128 x)
{
	const __m128 one = set1(1.f);
	x = _mm_min_ps(x, set1(88.3762626647949f));
	x = _mm_max_ps(x, set1(-88.3762626647949f));
	__m128 fx = _mm_add_ps(_mm_mul_ps(x, set1(1.44269504088896341f)), set1(0.5f));
	__m128 tmp = _mm_cvtepi32_ps(_mm_cvttps_epi32(fx));
	__m128 mask = _mm_and_ps(_mm_cmpgt_ps(tmp, fx), one);
	fx = _mm_sub_ps(tmp, mask);
	tmp = _mm_mul_ps(fx, set1(0.693359375f));
	__m128 z = _mm_mul_ps(fx, set1(-2.12194440e-4f));
	x = _mm_sub_ps(x, tmp);
	x = _mm_sub_ps(x, z);
	z = _mm_mul_ps(x, x);
	static const float p[6] = { 1.9875691500E-4f, 1.3981999507E-3f, 8.3334519073E-3f, 4.1665795894E-2f, 1.6666665459E-1f, 5.0000001201E-1f };
	__m128 y = set1(p[0]);
	for (int i = 1; i < 6; ++i) y = _mm_add_ps(_mm_mul_ps(y, x), set1(p[i]));
	y = _mm_add_ps(_mm_mul_ps(y, z), x);
	y = _mm_add_ps(y, one);
	__m128i n = _mm_slli_epi32(_mm_add_epi32(_mm_cvttps_epi32(fx), _mm_set1_epi32(0x7f)), 23);
	return _mm_mul_ps(y, cf(n));
}

static inline void mf_sincos_core(__m128 x, __m128i j, __m128* ysin_poly, __m128* ycos_poly)
{
	/* x already reduced; evaluate both polynomials */
	(void)j;
	__m128 z = _mm_mul_ps(x, x);
	__m128 y = set1(2.443315711809948E-005f);
	y = _mm_add_ps(_mm_mul_ps(y, z), set1(-1.388731625493765E-003f));
	y = _mm_add_ps(_mm_mul_ps(y, z), set1(4.166664568298827E-002f));
	y = _mm_mul_ps(y, z);
	y = _mm_mul_ps(y, z);
	y = _mm_sub_ps(y, _mm_mul_ps(z, set1(0.5f)));
	y = _mm_add_ps(y, set1(1.f));
	__m128 y2 = set1(-1.9515295891E-4f);
	y2 = _mm_add_ps(_mm_mul_ps(y2, z), set1(8.3321608736E-3f));
	y2 = _mm_add_ps(_mm_mul_ps(y2, z), set1(-1.6666654611E-1f));
	y2 = _mm_mul_ps(y2, z);
	y2 = _mm_mul_ps(y2, x);
	y2 = _mm_add_ps(y2, x);
	*ycos_poly = y; *ysin_poly = y2;
}

static inline __m128 mf_reduce(__m128 ax, __m128i* jout)
{
	__m128 y = _mm_mul_ps(ax, set1(1.27323954473516f));
	__m128i j = _mm_cvttps_epi32(y);
	j = _mm_and_si128(_mm_add_epi32(j, _mm_set1_epi32(1)), _mm_set1_epi32(~1));
	y = _mm_cvtepi32_ps(j);
	*jout = j;
	ax = _mm_add_ps(ax, _mm_mul_ps(y, set1(-0.78515625f)));
	ax = _mm_add_ps(ax, _mm_mul_ps(y, set1(-2.4187564849853515625e-4f)));
	ax = _mm_add_ps(ax, _mm_mul_ps(y, set1(-3.77489497744594108e-8f)));
	return ax;
}

static inline __m128 mf_sin(__m128 x)
{
	__m128 sign = _mm_and_ps(x, cf(_mm_set1_epi32((int)0x80000000)));
	__m128i j;
	__m128 r = mf_reduce(vabs(x), &j);
	__m128 swap = cf(_mm_slli_epi32(_mm_and_si128(j, _mm_set1_epi32(4)), 29));
	__m128 poly = cf(_mm_cmpeq_epi32(_mm_and_si128(j, _mm_set1_epi32(2)), _mm_setzero_si128()));
	sign = _mm_xor_ps(sign, swap);
	__m128 ys, yc;
	mf_sincos_core(r, j, &ys, &yc);
	__m128 y = _mm_add_ps(_mm_andnot_ps(poly, yc), _mm_and_ps(poly, ys));
	return _mm_xor_ps(y, sign);
}

static inline __m128 mf_cos(__m128 x)
{
	__m128i j;
	__m128 r = mf_reduce(vabs(x), &j);
	j = _mm_sub_epi32(j, _mm_set1_epi32(2));
	__m128 sign = cf(_mm_slli_epi32(_mm_andnot_si128(j, _mm_set1_epi32(4)), 29));
	__m128 poly = cf(_mm_cmpeq_epi32(_mm_and_si128(j, _mm_set1_epi32(2)), _mm_setzero_si128()));
	__m128 ys, yc;
	mf_sincos_core(r, j, &ys, &yc);
	__m128 y = _mm_add_ps(_mm_andnot_ps(poly, yc), _mm_and_ps(poly, ys));
	return _mm_xor_ps(y, sign);
}

static inline __m128 vlog(__m128 a) { return sel(_mm_cmpgt_ps(a, _mm_setzero_ps()), mf_log(a), _mm_setzero_ps()); }          /* RT.cpp:1919 */
static inline __m128 vpow(__m128 a, __m128 b)                                                                                /* RT.cpp:1936, SSE.h:1941-1948 */
{
	return sel(_mm_cmpgt_ps(a, EPS), mf_exp(_mm_mul_ps(mf_log(a), b)), _mm_setzero_ps());
}

/* VectorSinCos, SSE.h:3118-3172 */
static inline void vsincos(__m128 a, __m128* s_out, __m128* c_out)
{
	const float PI = 3.1415926535897932f;
	__m128 q = _mm_mul_ps(a, set1(1.0f / (2.0f * PI)));
	q = _mm_cvtepi32_ps(_mm_cvtps_epi32(q));
	__m128 x = _mm_sub_ps(a, _mm_mul_ps(set1(2.0f * PI), q));
	__m128 sign = _mm_and_ps(x, cf(_mm_set1_epi32((int)0x80000000)));
	__m128 c = _mm_or_ps(set1(PI), sign);
	__m128 absx = vabs(x);
	__m128 rflx = _mm_sub_ps(c, x);
	__m128 comp = _mm_cmpgt_ps(absx, set1(0.5f * PI));
	x = sel(comp, rflx, x);
	sign = sel(comp, set1(-1.f), set1(1.f));
	__m128 x2 = _mm_mul_ps(x, x);
	__m128 S = set1(-2.3889859e-08f);
	S = vmad(x2, S, set1(2.7525562e-06f));
	S = vmad(x2, S, set1(-0.00019840874f));
	S = vmad(x2, S, set1(0.0083333310f));
	S = vmad(x2, S, set1(-0.16666667f));
	S = vmad(x2, S, set1(1.0f));
	*s_out = _mm_mul_ps(S, x);
	__m128 C = set1(-2.6051615e-07f);
	C = vmad(x2, C, set1(2.4760495e-05f));
	C = vmad(x2, C, set1(-0.0013888378f));
	C = vmad(x2, C, set1(0.041666638f));
	C = vmad(x2, C, set1(-0.5f));
	C = vmad(x2, C, set1(1.0f));
	*c_out = _mm_mul_ps(C, sign);
}

/* integer helpers */
static inline __m128i inot(__m128i a) { return _mm_xor_si128(a, _mm_set1_epi32(-1)); }
static inline __m128i icmpge(__m128i a, __m128i b) { return inot(_mm_cmplt_epi32(a, b)); }     /* SSE.h:3364 */
static inline __m128i icmple(__m128i a, __m128i b) { return inot(_mm_cmpgt_epi32(a, b)); }     /* SSE.h:3365 */
static inline __m128i icmpne(__m128i a, __m128i b) { return inot(_mm_cmpeq_epi32(a, b)); }     /* SSE.h:3361 */
static inline __m128i iclamp(__m128i a, __m128i b, __m128i c) { return _mm_min_epi32(_mm_max_epi32(a, b), c); }   /* SSE.h:3400-3403 */
static inline __m128i isign(__m128i a) { return seli(icmpge(a, _mm_setzero_si128()), _mm_set1_epi32(1), _mm_set1_epi32(-1)); } /* SSE.h:3405 */
static inline int32_t safe_idiv(int32_t n, int32_t d)                                           /* RT.cpp:958-973 */
{
	if (d == 0) return 0;
	if (d == -1 && n == INT_MIN) return INT_MAX;
	return n / d;
}
static inline __m128i idiv(__m128i a, __m128i b)
{
	int32_t x[4], y[4], r[4];
	_mm_storeu_si128((__m128i*)x, a); _mm_storeu_si128((__m128i*)y, b);
	for (int i = 0; i < 4; ++i) r[i] = safe_idiv(x[i], y[i]);
	return _mm_loadu_si128((__m128i*)r);
}
/* per-lane variable shifts, VectorVMPlatformGeneric.h:61-89; an x86 shift masks its count to 5 bits */
static inline __m128i ishl(__m128i a, __m128i b)
{
	uint32_t x[4], y[4], r[4];
	_mm_storeu_si128((__m128i*)x, a); _mm_storeu_si128((__m128i*)y, b);
	for (int i = 0; i < 4; ++i) r[i] = x[i] << (y[i] & 31);
	return _mm_loadu_si128((__m128i*)r);
}
static inline __m128i ishr(__m128i a, __m128i b)
{
	uint32_t x[4], y[4], r[4];
	_mm_storeu_si128((__m128i*)x, a); _mm_storeu_si128((__m128i*)y, b);
	for (int i = 0; i < 4; ++i) r[i] = x[i] >> (y[i] & 31);
	return _mm_loadu_si128((__m128i*)r);
}
#define F2I(x) _mm_cvttps_epi32(x)      /* SSE.h:3409-3412 */
#define I2F(x) _mm_cvtepi32_ps(x)       /* SSE.h:3407 */

/* counter-based generator (this project's deterministic mode; see the header comment) */
static inline uint32_t rand_u32(uint64_t seed, uint32_t instance, uint32_t ordinal)
{
	uint32_t h = instance * 0x9E3779B1u + (uint32_t)seed;
	h ^= ordinal * 0x85EBCA77u + (uint32_t)(seed >> 32);
	h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
	h += ordinal * 0x27D4EB2Fu + instance;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
	return h;
}
/* VVMXorwowStep, VectorVMRuntime.cpp:468-483: four independent generators side by side, one per lane */
static inline __m128i xorwow_step(__m128i* st /* State[0..4], Counters */)
{
	__m128i t = st[4], s = st[0];
	st[4] = st[3]; st[3] = st[2]; st[2] = st[1]; st[1] = s;
	t = _mm_xor_si128(t, _mm_srli_epi32(t, 2));
	t = _mm_xor_si128(t, _mm_slli_epi32(t, 1));
	t = _mm_xor_si128(t, _mm_xor_si128(s, _mm_xor_si128(s, _mm_slli_epi32(s, 4))));     /* as written at :478: s cancels, s << 4 remains */
	st[0] = t;
	st[5] = _mm_add_epi32(st[5], _mm_set1_epi32(362437));
	return _mm_add_epi32(t, st[5]);
}
static inline __m128 rand4(__m128i* xorwow, uint32_t first_instance, uint32_t ordinal, __m128 a)    /* value mapping: RT.cpp:953-956 */
{
	__m128i bits;
	if (g_xorwow) bits = xorwow_step(xorwow);
	else {
		uint32_t r[4];
		for (int i = 0; i < 4; ++i) r[i] = rand_u32(g_rand_seed, first_instance + (uint32_t)i, ordinal);
		bits = _mm_loadu_si128((__m128i*)r);
	}
	__m128 f = cf(_mm_or_si128(_mm_srli_epi32(bits, 9), _mm_set1_epi32(0x3F800000)));
	return _mm_mul_ps(_mm_sub_ps(f, set1(1.f)), a);
}

/* ------------------------------------------------------------------------------------------------
 * built-in external functions
 * ---------------------------------------------------------------------------------------------- */
#include "perlin_table.inc"

static inline float p_grad3(int h, float x, float y, float z)     /* UnrealMath.cpp:3577-3602 */
{
	switch (h & 15) {
	case 0: return x + z;   case 1: return x + y;   case 2: return y + z;   case 3: return -x + y;
	case 4: return -x + z;  case 5: return -x - y;  case 6: return -y + z;  case 7: return x - y;
	case 8: return x - z;   case 9: return y - z;   case 10: return -x - z; case 11: return -y - z;
	case 12: return x + y;  case 13: return -x + y; case 14: return -y + z; default: return -y - z;
	}
}
static inline float p_smooth(float x) { return x * x * x * (x * (x * 6.0f - 15.0f) + 10.0f); }   /* UnrealMath.cpp:3605-3608 */
static inline float p_lerp(float a, float b, float t) { return a + t * (b - a); }                /* FMath::Lerp, UnrealMathUtility.h */

static float perlin3(float px, float py, float pz)               /* UnrealMath.cpp:3658-3699 */
{
	float xf = floorf(px), yf = floorf(py), zf = floorf(pz);
	/* (int32)floor & 255, :3665-3667: the cast is the truncating SSE conversion on the reference's x86 targets, which gives
	 * 0x80000000 (cell 0) for NaN and for a coordinate outside the int32 range; taken explicitly here, a plain C cast of such
	 * a value being undefined */
	int xi = _mm_cvttss_si32(_mm_set_ss(xf)) & 255, yi = _mm_cvttss_si32(_mm_set_ss(yf)) & 255, zi = _mm_cvttss_si32(_mm_set_ss(zf)) & 255;
	float x = px - xf, y = py - yf, z = pz - zf;
	float xm = x - 1.0f, ym = y - 1.0f, zm = z - 1.0f;
	const int32_t* P = k_perlin_perm;
	int a = P[xi] + yi, aa = P[a] + zi, ab = P[a + 1] + zi;
	int b = P[xi + 1] + yi, ba = P[b] + zi, bb = P[b + 1] + zi;
	float u = p_smooth(x), v = p_smooth(y), w = p_smooth(z);
	float r = 0.97f * p_lerp(
		p_lerp(p_lerp(p_grad3(P[aa], x, y, z), p_grad3(P[ba], xm, y, z), u),
		       p_lerp(p_grad3(P[ab], x, ym, z), p_grad3(P[bb], xm, ym, z), u), v),
		p_lerp(p_lerp(p_grad3(P[aa + 1], x, y, zm), p_grad3(P[ba + 1], xm, y, zm), u),
		       p_lerp(p_grad3(P[ab + 1], x, ym, zm), p_grad3(P[bb + 1], xm, ym, zm), u), v), w);
	return r < -1.0f ? -1.0f : (r < 1.0f ? r : 1.0f);   /* FMath::Clamp */
}

/* curl of three Perlin potentials taken at fixed offsets, central differences with step 1/64.
 * No counterpart in the reference tree (Niagara's curl-noise data interface lives in the plugin). */
static void curl_noise3(float x, float y, float z, float* ox, float* oy, float* oz)
{
	const float e = 0.015625f, inv2e = 32.0f;
	const float o1x = 31.416f, o1y = -47.853f, o1z = 12.793f;
	const float o2x = -233.145f, o2y = -113.408f, o2z = -185.31f;
	float p2_dy = perlin3(x + o2x, (y + e) + o2y, z + o2z) - perlin3(x + o2x, (y - e) + o2y, z + o2z);
	float p1_dz = perlin3(x + o1x, y + o1y, (z + e) + o1z) - perlin3(x + o1x, y + o1y, (z - e) + o1z);
	float p0_dz = perlin3(x, y, z + e) - perlin3(x, y, z - e);
	float p2_dx = perlin3((x + e) + o2x, y + o2y, z + o2z) - perlin3((x - e) + o2x, y + o2y, z + o2z);
	float p1_dx = perlin3((x + e) + o1x, y + o1y, z + o1z) - perlin3((x - e) + o1x, y + o1y, z + o1z);
	float p0_dy = perlin3(x, y + e, z) - perlin3(x, y - e, z);
	*ox = (p2_dy - p1_dz) * inv2e;
	*oy = (p0_dz - p2_dx) * inv2e;
	*oz = (p1_dx - p0_dy) * inv2e;
}

static float* extfn_next(vvm_extfn_context* c, int* adv)       /* GetNextRegister, VectorVM.h:350-355 */
{
	*adv = c->RegInc[c->RegReadCount];
	return (float*)c->RegisterData[c->RegReadCount++];
}

static void builtin_perlin3(vvm_extfn_context* c)
{
	int ax, ay, az, ao;
	float *x = extfn_next(c, &ax), *y = extfn_next(c, &ay), *z = extfn_next(c, &az), *o = extfn_next(c, &ao);
	for (int i = 0; i < c->NumInstances; ++i) {
		float r = perlin3(*x, *y, *z);
		x += ax; y += ay; z += az;
		*o = r; o += ao;
	}
}

static void builtin_curl_noise3(vvm_extfn_context* c)
{
	int ax, ay, az, a0, a1, a2;
	float *x = extfn_next(c, &ax), *y = extfn_next(c, &ay), *z = extfn_next(c, &az);
	float *o0 = extfn_next(c, &a0), *o1 = extfn_next(c, &a1), *o2 = extfn_next(c, &a2);
	for (int i = 0; i < c->NumInstances; ++i) {
		float rx, ry, rz;
		curl_noise3(*x, *y, *z, &rx, &ry, &rz);
		x += ax; y += ay; z += az;
		*o0 = rx; *o1 = ry; *o2 = rz;
		o0 += a0; o1 += a1; o2 += a2;
	}
}

/* ------------------------------------------------------------------------------------------------
 * batch state
 * ---------------------------------------------------------------------------------------------- */
typedef struct batch_state {
	__m128i*  register_data;
	uint8_t** reg_ptr;
	uint8_t*  reg_inc;
	uint8_t*  output_mask;            /* max_output_dataset x num_loops */
	uint8_t** output_mask_idx;
	uint32_t* starting_output_idx;
	uint32_t* num_output;
	uint32_t** ext_reg_data;
	uint8_t*  ext_reg_inc;
	__m128i*  dummy_regs;
	int32_t*  rand_counters;
	__m128i   xorwow[6];              /* FVectorVMBatchState::RandState: State[0..4], Counters */
	int start_instance, num_instances, chunk_idx;
	uint32_t max_loops;
} batch_state;

/* compaction tables, built once: survivors of a 4-group move to the front in lane order
 * (same content as VVM_PSHUFB_OUTPUT_TABLE / _TABLE16 / advance tables, VectorVMRuntime.cpp:132-186) */
static uint8_t g_shuf32[16][16], g_shuf16[16][16], g_adv32[16], g_adv16[16];
static int g_tables_ready = 0;
static void build_tables(void)
{
	if (g_tables_ready) return;
	for (int m = 0; m < 16; ++m) {
		int n = 0;
		memset(g_shuf32[m], 0xFF, 16); memset(g_shuf16[m], 0xFF, 16);
		for (int l = 0; l < 4; ++l) if (m & (1 << l)) {
			for (int b = 0; b < 4; ++b) g_shuf32[m][n * 4 + b] = (uint8_t)(l * 4 + b);
			for (int b = 0; b < 2; ++b) g_shuf16[m][n * 2 + b] = (uint8_t)(l * 2 + b);
			++n;
		}
		g_adv32[m] = (uint8_t)(n * 4); g_adv16[m] = (uint8_t)(n * 2);
	}
	g_tables_ready = 1;
}

/* SetupBatchStatePtrs, VectorVMRuntime.cpp:315-443 */
static int setup_batch(oracle_state* s, vvm_exec_context* ctx, batch_state* b, uint32_t max_loops)
{
	uint32_t T = s->num_temps, C = s->num_consts, I = s->num_inputs, O = s->num_output_buffers, M = s->max_output_dataset;
	size_t nptr = T + C + 2u * I + O;
	memset(b, 0, sizeof(*b));
	b->max_loops = max_loops;
	b->register_data = (__m128i*)aligned_alloc(64, (16u * (size_t)T * max_loops + 63u + 64u) & ~(size_t)63);
	b->reg_ptr = (uint8_t**)calloc(nptr + 1, sizeof(uint8_t*));
	b->reg_inc = (uint8_t*)malloc(nptr + 1);
	b->output_mask = (uint8_t*)malloc((size_t)M * max_loops + 16);
	b->output_mask_idx = (uint8_t**)calloc(M + 1u, sizeof(uint8_t*));
	b->starting_output_idx = (uint32_t*)calloc(M + 1u, 4);
	b->num_output = (uint32_t*)calloc(M + 1u, 4);
	b->ext_reg_data = (uint32_t**)calloc(s->max_ext_regs + 1u, sizeof(uint32_t*));
	b->ext_reg_inc = (uint8_t*)calloc(s->max_ext_regs + 1u, 1);
	b->dummy_regs = (__m128i*)aligned_alloc(16, 16u * (s->num_dummy_regs + 1u));
	for (uint32_t i = 0; i < M; ++i) b->output_mask_idx[i] = b->output_mask + (size_t)i * max_loops;
	memset(b->reg_inc, 16, nptr + 1);
	for (uint32_t i = 0; i < T; ++i) b->reg_ptr[i] = (uint8_t*)(b->register_data + (size_t)i * max_loops);
	for (uint32_t i = 0; i < C; ++i) { b->reg_ptr[T + i] = (uint8_t*)(s->const_buffers + i); b->reg_inc[T + i] = 0; }
	int noadv = 0;
	for (uint32_t i = 0; i < I; ++i) {
		uint8_t ds = s->input_map_idx[i];
		uint16_t src = s->input_map_src[i];
		if (ds >= ctx->DataSets.num) return VVM_E_INVALID_ARG;
		const vvm_dataset_meta* d = &ctx->DataSets.data[ds];
		int no_adv = src & 0x8000, half = src & 0x4000;
		src &= 0x3FFF;
		if ((int)src >= d->InputRegisters.num) return VVM_E_INVALID_ARG;
		uint32_t ri = T + C + i;
		if (no_adv) {
			b->reg_ptr[ri] = (uint8_t*)(s->const_buffers + C + noadv);
			++noadv;
			b->reg_inc[ri] = 0;
			if (half) {
				float v = load_half(((const uint16_t*)d->InputRegisters.data[src])[d->InstanceOffset]);
				_mm_storeu_ps((float*)b->reg_ptr[ri], _mm_set1_ps(v));
			} else {
				uint32_t v = ((const uint32_t*)d->InputRegisters.data[src])[d->InstanceOffset];
				_mm_storeu_si128((__m128i*)b->reg_ptr[ri], _mm_set1_epi32((int)v));
			}
		} else {
			b->reg_ptr[ri] = (uint8_t*)d->InputRegisters.data[src] + (size_t)d->InstanceOffset * (half ? 2u : 4u);
		}
		b->reg_ptr[ri + I] = b->reg_ptr[ri];
	}
	for (uint32_t i = 0; i < O; ++i) {
		uint8_t ds = s->out_dataset[i];
		uint16_t ty = s->out_type[i], dst = s->out_dst[i];
		uint32_t ri = T + C + 2u * I + i;
		if (ds >= ctx->DataSets.num || ty > 2) return VVM_E_INVALID_ARG;
		const vvm_dataset_meta* d = &ctx->DataSets.data[ds];
		if (dst == 0xFFFF) { b->reg_ptr[ri] = NULL; continue; }
		uint32_t idx = d->OutputRegisterTypeOffsets[ty] + dst;
		if ((int)idx >= d->OutputRegisters.num) return VVM_E_INVALID_ARG;
		b->reg_ptr[ri] = d->OutputRegisters.data[idx] + (size_t)d->InstanceOffset * (ty == 2 ? 2u : 4u);
	}
	return VVM_OK;
}

static void free_batch(batch_state* b)
{
	free(b->register_data); free(b->reg_ptr); free(b->reg_inc); free(b->output_mask); free(b->output_mask_idx);
	free(b->starting_output_idx); free(b->num_output); free(b->ext_reg_data); free(b->ext_reg_inc); free(b->dummy_regs);
	free(b->rand_counters);
}

/* SetBatchPointersForCorrectChunkOffsets, VectorVMRuntime.cpp:2103-2133 */
static void set_chunk_pointers(oracle_state* s, batch_state* b)
{
	uint32_t T = s->num_temps, C = s->num_consts, I = s->num_inputs, O = s->num_output_buffers;
	for (uint32_t i = 0; i < I; ++i) {
		if (b->reg_inc[T + C + i] != 0) {
			uint32_t stride = (s->input_map_src[i] & 0x4000) ? 2u : 4u;
			b->reg_ptr[T + C + i] = b->reg_ptr[T + C + I + i] + (size_t)b->start_instance * stride;
		}
	}
	if (b->start_instance != 0) {
		for (uint32_t i = 0; i < O; ++i) {
			uint32_t stride = s->out_type[i] == 2 ? 2u : 4u;
			uint32_t ds = g_strict ? 0u : s->out_dataset[i];
			if (b->reg_ptr[T + C + 2u * I + i])
				b->reg_ptr[T + C + 2u * I + i] += (size_t)b->num_output[ds] * stride;
		}
	}
	for (uint32_t i = 0; i < s->max_output_dataset; ++i) { b->starting_output_idx[i] = 0; b->num_output[i] = 0; }
}

/* ------------------------------------------------------------------------------------------------
 * special instruction handlers
 * ---------------------------------------------------------------------------------------------- */
static inline uint16_t u16at(const uint8_t* p, int i) { uint16_t v; memcpy(&v, p + 2 * i, 2); return v; }

/* VVM_acquireindex, VectorVMRuntime.cpp:824-878 */
static const uint8_t* do_acquireindex(oracle_state* s, vvm_exec_context* ctx, batch_state* b, const uint8_t* ip)
{
	uint16_t r = u16at(ip, 0);
	const uint32_t* in = (const uint32_t*)b->reg_ptr[r];
	uint32_t inc = (uint32_t)b->reg_inc[r] >> 2;
	uint8_t ds = ip[2];
	uint8_t* out = b->output_mask_idx[ds];
	uint32_t count = 0;
	int full = b->num_instances >> 2, left = b->num_instances - (full << 2);
	for (int i = 0; i < full; ++i) {
		uint32_t idx = (in[0] >> 31) | ((in[1] >> 31) << 1) | ((in[2] >> 31) << 2) | ((in[3] >> 31) << 3);
		in += inc;
		out[i] = (uint8_t)idx;
		count += g_adv32[idx];
	}
	if (left > 0) {
		uint32_t idx = 0;
		for (int l = 0; l < left; ++l) idx |= (in[l] >> 31) << l;
		out[full] = (uint8_t)idx;
		count += g_adv32[idx];
	}
	count >>= 2;
	int32_t prev = s->num_output_per_dataset[ds];
	s->num_output_per_dataset[ds] = prev + (int32_t)count;           /* InterlockedAdd with one writer */
	b->starting_output_idx[ds] = (uint32_t)(ctx->DataSets.data[ds].InstanceOffset + prev);
	b->num_output[ds] += count;
	return ip + 3;
}

/* VVM_Output32, VectorVMRuntime.cpp:744-822 */
static const uint8_t* do_output32(oracle_state* s, batch_state* b, const uint8_t* ip)
{
	(void)s;
	int n = ip[0];
	uint8_t ds = ip[1];
	uint32_t nout = b->num_output[ds];
	const uint8_t* src_idx = ip + 2;
	const uint8_t* dst_idx = src_idx + 2 * n;
	int loops = (b->num_instances + 3) >> 2;
	for (int j = 0; j < n; ++j) {
		uint16_t sr = u16at(src_idx, j), dr = u16at(dst_idx, j);
		int sinc = b->reg_inc[sr];
		uint8_t* src = b->reg_ptr[sr];
		uint8_t* dst = b->reg_ptr[dr];
		if (!dst) continue;
		if (nout == (uint32_t)b->num_instances) {
			if (src != dst) {
				if (sinc == 0) { uint32_t v; memcpy(&v, src, 4); for (uint32_t i = 0; i < nout; ++i) memcpy(dst + 4u * i, &v, 4); }
				else memmove(dst, src, 4u * (size_t)nout);
			}
		} else if (nout > 0) {
			if (sinc == 0) {
				uint32_t v; memcpy(&v, src, 4);
				for (uint32_t i = 0; i < nout; ++i) memcpy(dst + 4u * i, &v, 4);
			} else {
				const uint8_t* m = b->output_mask_idx[ds];
				/* the last groups are written element-wise so that nothing lands past the survivors:
				 * the reference stores a full 16 bytes per group (VectorVMRuntime.cpp:800) and so may write up
				 * to 12 don't-care bytes past the end; callers only compare [0, S). */
				for (int l = 0; l < loops; ++l) {
					uint8_t t = m[l];
					__m128i v = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)src), _mm_loadu_si128((const __m128i*)g_shuf32[t]));
					uint32_t tmp[4];
					_mm_storeu_si128((__m128i*)tmp, v);
					memcpy(dst, tmp, g_adv32[t]);
					src += 16;
					dst += g_adv32[t];
				}
			}
		}
	}
	return ip + 2 + 4 * n;
}

/* VVM_Output16 (float register -> half stream), VectorVMRuntime.cpp:659-742 */
static const uint8_t* do_output16(batch_state* b, const uint8_t* ip)
{
	int n = ip[0];
	uint8_t ds = ip[1];
	uint32_t nout = b->num_output[ds];
	const uint8_t* src_idx = ip + 2;
	const uint8_t* dst_idx = src_idx + 2 * n;
	int loops = (b->num_instances + 3) >> 2;
	for (int j = 0; j < n; ++j) {
		uint16_t sr = u16at(src_idx, j), dr = u16at(dst_idx, j);
		int sinc = b->reg_inc[sr];
		const float* src = (const float*)b->reg_ptr[sr];
		uint16_t* dst = (uint16_t*)b->reg_ptr[dr];
		if (!dst || nout == 0) continue;
		if (sinc == 0) {
			uint16_t h = store_half(src[0]);
			for (uint32_t i = 0; i < nout; ++i) dst[i] = h;
		} else {
			const uint8_t* m = b->output_mask_idx[ds];
			for (int l = 0; l < loops; ++l) {
				uint8_t t = (nout == (uint32_t)b->num_instances) ? 15 : m[l];
				for (int k = 0; k < 4; ++k) if (t & (1 << k)) {
					if (l * 4 + k < b->num_instances) *dst++ = store_half(src[k]);
				}
				src += 4;
			}
		}
	}
	return ip + 2 + 4 * n;
}

/* VVM_Output16_from_16 (half register stream -> half stream), VectorVMRuntime.cpp:579-657; the float-from-half
 * form is an empty stub in the reference (:573-577) and converts here unless strict. */
static const uint8_t* do_output_from_half(batch_state* b, const uint8_t* ip, int to_float)
{
	int n = ip[0];
	uint8_t ds = ip[1];
	uint32_t nout = b->num_output[ds];
	const uint8_t* src_idx = ip + 2;
	const uint8_t* dst_idx = src_idx + 2 * n;
	int loops = (b->num_instances + 3) >> 2;
	if (to_float && g_strict) return ip + 2 + 4 * n;
	for (int j = 0; j < n; ++j) {
		uint16_t sr = u16at(src_idx, j), dr = u16at(dst_idx, j);
		int sinc = b->reg_inc[sr];
		const uint16_t* src = (const uint16_t*)b->reg_ptr[sr];
		uint8_t* dst = b->reg_ptr[dr];
		if (!dst || nout == 0) continue;
		const uint8_t* m = b->output_mask_idx[ds];
		for (int l = 0; l < loops; ++l) {
			uint8_t t = (nout == (uint32_t)b->num_instances) ? 15 : m[l];
			for (int k = 0; k < 4; ++k) if ((t & (1 << k)) && l * 4 + k < b->num_instances) {
				uint16_t h = sinc ? src[k] : src[0];
				if (to_float) { float f = load_half(h); memcpy(dst, &f, 4); dst += 4; }
				else { memcpy(dst, &h, 2); dst += 2; }
			}
			if (sinc) src += 4;
		}
	}
	return ip + 2 + 4 * n;
}

/* VVM_update_id, VectorVMRuntime.cpp:1129-1402 */
static const uint8_t* do_update_id(vvm_exec_context* ctx, batch_state* b, const uint8_t* ip)
{
	const int32_t* r1 = (const int32_t*)b->reg_ptr[u16at(ip, 0)];
	uint8_t ds = ip[4];
	vvm_dataset_meta* d = &ctx->DataSets.data[ds];
	int nout = (int)b->num_output[ds];
	int nfreed = b->num_instances - nout;
	int max_id = -1;
	/* the reference takes the max over the first NumOutputInstances entries of the (uncompacted) ID register,
	 * VectorVMRuntime.cpp:1143-1174 */
	for (int i = 0; i < nout; ++i) if (r1[i] > max_id) max_id = r1[i];
	int start_freed = *d->NumFreeIDs;
	*d->NumFreeIDs = start_freed + nfreed;
	int32_t* id_table = d->IDTable->data;
	int32_t* free_table = d->FreeIDTable->data + start_freed;
	const uint8_t* m = b->output_mask_idx[ds];
	int counted = 0, freed = 0;
	for (int i = 0; i < b->num_instances; ++i) {
		if (m[i >> 2] & (1 << (i & 3))) id_table[r1[i]] = (int32_t)b->starting_output_idx[ds] + counted++;
		else free_table[freed++] = r1[i];
	}
	if (max_id != -1 && max_id > *d->MaxUsedID) *d->MaxUsedID = max_id;
	return ip + 5;
}

/* VVM_acquire_id, VectorVMRuntime.cpp:1404-1440 */
static const uint8_t* do_acquire_id(vvm_exec_context* ctx, batch_state* b, const uint8_t* ip)
{
	uint8_t ds = ip[4];
	vvm_dataset_meta* d = &ctx->DataSets.data[ds];
	int n = b->num_instances;
	int32_t* out = (int32_t*)b->reg_ptr[u16at(ip, 0)];
	int old_free = *d->NumFreeIDs;
	const int32_t* in = d->FreeIDTable->data + old_free - n;
	for (int i = 0; i < n; ++i) out[i] = in[n - i - 1];
	*d->NumFreeIDs = old_free - n;
	int start_spawned = *d->NumSpawnedIDs + n;
	*d->NumSpawnedIDs += n;
	memcpy(d->FreeIDTable->data + d->FreeIDTable->max - start_spawned, out, 4u * (size_t)n);
	int32_t* tag = (int32_t*)b->reg_ptr[u16at(ip, 1)];
	for (int i = 0; i < n; ++i) tag[i] = d->IDAcquireTag;
	return ip + 5;
}

/* VVM_external_func_call, VectorVMRuntime.cpp:1442-1489 */
static const uint8_t* do_extfn(oracle_state* s, vvm_exec_context* ctx, batch_state* b, const uint8_t* ip, int loops)
{
	int fn = u16at(ip, 0);
	oracle_extfn* e = &s->ext[fn];
	int nreg = e->num_inputs + e->num_outputs;
	uint32_t dummy = 0;
	for (int i = 0; i < nreg; ++i) {
		uint16_t r = u16at(ip + 2, i);
		if (r != 0xFFFF) { b->ext_reg_data[i] = (uint32_t*)b->reg_ptr[r]; b->ext_reg_inc[i] = b->reg_inc[r] >> 4; }
		else { b->ext_reg_data[i] = (uint32_t*)(b->dummy_regs + dummy++); b->ext_reg_inc[i] = 0; }
	}
	vvm_extfn_context c;
	memset(&c, 0, sizeof(c));
	c.RegisterData = b->ext_reg_data; c.RegInc = b->ext_reg_inc; c.RegReadCount = 0; c.NumRegisters = nreg;
	c.StartInstance = b->start_instance; c.NumInstances = b->num_instances; c.NumLoops = loops; c.PerInstanceFnInstanceIdx = 0;
	c.UserPtrTable = ctx->UserPtrTable.data; c.NumUserPtrs = ctx->UserPtrTable.num;
	c.RandCounters = &b->rand_counters; c.DataSets = ctx->DataSets.data; c.NumDataSets = ctx->DataSets.num;
	const vvm_ext_function* f = e->fn;
	if (f) {
		if (f->builtin_id == VVM_EXTFN_PERLIN3) builtin_perlin3(&c);
		else if (f->builtin_id == VVM_EXTFN_CURL_NOISE3) builtin_curl_noise3(&c);
		else if (f->host_fn) f->host_fn(&c);
	}
	return ip + 2 + 2 * nreg;
}

/* ------------------------------------------------------------------------------------------------
 * chunk execution -- VectorVMRuntime.cpp:2135-2175 with the dispatch loops of :1558-1904
 * ---------------------------------------------------------------------------------------------- */
#define RP(i) (b->reg_ptr[u16at(ip, (i))])
#define RI(i) ((int)b->reg_inc[u16at(ip, (i))])
#define LDF(p) _mm_loadu_ps((const float*)(p))
#define STF(p, v) _mm_storeu_ps((float*)(p), (v))

#define OP1(EXPR) { uint8_t *p0 = RP(0), *pd = RP(1); int i0 = RI(0); \
	for (int l = 0; l < loops; ++l) { __m128 a = LDF(p0); p0 += i0; __m128 r = (EXPR); STF(pd, r); pd += 16; } ip += 4; break; }
#define OP2(EXPR) { uint8_t *p0 = RP(0), *p1 = RP(1), *pd = RP(2); int i0 = RI(0), i1 = RI(1); \
	for (int l = 0; l < loops; ++l) { __m128 a = LDF(p0), bb = LDF(p1); p0 += i0; p1 += i1; __m128 r = (EXPR); STF(pd, r); pd += 16; } ip += 6; break; }
#define OP3(EXPR) { uint8_t *p0 = RP(0), *p1 = RP(1), *p2 = RP(2), *pd = RP(3); int i0 = RI(0), i1 = RI(1), i2 = RI(2); \
	for (int l = 0; l < loops; ++l) { __m128 a = LDF(p0), bb = LDF(p1), c = LDF(p2); p0 += i0; p1 += i1; p2 += i2; __m128 r = (EXPR); STF(pd, r); pd += 16; } ip += 8; break; }
#define OP4(EXPR) { uint8_t *p0 = RP(0), *p1 = RP(1), *p2 = RP(2), *p3 = RP(3), *pd = RP(4); int i0 = RI(0), i1 = RI(1), i2 = RI(2), i3 = RI(3); \
	for (int l = 0; l < loops; ++l) { __m128 a = LDF(p0), bb = LDF(p1), c = LDF(p2), d = LDF(p3); p0 += i0; p1 += i1; p2 += i2; p3 += i3; \
		__m128 r = (EXPR); STF(pd, r); pd += 16; } ip += 10; break; }
#define OP5(EXPR) { uint8_t *p0 = RP(0), *p1 = RP(1), *p2 = RP(2), *p3 = RP(3), *p4 = RP(4), *pd = RP(5); \
	int i0 = RI(0), i1 = RI(1), i2 = RI(2), i3 = RI(3), i4 = RI(4); \
	for (int l = 0; l < loops; ++l) { __m128 a = LDF(p0), bb = LDF(p1), c = LDF(p2), d = LDF(p3), e = LDF(p4); \
		p0 += i0; p1 += i1; p2 += i2; p3 += i3; p4 += i4; __m128 r = (EXPR); STF(pd, r); pd += 16; } ip += 12; break; }
/* integer views of the operands */
#define A ci(a)
#define B ci(bb)
#define Cc ci(c)
#define D ci(d)
#define FCMP(f) f(a, bb)
#define ICMP_LT _mm_cmplt_epi32(A, B)
#define ICMP_LE icmple(A, B)
#define ICMP_GT _mm_cmpgt_epi32(A, B)
#define ICMP_GE icmpge(A, B)
#define ICMP_EQ _mm_cmpeq_epi32(A, B)
#define ICMP_NE icmpne(A, B)

static void exec_chunk(oracle_state* s, vvm_exec_context* ctx, batch_state* b, int loops)
{
	set_chunk_pointers(s, b);
	const uint8_t* ip = s->bytecode;
	const int single = (loops == 1) && g_strict;
	uint32_t rand_ordinal = 0;
	const uint32_t inst0 = (uint32_t)(b->start_instance + g_rand_instance_base);
	for (;;) {
		uint8_t op = *ip++;
		switch (op) {
		case VVM_OP_done: return;
		case VVM_OP_add:  OP2(_mm_add_ps(a, bb))
		case VVM_OP_sub:  OP2(_mm_sub_ps(a, bb))
		case VVM_OP_mul:  OP2(_mm_mul_ps(a, bb))
		case VVM_OP_div:  OP2(vdiv(a, bb))
		case VVM_OP_mad:  OP3(vmad(a, bb, c))
		case VVM_OP_lerp: OP3(vlerp(a, bb, c))
		case VVM_OP_rcp:  OP1(vrcp(a))
		case VVM_OP_rsq:  OP1(vrsq(a))
		case VVM_OP_sqrt: OP1(vsqrtg(a))
		case VVM_OP_neg:  OP1(vneg(a))
		case VVM_OP_abs:  OP1(vabs(a))
		case VVM_OP_exp:  OP1(mf_exp(a))
		case VVM_OP_exp2: OP1(vexp2(a))
		case VVM_OP_log:  OP1(vlog(a))
		case VVM_OP_log2: OP1(vlog2(a))
		case VVM_OP_sin:  OP1(mf_sin(a))
		case VVM_OP_cos:  OP1(mf_cos(a))
		case VVM_OP_tan:  OP1(vtan(a))
		case VVM_OP_asin: OP1(vasin(a))
		case VVM_OP_acos: OP1(vacos(a))
		case VVM_OP_atan: OP1(vatan(a))
		case VVM_OP_atan2: OP2(vatan2(a, bb))
		case VVM_OP_ceil:  OP1(_mm_ceil_ps(a))
		case VVM_OP_floor: OP1(_mm_floor_ps(a))
		case VVM_OP_fmod:  OP2(vfmod(a, bb))
		case VVM_OP_frac:  OP1(vfrac(a))
		case VVM_OP_trunc: OP1(_mm_round_ps(a, _MM_FROUND_TRUNC))
		case VVM_OP_clamp: OP3(vclamp(a, bb, c))
		case VVM_OP_min:   OP2(_mm_min_ps(a, bb))
		case VVM_OP_max:   OP2(_mm_max_ps(a, bb))
		case VVM_OP_pow:   OP2(vpow(a, bb))
		case VVM_OP_round: OP1(_mm_round_ps(a, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC))
		case VVM_OP_sign:  OP1(vsign(a))
		case VVM_OP_step:  OP2(vstep(_mm_sub_ps(bb, a)))
		case VVM_OP_cmplt: OP2(_mm_cmplt_ps(a, bb))
		case VVM_OP_cmple: OP2(_mm_cmple_ps(a, bb))
		case VVM_OP_cmpgt: OP2(_mm_cmpgt_ps(a, bb))
		case VVM_OP_cmpge: OP2(_mm_cmpge_ps(a, bb))
		case VVM_OP_cmpeq: OP2(_mm_cmpeq_ps(a, bb))
		case VVM_OP_cmpneq: OP2(_mm_cmpneq_ps(a, bb))
		case VVM_OP_select: OP3(sel(a, bb, c))
		case VVM_OP_addi:  OP2(cf(_mm_add_epi32(A, B)))
		case VVM_OP_subi:  OP2(cf(_mm_sub_epi32(A, B)))
		case VVM_OP_muli:  OP2(cf(_mm_mullo_epi32(A, B)))
		case VVM_OP_divi:  OP2(cf(idiv(A, B)))
		case VVM_OP_clampi: OP3(cf(iclamp(A, B, Cc)))
		case VVM_OP_mini:  OP2(cf(_mm_min_epi32(A, B)))
		case VVM_OP_maxi:  OP2(cf(_mm_max_epi32(A, B)))
		case VVM_OP_absi:  OP1(cf(_mm_abs_epi32(A)))
		case VVM_OP_negi:  OP1(cf(_mm_sub_epi32(_mm_setzero_si128(), A)))
		case VVM_OP_signi: OP1(cf(isign(A)))
		case VVM_OP_cmplti: OP2(cf(ICMP_LT))
		case VVM_OP_cmplei: OP2(cf(ICMP_LE))
		case VVM_OP_cmpgti: OP2(cf(ICMP_GT))
		case VVM_OP_cmpgei: OP2(cf(ICMP_GE))
		case VVM_OP_cmpeqi: OP2(cf(ICMP_EQ))
		case VVM_OP_cmpneqi: OP2(cf(ICMP_NE))
		case VVM_OP_bit_and: case VVM_OP_logic_and: OP2(_mm_and_ps(a, bb))
		case VVM_OP_bit_or:  case VVM_OP_logic_or:  OP2(_mm_or_ps(a, bb))
		case VVM_OP_bit_xor: case VVM_OP_logic_xor: OP2(_mm_xor_ps(a, bb))
		case VVM_OP_bit_not: case VVM_OP_logic_not: OP1(cf(inot(A)))
		case VVM_OP_bit_lshift: OP2(cf(ishl(A, B)))
		case VVM_OP_bit_rshift: OP2(cf(ishr(A, B)))
		case VVM_OP_f2i: OP1(cf(F2I(a)))
		case VVM_OP_i2f: OP1(I2F(A))
		case VVM_OP_f2b: OP1(_mm_cmpgt_ps(a, _mm_setzero_ps()))
		case VVM_OP_b2f: OP1(sel(a, set1(1.f), _mm_setzero_ps()))
		case VVM_OP_i2b: OP1(cf(_mm_cmpgt_epi32(A, _mm_setzero_si128())))
		case VVM_OP_b2i: OP1(cf(seli(A, _mm_set1_epi32(1), _mm_setzero_si128())))
		/* merged ops, VectorVMRuntime.cpp:1999-2074 */
		case VVM_OP_cmplt_select:  OP4(sel(_mm_cmplt_ps(a, bb), c, d))
		case VVM_OP_cmple_select:  OP4(sel(_mm_cmple_ps(a, bb), c, d))
		case VVM_OP_cmpeq_select:  OP4(sel(_mm_cmpeq_ps(a, bb), c, d))
		case VVM_OP_cmplti_select: OP4(sel(cf(ICMP_LT), c, d))
		case VVM_OP_cmplei_select: OP4(sel(cf(ICMP_LE), c, d))
		case VVM_OP_cmpeqi_select: OP4(sel(cf(ICMP_EQ), c, d))
		case VVM_OP_cmplt_logic_and: OP3(_mm_and_ps(_mm_cmplt_ps(a, bb), c))
		case VVM_OP_cmple_logic_and: OP3(_mm_and_ps(_mm_cmple_ps(a, bb), c))
		case VVM_OP_cmpgt_logic_and: OP3(_mm_and_ps(_mm_cmpgt_ps(a, bb), c))
		case VVM_OP_cmpge_logic_and: OP3(_mm_and_ps(_mm_cmpge_ps(a, bb), c))
		case VVM_OP_cmpeq_logic_and: OP3(_mm_and_ps(_mm_cmpeq_ps(a, bb), c))
		case VVM_OP_cmpne_logic_and: OP3(_mm_and_ps(_mm_cmpneq_ps(a, bb), c))
		case VVM_OP_cmplti_logic_and: OP3(_mm_and_ps(cf(ICMP_LT), c))
		case VVM_OP_cmplei_logic_and: OP3(_mm_and_ps(cf(ICMP_LE), c))
		case VVM_OP_cmpgti_logic_and: OP3(_mm_and_ps(cf(ICMP_GT), c))
		case VVM_OP_cmpgei_logic_and: OP3(_mm_and_ps(cf(ICMP_GE), c))
		case VVM_OP_cmpeqi_logic_and: OP3(_mm_and_ps(cf(ICMP_EQ), c))
		case VVM_OP_cmpnei_logic_and: OP3(_mm_and_ps(cf(ICMP_NE), c))
		case VVM_OP_cmplt_logic_or: OP3(_mm_or_ps(_mm_cmplt_ps(a, bb), c))
		case VVM_OP_cmple_logic_or: OP3(_mm_or_ps(_mm_cmple_ps(a, bb), c))
		case VVM_OP_cmpgt_logic_or: OP3(_mm_or_ps(_mm_cmpgt_ps(a, bb), c))
		case VVM_OP_cmpge_logic_or: OP3(_mm_or_ps(_mm_cmpge_ps(a, bb), c))
		case VVM_OP_cmpeq_logic_or: OP3(_mm_or_ps(_mm_cmpeq_ps(a, bb), c))
		case VVM_OP_cmpne_logic_or: OP3(_mm_or_ps(_mm_cmpneq_ps(a, bb), c))
		case VVM_OP_cmplti_logic_or: OP3(_mm_or_ps(cf(ICMP_LT), c))
		case VVM_OP_cmplei_logic_or: OP3(_mm_or_ps(cf(ICMP_LE), c))
		case VVM_OP_cmpgti_logic_or: OP3(_mm_or_ps(cf(ICMP_GT), c))
		case VVM_OP_cmpgei_logic_or: OP3(_mm_or_ps(cf(ICMP_GE), c))
		case VVM_OP_cmpeqi_logic_or: OP3(_mm_or_ps(cf(ICMP_EQ), c))
		case VVM_OP_cmpnei_logic_or: OP3(_mm_or_ps(cf(ICMP_NE), c))
		case VVM_OP_mad_add:  OP4(_mm_add_ps(vmad(a, bb, c), d))
		case VVM_OP_mad_sub0: OP4(_mm_sub_ps(vmad(a, bb, c), d))
		case VVM_OP_mad_sub1: OP4(_mm_sub_ps(d, vmad(a, bb, c)))
		case VVM_OP_mad_mul:  OP4(_mm_mul_ps(vmad(a, bb, c), d))
		case VVM_OP_mad_sqrt: OP3(_mm_sqrt_ps(vmad(a, bb, c)))
		case VVM_OP_mad_mad0: OP5(vmad(d, e, vmad(a, bb, c)))
		case VVM_OP_mad_mad1: OP5(vmad(vmad(a, bb, c), d, e))
		case VVM_OP_mul_mad0: OP4(vmad(_mm_mul_ps(a, bb), c, d))
		case VVM_OP_mul_mad1: OP4(vmad(c, d, _mm_mul_ps(a, bb)))
		case VVM_OP_mul_add:  OP3(_mm_add_ps(_mm_mul_ps(a, bb), c))
		case VVM_OP_mul_sub0: OP3(_mm_sub_ps(_mm_mul_ps(a, bb), c))
		case VVM_OP_mul_sub1: OP3(_mm_sub_ps(c, _mm_mul_ps(a, bb)))
		case VVM_OP_mul_mul:  OP3(_mm_mul_ps(_mm_mul_ps(a, bb), c))
		case VVM_OP_mul_max:  OP3(_mm_max_ps(_mm_mul_ps(a, bb), c))
		case VVM_OP_add_mad1: OP4(vmad(c, d, _mm_add_ps(a, bb)))
		case VVM_OP_add_add:  OP3(_mm_add_ps(_mm_add_ps(a, bb), c))
		case VVM_OP_sub_cmplt1: OP3(_mm_cmplt_ps(c, _mm_sub_ps(a, bb)))
		case VVM_OP_sub_neg:  OP2(vneg(_mm_sub_ps(a, bb)))
		case VVM_OP_sub_mul:  OP3(_mm_mul_ps(_mm_sub_ps(a, bb), c))
		case VVM_OP_div_mad0: OP4(vmad(vdiv(a, bb), c, d))
		case VVM_OP_div_f2i:  OP2(cf(F2I(vdiv(a, bb))))
		case VVM_OP_div_mul:  OP3(_mm_mul_ps(vdiv(a, bb), c))
		case VVM_OP_muli_addi: OP3(cf(_mm_add_epi32(_mm_mullo_epi32(A, B), Cc)))
		case VVM_OP_addi_bit_rshift: OP3(cf(ishr(_mm_add_epi32(A, B), Cc)))
		case VVM_OP_addi_muli: OP3(cf(_mm_mullo_epi32(_mm_add_epi32(A, B), Cc)))
		case VVM_OP_i2f_div0: OP2(vdiv(I2F(A), bb))
		case VVM_OP_i2f_div1: OP2(vdiv(bb, I2F(A)))
		case VVM_OP_i2f_mul:  OP2(_mm_mul_ps(I2F(A), bb))
		case VVM_OP_i2f_mad0: OP3(vmad(I2F(A), bb, c))
		case VVM_OP_i2f_mad1: OP3(vmad(a, bb, I2F(Cc)))
		case VVM_OP_f2i_select1: OP3(cf(seli(A, F2I(bb), Cc)))
		case VVM_OP_f2i_maxi: OP2(cf(_mm_max_epi32(F2I(a), B)))
		case VVM_OP_f2i_addi: OP2(cf(_mm_add_epi32(F2I(a), B)))
		case VVM_OP_fmod_add: OP3(_mm_add_ps(vfmod(a, bb), c))
		case VVM_OP_bit_and_i2f: OP2(I2F(_mm_and_si128(A, B)))
		case VVM_OP_bit_rshift_bit_and: OP3(cf(_mm_and_si128(ishr(A, B), Cc)))
		case VVM_OP_neg_cmplt: OP2(_mm_cmplt_ps(vneg(a), bb))
		case VVM_OP_bit_or_muli: OP3(cf(_mm_mullo_epi32(_mm_or_si128(A, B), Cc)))
		case VVM_OP_bit_lshift_bit_or: OP3(cf(_mm_or_si128(ishl(A, B), Cc)))
		case VVM_OP_max_f2i: OP2(cf(F2I(_mm_max_ps(a, bb))))
		case VVM_OP_select_mul: OP4(_mm_mul_ps(sel(a, bb, c), d))
		case VVM_OP_select_add: OP4(_mm_add_ps(sel(a, bb, c), d))

		case VVM_OP_mul_2x: {   /* VectorVMRuntime.cpp:1743-1777 */
			uint8_t *p0 = RP(0), *p1 = RP(1), *d0 = RP(2), *d1 = RP(3); int i0 = RI(0), i1 = RI(1);
			for (int l = 0; l < loops; ++l) { __m128 r = _mm_mul_ps(LDF(p0), LDF(p1)); p0 += i0; p1 += i1; STF(d0, r); STF(d1, r); d0 += 16; d1 += 16; }
			ip += 8; break; }
		case VVM_OP_b2i_2x: {   /* VectorVMRuntime.cpp:1875-1904, :2055 */
			uint8_t *p0 = RP(0), *d0 = RP(1), *d1 = RP(2); int i0 = RI(0);
			for (int l = 0; l < loops; ++l) { __m128 r = cf(seli(ci(LDF(p0)), _mm_set1_epi32(1), _mm_setzero_si128())); p0 += i0; STF(d0, r); STF(d1, r); d0 += 16; d1 += 16; }
			ip += 6; break; }
		case VVM_OP_sin_cos: {  /* VectorVMRuntime.cpp:1491-1516 */
			uint8_t *p0 = RP(0), *d0 = RP(1), *d1 = RP(2); int i0 = RI(0);
			for (int l = 0; l < loops; ++l) { __m128 sn, cs; vsincos(LDF(p0), &sn, &cs); p0 += i0; STF(d0, sn); STF(d1, cs); d0 += 16; d1 += 16; }
			ip += 6; break; }
		case VVM_OP_random: {   /* VectorVMRuntime.cpp:990-1016 */
			uint8_t *p0 = RP(0), *d0 = RP(1); int i0 = RI(0);
			for (int l = 0; l < loops; ++l) { STF(d0, rand4(b->xorwow, inst0 + 4u * (uint32_t)l, rand_ordinal, LDF(p0))); p0 += i0; d0 += 16; }
			++rand_ordinal; ip += 4; break; }
		case VVM_OP_randomi: {  /* VectorVMRuntime.cpp:1018-1043; intended semantics, see header */
			uint8_t *p0 = RP(0), *d0 = RP(1); int i0 = RI(0);
			for (int l = 0; l < loops; ++l) { STF(d0, cf(F2I(rand4(b->xorwow, inst0 + 4u * (uint32_t)l, rand_ordinal, I2F(ci(LDF(p0))))))); p0 += i0; d0 += 16; }
			++rand_ordinal; ip += 4; break; }
		case VVM_OP_random_add: { /* VectorVMRuntime.cpp:1045-1077 */
			uint8_t *p0 = RP(0), *p1 = RP(1), *d0 = RP(2); int i0 = RI(0), i1 = RI(1);
			for (int l = 0; l < loops; ++l) { STF(d0, _mm_add_ps(rand4(b->xorwow, inst0 + 4u * (uint32_t)l, rand_ordinal, LDF(p0)), LDF(p1))); p0 += i0; p1 += i1; d0 += 16; }
			++rand_ordinal; ip += 6; break; }
		case VVM_OP_random_2x: { /* VectorVMRuntime.cpp:1079-1111 */
			uint8_t *p0 = RP(0), *d0 = RP(1), *d1 = RP(2); int i0 = RI(0);
			for (int l = 0; l < loops; ++l) { __m128 a = LDF(p0); p0 += i0;
				STF(d0, rand4(b->xorwow, inst0 + 4u * (uint32_t)l, rand_ordinal, a)); STF(d1, rand4(b->xorwow, inst0 + 4u * (uint32_t)l, rand_ordinal + 1, a)); d0 += 16; d1 += 16; }
			rand_ordinal += 2; ip += 6; break; }
		case VVM_OP_exec_index: case VVM_OP_exec_indexf: {   /* VectorVMRuntime.cpp:880-923 */
			uint8_t* d0 = RP(0);
			__m128i v = _mm_add_epi32(_mm_set1_epi32(single ? 0 : b->start_instance + g_exec_index_base), _mm_setr_epi32(0, 1, 2, 3));
			for (int l = 0; l < loops; ++l) {
				if (op == VVM_OP_exec_index) _mm_storeu_si128((__m128i*)d0, v); else STF(d0, I2F(v));
				v = _mm_add_epi32(v, _mm_set1_epi32(4)); d0 += 16; }
			ip += 2; break; }
		case VVM_OP_exec_index_addi: {   /* VectorVMRuntime.cpp:925-951 */
			uint8_t *p0 = RP(0), *d0 = RP(1); int i0 = RI(0);
			__m128i v = _mm_add_epi32(_mm_set1_epi32(single ? 0 : b->start_instance + g_exec_index_base), _mm_setr_epi32(0, 1, 2, 3));
			for (int l = 0; l < loops; ++l) { _mm_storeu_si128((__m128i*)d0, _mm_add_epi32(v, ci(LDF(p0)))); p0 += i0;
				v = _mm_add_epi32(v, _mm_set1_epi32(4)); d0 += 16; }
			ip += 4; break; }
		case VVM_OP_half_to_float: {     /* VectorVMRuntime.cpp:1113-1127 */
			uint8_t *p0 = RP(0), *d0 = RP(1); int i0 = RI(0) >> 1;
			for (int l = 0; l < loops; ++l) { const uint16_t* h = (const uint16_t*)p0; float f[4];
				for (int k = 0; k < 4; ++k) f[k] = load_half(h[i0 ? k : 0]);
				STF(d0, _mm_loadu_ps(f)); p0 += i0; d0 += 16; }
			ip += 4; break; }
		case VVM_OP_acquireindex: ip = do_acquireindex(s, ctx, b, ip); break;
		case VVM_OP_outputdata_float: case VVM_OP_outputdata_int32: ip = do_output32(s, b, ip); break;
		case VVM_OP_outputdata_half: ip = do_output16(b, ip); break;
		case VVM_OP_outputdata_float_from_half: ip = do_output_from_half(b, ip, 1); break;
		case VVM_OP_outputdata_half_from_half: ip = do_output_from_half(b, ip, 0); break;
		case VVM_OP_external_func_call: ip = do_extfn(s, ctx, b, ip, loops); break;
		case VVM_OP_update_id: ip = do_update_id(ctx, b, ip); break;
		case VVM_OP_acquire_id: ip = do_acquire_id(ctx, b, ip); break;
		case VVM_OP_enter_stat_scope: ip += 1; break;     /* VectorVMRuntime.cpp:1990 */
		default: break;                                   /* no-op stubs: inputs, noise*, fasi/iasf, exit_stat_scope */
		}
	}
}

/* ------------------------------------------------------------------------------------------------
 * ExecVectorVMState -- VectorVMRuntime.cpp:2182-2309
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API int vvm_oracle_exec(vvm_exec_context* ctx)
{
	if (!ctx || !ctx->VVMState) return VVM_E_INVALID_ARG;
	oracle_state* s = (oracle_state*)ctx->VVMState;
	build_tables();
	if ((uint32_t)ctx->ExtFunctionTable.num != s->num_ext) return VVM_E_EXTFN_MISMATCH;
	for (uint32_t i = 0; i < s->num_ext; ++i) s->ext[i].fn = ctx->ExtFunctionTable.data[i];
	for (uint32_t i = 0; i < s->max_output_dataset; ++i) s->num_output_per_dataset[i] = 0;
	if (!s->map_cache_ready) {
		int rc = build_map_caches(s, ctx);
		if (rc != VVM_OK) return rc;
		s->map_cache_ready = 1;
	}
	for (uint32_t i = 0; i < s->num_consts; ++i) {
		uint32_t v = ((const uint32_t*)ctx->ConstantTableData[s->const_map_idx[i]])[s->const_map_src[i]];
		s->const_buffers[i] = _mm_set1_epi32((int)v);
	}
	/* chunking, VectorVMRuntime.cpp:2218-2261 */
	uint32_t max_loops;
	{
		size_t page = (size_t)g_page_kb << 10;
		uint32_t total_loops = ((uint32_t)ctx->NumInstances + 3) >> 2;
		if (total_loops < 1) total_loops = 1;
		size_t per_loop = 16u * (size_t)s->num_temps + s->max_output_dataset;
		size_t avail = page - s->batch_overhead_size;
		size_t need = (((size_t)total_loops + 3) & ~(size_t)3) * per_loop;
		int chunks;
		if (avail < need) {
			chunks = (int)((need + avail - 1) / avail);
			max_loops = (total_loops + (uint32_t)chunks - 1) / (uint32_t)chunks;
		} else { chunks = 1; max_loops = total_loops; }
		ctx->Internal.PerBatchRegisterDataBytesRequired = (uint32_t)(max_loops * per_loop);
		ctx->Internal.NumBytesRequiredPerBatch = s->batch_overhead_size + ctx->Internal.PerBatchRegisterDataBytesRequired;
		ctx->Internal.MaxChunksPerBatch = (uint32_t)chunks;
		ctx->Internal.MaxInstancesPerChunk = max_loops << 2;
	}
	if (!s->bytecode || s->num_bytecode_bytes == 0) return VVM_E_NO_BYTECODE;
	batch_state b;
	int rc = setup_batch(s, ctx, &b, max_loops);
	if (rc != VVM_OK) { free_batch(&b); return rc; }
	memcpy(b.xorwow, g_xorwow_start, sizeof b.xorwow);      /* SetupRandStateForBatch: once per Exec (:2271-2274) */
	int per_chunk = (int)ctx->Internal.MaxInstancesPerChunk;
	int nchunks = (ctx->NumInstances + per_chunk - 1) / per_chunk;
	int start = 0;
	for (int c = 0; c < nchunks; ++c, start += per_chunk) {
		int n = ctx->NumInstances - start < per_chunk ? ctx->NumInstances - start : per_chunk;
		int loops = ((n + 3) & ~3) >> 2;
		b.chunk_idx = c; b.start_instance = start; b.num_instances = n;
		if (loops >= 1) exec_chunk(s, ctx, &b, loops);
	}
	free_batch(&b);
	for (uint32_t i = 0; i < s->max_output_dataset && (int)i < ctx->DataSets.num; ++i)
		ctx->DataSets.data[i].DataSetAccessIndex = s->num_output_per_dataset[i] - 1;
	return VVM_OK;
}

/* scalar entry points for per-function tests */
ORACLE_API float    vvm_oracle_perlin3(float x, float y, float z) { return perlin3(x, y, z); }
ORACLE_API void     vvm_oracle_curl_noise3(float x, float y, float z, float* o) { curl_noise3(x, y, z, o, o + 1, o + 2); }
ORACLE_API uint32_t vvm_oracle_rand_u32(uint64_t seed, uint32_t instance, uint32_t ordinal) { return rand_u32(seed, instance, ordinal); }
ORACLE_API uint16_t vvm_oracle_store_half(float f) { return store_half(f); }
ORACLE_API float    vvm_oracle_load_half(uint16_t h) { return load_half(h); }
/* the compaction tables the output instructions shuffle by, for the test that compares them with the reference's own
 * (VVM_PSHUFB_OUTPUT_TABLE, _TABLE16 and the two advance tables, VectorVMRuntime.cpp:132-186): 256 + 256 + 16 + 16 bytes */
ORACLE_API void     vvm_oracle_compaction_tables(uint8_t* shuf32, uint8_t* shuf16, uint8_t* adv32, uint8_t* adv16)
{
	build_tables();
	memcpy(shuf32, g_shuf32, 256); memcpy(shuf16, g_shuf16, 256); memcpy(adv32, g_adv32, 16); memcpy(adv16, g_adv16, 16);
}
