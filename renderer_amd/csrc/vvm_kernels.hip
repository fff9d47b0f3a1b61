// vvm_kernels.hip -- the gfx950 VectorVM interpreter kernel.
//
// Replaces the reference's chunk loop (ExecVectorVMState step 8, VectorVMRuntime.cpp:2276-2295, with
// ExecChunkMultipleLoops :2157-2175 and the per-op dispatch loops :1558-1904).  Where the reference runs
// chunks serially and every op makes a 4-wide SSE pass over a chunk, here a 256-thread workgroup owns a
// tile of 256*V consecutive instances, every lane owns V consecutive instances (V = 4: one 128-bit access
// per stream per lane), and the instruction stream is decoded once per wave from scalar memory.
//
//   registers   per-lane virtual registers live in a VGPR vector indexed through the scalar GPR-index mode
//               (uniform index), the rest in LDS laid out [register][thread]
//   inputs      VVM_DOP_LOADN issues up to 8 stream loads back to back before the first use
//   acquireindex  sign bit of the predicate -> wave ballots + popcounts -> 4-wave scan in LDS -> decoupled
//               look-back over per-tile words in global memory for the exclusive prefix across tiles.
//               Survivors keep instance order (VectorVMRuntime.cpp:786-804, 824-878).
//   outputs     written at InstanceOffset + prefix + rank; a tile with no deaths stores 128 bits per lane
//
// Arithmetic follows the reference's SSE math layer lane by lane (citations at each helper); it is compiled
// with -ffp-contract=off so that every multiply-add keeps two roundings (UE_PLATFORM_MATH_USE_FMA3 = 0).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vvm_abi.h"
#include "vvm_device.h"
#include "vvm_kernels.h"

typedef float    f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint16_t h4u __attribute__((ext_vector_type(4), aligned(2)));
typedef float    f4a __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

#define DEV __device__ __forceinline__
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Uniform tables (program, constants, stream pointers, descriptors) are never written while a launch runs, so they
// are read through the constant address space: the compiler then issues scalar loads and keeps the values in SGPRs,
// which is what lets a register index drive the GPR-index mode instead of a per-lane select chain.
typedef const __attribute__((address_space(4))) uint32_t* cu32p;
typedef const __attribute__((address_space(4))) uint64_t* cu64p;
// (the builtin returns a signed int: the cast keeps a 64-bit composition from sign-extending its low half)
#define SCALAR(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint32_t sload32(const void* p, uint32_t i) { return SCALAR(((cu32p)(uintptr_t)p)[i]); }
__device__ __forceinline__ uint64_t sload64(const void* p, uint32_t i)
{
	const uint64_t v = ((cu64p)(uintptr_t)p)[i];
	return ((uint64_t)SCALAR((uint32_t)(v >> 32)) << 32) | (uint64_t)SCALAR((uint32_t)v);
}

// 2^K consecutive 64-bit table entries with one scalar load (the table must be aligned to the size of the load)
typedef uint32_t su32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t su32x16 __attribute__((ext_vector_type(16)));
// four consecutive dwords at a 32-bit byte offset (one scalar load that takes the offset as it is)
typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sload32x4_at(const void* p, uint32_t byte_offset, uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3)
{
	const su32x4 v = *(const __attribute__((address_space(4))) su32x4*)((uintptr_t)p + (uint64_t)byte_offset);
	w0 = SCALAR(v[0]); w1 = SCALAR(v[1]); w2 = SCALAR(v[2]); w3 = SCALAR(v[3]);
}
__device__ __forceinline__ void sload64x4(const void* p, uint32_t first, uint64_t (&out)[4])
{
	const su32x8 v = *(const __attribute__((address_space(4))) su32x8*)(uintptr_t)((const uint64_t*)p + first);
#pragma unroll
	for (int i = 0; i < 4; ++i) out[i] = ((uint64_t)SCALAR(v[2 * i + 1]) << 32) | (uint64_t)SCALAR(v[2 * i]);
}
__device__ __forceinline__ void sload64x8(const void* p, uint32_t first, uint64_t (&out)[8])
{
	const su32x16 v = *(const __attribute__((address_space(4))) su32x16*)(uintptr_t)((const uint64_t*)p + first);
#pragma unroll
	for (int i = 0; i < 8; ++i) out[i] = ((uint64_t)SCALAR(v[2 * i + 1]) << 32) | (uint64_t)SCALAR(v[2 * i]);
}

// VGPR-resident virtual registers: one vector per instance slot of the lane, R<k>[i] = register i of the lane's k-th
// instance.  The vectors are plain local values that are passed to helpers BY VALUE and assigned back: handed around by
// reference they stay in a stack object whose run-time-indexed element stores the compiler leaves in scratch memory; as
// values each one is an SSA vector indexed through the scalar GPR-index mode (s_set_gpr_idx_on + v_mov_b32).
typedef float vec16f __attribute__((ext_vector_type(16)));
typedef float vec32f __attribute__((ext_vector_type(32)));
// Resident registers per kernel shape.  SPILL marks the kernels for scripts with many registers: at four instances per lane
// they hold 32 registers in VGPRs (and further ones in LDS slots) at half the occupancy of the 16-register kernel.
template <int V, bool SPILL> struct RegTraits;
template <> struct RegTraits<4, false> { typedef vec16f vec; static constexpr uint32_t RV = 16; };
template <> struct RegTraits<4, true> { typedef vec32f vec; static constexpr uint32_t RV = 32; };
template <> struct RegTraits<8, false> { typedef vec16f vec; static constexpr uint32_t RV = 16; };
// <8, true> is the LIGHT eight-wide kernel, for scripts that get by with eight registers: no LDS register slots (the flag means
// something else at this width), a 64-VGPR register file, six staging slots; 168 VGPRs and 50 KB of LDS per workgroup, so that
// three workgroups fit a CU instead of two.  The interpreter is bound by the latency of a single wave's instruction stream
// (a CU with one workgroup interprets half as fast as with two, with four of the four-wide kernel twice as fast as with two),
// so a third resident workgroup is what hides the tile's memory phases.
// (ten elements: registers 0..7 and the two that hold riding outputs, VVM_LIGHT_RIDE_REG; the backend turns a run-time index into an eight-element vector into compare-and-select
// chains instead of the GPR-index mode, half as fast per interpreted instruction as the sixteen-element file; from nine on it indexes)
typedef float vec8f __attribute__((ext_vector_type(10)));
template <> struct RegTraits<8, true> { typedef vec8f vec; static constexpr uint32_t RV = 8; };
#define IS_LIGHT(V, SPILL) ((V) == 8 && (SPILL))
// Two acquired datasets in one pass ("chains": each has its own keep mask, ranks, count and look-back words): the general
// eight-wide kernel and the four-wide kernel with 32 registers have the registers to spare for the second set.
#define HAS_TWO_CHAINS(V, SPILL, FULL) (!(FULL) && (((V) == 8 && !(SPILL)) || ((V) == 4 && (SPILL))))
#define VVM_LIGHT_STAGE_SLOTS 6u

// Instance slots of a lane.  V = 1: the lane's one instance.  V = 4: four consecutive instances, [4 tid, 4 tid + 4) of the
// tile.  V = 8: two such groups, the second one VVM_HALF_TILE instances after the first: the tile is two halves of 1024
// instances, and slots 0..3 / 4..7 of lane `tid` are instances [4 tid, 4 tid + 4) of the first / second half.  Every memory
// access stays a 128-bit access of four consecutive instances per lane, and instance order is (half, lane, slot).
#define VVM_HALF_TILE (VVM_BLOCK_THREADS * 4u)
template <int V> __device__ __forceinline__ constexpr uint32_t lane_stride() { return V == 8 ? 4u : (uint32_t)V; }     // instances between neighbouring lanes
template <int V> __device__ __forceinline__ constexpr uint32_t slot_offset(int k) { return (V == 8 && k >= 4) ? VVM_HALF_TILE - 4u + (uint32_t)k : (uint32_t)k; }
template <> struct RegTraits<1, false> { typedef vec32f vec; static constexpr uint32_t RV = 32; };
template <> struct RegTraits<1, true> { typedef vec32f vec; static constexpr uint32_t RV = 32; };
template <typename VEC> __device__ __forceinline__ float vget(VEC v, uint32_t i) { return v[i]; }
template <typename VEC> __device__ __forceinline__ VEC vset(VEC v, uint32_t i, float x) { v[i] = x; return v; }

// LDS carve (dynamic region only, every offset a multiple of 16)
//   [0, 256)      control words (uint32), see the LC_ constants
//   [256, 768)    Perlin permutation table
//   kernels with the noise built-ins (FULL) only:
//   [768, 1280)   the same table as gradient selectors: (entry & 15) * 16, the byte offset of the entry's row below
//   [1280, 1536)  sixteen gradient rows of four dwords (see p_grad3)
//   S = 896, FULL: 1664
//   [S, ...)      output staging: VVM_STAGE_SLOTS slots of one tile's worth of 32-bit values, then one sink dword per thread,
//                 then one dword per thread with the lane's acquireindex result (kept for update_id)
//   then          register spill slots
#define LDS_PERLIN_OFF      256
#define LDS_PERLIN_SEL_OFF  768
#define LDS_PERLIN_GRAD_OFF 1280
#define LDS_STAGE_OFF(FULL)   ((FULL) ? 1664u : 896u)
// (LIGHT: the eight-wide kernel with six staging slots, see RegTraits<8, true>)
#define LDS_SLOTS(LIGHT) ((LIGHT) ? VVM_LIGHT_STAGE_SLOTS : (uint32_t)VVM_STAGE_SLOTS)
#define LDS_STAGE_SINK_OFF(V, FULL, LIGHT) (LDS_STAGE_OFF(FULL) + LDS_SLOTS(LIGHT) * VVM_BLOCK_THREADS * (V) * 4u)
#define LDS_STAGE_KEEP_OFF(V, FULL, LIGHT) (LDS_STAGE_SINK_OFF(V, FULL, LIGHT) + VVM_BLOCK_THREADS * 4u)
// (LIGHT: and 1 KB more.  Wave 0 parks the look-back words it polled ahead of time there and in the sink words, which nothing uses
// before the tile's first output, while the tile's program runs: 2 KB.  A light workgroup may use 53760 bytes: LDS is handed out in
// units of 1280 bytes, and three workgroups share a CU's 160 KB.)
#define LDS_STAGE_BYTES(V, LIGHT)  (LDS_SLOTS(LIGHT) * VVM_BLOCK_THREADS * (V) * 4u + 2u * VVM_BLOCK_THREADS * 4u + ((LIGHT) ? 1024u : 0u))
#define LDS_LIGHT_PRE_OFF (LDS_STAGE_KEEP_OFF(8, false, true) + VVM_BLOCK_THREADS * 4u)
static_assert(LDS_STAGE_OFF(false) + LDS_STAGE_BYTES(8, true) <= 53760u, "the light kernel's LDS no longer fits three workgroups per CU");
#define LDS_SPILL_OFF(V, FULL, LIGHT) (LDS_STAGE_OFF(FULL) + LDS_STAGE_BYTES(V, LIGHT))

DEV uint32_t fbits(float f) { return __float_as_uint(f); }
DEV float    bitsf(uint32_t u) { return __uint_as_float(u); }
DEV float    mask(bool b) { return bitsf(b ? 0xFFFFFFFFu : 0u); }

// ---- lane arithmetic ---------------------------------------------------------------------------------
// correctly rounded square root (_mm_sqrt_ps); HIP's __fsqrt_rn is the hardware approximation unless OCML's rounded
// operations are enabled, the plain sqrtf is IEEE under the compiler's default -fhip-fp32-correctly-rounded-divide-sqrt
DEV float ieee_sqrt(float a) { return __builtin_sqrtf(a); }
DEV float f_sel(float m, float a, float b) { uint32_t M = fbits(m); return bitsf((M & fbits(a)) | (~M & fbits(b))); }   // SSE.h:1805-1810
DEV float f_min(float a, float b) { return (a < b) ? a : b; }                  // _mm_min_ps: second operand when unordered or equal, SSE.h:2380-2383
DEV float f_max(float a, float b) { return (a > b) ? a : b; }                  // SSE.h:2404-2407
DEV float f_abs(float a) { return bitsf(fbits(a) & 0x7FFFFFFFu); }             // SSE.h:1287-1290
DEV float f_neg(float a) { return 0.0f - a; }                                  // SSE.h:1311-1314
DEV float f_mad(float a, float b, float c) { return (a * b) + c; }             // SSE.h:1411-1418 with FMA3 off
// Every helper below computes its value unconditionally and then selects: a lazily evaluated `cond ? f(x) : 0` leaves a
// per-lane branch in the interpreter loop, and one per-lane branch anywhere in that loop makes the compiler rewrite all of
// its (wave-uniform) control flow into flag-guarded if-chains.
DEV float f_div(float a, float b) { const float q = a / b; return (f_abs(b) > 1.e-8f) ? q : 0.0f; }                     // RT.cpp:1909
DEV float f_sqrtg(float a) { const float q = ieee_sqrt(a); return (a > 1.e-8f) ? q : 0.0f; }                           // RT.cpp:1914
DEV float f_rsq(float a) { const float q = 1.0f / ieee_sqrt(a); return (a > 1.e-8f) ? q : 0.0f; }                      // RT.cpp:1913, SSE.h:2014-2022
DEV float f_rcp(float a) { const float q = __builtin_amdgcn_rcpf(a); return (f_abs(a) > 1.e-8f) ? q : 0.0f; }           // RT.cpp:1912 (hardware estimate, SSE.h:2121-2125)
DEV float f_lerp(float a, float b, float c) { return f_mad(b, c, a * (1.0f - c)); }           // VCOMMON.inl:532-536
DEV float f_clamp(float a, float b, float c) { return f_min(f_max(a, b), c); }                // VCOMMON.inl:495-498
DEV float f_sign(float a) { return (a >= 0.0f) ? 1.0f : -1.0f; }                              // SSE.h:2911-2915
DEV float f_step(float a) { return (a >= 0.0f) ? 1.0f : 0.0f; }                               // SSE.h:2923-2927
DEV float f_frac(float a) { return a - truncf(a); }                                           // VCOMMON.inl:883-886
// libm-class functions stay out of line: inlined V times each into the dispatch switch they would set the kernel's register
// allocation (and with it the occupancy of every script, including the ones that never call them)
__device__ __noinline__ float lib_fmod(float a, float b) { return fmodf(a, b); }
__device__ __noinline__ float lib_exp2(float a) { return powf(2.0f, a); }                       // SSE.h:2960-2972, GPM.h:491
__device__ __noinline__ float lib_log2(float a) { return logf(a) * 1.4426950f; }                // SSE.h:3015-3027, GPM.h:502
__device__ __noinline__ float lib_tan(float a) { return tanf(a); }                              // SSE.h:3188-3201
__device__ __noinline__ float lib_atan(float a) { return atanf(a); }                            // SSE.h:3279-3292
DEV float f_fmod(float a, float b) { const float q = lib_fmod(a, b); return (f_abs(b) <= 1.e-8f) ? 0.0f : q; }   // SSE.h:2867-2886
// _mm_cvttps_epi32 / _mm_cvtps_epi32: 0x80000000 for NaN and out-of-range (SSE.h:3409-3412)
DEV int32_t cvtt(float a) { const int32_t q = (int32_t)a; return (a >= -2147483648.0f && a < 2147483648.0f) ? q : (int32_t)0x80000000; }
DEV int32_t cvtr(float a) { const int32_t q = __float2int_rn(a); return (a >= -2147483648.0f && a < 2147483648.0f) ? q : (int32_t)0x80000000; }
DEV float   i2f(int32_t a) { return (float)a; }                                               // SSE.h:3407

// atan2 as the reference computes it (GenericPlatformMath.cpp:33-76): reduce to the first octant, where the ratio of the smaller to
// the larger magnitude lies in [0, 1]; the angle there is ratio * P(ratio^2) with the reference's six minimax coefficients, evaluated
// by Horner's rule in single precision in this order (bit parity depends on it); then undo the reduction by three reflections.
DEV float first_octant_angle(float ratio)
{
	const float coeff[7] = {+7.2128853633444123e-03f, -3.5059680836411644e-02f, +8.1675882859940430e-02f, -1.3374657325451267e-01f,
	                        +1.9856563505717162e-01f, -3.3324998579202170e-01f, 1.0f};
	const float square = ratio * ratio;
	float poly = coeff[0];
#pragma unroll
	for (int i = 1; i < 7; ++i) poly = poly * square + coeff[i];
	return poly * ratio;
}
DEV float ue_atan2(float y, float x)
{
	const float pi = 3.1415926535897932f;
	const float mag_x = f_abs(x), mag_y = f_abs(y);
	const bool steep = mag_y > mag_x;                                   // above the diagonal: the roles of x and y swap
	const float larger = steep ? mag_y : mag_x, smaller = steep ? mag_x : mag_y;
	float angle = first_octant_angle(smaller / larger);
	angle = steep ? (0.5f * pi) - angle : angle;                        // reflect about the diagonal
	angle = (x < 0.0f) ? pi - angle : angle;                            // ... about the vertical axis
	angle = (y < 0.0f) ? -angle : angle;                                // ... about the horizontal axis
	return (larger == 0.f) ? 0.f : angle;                               // atan2(0, 0) = 0 (:47-50)
}
DEV float f_acos(float a) { return ue_atan2(f_sqrtg((1.0f - a) * (1.0f + a)), a); }           // RT.cpp:1924
DEV float f_asin(float a) { return (3.14159265359f * 0.25f) - f_acos(a); }                    // RT.cpp:1925 with :111

// sse_mathfun sin/cos/exp/log: third-party header outside the reference tree (Core/Private/Math/UnrealMathSSE.cpp:8-9);
// restated from its published algorithm (Cephes single-precision polynomials), same operation order.
DEV float mf_log(float x)
{
	const bool invalid = (x <= 0.0f);
	x = (x > bitsf(0x00800000u)) ? x : bitsf(0x00800000u);                 // _mm_max_ps(x, min_norm_pos)
	int32_t e0 = (int32_t)(fbits(x) >> 23);
	x = bitsf((fbits(x) & ~0x7f800000u) | fbits(0.5f));
	e0 -= 0x7f;
	float e = (float)e0 + 1.0f;
	const bool lt = x < 0.707106781186547524f;
	float tmp = lt ? x : 0.0f;
	x = x - 1.0f;
	e = e - (lt ? 1.0f : 0.0f);
	x = x + tmp;
	float z = x * x;
	float y = 7.0376836292E-2f;
	y = y * x + -1.1514610310E-1f;
	y = y * x + 1.1676998740E-1f;
	y = y * x + -1.2420140846E-1f;
	y = y * x + 1.4249322787E-1f;
	y = y * x + -1.6668057665E-1f;
	y = y * x + 2.0000714765E-1f;
	y = y * x + -2.4999993993E-1f;
	y = y * x + 3.3333331174E-1f;
	y = y * x;
	y = y * z;
	y = y + e * -2.12194440e-4f;
	y = y - z * 0.5f;
	tmp = e * 0.693359375f;
	x = x + y;
	x = x + tmp;
	return invalid ? bitsf(0xFFFFFFFFu) : x;                                // x | invalid_mask
}

DEV float mf_exp(float x)
{
	x = (x < 88.3762626647949f) ? x : 88.3762626647949f;                   // _mm_min_ps
	x = (x > -88.3762626647949f) ? x : -88.3762626647949f;                 // _mm_max_ps
	float fx = x * 1.44269504088896341f + 0.5f;
	float tmp = (float)cvtt(fx);
	fx = tmp - ((tmp > fx) ? 1.0f : 0.0f);
	tmp = fx * 0.693359375f;
	float z = fx * -2.12194440e-4f;
	x = x - tmp;
	x = x - z;
	z = x * x;
	float y = 1.9875691500E-4f;
	y = y * x + 1.3981999507E-3f;
	y = y * x + 8.3334519073E-3f;
	y = y * x + 4.1665795894E-2f;
	y = y * x + 1.6666665459E-1f;
	y = y * x + 5.0000001201E-1f;
	y = y * z + x;
	y = y + 1.0f;
	const uint32_t n = (uint32_t)(cvtt(fx) + 0x7f) << 23;
	return y * bitsf(n);
}

DEV float mf_reduce(float ax, int32_t& j)
{
	float y = ax * 1.27323954473516f;
	j = (cvtt(y) + 1) & ~1;
	y = (float)j;
	ax = ax + y * -0.78515625f;
	ax = ax + y * -2.4187564849853515625e-4f;
	ax = ax + y * -3.77489497744594108e-8f;
	return ax;
}
DEV void mf_polys(float x, float& ysin, float& ycos)
{
	float z = x * x;
	float y = 2.443315711809948E-005f;
	y = y * z + -1.388731625493765E-003f;
	y = y * z + 4.166664568298827E-002f;
	y = y * z;
	y = y * z;
	y = y - z * 0.5f;
	ycos = y + 1.0f;
	float y2 = -1.9515295891E-4f;
	y2 = y2 * z + 8.3321608736E-3f;
	y2 = y2 * z + -1.6666654611E-1f;
	y2 = y2 * z;
	y2 = y2 * x;
	ysin = y2 + x;
}
DEV float mf_sin(float x)
{
	uint32_t sign = fbits(x) & 0x80000000u;
	int32_t j;
	float r = mf_reduce(f_abs(x), j);
	sign ^= (uint32_t)(j & 4) << 29;
	float ys, yc;
	mf_polys(r, ys, yc);
	float y = ((j & 2) == 0) ? (0.0f + ys) : (yc + 0.0f);
	return bitsf(fbits(y) ^ sign);
}
DEV float mf_cos(float x)
{
	int32_t j;
	float r = mf_reduce(f_abs(x), j);
	j -= 2;
	uint32_t sign = (uint32_t)(~j & 4) << 29;
	float ys, yc;
	mf_polys(r, ys, yc);
	float y = ((j & 2) == 0) ? (0.0f + ys) : (yc + 0.0f);
	return bitsf(fbits(y) ^ sign);
}
DEV float f_log(float a) { const float q = mf_log(a); return (a > 0.0f) ? q : 0.0f; }                            // RT.cpp:1919
DEV float f_pow(float a, float b) { const float q = mf_exp(mf_log(a) * b); return (a > 1.e-8f) ? q : 0.0f; }     // RT.cpp:1936, SSE.h:1941-1948

DEV void f_sincos(float a, float& s_out, float& c_out)                                        // SSE.h:3118-3172
{
	const float PI = 3.1415926535897932f;
	float q = a * (1.0f / (2.0f * PI));
	q = (float)cvtr(q);
	float x = a - ((2.0f * PI) * q);
	const uint32_t sgn = fbits(x) & 0x80000000u;
	const float c = bitsf(fbits(PI) | sgn);
	const float rflx = c - x;
	const bool comp = f_abs(x) > (0.5f * PI);
	x = comp ? rflx : x;
	const float sign = comp ? -1.0f : 1.0f;
	const float x2 = x * x;
	float S = -2.3889859e-08f;
	S = f_mad(x2, S, 2.7525562e-06f);
	S = f_mad(x2, S, -0.00019840874f);
	S = f_mad(x2, S, 0.0083333310f);
	S = f_mad(x2, S, -0.16666667f);
	S = f_mad(x2, S, 1.0f);
	s_out = S * x;
	float C = -2.6051615e-07f;
	C = f_mad(x2, C, 2.4760495e-05f);
	C = f_mad(x2, C, -0.0013888378f);
	C = f_mad(x2, C, 0.041666638f);
	C = f_mad(x2, C, -0.5f);
	C = f_mad(x2, C, 1.0f);
	c_out = C * sign;
}

DEV int32_t i_div(int32_t n, int32_t d)                                                        // RT.cpp:958-973
{
	const bool zero = (d == 0), wrap = (d == -1 && n == (int32_t)0x80000000);
	const int32_t q = n / ((zero || wrap) ? 1 : d);
	return zero ? 0 : (wrap ? 0x7FFFFFFF : q);
}
DEV int32_t i_min(int32_t a, int32_t b) { return a < b ? a : b; }
DEV int32_t i_max(int32_t a, int32_t b) { return a > b ? a : b; }
DEV int32_t i_abs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }             // _mm_abs_epi32: INT_MIN stays
DEV uint32_t i_shl(uint32_t a, uint32_t b) { return a << (b & 31u); }                         // VectorVMPlatformGeneric.h:76-89
DEV uint32_t i_shr(uint32_t a, uint32_t b) { return a >> (b & 31u); }                         // VectorVMPlatformGeneric.h:61-74

// half <-> float, GenericPlatformMath.h:34-61 and :68-123 (round to nearest even)
DEV float load_half(uint32_t h)
{
	const uint32_t shifted_exp = 0x7c00u << 13;
	uint32_t u = (h & 0x7fffu) << 13;
	const uint32_t e = shifted_exp & u;
	u += (127u - 15u) << 23;
	const uint32_t inf_nan = u + ((128u - 16u) << 23);
	const uint32_t denorm = fbits(bitsf(u + (1u << 23)) - bitsf(113u << 23));
	u = (e == shifted_exp) ? inf_nan : ((e == 0) ? denorm : u);
	u |= (h & 0x8000u) << 16;
	return bitsf(u);
}
DEV uint32_t store_half(float x)
{
	uint32_t u = fbits(x);
	const uint32_t f32infty = 255u << 23, f16max = (127u + 16u) << 23, dm = ((127u - 15u) + (23u - 10u) + 1u) << 23;
	const uint32_t sign = u & 0x80000000u;
	u ^= sign;
	const uint32_t o_big = (u > f32infty) ? 0x7e00u : 0x7c00u;
	const uint32_t o_small = (fbits(bitsf(u) + bitsf(dm)) - dm) & 0xFFFFu;
	const uint32_t o_norm = ((u + ((uint32_t)(15 - 127) << 23) + 0xfffu + ((u >> 13) & 1u)) >> 13) & 0xFFFFu;
	const uint32_t o = (u >= f16max) ? o_big : ((u < (113u << 23)) ? o_small : o_norm);
	return o | (sign >> 16);
}

// counter-based generator keyed by (seed, instance, ordinal of the random op); value mapping RT.cpp:953-956
DEV uint32_t rand_u32(uint32_t seed_lo, uint32_t seed_hi, uint32_t instance, uint32_t ordinal)
{
	uint32_t h = instance * 0x9E3779B1u + seed_lo;
	h ^= ordinal * 0x85EBCA77u + seed_hi;
	h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
	h += ordinal * 0x27D4EB2Fu + instance;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
	return h;
}
DEV float rand01(uint32_t r) { return bitsf((r >> 9) | 0x3F800000u) - 1.0f; }

// ---- noise built-ins ---------------------------------------------------------------------------------
// Gradient: one of the sixteen sums (+-x or +-y) + (+-y or +-z) of UnrealMath.cpp:3577-3602, e.g. case 3 "-x + y", case 9 "y - z".
// Per lane the case differs, so instead of a switch (sixteen masked regions per call) the two terms are selected: bit h of
// each constant says whether the first term is y (else x), whether it is negated, whether the second is y (else z), whether
// it is negated.  a - b and -a + b are the IEEE sums a + (-b) and (-a) + b, so the result is bit-identical to the switch.
// What the hash chooses is tabulated per workgroup in LDS: row h & 15 holds {first is y (all ones) else x, sign bit if the first
// term is negated, second is y else z, sign bit if the second is negated}, built from the four bit sets below by perlin_rows;
// a second copy of the permutation table holds the rows' byte offsets, so a corner costs two LDS reads, two selects, two sign
// flips and the sum.  (Extracting the four choices from the bit sets per corner took four more instructions, a quarter of
// an evaluation.)
#define BITOP3(a, b, c, table) ((uint32_t)__builtin_amdgcn_bitop3_b32((int)(a), (int)(b), (int)(c), (table)))
typedef const __attribute__((address_space(3))) uint8_t* lds_u8p;
typedef uint32_t lds_row __attribute__((ext_vector_type(4)));
// The noise tables are addressed by their absolute LDS offsets (a lookup is then one read whose address is the index itself,
// the table's offset riding in the instruction).  That holds because these kernels declare no static LDS, so the dynamic
// region starts at LDS address 0; the kernel checks it where it fills the tables and reports a launch error otherwise.
#define PERLIN_LDS ((lds_u8p)(unsigned int)LDS_PERLIN_OFF)
DEV lds_row perlin_row(uint32_t h)
{
	lds_row r;
	r.x = (uint32_t)__builtin_amdgcn_sbfe((int)0xCA44u, h, 1u);
	r.y = ((0xEC78u >> h) & 1u) << 31;
	r.z = (uint32_t)__builtin_amdgcn_sbfe((int)0x30AAu, h, 1u);
	r.w = ((0x8FA0u >> h) & 1u) << 31;
	return r;
}
// any function of three bit operands is one instruction on this target: the table byte is the function applied to 0xF0, 0xCC, 0xAA
// (0xCA: a & b | ~a & c)
DEV float p_grad3(uint32_t corner, float x, float y, float z)
{
	const lds_u8p P = PERLIN_LDS;
	const uint32_t row = P[(LDS_PERLIN_SEL_OFF - LDS_PERLIN_OFF) + corner];
	const lds_row g = *(const __attribute__((address_space(3))) lds_row*)(P + (LDS_PERLIN_GRAD_OFF - LDS_PERLIN_OFF) + row);
	const uint32_t first = BITOP3(g.x, fbits(y), fbits(x), 0xCA) ^ g.y;
	const uint32_t second = BITOP3(g.z, fbits(y), fbits(z), 0xCA) ^ g.w;
	return bitsf(first) + bitsf(second);
}
DEV float p_smooth(float x) { return x * x * x * (x * (x * 6.0f - 15.0f) + 10.0f); }          // UnrealMath.cpp:3605-3608
DEV float p_lerp(float a, float b, float t) { return a + t * (b - a); }                       // UnrealMathUtility.h:1187-1190

// The permutation table is named as an LDS address: through a generic pointer every one of the fourteen lookups of an
// evaluation paid a 64-bit address, a null test and a select on top of the read (42 of 282 instructions).
__device__ __noinline__ float perlin3(float px, float py, float pz)                             // UnrealMath.cpp:3658-3699
{
	const lds_u8p P = PERLIN_LDS;
	const float xf = floorf(px), yf = floorf(py), zf = floorf(pz);
	// (int32)floor & 255 as the reference's cast comes out on x86 (:3665-3667): 0x80000000 for NaN and for what does not fit, so
	// the cell index of such a coordinate is 0.  The device's conversion saturates instead: it agrees except above the range, where
	// it yields 0x7FFFFFFF (no float below 2^31 converts to that value, the nearest is 2147483520)
	const int qx = (int)xf, qy = (int)yf, qz = (int)zf;
	const int xi = qx == 0x7FFFFFFF ? 0 : qx & 255, yi = qy == 0x7FFFFFFF ? 0 : qy & 255, zi = qz == 0x7FFFFFFF ? 0 : qz & 255;
	const float x = px - xf, y = py - yf, z = pz - zf;
	const float xm = x - 1.0f, ym = y - 1.0f, zm = z - 1.0f;
	const int a = P[xi] + yi, aa = P[a] + zi, ab = P[a + 1] + zi;
	const int b = P[xi + 1] + yi, ba = P[b] + zi, bb = P[b + 1] + zi;
	const float u = p_smooth(x), v = p_smooth(y), w = p_smooth(z);
	const float r = 0.97f * p_lerp(
		p_lerp(p_lerp(p_grad3(aa, x, y, z), p_grad3(ba, xm, y, z), u),
		       p_lerp(p_grad3(ab, x, ym, z), p_grad3(bb, xm, ym, z), u), v),
		p_lerp(p_lerp(p_grad3(aa + 1, x, y, zm), p_grad3(ba + 1, xm, y, zm), u),
		       p_lerp(p_grad3(ab + 1, x, ym, zm), p_grad3(bb + 1, xm, ym, zm), u), v), w);
	return r < -1.0f ? -1.0f : (r < 1.0f ? r : 1.0f);
}

// curl of three Perlin potentials at fixed offsets, central differences with step 1/64
__device__ __noinline__ void curl_noise3(float x, float y, float z, float& ox, float& oy, float& oz)
{
	const float e = 0.015625f, inv2e = 32.0f;
	const float o1x = 31.416f, o1y = -47.853f, o1z = 12.793f;
	const float o2x = -233.145f, o2y = -113.408f, o2z = -185.31f;
	const float p2_dy = perlin3(x + o2x, (y + e) + o2y, z + o2z) - perlin3(x + o2x, (y - e) + o2y, z + o2z);
	const float p1_dz = perlin3(x + o1x, y + o1y, (z + e) + o1z) - perlin3(x + o1x, y + o1y, (z - e) + o1z);
	const float p0_dz = perlin3(x, y, z + e) - perlin3(x, y, z - e);
	const float p2_dx = perlin3((x + e) + o2x, y + o2y, z + o2z) - perlin3((x - e) + o2x, y + o2y, z + o2z);
	const float p1_dx = perlin3((x + e) + o1x, y + o1y, z + o1z) - perlin3((x - e) + o1x, y + o1y, z + o1z);
	const float p0_dy = perlin3(x, y + e, z) - perlin3(x, y - e, z);
	ox = (p2_dy - p1_dz) * inv2e;
	oy = (p0_dz - p2_dx) * inv2e;
	oz = (p1_dx - p0_dy) * inv2e;
}

// one component of curl_noise3 (0 = x, 1 = y, 2 = z): four Perlin evaluations, same arithmetic as the full form
__device__ __noinline__ float curl_noise3_component(float x, float y, float z, uint32_t which)
{
	const float e = 0.015625f, inv2e = 32.0f;
	const float o1x = 31.416f, o1y = -47.853f, o1z = 12.793f;
	const float o2x = -233.145f, o2y = -113.408f, o2z = -185.31f;
	if (which == 0) {
		const float p2_dy = perlin3(x + o2x, (y + e) + o2y, z + o2z) - perlin3(x + o2x, (y - e) + o2y, z + o2z);
		const float p1_dz = perlin3(x + o1x, y + o1y, (z + e) + o1z) - perlin3(x + o1x, y + o1y, (z - e) + o1z);
		return (p2_dy - p1_dz) * inv2e;
	} else if (which == 1) {
		const float p0_dz = perlin3(x, y, z + e) - perlin3(x, y, z - e);
		const float p2_dx = perlin3((x + e) + o2x, y + o2y, z + o2z) - perlin3((x - e) + o2x, y + o2y, z + o2z);
		return (p0_dz - p2_dx) * inv2e;
	}
	const float p1_dx = perlin3((x + e) + o1x, y + o1y, z + o1z) - perlin3((x - e) + o1x, y + o1y, z + o1z);
	const float p0_dy = perlin3(x, y + e, z) - perlin3(x, y - e, z);
	return (p1_dx - p0_dy) * inv2e;
}

// ---- wave helpers --------------------------------------------------------------------------------------
// lane LANE of v := x (wave-uniform); the other lanes keep their value.  The lane number is an immediate: a VALU
// instruction of this target reads at most one scalar register.
template <int LANE> DEV uint32_t write_lane(uint32_t v, uint32_t x)
{
	asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(x), "n"(LANE));
	return v;
}
template <int G> DEV uint32_t fill_uniforms(uint32_t uni, const void* consts, uint32_t count)
{
	if constexpr (G < (int)VVM_MAX_UNIFORMS / 4) {
		if ((uint32_t)(4 * G) < count) {
			uni = write_lane<4 * G + 0>(uni, sload32(consts, 4 * G + 0));
			uni = write_lane<4 * G + 1>(uni, sload32(consts, 4 * G + 1));
			uni = write_lane<4 * G + 2>(uni, sload32(consts, 4 * G + 2));
			uni = write_lane<4 * G + 3>(uni, sload32(consts, 4 * G + 3));
			uni = fill_uniforms<G + 1>(uni, consts, count);
		}
	}
	return uni;
}
DEV uint32_t lanes_below(u64 m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

// An equality test the compiler cannot merge with its neighbours: a chain of such tests stays a chain, in the order written
// (plain comparisons of one value are turned into a switch and lowered as a balanced tree).
DEV bool nin_is(uint32_t nin, uint32_t n) { asm volatile("" : "+s"(nin)); return nin == n; }

// ---- per-tile interpreter state ----------------------------------------------------------------------
#define RVEC typename RegTraits<V, SPILL>::vec
// (eight vectors are handed around whatever V is; the ones past V are never touched and cost nothing)
#define CREGS RVEC R0, RVEC R1, RVEC R2, RVEC R3, RVEC R4, RVEC R5, RVEC R6, RVEC R7
#define RARGS R0, R1, R2, R3, R4, R5, R6, R7

// (Fields of the Exec descriptor that an instruction needs rarely are read from the descriptor where they are used, with a
// scalar load: a value kept in a scalar register for the whole tile costs more than the occasional load.)
#define DESC_U32(dp, field) sload32((dp), offsetof(VvmExecDesc, field) / 4)
#define DESC_PTR(dp, field) sload64((dp), offsetof(VvmExecDesc, field) / 8)
// the tables that travel with the descriptor (consts, in_ptrs, load_ptrs, out_list): their fields hold offsets from the
// descriptor when the block came as a kernel argument, whose address the host does not know (VVM_DESC_RELATIVE)
#define DESC_TABLE(c, field) (DESC_PTR((c).dp, field) + (c).rel)
struct Ctx {
	const void*     dp;           // the Exec's VvmExecDesc
	uint64_t        rel;          // what the descriptor's table fields are relative to (0: they are addresses)
	const uint32_t* consts;
	const uint64_t* out_list;
	float*          spill;        // LDS, [slot][thread] (V floats per thread and slot)
	uint32_t tid;
	uint32_t wave;                // tid / 64 as a scalar
	uint32_t inst0;               // index (within the Exec) of the lane's first instance
	uint32_t num_instances;
	uint32_t tile_first;          // index of the tile's first instance
	uint32_t valid;               // instances of the tile that exist (the Exec's last tile may be partial)
	uint32_t phase;               // tiles this workgroup has started; its parity selects the LDS word of double-buffered exchanges
	uint32_t reg_shift;           // four-wide kernel with the hand-written routines: log2 of the VGPRs per register of the Exec's program (VVM_DESC_REGS_OF_4)
};

// The 32 random bits of draw `ordinal` for instance `inst` of the Exec: from the words laid out by vvm_xorwow_fill_kernel when the
// Exec asks for the reference's generator (FULL kernels only: the host sends such an Exec there), else from the counter-based one.
template <bool FULL> DEV uint32_t draw_bits(const Ctx& c, uint32_t inst, uint32_t ordinal)
{
	if constexpr (FULL) {
		const uint64_t words = DESC_PTR(c.dp, rand_words);
		if (words != 0) {
			const uint64_t at = (uint64_t)ordinal * DESC_U32(c.dp, rand_stride) + inst;      // (inst < rand_stride: the stride covers whole tiles)
			return *(const uint32_t*)(words + at * 4u);
		}
	}
	return rand_u32(DESC_U32(c.dp, rand_seed_lo), DESC_U32(c.dp, rand_seed_hi), DESC_U32(c.dp, rand_instance_base) + inst, ordinal);
}

// Register access.  The index is wave-uniform.  The VGPR vectors are read and written unconditionally (a branch around an
// element insert makes the compiler keep two copies of the whole vector); a write to a register that lives in LDS puts
// the old value back into the element it aliases.
template <int V, bool SPILL> DEV void reg_read(CREGS, const Ctx& c, uint32_t idx, float (&out)[V])
{
	constexpr uint32_t RV = RegTraits<V, SPILL>::RV;
	const uint32_t i = idx & (IS_LIGHT(V, SPILL) ? 15u : RV - 1);      // (light: eight registers and the two riding ones)
	out[0] = vget(R0, i);
	if constexpr (V >= 4) { out[1] = vget(R1, i); out[2] = vget(R2, i); out[3] = vget(R3, i); }
	if constexpr (V == 8) { out[4] = vget(R4, i); out[5] = vget(R5, i); out[6] = vget(R6, i); out[7] = vget(R7, i); }
	if constexpr (SPILL && V != 8) {      // (the eight-instance kernels have no LDS register slots)
		if (idx >= RV) {
			const float* p = c.spill + ((idx - RV) * VVM_BLOCK_THREADS + c.tid) * V;
			if constexpr (V == 4) { const f4a v = *(const f4a*)p; out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
			else out[0] = p[0];
		}
	}
}
template <int V, bool SPILL> DEV RVEC reg_put(RVEC Rk, bool enable, uint32_t idx, float x)
{
	constexpr uint32_t RV = RegTraits<V, SPILL>::RV;
	const uint32_t i = idx & (IS_LIGHT(V, SPILL) ? 15u : RV - 1);
	if constexpr (SPILL && V != 8) enable = enable && (idx < RV);
	return vset(Rk, i, enable ? x : vget(Rk, i));
}
template <int V, bool SPILL> DEV void reg_spill(const Ctx& c, bool enable, uint32_t idx, const float (&in)[V])
{
	constexpr uint32_t RV = RegTraits<V, SPILL>::RV;
	if constexpr (SPILL && V != 8) {
		if (enable && idx >= RV) {
			float* p = c.spill + ((idx - RV) * VVM_BLOCK_THREADS + c.tid) * V;
			if constexpr (V == 4) { f4a v; v.x = in[0]; v.y = in[1]; v.z = in[2]; v.w = in[3]; *(f4a*)p = v; }
			else p[0] = in[0];
		}
	}
}
// statement: if `enable` (wave-uniform), write the V values of `arr` to virtual register `idx`.  The element insert itself is
// unconditional (a disabled write re-inserts the old value) so that no control flow ever merges two versions of a vector.
#define REG_WRITE_IF(enable, idx, arr) do { \
		const bool en_ = (enable); const uint32_t idx_ = (idx); \
		R0 = reg_put<V, SPILL>(R0, en_, idx_, (arr)[0]); \
		if constexpr (V >= 4) { R1 = reg_put<V, SPILL>(R1, en_, idx_, (arr)[1]); R2 = reg_put<V, SPILL>(R2, en_, idx_, (arr)[2]); R3 = reg_put<V, SPILL>(R3, en_, idx_, (arr)[3]); } \
		if constexpr (V == 8) { R4 = reg_put<V, SPILL>(R4, en_, idx_, (arr)[4]); R5 = reg_put<V, SPILL>(R5, en_, idx_, (arr)[5]); R6 = reg_put<V, SPILL>(R6, en_, idx_, (arr)[6]); R7 = reg_put<V, SPILL>(R7, en_, idx_, (arr)[7]); } \
		reg_spill<V, SPILL>(c, en_, idx_, (arr)); \
	} while (0)
#define REG_WRITE(idx, arr) REG_WRITE_IF(true, idx, arr)

// ---- memory access ------------------------------------------------------------------------------------
// No branch in this kernel depends on the lane: every `if` and every loop condition is wave-uniform, and the kernel is compiled
// so that such branches stay plain scalar branches.  One per-lane branch would make the compiler rewrite the control flow of
// everything around it into flag-guarded if-chains, interpreter loop included (about three times the branches per interpreted
// instruction).  Whatever has to happen for some lanes only goes through the bounds check of a buffer descriptor instead:
// a lane that must not take part addresses past the end of the descriptor's range, where the hardware drops stores and
// atomics and returns zero for loads (multi-dword accesses are checked per dword).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define VVM_RSRC_FLAGS 0x00020000      // raw buffer, 32-bit data format (descriptor dword 3 for this target family)
#define AUX_STREAM 2                   // nt: streamed once, keep it out of the way of the caches
#ifndef AUX_STREAM_LOAD
#define AUX_STREAM_LOAD AUX_STREAM
#endif
#ifndef AUX_STREAM_STORE
#define AUX_STREAM_STORE AUX_STREAM
#endif
#define AUX_AGENT  16                  // sc1: device-scope visibility, what a relaxed agent-scope atomic access uses here
DEV rsrc_t make_rsrc(uint64_t base, uint32_t num_bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)num_bytes, VVM_RSRC_FLAGS); }

// The tile's part of an input stream: the lane reads the elements of its instance slots; elements past the Exec's end read 0.
template <int V> DEV void load_stream32(const Ctx& c, uint64_t base, float (&out)[V])
{
	const rsrc_t rs = make_rsrc(base + 4ull * c.tile_first, 4u * c.valid);
	if constexpr (V >= 4) {
		const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(c.tid * 16u), 0, AUX_STREAM_LOAD);
		out[0] = bitsf(v.x); out[1] = bitsf(v.y); out[2] = bitsf(v.z); out[3] = bitsf(v.w);
		if constexpr (V == 8) {
			const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(VVM_HALF_TILE * 4u + c.tid * 16u), 0, AUX_STREAM_LOAD);
			out[4] = bitsf(u.x); out[5] = bitsf(u.y); out[6] = bitsf(u.z); out[7] = bitsf(u.w);
		}
	} else {
		out[0] = bitsf(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(c.tid * 4u), 0, AUX_STREAM_LOAD));
	}
}
template <int V> DEV void load_stream16(const Ctx& c, uint64_t base, float (&out)[V])    // raw half bits, zero-extended
{
	const rsrc_t rs = make_rsrc(base + 2ull * c.tile_first, 2u * c.valid);
#pragma unroll
	for (int k = 0; k < V; ++k)
		out[k] = bitsf((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)((c.tid * lane_stride<V>() + slot_offset<V>(k)) * 2u), 0, AUX_STREAM_LOAD));
}
// 64-bit device-scope store by the first lane of the calling wave
DEV void store_u64_first_lane(gu64* addr, u64 value, uint32_t lane)
{
	const rsrc_t rs = make_rsrc((uint64_t)addr, 8u);
	u32x2 v; v.x = (uint32_t)value; v.y = (uint32_t)(value >> 32);
	__builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)(lane == 0 ? 0u : 8u), 0, AUX_AGENT);
}
// device-scope fetch-and-add by the first lane of the calling wave; returns the old value as a scalar
DEV uint32_t fetch_add_first_lane(gu32* addr, uint32_t value, uint32_t lane)
{
	const rsrc_t rs = make_rsrc((uint64_t)addr, 4u);
	const int old = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)value, rs, (int)(lane == 0 ? 0u : 4u), 0, AUX_AGENT);
	return SCALAR(old);
}

// Decoupled look-back over the Exec's tile words; called by wave 0 only.  Returns the exclusive prefix of `tile`.
// Each round reads the VVM_LB_WINDOW tiles before `first` (VVM_LB_WINDOW / 64 words per lane, all loads in flight together),
// waits until every one of them carries this launch's tag, and sums the counts back to the nearest tile whose word already
// holds an inclusive prefix.
#define VVM_LB_WINDOW 256
#define VVM_LB_WORDS_PER_LANE (VVM_LB_WINDOW / 64)
// The loads of one poll: lane l reads the words of tiles first - l - 64 j.
DEV void look_back_poll(gu64* status, int32_t first, uint32_t lane, u64 (&w)[VVM_LB_WORDS_PER_LANE])
{
#pragma unroll
	for (int j = 0; j < VVM_LB_WORDS_PER_LANE; ++j)
		w[j] = __hip_atomic_load(status + max(first - (int32_t)lane - 64 * j, 0), RLX_AGENT);
}
// `pre` holds a first poll that the caller issued earlier (have_pre), so that its round trip overlaps other work.
DEV uint32_t look_back(gu64* status, uint32_t tile, uint32_t tag, uint32_t lane, VvmLaunchState* ls, uint32_t& rounds, uint32_t& polls,
                       const u64 (&pre)[VVM_LB_WORDS_PER_LANE], bool have_pre)
{
	constexpr int WPL = VVM_LB_WORDS_PER_LANE;
	uint32_t sum = 0;
	int32_t first = (int32_t)tile - 1;
	for (;;) {
		u64 w[WPL];
		uint32_t spins = 0;
		++rounds;
		for (;;) {
			bool ok = true;
			if (have_pre) {
#pragma unroll
				for (int j = 0; j < WPL; ++j) w[j] = pre[j];
				have_pre = false;
			} else {
				++polls;
				look_back_poll(status, first, lane, w);
			}
#pragma unroll
			for (int j = 0; j < WPL; ++j) {
				const int32_t mine = first - (int32_t)lane - 64 * j;
				w[j] = (mine >= 0) ? w[j] : (((u64)tag << 34) | (VVM_LB_INCLUSIVE << 32));     // before the first tile: prefix 0
				ok = ok && ((uint32_t)(w[j] >> 34) == tag);
			}
			if (__all(ok)) break;
			if (++spins > (1u << 22)) {     // a predecessor never published: give up rather than hang the device
				__hip_atomic_store((gu32*)&ls->pad, 1u, RLX_AGENT);
				return sum;
			}
			__builtin_amdgcn_s_sleep(1);
		}
		uint32_t acc = 0;
		bool found = false;
#pragma unroll
		for (int j = 0; j < WPL; ++j) {
			const u64 inc_mask = __ballot(((w[j] >> 32) & 3ull) == VVM_LB_INCLUSIVE);
			const uint32_t nearest = inc_mask ? (uint32_t)__builtin_ctzll(inc_mask) : 63u;
			acc += (!found && lane <= nearest) ? (uint32_t)w[j] : 0u;
			found = found || (inc_mask != 0);
		}
		sum += wave_sum(acc);
		if (found) return sum;
		first -= VVM_LB_WINDOW;
	}
}

// LDS control words
#define LC_WAVE_TOTALS 0      // [2][2][4]  survivor count of each wave (per half of the tile), by parity of the tile
#define LC_PREFIX      16     // [2]     resolved prefix, by parity of the resolve
#define LC_TICKET      18     // [2]     claimed ticket, by parity of the tile
#define LC_STATS       24     // [8]     profiling sums
#define LC_WAVE_TOTALS1 32    // [2][2][4]  the same two for the second chain
#define LC_PREFIX1     48     // [2]

// acquireindex, first half: rank the tile's survivors and publish the tile's count.
// keep_bits: bit k = instance k of the lane survives.  Returns (rank of the lane's first survivor within the tile, survivors
// in the tile).  Survivors keep instance order (VectorVMRuntime.cpp:786-804, 824-878): a lane's instances are consecutive,
// lanes are in instance order, so the rank is the number of surviving instances of all lower lanes and waves.
// With eight instances per lane the tile's two halves are ranked one after the other (all of the first half's survivors come
// first): HALVES = 2, bits 0..3 / 4..7 of keep_bits, and .y of the result is the rank of the lane's first survivor of the
// second half.
template <int HALVES> DEV uint3 acquire_rank_and_publish(uint32_t keep_bits, const Ctx& c, uint32_t* lds_ctrl, gu64* status, uint32_t tile_local, uint32_t tag, uint32_t dbg,
                                                         uint32_t chain = 0)
{
	const uint32_t lane = c.tid & 63u;
	uint32_t below[2] = {0, 0}, wave_total[2] = {0, 0};
#pragma unroll
	for (int h = 0; h < HALVES; ++h)
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const u64 b = __ballot((keep_bits >> (4 * h + k)) & 1u);
			below[h] += lanes_below(b);
			wave_total[h] += (uint32_t)__popcll(b);
		}
	uint32_t* totals = lds_ctrl + (chain ? LC_WAVE_TOTALS1 : LC_WAVE_TOTALS) + 8u * (c.phase & 1u);
	totals[c.wave] = wave_total[0];            // every lane writes the wave's value
	if constexpr (HALVES == 2) totals[4u + c.wave] = wave_total[1];
	__syncthreads();                        // (the words alternate between tiles, so one barrier is enough)
	uint32_t wave_off[2] = {0, 0}, total[2] = {0, 0};
#pragma unroll
	for (int h = 0; h < HALVES; ++h)
#pragma unroll
		for (uint32_t w = 0; w < VVM_BLOCK_THREADS / 64; ++w) {
			const uint32_t x = totals[4u * h + w];
			wave_off[h] += (w < c.wave) ? x : 0u;
			total[h] += x;
		}
	const uint32_t total0 = SCALAR(total[0]);
	const uint32_t all = (HALVES == 2) ? total0 + SCALAR(total[1]) : total0;
	if (c.wave == 0 && !(tile_local == 0 && (dbg & 4u))) {
		// test hook (debug instantiations only): odd tiles publish their count about 50 us late, so that the tiles behind them
		// poll words that this launch has not written yet and a word left by an earlier launch would be taken if its tag matched
		if ((dbg & 1024u) && (tile_local & 1u))
			for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127);
		// the first tile's count is its inclusive prefix already
		const u64 flag = (tile_local == 0) ? VVM_LB_INCLUSIVE : VVM_LB_AGGREGATE;
		store_u64_first_lane(status + tile_local, ((u64)tag << 34) | (flag << 32) | all, lane);
	}
	return make_uint3(wave_off[0] + below[0], total0 + wave_off[1] + below[1], all);
}

// What is left to do for a tile whose program has run: its outputs sit compacted in the LDS staging slots, and its prefix
// may still be unresolved.  All fields are wave-uniform.
struct Pending {
	bool     active;
	bool     unresolved;   // the tile's count is published, its exclusive prefix not yet known
	uint32_t total;        // survivors in the tile
	uint32_t base;         // exclusive prefix (valid once resolved)
	bool     last;         // the Exec's last tile: it writes the dataset's survivor count
	uint32_t stage_end;    // output-list index one past the last staged slot (0: nothing staged); the slots in use are the
	                       // entries of its group of VVM_STAGE_SLOTS up to there
	uint32_t tile_local;
	gu64*    status;
	const uint64_t* out_list;      // the Exec's output list; the entry before it is the address of the survivor count
	// the second chain of a program that acquires two datasets (kernels with HAS_TWO_CHAINS): its look-back words follow the
	// first chain's, `tiles` words on
	bool     unresolved1;
	uint32_t total1, base1, tiles;
	uint32_t ride;                 // per lane, eight instances per lane: the lane's acquireindex result (for the light kernel's outputs that wait in
	                               // registers, and for vvm_through_kernel): keep bits [0:8), rank of the lane's first survivor [8:20), of its first
	                               // one in the second half [20:32)
};
#define STAGED_COUNT(end) ((end) ? (((end) - 1u) & (VVM_STAGE_SLOTS - 1u)) + 1u : 0u)
#define STAGED_FIRST(end) ((end) ? ((end) - 1u) & ~(VVM_STAGE_SLOTS - 1u) : 0u)

// Finishing a tile takes two steps, both run by the NEXT tile while its input loads are in flight (also when the staging
// slots run out, and after the last tile):
//   resolve_tile  the look-back for the tile's exclusive prefix (acquireindex, second half).  Its polls queue behind the
//                 input loads, so the two latencies overlap instead of adding up.  The Exec's last tile also writes the
//                 dataset's survivor count.  Contains one barrier, after which every thread's staged values are visible.
//   store_tile    survivor i of the tile (the survivors were compacted when they were staged) goes to element base + i of
//                 each staged output.  A thread moves V consecutive values, so every tile stores 128 bits per lane whatever
//                 its death rate.  Issued after the next tile's loads have arrived: stores and loads share one completion
//                 counter, and a wait for loads would otherwise also wait for these stores.
template <int V, bool TWO = false> DEV void resolve_tile(Pending& p, const Ctx& c, uint32_t* lds_ctrl, uint32_t& resolves, uint32_t tag, VvmLaunchState* ls, uint32_t dbg,
                                     const u64 (&pre)[VVM_LB_WORDS_PER_LANE], bool have_pre)
{
	const bool need0 = p.unresolved && p.tile_local != 0;
	const bool need1 = TWO && p.unresolved1 && p.tile_local != 0;
	if (need0 || need1) {
		uint32_t* slot = lds_ctrl + LC_PREFIX + (resolves & 1u);
		uint32_t* slot1 = lds_ctrl + LC_PREFIX1 + (resolves & 1u);
		++resolves;
		if (c.wave == 0 && need0) {
			const uint32_t lane = c.tid & 63u;
			uint32_t rounds = 0, polls = 0;
			const u64 t0 = (dbg & 128u) ? __builtin_amdgcn_s_memtime() : 0;
			const uint32_t ex = (dbg & 32u) ? p.tile_local * (VVM_BLOCK_THREADS * V) : look_back(p.status, p.tile_local, tag, lane, ls, rounds, polls, pre, have_pre);
			*slot = ex;
			store_u64_first_lane(p.status + p.tile_local, ((u64)tag << 34) | (VVM_LB_INCLUSIVE << 32) | (ex + p.total), lane);
			if (dbg & 128u) {      // per-workgroup sums (all lanes of wave 0 write the same values)
				uint32_t* st = lds_ctrl + LC_STATS;
				const uint32_t s0 = st[0] + 1u, s1 = st[1] + rounds, s2 = st[2] + polls, s3 = st[3] + (uint32_t)((__builtin_amdgcn_s_memtime() - t0) >> 4);
				st[0] = s0; st[1] = s1; st[2] = s2; st[3] = s3;
			}
		}
		if constexpr (TWO) {
			if (c.wave == 1 && need1) {      // the second chain's look-back runs beside the first one's, on the next wave
				const uint32_t lane = c.tid & 63u;
				uint32_t rounds = 0, polls = 0;
				gu64* status1 = p.status + p.tiles;
				const u64 none[VVM_LB_WORDS_PER_LANE] = {};
				const uint32_t ex = look_back(status1, p.tile_local, tag, lane, ls, rounds, polls, none, false);
				*slot1 = ex;
				store_u64_first_lane(status1 + p.tile_local, ((u64)tag << 34) | (VVM_LB_INCLUSIVE << 32) | (ex + p.total1), lane);
			}
		}
		__syncthreads();
		if (need0) p.base = SCALAR(*slot);
		if (need1) p.base1 = SCALAR(*slot1);
	} else {
		__syncthreads();
	}
	if (p.unresolved && p.tile_local == 0) p.base = 0;
	if (TWO && p.unresolved1 && p.tile_local == 0) p.base1 = 0;
	if (p.unresolved && p.last && c.wave == 0 && !(dbg & 16u)) *(int32_t*)sload64(p.out_list - 1, 0) = (int32_t)(p.base + p.total);
	if (TWO && p.unresolved1 && p.last && c.wave == 0) *(int32_t*)sload64(p.out_list - 2, 0) = (int32_t)(p.base1 + p.total1);
	p.unresolved = false;
	p.unresolved1 = false;
}
// One staging slot to its destination: entry `e` of the output list, 32-bit values at src[0..V).  off4 = 4 * base and
// rec4 = 4 * total come from the caller (the same for every slot).  Elements at and past `total` are dropped by the bounds
// check; so is everything when the entry has no destination.
#define VVM_RSRC_NOWHERE 0xFFFFFFF0u      // a byte offset past the end of every descriptor used here: the access is dropped

// update_id for one tile (VectorVMRuntime.cpp:1175-1380), run when the tile's output position `base` is known.
//   q[0..V)  the lane's ID-register values; kw = the lane's acquireindex result: rank[0:16) keep[16:24) exists[24:32)
//   survivor:  IDTable[id] = InstanceOffset + its output index            (:1191-1193 and the per-mask cases)
//   dead:      FreeIDTable[NumFreeIDs at start + number of dead instances before it] = id
// The reference appends each chunk's freed IDs in instance order and the chunks in order (:1175, 1181), which is the order
// of the dead instances of the whole Exec.  For the later MaxUsedID pass: the ID of every instance, and the number of
// survivors before each instance that starts one of the reference's chunks.
template <int V> DEV void apply_update_id(const void* idp, const float* q, uint32_t kw, uint32_t first_instance, uint32_t base)
{
#define IDF_U32(field) sload32(idp, offsetof(VvmIdDesc, field) / 4)
#define IDF_PTR(field) sload64(idp, offsetof(VvmIdDesc, field) / 8)
	const uint32_t id_table_num = IDF_U32(id_table_num);
	const rsrc_t rs_ids = make_rsrc(IDF_PTR(id_table), 4u * id_table_num);
	const rsrc_t rs_free = make_rsrc(IDF_PTR(free_table), 4u * IDF_U32(free_table_max));
	const rsrc_t rs_prefix = make_rsrc(IDF_PTR(chunk_prefix), 4u * IDF_U32(num_chunks));
	const rsrc_t rs_scratch = make_rsrc(IDF_PTR(id_scratch), 4u * IDF_U32(num_instances));
	const uint32_t instance_offset = IDF_U32(instance_offset), num_free_start = IDF_U32(push_start), chunk = IDF_U32(chunk_instances);
	const uint32_t rank = kw & 0xFFFFu, keep = (kw >> 16) & 0xFFu, exists = kw >> 24;
#pragma unroll
	for (int k = 0; k < V; ++k) {
		const uint32_t id = fbits(q[k]);
		const uint32_t g = first_instance + (uint32_t)k;
		// (bitwise combinations of the per-lane conditions: a short-circuit would become a per-lane branch)
		const uint32_t ok = (exists >> k) & 1u, alive = (keep >> k) & 1u;
		const uint32_t before = base + rank + (uint32_t)__popc(keep & ((1u << k) - 1u));     // survivors before this instance
		__builtin_amdgcn_raw_buffer_store_b32(instance_offset + before, rs_ids, (int)((ok & alive & (uint32_t)(id < id_table_num)) ? id * 4u : VVM_RSRC_NOWHERE), 0, 0);
		__builtin_amdgcn_raw_buffer_store_b32(id, rs_free, (int)((ok & (alive ^ 1u)) ? (num_free_start + g - before) * 4u : VVM_RSRC_NOWHERE), 0, 0);
		uint32_t ci = g / chunk;
		asm volatile("" : "+v"(ci));          // computed for every lane, then selected: otherwise the division ends up behind a per-lane branch
		__builtin_amdgcn_raw_buffer_store_b32(before, rs_prefix, (int)((ok & (uint32_t)(g == ci * chunk)) ? ci * 4u : VVM_RSRC_NOWHERE), 0, 0);
		__builtin_amdgcn_raw_buffer_store_b32(id, rs_scratch, (int)(g * 4u), 0, 0);          // (past the last instance: dropped)
	}
#undef IDF_U32
#undef IDF_PTR
}

struct StoreArgs {
	uint32_t i0;               // the thread's first element within the tile
	uint64_t off4;             // 4 * base
	uint32_t rec4;             // 4 * total
	uint32_t base, tile_first; // for update_id
	const uint32_t* keep_word; // the thread's acquireindex result in LDS
	uint64_t off4b;            // the same two for the entries of the second chain (VVM_OUT_CHAIN1)
	uint32_t rec4b;
};
// LAST: the group's last slot, the only one that can hold an update_id (only the FULL kernels run scripts with ID instructions)
template <int V, bool ANY16, bool LAST, bool TWO> DEV void store_slot(uint64_t e, const float* q, const StoreArgs& a)
{
	const uint32_t i0 = a.i0;
	const bool second = TWO && (e & VVM_OUT_CHAIN1);
	const uint64_t off4 = second ? a.off4b : a.off4;
	const uint32_t rec4 = second ? a.rec4b : a.rec4;
	const uint64_t ptr = e & VVM_OUT_ADDR_MASK;
	if constexpr (ANY16 && LAST && V != 8) {
		if (e & VVM_OUT_UPDATE_ID) {
			apply_update_id<V>((const void*)ptr, q, *a.keep_word, a.tile_first + i0, a.base);
			return;
		}
	}
	if (ANY16 && (e & VVM_OUT_IS16)) {       // 16-bit destination, the staged value holds the bits
		const rsrc_t rs = make_rsrc(ptr + (off4 >> 1), ptr ? (rec4 >> 1) : 0u);
#pragma unroll
		for (int k = 0; k < V; ++k)
			__builtin_amdgcn_raw_buffer_store_b16((uint16_t)fbits(q[slot_offset<V>(k)]), rs, (int)((i0 + slot_offset<V>(k)) * 2u), 0, AUX_STREAM_STORE);
		return;
	}
	const rsrc_t rs = make_rsrc(ptr + off4, ptr ? rec4 : 0u);
	if constexpr (V >= 4) {
		const f4a x = *(const f4a*)q;
		u32x4 y; y.x = fbits(x.x); y.y = fbits(x.y); y.z = fbits(x.z); y.w = fbits(x.w);
		__builtin_amdgcn_raw_buffer_store_b128(y, rs, (int)(i0 * 4u), 0, AUX_STREAM_STORE);
		if constexpr (V == 8) {
			const f4a x2 = *(const f4a*)(q + VVM_HALF_TILE);
			u32x4 y2; y2.x = fbits(x2.x); y2.y = fbits(x2.y); y2.z = fbits(x2.z); y2.w = fbits(x2.w);
			__builtin_amdgcn_raw_buffer_store_b128(y2, rs, (int)((VVM_HALF_TILE + i0) * 4u), 0, AUX_STREAM_STORE);
		}
	} else {
		__builtin_amdgcn_raw_buffer_store_b32(fbits(q[0]), rs, (int)(i0 * 4u), 0, AUX_STREAM_STORE);
	}
}
template <int V, bool ANY16, bool FULL, bool TWO> DEV void store_slots(const uint64_t (&entry)[VVM_STAGE_SLOTS], uint32_t n, const float* src, const StoreArgs& a)
{
	// the staged slots are 0 .. n-1: enter the chain at slot n-1 and fall through to slot 0 (no per-slot loop test)
#define VVM_SLOT(s) store_slot<V, ANY16, FULL && (s) == VVM_STAGE_SLOTS - 1, TWO>(entry[s], src + (s) * (VVM_BLOCK_THREADS * V), a)
	switch (n) {
	case 8: VVM_SLOT(7); [[fallthrough]];
	case 7: VVM_SLOT(6); [[fallthrough]];
	case 6: VVM_SLOT(5); [[fallthrough]];
	case 5: VVM_SLOT(4); [[fallthrough]];
	case 4: VVM_SLOT(3); [[fallthrough]];
	case 3: VVM_SLOT(2); [[fallthrough]];
	case 2: VVM_SLOT(1); [[fallthrough]];
	case 1: VVM_SLOT(0); [[fallthrough]];
	default: break;
	}
#undef VVM_SLOT
}
// ride (LIGHT): the values of the tile's two riding outputs, [output][instance slot] (registers VVM_LIGHT_RIDE_REG, +1 of the file)
// The light kernel keeps a lane's acquireindex word (Pending::ride's layout) here between the tile that made it and the next one, which
// stores that tile's outputs: a register held across the whole tile program is one more than the kernel has.  Written and read by
// the same thread.
DEV uint32_t* light_ride_word(unsigned char* lds, uint32_t tid) { return (uint32_t*)(lds + LDS_STAGE_KEEP_OFF(8, false, true)) + tid; }
template <int V, bool FULL, bool LIGHT, bool TWO> DEV void store_tile(Pending& p, const Ctx& c, unsigned char* lds, uint32_t dbg, const float (&ride)[16])
{
	static_assert(VVM_STAGE_SLOTS == 8, "store_slots is written for eight slots");
	if constexpr (LIGHT) {
		// for vvm_through_kernel, when the Exec has streams that stay out of its program: the lane's acquireindex result of this tile
		// (the light kernel keeps the word for its riding outputs anyway; the general eight-wide kernel stores it at the end of run_tile)
		const uint64_t ride_base = sload64(p.out_list - 3, 0);
		if (ride_base != 0) {
			const rsrc_t rr = make_rsrc(ride_base + (uint64_t)p.tile_local * (VVM_BLOCK_THREADS * 4u), VVM_BLOCK_THREADS * 4u);
			__builtin_amdgcn_raw_buffer_store_b32(*light_ride_word(lds, c.tid), rr, (int)(c.tid * 4u), 0, 0);
		}
	}
	const uint32_t n_staged = STAGED_COUNT(p.stage_end);
	if (n_staged != 0 && !(dbg & 8u)) {
		// slot s holds the values for entry first + s of the Exec's output list: all eight entries arrive with one scalar load
		uint64_t entry[VVM_STAGE_SLOTS];
		sload64x8(p.out_list, STAGED_FIRST(p.stage_end), entry);
		StoreArgs a;
		a.i0 = c.tid * lane_stride<V>(); a.off4 = 4ull * p.base; a.rec4 = 4u * p.total; a.base = p.base; a.tile_first = p.tile_local * (VVM_BLOCK_THREADS * V);
		a.off4b = TWO ? 4ull * p.base1 : 0; a.rec4b = TWO ? 4u * p.total1 : 0;
		a.keep_word = (const uint32_t*)(lds + LDS_STAGE_KEEP_OFF(V, FULL, LIGHT)) + c.tid;
		const float* src = (const float*)(lds + LDS_STAGE_OFF(FULL)) + a.i0;
		if (entry[0] & VVM_OUT_GROUP16) store_slots<V, true, FULL, TWO>(entry, n_staged, src, a);
		else store_slots<V, false, FULL, TWO>(entry, n_staged, src, a);
		if constexpr (LIGHT) {
			// Entries six and seven of a light program's list: the staging slots have no room for the values of a tile that keeps more than
			// three quarters of its instances, those are still in the registers the tile left them in (the program that runs now writes
			// those only after this point).  A full tile's values go out as they lie, 128 bits per half; a tile that lost some, compacted
			// within the lane; a tile that kept at most three quarters left them in the unused ends of the staging slots.
#pragma unroll
			for (uint32_t h = 0; h < 2; ++h) {
				const uint64_t ptr = entry[VVM_LIGHT_GROUP_SLOTS + h] & VVM_OUT_ADDR_MASK;
				if (ptr) {
					const rsrc_t rs = make_rsrc(ptr + a.off4, a.rec4);
					const uint32_t v[8] = {fbits(ride[8 * h + 0]), fbits(ride[8 * h + 1]), fbits(ride[8 * h + 2]), fbits(ride[8 * h + 3]),
					                       fbits(ride[8 * h + 4]), fbits(ride[8 * h + 5]), fbits(ride[8 * h + 6]), fbits(ride[8 * h + 7])};
					// (Addresses and offsets below are made from the thread index and the lane's acquireindex word HERE, from values the
					// compiler cannot trace back: made where it likes, it makes them once per kernel or per tile, far ahead, and keeps them in
					// scratch memory for want of registers, so that every store below would wait for a scratch load first.)
					// The thread index is counted again and the acquireindex word comes from LDS (light_ride_word) for the same reason: both would
					// otherwise be among the values that wait in scratch memory.
					uint32_t tid_here;
					asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n v_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid_here));
					tid_here += c.wave * 64u;
					const uint32_t ride_here = *light_ride_word(lds, tid_here);
					if (p.total <= VVM_LIGHT_STASH_MAX) {
						// run_tile has left the survivors' values compacted in the unused ends of the staging slots (see there): survivor
						// r is entry r % 512 of the end of the output's slot r / 512; 128 bits per lane and round, like a staged output
						const float* stash = (const float*)(lds + LDS_STAGE_OFF(FULL)) + (VVM_LIGHT_STASH_SLOTS * h) * (VVM_BLOCK_THREADS * V) + VVM_LIGHT_STASH_MAX;
						const f4a x0 = *(const f4a*)(stash + (tid_here >> 7) * (VVM_BLOCK_THREADS * V) + (tid_here & 127u) * 4u);
						const f4a x1 = *(const f4a*)(stash + 2u * (VVM_BLOCK_THREADS * V) + (tid_here & 127u) * 4u);
						u32x4 y0, y1;
						y0.x = fbits(x0.x); y0.y = fbits(x0.y); y0.z = fbits(x0.z); y0.w = fbits(x0.w);
						y1.x = fbits(x1.x); y1.y = fbits(x1.y); y1.z = fbits(x1.z); y1.w = fbits(x1.w);
						__builtin_amdgcn_raw_buffer_store_b128(y0, rs, (int)(tid_here * 16u), 0, AUX_STREAM_STORE);
						__builtin_amdgcn_raw_buffer_store_b128(y1, rs, (int)((1024u + tid_here * 4u) * 4u), 0, AUX_STREAM_STORE);      // (lanes 128 and up: past the descriptor's end)
					} else if (p.total == VVM_BLOCK_THREADS * V) {
						u32x4 lo, hi;
						lo.x = v[0]; lo.y = v[1]; lo.z = v[2]; lo.w = v[3];
						hi.x = v[4]; hi.y = v[5]; hi.z = v[6]; hi.w = v[7];
						__builtin_amdgcn_raw_buffer_store_b128(lo, rs, (int)(a.i0 * 4u), 0, AUX_STREAM_STORE);
						__builtin_amdgcn_raw_buffer_store_b128(hi, rs, (int)((VVM_HALF_TILE + a.i0) * 4u), 0, AUX_STREAM_STORE);
#ifdef VVM_EXPERIMENT_RIDE_WITHIN_LANE
					} else {
						// (the path this kernel had before the one below, kept to measure against)
						uint32_t off[6], w1v[2], wsv[2];
						u32x4 q4[2];
#pragma unroll
						for (int half = 0; half < 2; ++half) {
							const uint32_t m = (ride_here >> (4 * half)) & 15u;
							const uint32_t r4 = (half ? (ride_here >> 20) : ((ride_here >> 8) & 0xFFFu)) * 4u;
							const uint32_t a0 = v[4 * half], a1 = v[4 * half + 1], a2 = v[4 * half + 2], a3 = v[4 * half + 3];
							const bool k0 = (m & 1u) != 0, k1 = (m & 2u) != 0, k2 = (m & 4u) != 0;
							const uint32_t w0 = k0 ? a0 : k1 ? a1 : k2 ? a2 : a3;
							const uint32_t w1 = (m & 3u) == 3u ? a1 : (k2 & (k0 != k1)) ? a2 : a3;
							const uint32_t w2 = (m & 7u) == 7u ? a2 : a3;
							const uint32_t cnt = (uint32_t)__popc(m);
							q4[half].x = w0; q4[half].y = w1; q4[half].z = w2; q4[half].w = a3;
							w1v[half] = w1; wsv[half] = cnt == 3u ? w2 : w0;
							off[3 * half] = cnt == 4u ? r4 : VVM_RSRC_NOWHERE;
							off[3 * half + 1] = (cnt & 6u) == 2u ? r4 : VVM_RSRC_NOWHERE;
							off[3 * half + 2] = (cnt & 1u) ? r4 + (cnt & 2u) * 4u : VVM_RSRC_NOWHERE;
						}
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int half = 0; half < 2; ++half) {
							u32x2 q2; q2.x = q4[half].x; q2.y = w1v[half];
							__builtin_amdgcn_raw_buffer_store_b128(q4[half], rs, (int)off[3 * half], 0, AUX_STREAM_STORE);
							__builtin_amdgcn_raw_buffer_store_b64(q2, rs, (int)off[3 * half + 1], 0, AUX_STREAM_STORE);
							__builtin_amdgcn_raw_buffer_store_b32(wsv[half], rs, (int)off[3 * half + 2], 0, AUX_STREAM_STORE);
						}
						__builtin_amdgcn_sched_barrier(0);
					}
#else
					} else {
						// A tile that keeps more than three quarters of its instances, but not all: the staging slots had no room for these
						// values when the tile ran; they have now (every staged value has been read above), so the survivors' values go
						// through the output's own slot like a staged output's (two more barriers; all conditions up to here are the same for
						// every thread of the workgroup) and leave with 128 bits per lane and half.  (Before: compacted within the lane and
						// stored as one 128-, 64- and / or 32-bit store per half: six store instructions per lane and output, most of them
						// to nowhere.)
						float* const slot = (float*)(lds + LDS_STAGE_OFF(FULL)) + h * (VVM_BLOCK_THREADS * V);
						float* const sink = (float*)(lds + LDS_STAGE_SINK_OFF(V, FULL, true)) + tid_here;
						__syncthreads();
						uint32_t r = (ride_here >> 8) & 0xFFFu;
#pragma unroll
						for (int k = 0; k < 8; ++k) {
							if (k == 4) r = ride_here >> 20;
							const bool keep = (ride_here >> k) & 1u;
							float* q = keep ? (slot + r) : sink;
							*q = ride[8 * h + k];
							r += keep ? 1u : 0u;
						}
						__syncthreads();
						const f4a x0 = *(const f4a*)(slot + tid_here * 4u);
						const f4a x1 = *(const f4a*)(slot + VVM_HALF_TILE + tid_here * 4u);
						u32x4 y0, y1;
						y0.x = fbits(x0.x); y0.y = fbits(x0.y); y0.z = fbits(x0.z); y0.w = fbits(x0.w);
						y1.x = fbits(x1.x); y1.y = fbits(x1.y); y1.z = fbits(x1.z); y1.w = fbits(x1.w);
						__builtin_amdgcn_raw_buffer_store_b128(y0, rs, (int)(tid_here * 16u), 0, AUX_STREAM_STORE);
						__builtin_amdgcn_raw_buffer_store_b128(y1, rs, (int)((VVM_HALF_TILE + tid_here * 4u) * 4u), 0, AUX_STREAM_STORE);
					}
#endif
				}
			}
		}
	}
	p.stage_end = 0;
	p.active = false;
}

// Operand fetch.  The class and the index are wave-uniform, so every branch below is a scalar branch.  Two classes cover
// almost every operand of a linked program and are tested first: a per-lane register (GPR-index read) and a uniform register
// (lane i of `uni` holds constant i of the script; a lane read with a uniform index yields it as a scalar).
template <int V> DEV void fetch_slow(const Ctx& c, uint32_t o, float (&out)[V])
{
	// only uniform values are left here (scalar loads): advancing streams are in registers before the interpreter loop starts,
	// so the loop issues no vector memory instruction and never has to wait for one (a wait would also cover the output
	// stores of the previous tile, which are still draining)
	const uint32_t cls = VVM_OPND_CLS(o), idx = VVM_OPND_IDX(o);
	uint32_t u = 0;
	if (cls == VVM_OPC_CONST) u = sload32(c.consts, idx);
	else if (cls == VVM_OPC_NOADV) u = sload32((const void*)sload64((const void*)DESC_TABLE(c, in_ptrs), idx), 0);      // element [InstanceOffset], VectorVMRuntime.cpp:381-401
	else if (cls == VVM_OPC_NOADV_HALF) {                                                     // converted once, like VectorVMRuntime.cpp:388-394
		const uint64_t a = sload64((const void*)DESC_TABLE(c, in_ptrs), idx);
		const uint32_t w = sload32((const void*)(a & ~3ull), 0);
		u = fbits(load_half((a & 2ull) ? (w >> 16) : (w & 0xFFFFu)));
	}
#pragma unroll
	for (int k = 0; k < V; ++k) out[k] = bitsf(u);
}
template <int V, bool SPILL> DEV void fetch(CREGS, uint32_t uni, const Ctx& c, uint32_t o, float (&out)[V])
{
	if (o < (1u << 13)) {                       // VVM_OPC_VREG
		reg_read<V, SPILL>(RARGS, c, o, out);
	} else if (o < (2u << 13)) {                // VVM_OPC_UNIF
		const float v = bitsf((uint32_t)__builtin_amdgcn_readlane((int)uni, (int)(o & 63u)));
#pragma unroll
		for (int k = 0; k < V; ++k) out[k] = v;
	} else {
		fetch_slow<V>(c, o, out);
	}
}

struct KernelEnv {
	const VvmExecDesc* descs;      // null: the launch has one Exec, whose parameter block is `block` below
	uint32_t num_execs;
	uint32_t total_tiles;
	VvmLaunchState* ls;
	uint32_t lds_spill_slots;
	uint32_t debug_skip;     // bring-up aid (VVM_DEBUG_SKIP): bit 0 tiles, 1 stream loads, 2 tile-0 look-back word, 3 stream stores, 4 counts, 5 look-back (assume no deaths),
	                         // bit 10: odd tiles publish their count late (test hook)
	// Tag that makes a look-back word valid for this launch.  0: the epoch word of `ls`, which advances with every launch made
	// through these launch words (a state's launches: its look-back words are only ever written through its own launch words).
	// Launches that share one array of look-back words but not one set of launch words (a batch) get their tags from the
	// host, one counter per array, so that no word written earlier can carry the tag of a later launch.
	uint32_t tag;
	// Not 0 (FULL kernels only): the launch's Execs are the passes of ONE Exec over the same instances (a script that acquires
	// several datasets: one program per dataset).  Every ticket is one tile of all of them: the workgroup runs the passes on the
	// tile one after the other, so that the streams the passes share come from memory once; total_tiles counts tiles, not
	// tiles x passes, and every pass has its own look-back words.
	uint32_t fused;
	uint32_t run_shift;      // log2 of the run of adjacent tiles a ticket class works through (see the ticket draw)
	// A single Exec's parameter block (descriptor and tables, VVM_DESC_RELATIVE) rides in the kernel-argument segment: no
	// host-to-device copy ahead of the launch (a copy is a launch of its own on the stream, and cost more than half of the
	// 13 us between two Execs).  The kernel never names this member; it reads the segment through scalar loads.
	alignas(64) uint8_t block[VVM_INLINE_PARAM_BYTES];
};

// The cases of the interpreter's opcode switches, one list per operand count (the lists are used by run_tile and by the compiled path of the
// eight-wide kernels; they expand where LANES, A .. E, IA .., SETF / SETI / SETM, r, c, aux, aux2, inst_i, V and FULL are defined)
#define VVM_CASES_NIN2 \
				case VVM_OP_add: SETF(A + B); \
				case VVM_OP_sub: SETF(A - B); \
				case VVM_OP_mul: SETF(A * B); \
				case VVM_OP_div: SETF(f_div(A, B)); \
				case VVM_OP_atan2: if constexpr (FULL) SETF(ue_atan2(A, B)); \
				case VVM_OP_fmod: if constexpr (FULL) SETF(f_fmod(A, B)); \
				case VVM_OP_min: SETF(f_min(A, B)); \
				case VVM_OP_max: SETF(f_max(A, B)); \
				case VVM_OP_pow: if constexpr (FULL) SETF(f_pow(A, B)); \
				case VVM_OP_step: SETF(f_step(B - A));                     /* RT.cpp:1939 */ \
				case VVM_OP_cmplt: SETM(A < B); \
				case VVM_OP_cmple: SETM(A <= B); \
				case VVM_OP_cmpgt: SETM(A > B); \
				case VVM_OP_cmpge: SETM(A >= B); \
				case VVM_OP_cmpeq: SETM(A == B); \
				case VVM_OP_cmpneq: SETM(A != B); \
				case VVM_OP_addi: SETI(UA + UB); \
				case VVM_OP_subi: SETI(UA - UB); \
				case VVM_OP_muli: SETI(UA * UB); \
				case VVM_OP_divi: SETI(i_div(IA, IB)); \
				case VVM_OP_mini: SETI(i_min(IA, IB)); \
				case VVM_OP_maxi: SETI(i_max(IA, IB)); \
				case VVM_OP_cmplti: SETM(IA < IB); \
				case VVM_OP_cmplei: SETM(IA <= IB); \
				case VVM_OP_cmpgti: SETM(IA > IB); \
				case VVM_OP_cmpgei: SETM(IA >= IB); \
				case VVM_OP_cmpeqi: SETM(IA == IB); \
				case VVM_OP_cmpneqi: SETM(IA != IB); \
				case VVM_OP_bit_and: case VVM_OP_logic_and: SETI(UA & UB); \
				case VVM_OP_bit_or: case VVM_OP_logic_or: SETI(UA | UB); \
				case VVM_OP_bit_xor: case VVM_OP_logic_xor: SETI(UA ^ UB); \
				case VVM_OP_bit_lshift: SETI(i_shl(UA, UB)); \
				case VVM_OP_bit_rshift: SETI(i_shr(UA, UB)); \
				case VVM_OP_sub_neg: SETF(f_neg(A - B)); \
				case VVM_OP_div_f2i: SETI(cvtt(f_div(A, B))); \
				case VVM_OP_i2f_div0: SETF(f_div(i2f(IA), B)); \
				case VVM_OP_i2f_div1: SETF(f_div(B, i2f(IA))); \
				case VVM_OP_i2f_mul: SETF(i2f(IA) * B); \
				case VVM_OP_f2i_maxi: SETI(i_max(cvtt(A), IB)); \
				case VVM_OP_f2i_addi: SETI((uint32_t)cvtt(A) + UB); \
				case VVM_OP_bit_and_i2f: SETF(i2f((int32_t)(UA & UB))); \
				case VVM_OP_neg_cmplt: SETM(f_neg(A) < B); \
				case VVM_OP_max_f2i: SETI(cvtt(f_max(A, B))); \
				case VVM_OP_random_add: SETF(rand01(draw_bits<FULL>(c, inst_i + slot_offset<V>(k), aux2)) * A + B); \
				default: { _Pragma("unroll") LANES r[k] = 0.0f; } break;
#define VVM_CASES_NIN3 \
				case VVM_OP_mad: SETF(f_mad(A, B, C)); \
				case VVM_OP_lerp: SETF(f_lerp(A, B, C)); \
				case VVM_OP_clamp: SETF(f_clamp(A, B, C)); \
				case VVM_OP_select: SETF(f_sel(A, B, C)); \
				case VVM_OP_clampi: SETI(i_min(i_max(IA, IB), IC)); \
				case VVM_OP_cmplt_logic_and: SETI((A < B ? ~0u : 0u) & UC); \
				case VVM_OP_cmple_logic_and: SETI((A <= B ? ~0u : 0u) & UC); \
				case VVM_OP_cmpgt_logic_and: SETI((A > B ? ~0u : 0u) & UC); \
				case VVM_OP_cmpge_logic_and: SETI((A >= B ? ~0u : 0u) & UC); \
				case VVM_OP_cmpeq_logic_and: SETI((A == B ? ~0u : 0u) & UC); \
				case VVM_OP_cmpne_logic_and: SETI((A != B ? ~0u : 0u) & UC); \
				case VVM_OP_cmplti_logic_and: SETI((IA < IB ? ~0u : 0u) & UC); \
				case VVM_OP_cmplei_logic_and: SETI((IA <= IB ? ~0u : 0u) & UC); \
				case VVM_OP_cmpgti_logic_and: SETI((IA > IB ? ~0u : 0u) & UC); \
				case VVM_OP_cmpgei_logic_and: SETI((IA >= IB ? ~0u : 0u) & UC); \
				case VVM_OP_cmpeqi_logic_and: SETI((IA == IB ? ~0u : 0u) & UC); \
				case VVM_OP_cmpnei_logic_and: SETI((IA != IB ? ~0u : 0u) & UC); \
				case VVM_OP_cmplt_logic_or: SETI((A < B ? ~0u : 0u) | UC); \
				case VVM_OP_cmple_logic_or: SETI((A <= B ? ~0u : 0u) | UC); \
				case VVM_OP_cmpgt_logic_or: SETI((A > B ? ~0u : 0u) | UC); \
				case VVM_OP_cmpge_logic_or: SETI((A >= B ? ~0u : 0u) | UC); \
				case VVM_OP_cmpeq_logic_or: SETI((A == B ? ~0u : 0u) | UC); \
				case VVM_OP_cmpne_logic_or: SETI((A != B ? ~0u : 0u) | UC); \
				case VVM_OP_cmplti_logic_or: SETI((IA < IB ? ~0u : 0u) | UC); \
				case VVM_OP_cmplei_logic_or: SETI((IA <= IB ? ~0u : 0u) | UC); \
				case VVM_OP_cmpgti_logic_or: SETI((IA > IB ? ~0u : 0u) | UC); \
				case VVM_OP_cmpgei_logic_or: SETI((IA >= IB ? ~0u : 0u) | UC); \
				case VVM_OP_cmpeqi_logic_or: SETI((IA == IB ? ~0u : 0u) | UC); \
				case VVM_OP_cmpnei_logic_or: SETI((IA != IB ? ~0u : 0u) | UC); \
				case VVM_OP_mad_sqrt: SETF(ieee_sqrt(f_mad(A, B, C)));      /* unguarded, RT.cpp:2033 */ \
				case VVM_OP_mul_add: SETF((A * B) + C); \
				case VVM_OP_mul_sub0: SETF((A * B) - C); \
				case VVM_OP_mul_sub1: SETF(C - (A * B)); \
				case VVM_OP_mul_mul: SETF((A * B) * C); \
				case VVM_OP_mul_max: SETF(f_max(A * B, C)); \
				case VVM_OP_add_add: SETF((A + B) + C); \
				case VVM_OP_sub_cmplt1: SETM(C < (A - B)); \
				case VVM_OP_sub_mul: SETF((A - B) * C); \
				case VVM_OP_div_mul: SETF(f_div(A, B) * C); \
				case VVM_OP_muli_addi: SETI(UA * UB + UC); \
				case VVM_OP_addi_bit_rshift: SETI(i_shr(UA + UB, UC)); \
				case VVM_OP_addi_muli: SETI((UA + UB) * UC); \
				case VVM_OP_i2f_mad0: SETF(f_mad(i2f(IA), B, C)); \
				case VVM_OP_i2f_mad1: SETF(f_mad(A, B, i2f(IC))); \
				case VVM_OP_f2i_select1: SETI((UA & (uint32_t)cvtt(B)) | (~UA & UC)); \
				case VVM_OP_fmod_add: if constexpr (FULL) SETF(f_fmod(A, B) + C); \
				case VVM_OP_bit_rshift_bit_and: SETI(i_shr(UA, UB) & UC); \
				case VVM_OP_bit_or_muli: SETI((UA | UB) * UC); \
				case VVM_OP_bit_lshift_bit_or: SETI(i_shl(UA, UB) | UC); \
				case VVM_OP_external_func_call: \
					/* aux = built-in id, aux2 = which output this device instruction produces (VectorVMRuntime.cpp:1450-1464 lists the */ \
					/* operands as inputs then outputs; the linker emits one instruction per output in use) */ \
					if constexpr (FULL) { \
						if (aux == VVM_EXTFN_PERLIN3) { _Pragma("unroll") LANES r[k] = perlin3(A, B, C); } \
						else { _Pragma("unroll") LANES r[k] = curl_noise3_component(A, B, C, aux2); } \
					} \
					break; \
				default: { _Pragma("unroll") LANES r[k] = 0.0f; } break;
#define VVM_CASES_NIN1 \
				case VVM_DOP_COPY: SETF(A); \
				case VVM_OP_rcp: SETF(f_rcp(A)); \
				case VVM_OP_rsq: SETF(f_rsq(A)); \
				case VVM_OP_sqrt: SETF(f_sqrtg(A)); \
				case VVM_OP_neg: SETF(f_neg(A)); \
				case VVM_OP_abs: SETF(f_abs(A)); \
				case VVM_OP_exp: if constexpr (FULL) SETF(mf_exp(A)); \
				case VVM_OP_exp2: if constexpr (FULL) SETF(lib_exp2(A)); \
				case VVM_OP_log: if constexpr (FULL) SETF(f_log(A)); \
				case VVM_OP_log2: if constexpr (FULL) SETF(lib_log2(A)); \
				case VVM_OP_sin: if constexpr (FULL) SETF(mf_sin(A)); \
				case VVM_OP_cos: if constexpr (FULL) SETF(mf_cos(A)); \
				case VVM_OP_tan: if constexpr (FULL) SETF(lib_tan(A)); \
				case VVM_OP_asin: if constexpr (FULL) SETF(f_asin(A)); \
				case VVM_OP_acos: if constexpr (FULL) SETF(f_acos(A)); \
				case VVM_OP_atan: if constexpr (FULL) SETF(lib_atan(A)); \
				case VVM_OP_ceil: SETF(ceilf(A)); \
				case VVM_OP_floor: SETF(floorf(A)); \
				case VVM_OP_frac: SETF(f_frac(A)); \
				case VVM_OP_trunc: SETF(truncf(A)); \
				case VVM_OP_round: SETF(rintf(A));                         /* SSE.h:2819-2822 (nearest even) */ \
				case VVM_OP_sign: SETF(f_sign(A)); \
				case VVM_OP_absi: SETI(i_abs(IA)); \
				case VVM_OP_negi: SETI(0u - UA); \
				case VVM_OP_signi: SETI(IA >= 0 ? 1 : -1); \
				case VVM_OP_bit_not: case VVM_OP_logic_not: SETI(~UA); \
				case VVM_OP_f2i: SETI(cvtt(A)); \
				case VVM_OP_i2f: SETF(i2f(IA)); \
				case VVM_OP_f2b: SETM(A > 0.0f); \
				case VVM_OP_b2f: SETF(f_sel(A, 1.0f, 0.0f)); \
				case VVM_OP_i2b: SETM(IA > 0); \
				case VVM_OP_b2i: SETI(UA & 1u); \
				case VVM_OP_sin_cos: if constexpr (FULL) { _Pragma("unroll") LANES { float sn, cs; f_sincos(A, sn, cs); r[k] = aux ? cs : sn; } } break;   /* aux: 0 sine, 1 cosine */ \
				case VVM_OP_half_to_float: SETF(load_half(UA));             /* RT.cpp:1113-1127 */ \
				case VVM_OP_exec_index_addi: SETI(DESC_U32(c.dp, exec_index_base) + inst_i + slot_offset<V>(k) + UA);             /* RT.cpp:925-951 */ \
				/* aux2 = ordinal of the random op within the script */ \
				case VVM_OP_random: SETF(rand01(draw_bits<FULL>(c, inst_i + slot_offset<V>(k), aux2)) * A); \
				case VVM_OP_randomi: SETI(cvtt(rand01(draw_bits<FULL>(c, inst_i + slot_offset<V>(k), aux2)) * i2f(IA))); \
				default: { _Pragma("unroll") LANES r[k] = 0.0f; } break;
#define VVM_CASES_NIN4 \
				/* merged ops, VectorVMRuntime.cpp:1999-2074 */ \
				case VVM_OP_cmplt_select: SETF(f_sel(mask(A < B), C, D)); \
				case VVM_OP_cmple_select: SETF(f_sel(mask(A <= B), C, D)); \
				case VVM_OP_cmpeq_select: SETF(f_sel(mask(A == B), C, D)); \
				case VVM_OP_cmplti_select: SETF(f_sel(mask(IA < IB), C, D)); \
				case VVM_OP_cmplei_select: SETF(f_sel(mask(IA <= IB), C, D)); \
				case VVM_OP_cmpeqi_select: SETF(f_sel(mask(IA == IB), C, D)); \
				case VVM_OP_mad_add: SETF(f_mad(A, B, C) + D); \
				case VVM_OP_mad_sub0: SETF(f_mad(A, B, C) - D); \
				case VVM_OP_mad_sub1: SETF(D - f_mad(A, B, C)); \
				case VVM_OP_mad_mul: SETF(f_mad(A, B, C) * D); \
				case VVM_OP_mul_mad0: SETF(f_mad(A * B, C, D)); \
				case VVM_OP_mul_mad1: SETF(f_mad(C, D, A * B)); \
				case VVM_OP_add_mad1: SETF(f_mad(C, D, A + B)); \
				case VVM_OP_div_mad0: SETF(f_mad(f_div(A, B), C, D)); \
				case VVM_OP_select_mul: SETF(f_sel(A, B, C) * D); \
				case VVM_OP_select_add: SETF(f_sel(A, B, C) + D); \
				default: { _Pragma("unroll") LANES r[k] = 0.0f; } break;
#define VVM_CASES_NIN0 \
				case VVM_OP_exec_index: SETI(DESC_U32(c.dp, exec_index_base) + inst_i + slot_offset<V>(k));                       /* RT.cpp:880-900 */ \
				case VVM_OP_exec_indexf: SETF(i2f((int32_t)(DESC_U32(c.dp, exec_index_base) + inst_i + slot_offset<V>(k))));      /* RT.cpp:902-923 */ \
				case VVM_OP_acquire_id: if constexpr (FULL) { \
					/* VectorVMRuntime.cpp:1404-1440.  Instance g of the Exec takes FreeIDTable[NumFreeIDs - 1 - g]: each chunk pops its */ \
					/* IDs from the end of the free list, reversed (:1417-1421), and the chunks pop in order.  The IDs are also logged as */ \
					/* spawned: chunk c's, in instance order, at FreeIDTable[Max - (NumSpawnedIDs + instances up to the end of chunk c)] */ \
					/* (:1429-1432).  aux 1 is the instruction's second destination, the tag (:1435). */ \
					const void* idp = (const void*)(DESC_PTR(c.dp, ids) + (uint64_t)aux2 * sizeof(VvmIdDesc));      /* (aux2: the dataset's descriptor) */ \
					if (aux == 1) { \
						const uint32_t tag_value = sload32(idp, offsetof(VvmIdDesc, acquire_tag) / 4); \
						_Pragma("unroll") LANES r[k] = bitsf(tag_value); \
					} else { \
						const uint32_t table_max = sload32(idp, offsetof(VvmIdDesc, free_table_max) / 4); \
						const rsrc_t rs_free = make_rsrc(sload64(idp, offsetof(VvmIdDesc, free_table) / 8), 4u * table_max); \
						const rsrc_t rs_read = make_rsrc(sload64(idp, offsetof(VvmIdDesc, free_read) / 8), 4u * table_max); \
						const uint32_t num_free = sload32(idp, offsetof(VvmIdDesc, num_free_start) / 4), num_spawned = sload32(idp, offsetof(VvmIdDesc, num_spawned_start) / 4); \
						const uint32_t chunk = sload32(idp, offsetof(VvmIdDesc, chunk_instances) / 4); \
						_Pragma("unroll") LANES { \
							const uint32_t g = inst_i + slot_offset<V>(k); \
							const uint32_t ok = (uint32_t)(g < c.num_instances) & (uint32_t)(g < num_free); \
							const uint32_t id = __builtin_amdgcn_raw_buffer_load_b32(rs_read, (int)(ok ? (num_free - 1u - g) * 4u : VVM_RSRC_NOWHERE), 0, 0); \
							const uint32_t ci = g / chunk; \
							const uint32_t chunk_end = min(c.num_instances, (ci + 1u) * chunk); \
							uint32_t pos = table_max - (num_spawned + chunk_end) + (g - ci * chunk); \
							asm volatile("" : "+v"(pos));      /* computed for every lane, then selected: otherwise the division ends up behind a per-lane branch */ \
							__builtin_amdgcn_raw_buffer_store_b32(id, rs_free, (int)(ok ? pos * 4u : VVM_RSRC_NOWHERE), 0, 0); \
							r[k] = bitsf(id); \
						} \
					} \
				} break; \
				default: { _Pragma("unroll") LANES r[k] = 0.0f; } break;
#define VVM_CASES_NIN5 \
				case VVM_OP_mad_mad0: SETF(f_mad(D, E, f_mad(A, B, C))); \
				case VVM_OP_mad_mad1: SETF(f_mad(f_mad(A, B, C), D, E)); \
				default: { _Pragma("unroll") LANES r[k] = 0.0f; } break;

// ------------------------------------------------------------------------------------------------------
template <int V, bool SPILL, bool FULL, bool RIDE = false>
DEV void run_tile(const VvmIns* prog, gu64* status, bool last_tile, const Ctx& c, unsigned char* lds, uint32_t tile_local, uint32_t tag, VvmLaunchState* ls, uint32_t dbg, Pending& pend,
                  uint32_t& resolves, const u64 (&pre)[VVM_LB_WORDS_PER_LANE], bool have_pre)
{
	// (four instances per lane or one; the eight-wide kernels have a tile program of their own, vvm_eight_wide.h)
	static_assert(V != 8, "eight instances per lane: run_tile8");
	RVEC R0 = 0.0f, R1 = 0.0f, R2 = 0.0f, R3 = 0.0f, R4 = 0.0f, R5 = 0.0f, R6 = 0.0f, R7 = 0.0f;      // the register file starts every tile as zeros
	uint32_t* lds_ctrl = (uint32_t*)lds;
	const float no_ride[16] = {};      // (store_tile's riding outputs exist in the light eight-wide kernel only)
	// acquireindex state of the tile: all wave-uniform except keep and rank
	uint32_t t_keep = 0, t_rank = 0, t_rank_b = 0;      // (t_rank_b: the second half of an eight-instance tile)
	uint32_t t_total = 0, t_base = 0;
	bool t_pending = false;          // the tile's count is published, its prefix not yet resolved
	// the same for the second chain of a program that acquires two datasets (acquireindex's aux and an output's aux2 say which)
	constexpr bool TWO = HAS_TWO_CHAINS(V, SPILL, FULL);
	uint32_t u_keep = 0, u_rank = 0, u_rank_b = 0, u_total = 0, u_base = 0;
	bool u_pending = false;
	const uint32_t exec_tiles = TWO ? DESC_U32(c.dp, num_tiles) : 0u;
	uint32_t stage_end = 0;          // output-list index one past the last slot this tile has staged (0: nothing staged)

	// Uniform registers: lane i of `uni` holds constant i.  Filled by scalar loads and lane writes (the table is padded to a
	// multiple of four dwords): a vector load would leave a pending-load wait on every later read of `uni`, and that wait
	// would also cover the previous tile's output stores.
	const uint32_t uni = fill_uniforms<0>(0u, c.consts, DESC_U32(c.dp, num_unif));
	const u64 tl0 = (dbg & 128u) ? __builtin_amdgcn_s_memtime() : 0;

	// Prologue: the program starts with its VVM_DOP_LOADN / VVM_DOP_LOADN16 instructions.  They run ahead of the main loop so
	// that the main loop has exactly one place that inserts into the register vectors (with two, the compiler keeps a second
	// copy of the whole register file alive across the loop).
	//   aux = count (1..8), aux2 = first destination register, w1 = first of eight slots of the load list: the eight stream
	//   addresses arrive with one scalar load, then all stream loads are issued back to back.
	//   Unused slots write (zeros) to the last used slot's register, and the slots are written from the last one down: every
	//   write is a plain insert, none has to read the register it might have to preserve.
#define VVM_ISSUE_LOADS(W0, W1, ST) do { \
		const uint32_t lop_ = (W0) & 0xFFu, cnt_ = ((W0) >> 16) & 0xFFu; \
		uint64_t ptr_[8]; \
		sload64x8(load_ptrs, (W1), ptr_); \
		_Pragma("unroll") for (int j = 0; j < 8; ++j) { \
			_Pragma("unroll") for (int k = 0; k < V; ++k) (ST)[j][k] = 0.0f; \
			if ((uint32_t)j < cnt_ && !(dbg & 2u)) { \
				if (lop_ == VVM_DOP_LOADN) load_stream32<V>(c, ptr_[j], (ST)[j]); \
				else load_stream16<V>(c, ptr_[j], (ST)[j]); \
			} \
		} \
	} while (0)
#define VVM_WRITE_LOADS(W0, ST) do { \
		const uint32_t cnt_ = ((W0) >> 16) & 0xFFu, dst_ = (W0) >> 24; \
		_Pragma("unroll") for (int j = 7; j >= 0; --j) REG_WRITE(dst_ + min((uint32_t)j, cnt_ - 1u), (ST)[j]); \
	} while (0)
	const void* load_ptrs = (const void*)DESC_TABLE(c, load_ptrs);
	uint32_t pc0 = 0;
	for (;; ++pc0) {
		const uint32_t w0 = sload32(prog, pc0 * (VVM_INS_STRIDE / 4u) + 0);
		const uint32_t lop = w0 & 0xFFu;
		if (lop != VVM_DOP_LOADN && lop != VVM_DOP_LOADN16) break;
		const uint32_t w1 = sload32(prog, pc0 * (VVM_INS_STRIDE / 4u) + 1);
		float st[8][V];
		VVM_ISSUE_LOADS(w0, w1, st);
		VVM_WRITE_LOADS(w0, st);
	}
#undef VVM_ISSUE_LOADS
#undef VVM_WRITE_LOADS

	// The program is a sequence of runs of ordinary instructions separated by the few special ones (acquireindex, outputs,
	// the end).  The inner loop interprets one run; the special instructions are handled between runs.  The next instruction
	// is always fetched while the current one executes.
	//
	// Every ordinary instruction ends at the single register write at the bottom of the inner loop: each device instruction has
	// exactly one destination register (the linker splits the reference's two-output ops and the three-output noise
	// built-in).  With one insert site the register vectors stay in place; a second one makes the compiler keep a copy of the
	// whole register file.
	const u64 tl1 = (dbg & 128u) ? __builtin_amdgcn_s_memtime() : 0;
	VvmIns nxt;
	uint32_t pc = pc0 * VVM_INS_STRIDE;       // byte offset of the next instruction to fetch (the linker's four words of each VvmDevIns)
	sload32x4_at(prog, pc, nxt.w0, nxt.w1, nxt.w2, nxt.w3);
	pc += VVM_INS_STRIDE;
	for (;;) {
		uint32_t op, aux, aux2, o0, o1, o2, o3, o4, o5;
		for (;;) {
			const VvmIns ins = nxt;
			sload32x4_at(prog, pc, nxt.w0, nxt.w1, nxt.w2, nxt.w3);
			pc += VVM_INS_STRIDE;
			const uint32_t w0 = ins.w0;
			op = w0 & 0xFFu; aux = (w0 >> 16) & 0xFFu; aux2 = w0 >> 24;
			const uint32_t nin = (w0 >> 8) & 7u;
			o0 = ins.w1 & 0xFFFFu; o1 = ins.w1 >> 16; o2 = ins.w2 & 0xFFFFu; o3 = ins.w2 >> 16; o4 = ins.w3 & 0xFFFFu; o5 = ins.w3 >> 16;

			float r[V];
			const uint32_t dst = o5;
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
// the most common ops of an operand count are tested first, one after the other, ahead of the switch over the rest
#define HOTF(opcode, expr) if (nin_is(op, opcode)) { _Pragma("unroll") LANES r[k] = (expr); } else
#define HOTM(opcode, expr) if (nin_is(op, opcode)) { _Pragma("unroll") LANES r[k] = mask(expr); } else
			// One group of cases per operand count: the operands an instruction has are fetched, no others exist (a common fetch
			// sequence in front of one big switch leaves the compiler with placeholder values to materialize for the missing ones).
			// (tested in order of how common the operand counts are: a chain of equality tests, not a balanced tree)
			if (nin_is(nin, 2u)) {
				float a[V], b[V];
				fetch<V, SPILL>(RARGS, uni, c, o0, a);
				fetch<V, SPILL>(RARGS, uni, c, o1, b);
				HOTF(VVM_OP_mul, A * B) HOTF(VVM_OP_add, A + B) HOTF(VVM_OP_sub, A - B) HOTM(VVM_OP_cmplt, A < B)
				switch (op) {
				VVM_CASES_NIN2
				}
			}
			else if (nin_is(nin, 3u)) {
				float a[V], b[V], cc[V];
				fetch<V, SPILL>(RARGS, uni, c, o0, a);
				fetch<V, SPILL>(RARGS, uni, c, o1, b);
				fetch<V, SPILL>(RARGS, uni, c, o2, cc);
				HOTF(VVM_OP_mad, f_mad(A, B, C)) HOTF(VVM_OP_select, f_sel(A, B, C)) HOTF(VVM_OP_mul_add, (A * B) + C)
				switch (op) {
				VVM_CASES_NIN3
				}
			}
			else if (nin_is(nin, 1u)) {
				float a[V];
				fetch<V, SPILL>(RARGS, uni, c, o0, a);
				switch (op) {
				VVM_CASES_NIN1
				}
			}
			else if (nin_is(nin, 4u)) {
				float a[V], b[V], cc[V], dd[V];
				fetch<V, SPILL>(RARGS, uni, c, o0, a);
				fetch<V, SPILL>(RARGS, uni, c, o1, b);
				fetch<V, SPILL>(RARGS, uni, c, o2, cc);
				fetch<V, SPILL>(RARGS, uni, c, o3, dd);
				HOTF(VVM_OP_mad_mul, f_mad(A, B, C) * D) HOTF(VVM_OP_mad_add, f_mad(A, B, C) + D)
				switch (op) {
				VVM_CASES_NIN4
				}
			}
			else if (nin_is(nin, 0u)) {
				switch (op) {
				VVM_CASES_NIN0
				}
			}
			else if (nin_is(nin, 5u)) {
				float a[V], b[V], cc[V], dd[V], ee[V];
				fetch<V, SPILL>(RARGS, uni, c, o0, a);
				fetch<V, SPILL>(RARGS, uni, c, o1, b);
				fetch<V, SPILL>(RARGS, uni, c, o2, cc);
				fetch<V, SPILL>(RARGS, uni, c, o3, dd);
				fetch<V, SPILL>(RARGS, uni, c, o4, ee);
				switch (op) {
				VVM_CASES_NIN5
				}
			}
			else goto leave_run;      // VVM_NIN_SPECIAL: done, or a device-only instruction (handled below)
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
#undef HOTF
#undef HOTM
			// the one place where the interpreter writes registers
			REG_WRITE(VVM_OPND_IDX(dst), r);
		}
leave_run:

		if (op == VVM_DOP_ACQUIRE) {
			// sign bit of the predicate = keep (VectorVMRuntime.cpp:838-841); instances past the end never survive (:846-857)
			// aux2 != 0: the linker has folded the compare that produced the predicate into this instruction (the mask had no other
			// reader): keep = o0 <cmp> o1, the same bit the compare's all-ones mask would carry in its sign
			float p[V];
			fetch<V, SPILL>(RARGS, uni, c, o0, p);
			if (aux2 != VVM_ACQ_SIGN) {
				float q[V];
				fetch<V, SPILL>(RARGS, uni, c, o1, q);
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
			const uint3 rt = acquire_rank_and_publish<(V == 8 ? 2 : 1)>(kb, c, lds_ctrl, status + (size_t)chain * exec_tiles, tile_local, tag, dbg, chain);
			if (TWO && chain) { u_keep = kb; u_rank = rt.x; u_rank_b = rt.y; u_total = SCALAR(rt.z); u_pending = true; }
			else { t_keep = kb; t_rank = rt.x; t_rank_b = rt.y; t_total = SCALAR(rt.z); t_pending = true; }
		}
		// The previous tile is finished here: after this tile has published its own count (every later tile's look-back waits
		// for that word, so it goes out first), and before this tile needs the staging slots.  The first poll of the previous
		// tile's look-back has been in flight since this tile was claimed.
		if (pend.active) {
			resolve_tile<V, TWO>(pend, c, lds_ctrl, resolves, tag, ls, dbg, pre, have_pre);
			store_tile<V, FULL, false, TWO>(pend, c, lds, dbg, no_ride);
			__syncthreads();      // every thread has read its staged values: the slots may be written again
		}
		if (op == VVM_DOP_ACQUIRE) continue;
		if (op == VVM_OP_done) break;

		if (op == VVM_DOP_FLUSH) {
			// the next output instruction starts a new group of staging slots: write out the staged group now (this waits for the
			// tile's look-back)
			Pending cur;
			cur.active = true; cur.unresolved = t_pending; cur.total = t_total; cur.base = t_base;
			cur.stage_end = stage_end; cur.last = last_tile;
			cur.tile_local = tile_local; cur.status = status; cur.out_list = c.out_list; cur.ride = 0;
			cur.unresolved1 = u_pending; cur.total1 = u_total; cur.base1 = u_base; cur.tiles = exec_tiles;
			resolve_tile<V, TWO>(cur, c, lds_ctrl, resolves, tag, ls, dbg, pre, false);
			store_tile<V, FULL, false, TWO>(cur, c, lds, dbg, no_ride);
			__syncthreads();      // the slots are written again right away
			t_base = cur.base; t_pending = false; stage_end = 0;
			u_base = cur.base1; u_pending = false;
			continue;
		}

		if (FULL && V != 8 && op == VVM_DOP_UPDATE_ID) {
			// the ID register of every instance of the tile goes to a staging slot as it is, with the lane's acquireindex result
			// next to it; apply_update_id does the rest once the tile's output position is known
			const uint32_t slot_index = o4 | (o5 << 16);
			float v[V];
			fetch<V, SPILL>(RARGS, uni, c, o0, v);
			float* slot = (float*)(lds + LDS_STAGE_OFF(FULL)) + (slot_index & (VVM_STAGE_SLOTS - 1u)) * (VVM_BLOCK_THREADS * V) + c.tid * V;
			if constexpr (V >= 4) { f4a x; x.x = v[0]; x.y = v[1]; x.z = v[2]; x.w = v[3]; *(f4a*)slot = x; }
			else slot[0] = v[0];
			uint32_t exists = 0;
#pragma unroll
			for (int k = 0; k < V; ++k) exists |= ((c.inst0 + slot_offset<V>(k) < c.num_instances) ? 1u : 0u) << k;
			((uint32_t*)(lds + LDS_STAGE_KEEP_OFF(V, FULL, IS_LIGHT(V, SPILL))))[c.tid] = t_rank | (t_keep << 16) | (exists << 24);
			stage_end = max(stage_end, slot_index + 1u);      // (the group's last slot: outputs staged after this one lie below it)
			continue;
		}

		// Output instruction: aux = pair count (1..3), sources o0..o2, w3 = output-list index of the first pair; the pair's staging
		// slot is that index modulo VVM_STAGE_SLOTS.  The survivors' values are compacted into the slot; store_tile writes them
		// out later.  Without an acquireindex before it nothing survives (VectorVMRuntime.cpp:565, 769): everything goes to
		// the sink, and the host leaves such an output's list entry empty.
		{
			const bool second = TWO && aux2 != 0;      // the output's chain
			const uint32_t o_total = second ? u_total : t_total, o_keep = second ? u_keep : t_keep, o_rank = second ? u_rank : t_rank, o_rank_b = second ? u_rank_b : t_rank_b;
			const bool all_alive = (o_total == VVM_BLOCK_THREADS * V);
			const uint32_t kind = op - VVM_DOP_OUTPUT32;          // 0: 32-bit value  1: float -> half  2: low 16 bits as they are  3: half bits -> float
			const uint32_t first_slot = o4 | (o5 << 16);
			float* slot = (float*)(lds + LDS_STAGE_OFF(FULL)) + (first_slot & (VVM_STAGE_SLOTS - 1u)) * (VVM_BLOCK_THREADS * V);
			float* sink = (float*)(lds + LDS_STAGE_SINK_OFF(V, FULL, IS_LIGHT(V, SPILL))) + c.tid;
			uint64_t srcs = (uint64_t)o0 | ((uint64_t)o1 << 16) | ((uint64_t)o2 << 32);
			for (uint32_t j = 0; j < aux; ++j, srcs >>= 16, slot += VVM_BLOCK_THREADS * V) {
				float v[V];
				fetch<V, SPILL>(RARGS, uni, c, (uint32_t)srcs & 0xFFFFu, v);
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
					if constexpr (V >= 4) { f4a x; x.x = v[0]; x.y = v[1]; x.z = v[2]; x.w = v[3]; *(f4a*)(slot + c.tid * 4) = x; }
					else slot[c.tid] = v[0];
					if constexpr (V == 8) { f4a x; x.x = v[4]; x.y = v[5]; x.z = v[6]; x.w = v[7]; *(f4a*)(slot + VVM_HALF_TILE + c.tid * 4) = x; }
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
	pend.ride = t_keep | (t_rank << 8) | (t_rank_b << 20);
	if constexpr (RIDE) {
		// the same word for vvm_through_kernel from the general eight-wide kernel, straight from the tile that made it (see store_tile);
		// also from the instantiation of the four-wide kernel with the transcendental and noise functions that lean programs run on
		// (RIDE; its t_rank_b nobody reads.  An instantiation of its own: in the kernel every such script runs on, the store cost 23 more
		// spilled scalars and 0.5 to 1 % of the transcendental configuration's time)
		const uint64_t ride_base = sload64(c.out_list - 3, 0);
		if (ride_base != 0) {
			const rsrc_t rr = make_rsrc(ride_base + (uint64_t)tile_local * (VVM_BLOCK_THREADS * 4u), VVM_BLOCK_THREADS * 4u);
			__builtin_amdgcn_raw_buffer_store_b32(t_keep | (t_rank << 8) | (t_rank_b << 20), rr, (int)(c.tid * 4u), 0, 0);
		}
	}
}

#include "vvm_threaded.h"
#include "vvm_eight_wide.h"

// DBG: the bring-up switches of KernelEnv::debug_skip are compiled in (one instantiation only); otherwise they cost nothing.
template <int V, bool SPILL, bool FULL, bool DBG = false, bool RIDE = false>
__global__ void __launch_bounds__(VVM_BLOCK_THREADS, (IS_LIGHT(V, SPILL) ? 3 : V == 8 || (V == 4 && SPILL) ? 2 : FULL ? 3 : (V == 4 ? 4 : 8))) vvm_exec_kernel(KernelEnv env)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const uint32_t dbgw = DBG ? env.debug_skip : 0u;
	uint32_t* lds_ctrl = (uint32_t*)lds;
	uint8_t* lds_perlin = lds + LDS_PERLIN_OFF;
	float* lds_spill = (float*)(lds + LDS_SPILL_OFF(V, FULL, IS_LIGHT(V, SPILL)));
	// (eight instances per lane: the register file, in fixed VGPRs from here to the end of the kernel; the light kernel's riding outputs
	// wait in it from one tile to the next)
	constexpr bool THREADED = V == 8 || (V == 4 && SPILL);      // the kernels whose tile program is run_tile8
	FileOf<(THREADED ? V : 8), IS_LIGHT(V, SPILL), THREADED && FULL> file8;
	if constexpr (THREADED) file_clear(file8);
	Pending pend;
	pend.active = false; pend.unresolved = false; pend.last = false; pend.total = 0; pend.base = 0; pend.stage_end = 0; pend.tile_local = 0;
	pend.status = nullptr; pend.out_list = nullptr; pend.ride = 0;
	pend.unresolved1 = false; pend.total1 = 0; pend.base1 = 0; pend.tiles = 0;
	const uint32_t tid = threadIdx.x;
	const uint32_t wave = SCALAR(tid >> 6), lane = tid & 63u;
	// ticket counter of this workgroup's class (see VvmLaunchState); with fewer workgroups than classes only the first
	// gridDim.x counters are in use
	const uint32_t classes = min(gridDim.x, VVM_TICKET_CLASSES);
	// Which class a workgroup draws from must not follow from where it runs.  Workgroups go to the XCDs in turn (workgroup b to XCD
	// b % 8), so with class b % 16 all workgroups of a class sit on ONE XCD; when two such launches share the chip (two streams, or two
	// processes) and each holds the slots of some XCDs while its workgroups for the others wait to be placed, each has classes nobody
	// draws from, the tiles of those classes are never claimed, every later tile's look-back waits for them, and neither launch ever
	// frees a slot: both sit until the look-back's poll limit ends them (seen with two processes of 4096 tiles each on one GPU, and
	// once with two 64 Mi-instance Execs of one process on two streams).  With class (b + b / 16) % 16 the sixteen workgroups b = 16 g ..
	// 16 g + 15 still take sixteen different classes (tiles start in index order as before), and every XCD holds workgroups of every
	// class as soon as it holds sixteen of the launch's: a launch that is only partly placed claims every tile, so it finishes and
	// makes room.  (It also spreads a class's tiles over the XCDs instead of tying them to one: 1 to 2 % faster on Execs of 64 Mi
	// instances; an Exec small enough to find its streams in the XCDs' L2 from the Exec before loses that, 19.5 -> 20.8 us at 1 Mi.)
	const uint32_t cls = gridDim.x >= 2u * VVM_TICKET_CLASSES ? (blockIdx.x + (blockIdx.x >> 4)) % VVM_TICKET_CLASSES : blockIdx.x % classes;
	const uint32_t tag = env.tag ? env.tag : (SCALAR(__hip_atomic_load((gu32*)&env.ls->epoch, RLX_AGENT)) & 0x3FFFFFFFu);
	bool perlin_ready = false;
	uint32_t resolves = 0, started = 0;
	// (where the descriptors are, and what the table fields of a descriptor are relative to, see KernelEnv::block: per tile, below)

	if (wave == 0) lds_ctrl[LC_STATS + (lane & 7u)] = 0;
	// A tile is claimed when the workgroup is ready to start it, never ahead of time: the look-back of a tile waits for the
	// counts of all earlier tiles, so a tile that is claimed but not yet running stalls every later one.
	uint32_t next_floor = 0;      // no draw of this workgroup's class can yield a tile below this one any more
	for (uint32_t phase = 0;; ++phase) {
		// (a class hands out increasing tiles: when even its next draw lies past the end, the draw and its round trip are left
		// out; a launch with one round of tiles ends 1.5 us sooner)
		if (next_floor >= env.total_tiles) break;
		const u64 tk0 = (dbgw & 128u) ? __builtin_amdgcn_s_memtime() : 0;
		uint32_t* slot = lds_ctrl + LC_TICKET + (phase & 1u);
		// Wave 0 draws the ticket and, in the shadow of that round trip, issues the first look-back poll for the tile the
		// workgroup has just finished; the poll is examined after this tile's stream loads have been issued (resolve_tile).
		u64 pre[VVM_LB_WORDS_PER_LANE] = {};
		const bool have_pre = pend.active && pend.unresolved && pend.tile_local != 0 && !(dbgw & (32u | 512u));
		if (wave == 0) {
			const rsrc_t rs = make_rsrc((uint64_t)&env.ls->tickets[cls * (VVM_TICKET_STRIDE_BYTES / 4u)], 4u);
			const int drawn = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rs, (int)(lane == 0 ? 0u : 4u), 0, AUX_AGENT);
			if (have_pre) look_back_poll(pend.status, (int32_t)pend.tile_local - 1, lane, pre);
			*slot = SCALAR(drawn);
		}
		__syncthreads();      // (the word alternates between tiles, so one barrier is enough)
		const uint32_t draw = SCALAR(*slot);
		if ((dbgw & 128u) && wave == 0) {
			uint32_t* st = lds_ctrl + LC_STATS;
			const uint32_t s4 = st[4] + (uint32_t)((__builtin_amdgcn_s_memtime() - tk0) >> 4), s7 = st[7] + 1u;
			st[4] = s4; st[7] = s7;
		}
		// draw c of class k is tile (c / RUN) * RUN * classes + k * RUN + c % RUN: a class works through runs of RUN adjacent tiles
		// (RUN = 1: tile c * classes + k); increasing in c, so the class is done at the first tile past the end.  RUN = 1 << run_shift
		// comes with the launch: 2 when there are several rounds of tiles (measured on the 64 Mi fountain Exec: RUN 1 0.770 ms,
		// 2 0.758 ms, 4 0.760 ms, 8 0.762 ms, 16 0.771 ms), 1 for small launches (with RUN 2 half of the workgroups would take
		// two tiles each and the other half none)
		const uint32_t rs = env.run_shift;
		const uint32_t ticket = ((draw >> rs) * classes << rs) + (cls << rs) + (draw & ((1u << rs) - 1u));
		if (ticket >= env.total_tiles) break;
		next_floor = (((draw + 1u) >> rs) * classes << rs) + (cls << rs) + ((draw + 1u) & ((1u << rs) - 1u));

		// Both are re-read from the kernel-argument segment for every tile (one scalar load): kept across the tile they cost four
		// scalar registers that the four-wide kernel does not have to spare (1.7 % on the half-dying workload).
		uint64_t karg = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
		asm volatile("" : "+s"(karg));      // (opaque per tile, or the loads below are hoisted out of the loop again)
		const uint64_t given = sload64((const void*)karg, offsetof(KernelEnv, descs) / 8);
		const uint64_t block_addr = karg + offsetof(KernelEnv, block);
		const VvmExecDesc* descs = given ? (const VvmExecDesc*)given : (const VvmExecDesc*)block_addr;
		const uint64_t rel = given ? 0ull : block_addr;
		// which Exec owns this ticket: the last descriptor with first_tile <= ticket; fused passes: all of them, in turn
		uint32_t e = 0, e_end = 1;
		bool fused = false;
		if constexpr (FULL) fused = env.fused != 0;
		if (fused) e_end = env.num_execs;
		else if (env.num_execs > 1) {
			uint32_t lo = 0, hi = env.num_execs;
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (sload32(descs + mid, offsetof(VvmExecDesc, first_tile) / 4) <= ticket) lo = mid; else hi = mid;
			}
			e = lo;
		}
		bool first_pass = true;
		do {
		const void* dp = (const void*)(descs + e);
		const uint32_t num_instances = DESC_U32(dp, num_instances);
		const uint32_t tile_local = ticket - DESC_U32(dp, first_tile);
		const uint32_t tile_first = tile_local * (VVM_BLOCK_THREADS * V);
		const bool last_tile = (tile_local + 1u == DESC_U32(dp, num_tiles));

		if (FULL && (DESC_U32(dp, flags) & VVM_DESC_USES_PERLIN) && !perlin_ready) {
			const uint8_t* perlin = (const uint8_t*)DESC_PTR(dp, perlin);
			for (uint32_t i = tid; i < 512; i += VVM_BLOCK_THREADS) {
				const uint8_t h = perlin[i];
				lds_perlin[i] = h;
				lds_perlin[(LDS_PERLIN_SEL_OFF - LDS_PERLIN_OFF) + i] = (uint8_t)((h & 15u) << 4);
			}
			((lds_row*)(lds + LDS_PERLIN_GRAD_OFF))[tid & 15u] = perlin_row(tid & 15u);      // (every thread writes: no per-lane branch)
			// the noise functions address these tables absolutely (see PERLIN_LDS); wave-uniform test, every thread stores the same word
			if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)lds != 0u) __hip_atomic_store((gu32*)&env.ls->pad, 2u, RLX_AGENT);
			perlin_ready = true;
			__syncthreads();
		}

		Ctx c;
		c.dp = dp; c.rel = rel; c.consts = (const uint32_t*)DESC_TABLE(c, consts); c.out_list = (const uint64_t*)DESC_TABLE(c, out_list);
		c.spill = lds_spill; c.tid = tid; c.wave = wave;
		c.inst0 = tile_first + tid * lane_stride<V>();
		c.num_instances = num_instances;
		c.tile_first = tile_first;
		c.valid = min(num_instances - tile_first, (uint32_t)(VVM_BLOCK_THREADS * V));
		c.phase = FULL ? started++ : phase;      // (tiles this workgroup has started: with fused passes more than one per ticket)
		c.reg_shift = (THREADED && V == 4 && (DESC_U32(dp, flags) & VVM_DESC_REGS_OF_4)) ? 2u : 3u;
		if (!(dbgw & 1u)) {
			if constexpr (THREADED)
				run_tile8<V, IS_LIGHT(V, SPILL), FULL, RIDE>((const void*)DESC_PTR(dp, prog), (gu64*)DESC_PTR(dp, status), last_tile, c, lds, tile_local, tag, env.ls, dbgw, pend, resolves, pre, have_pre && first_pass, file8);
			else
				run_tile<V, SPILL, FULL, RIDE>((const VvmIns*)DESC_PTR(dp, prog), (gu64*)DESC_PTR(dp, status), last_tile, c, lds, tile_local, tag, env.ls, dbgw, pend, resolves, pre, have_pre && first_pass);
		}
		first_pass = false;
		} while (FULL && ++e < e_end);
	}
	if (pend.active) {
		Ctx c;
		c.dp = nullptr; c.rel = 0; c.consts = nullptr; c.out_list = nullptr; c.spill = lds_spill;
		c.tid = tid; c.wave = wave; c.inst0 = 0; c.num_instances = 0; c.tile_first = 0; c.valid = 0; c.phase = 0; c.reg_shift = 3u;
		const u64 none[VVM_LB_WORDS_PER_LANE] = {};
		resolve_tile<V, HAS_TWO_CHAINS(V, SPILL, FULL)>(pend, c, lds_ctrl, resolves, tag, env.ls, dbgw, none, false);
		float ride[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) { if constexpr (IS_LIGHT(V, SPILL)) ride[k] = file8.f2[k]; else ride[k] = 0.0f; }
		store_tile<V, FULL, IS_LIGHT(V, SPILL), HAS_TWO_CHAINS(V, SPILL, FULL)>(pend, c, lds, dbgw, ride);
	}

	// the last workgroup to leave re-arms the launch words for the next launch
	__syncthreads();
	if (wave == 0) {
		if (dbgw & 128u)
			for (int i = 0; i < 8; ++i) {
				const rsrc_t rs = make_rsrc((uint64_t)&env.ls->stats[i], 8u);      // 64-bit add by the first lane
				(void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)lds_ctrl[LC_STATS + i], rs, (int)(lane == 0 ? 0u : 8u), 0, AUX_AGENT);
			}
		const uint32_t n = fetch_add_first_lane((gu32*)&env.ls->done, 1u, lane);
		if (n + 1 == gridDim.x) {      // (every lane of the wave stores the same values)
			if (!env.tag) {
				uint32_t next = (tag + 1u) & 0x3FFFFFFFu;
				if (next == 0) next = 1;
				__hip_atomic_store((gu32*)&env.ls->epoch, next, RLX_AGENT);
			}
			__hip_atomic_store((gu32*)&env.ls->done, 0u, RLX_AGENT);
			for (uint32_t x = 0; x < VVM_TICKET_CLASSES; ++x)
				__hip_atomic_store((gu32*)&env.ls->tickets[x * (VVM_TICKET_STRIDE_BYTES / 4u)], 0u, RLX_AGENT);
		}
	}
}

// The streams a script hands through unchanged, moved behind the Exec's own kernel (see VvmThroughArgs): workgroup (tile, b) moves
// pairs [P b, P b + P) of one tile of eight instances per lane, P = VVM_THROUGH_PAIRS_PER_BLOCK.  A tile whose instances all survive is copied as it lies, 128 bits
// per lane and half; otherwise every survivor's value goes to its rank in LDS (dead ones to the thread's sink word) and the tile's
// survivors leave from there, 128 bits per lane, cut at the tile's count by the descriptor's bound.  Survivors keep instance order
// (VectorVMRuntime.cpp:786-804) because the ranks are the acquireindex's own.
// SRC / DST: where pair i's stream addresses come from (the kernel's own arguments, or an Exec's block in device memory: scalar
// loads either way, the index is uniform).
// V = 8: the tile layout of the eight-wide kernels (two halves of 1024 instances, a lane holds four consecutive instances of each);
// V = 4: tiles of 1024 instances, four consecutive ones per lane.
template <int V, typename SRC, typename DST>
DEV void through_tile(const unsigned long long* status, const uint32_t* ride_words, uint32_t num_instances, uint32_t num_pairs, uint32_t tile, uint32_t p0,
                      uint32_t (&packed)[VVM_THROUGH_PAIRS_PER_BLOCK][VVM_BLOCK_THREADS * 8u + VVM_BLOCK_THREADS], SRC src, DST dst)
{
	constexpr uint32_t TILE = VVM_BLOCK_THREADS * (uint32_t)V, PPB = VVM_THROUGH_PAIRS_PER_BLOCK;
	const uint32_t tid = threadIdx.x;
	const uint32_t inclusive = (uint32_t)sload64(status, tile);
	const uint32_t base = tile ? (uint32_t)sload64(status, tile - 1u) : 0u;
	const uint32_t total = inclusive - base;
	if (total == 0 || p0 >= num_pairs) return;      // (uniform)
	const uint32_t first = tile * TILE;
	const uint32_t valid = min(num_instances - first, TILE);
	if constexpr (PPB == 1) {
		// a pair of 16-bit streams: element by element (the streams need not be aligned beyond their elements), every survivor
		// straight to its place, dead ones past the end of the descriptor
		const uint64_t d0 = dst(p0);
		if (d0 & VVM_THROUGH_IS16) {
			const rsrc_t rs = make_rsrc(src(p0) + 2ull * first, 2u * valid);
			const rsrc_t rd = make_rsrc((d0 & ~VVM_THROUGH_IS16) + 2ull * base, 2u * total);
			const uint32_t ride = (total == TILE) ? 0u : ride_words[(size_t)tile * VVM_BLOCK_THREADS + tid];
			uint16_t v[V];
#pragma unroll
			for (int k = 0; k < V; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)((tid * 4u + slot_offset<V>(k)) * 2u), 0, AUX_STREAM_LOAD);
			uint32_t r = (ride >> 8) & 0xFFFu;
#pragma unroll
			for (int k = 0; k < V; ++k) {
				if (k == 4) r = ride >> 20;
				const bool keep = (total == TILE) || ((ride >> k) & 1u);
				const uint32_t at = (total == TILE) ? tid * 4u + slot_offset<V>(k) : r;
				__builtin_amdgcn_raw_buffer_store_b16(v[k], rd, (int)(keep ? at * 2u : VVM_RSRC_NOWHERE), 0, AUX_STREAM_STORE);
				r += keep ? 1u : 0u;
			}
			return;
		}
	}
	u32x4 lo[PPB], hi[PPB];
#pragma unroll
	for (uint32_t j = 0; j < PPB; ++j) {
		const bool used = p0 + j < num_pairs;
		const rsrc_t rs = make_rsrc(src(used ? p0 + j : p0) + 4ull * first, used ? 4u * valid : 0u);
		lo[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tid * 16u), 0, AUX_STREAM_LOAD);
		if constexpr (V == 8) hi[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(VVM_HALF_TILE * 4u + tid * 16u), 0, AUX_STREAM_LOAD);
		else hi[j] = lo[j];
	}
	if (total == TILE) {
#pragma unroll
		for (uint32_t j = 0; j < PPB; ++j) {
			const bool used = p0 + j < num_pairs;
			const rsrc_t rd = make_rsrc(dst(used ? p0 + j : p0) + 4ull * base, used ? 4u * TILE : 0u);
			__builtin_amdgcn_raw_buffer_store_b128(lo[j], rd, (int)(tid * 16u), 0, AUX_STREAM_STORE);
			if constexpr (V == 8) __builtin_amdgcn_raw_buffer_store_b128(hi[j], rd, (int)(VVM_HALF_TILE * 4u + tid * 16u), 0, AUX_STREAM_STORE);
		}
		return;
	}
	const uint32_t ride = ride_words[(size_t)tile * VVM_BLOCK_THREADS + tid];
#pragma unroll
	for (uint32_t j = 0; j < PPB; ++j) {
		const uint32_t v[8] = {lo[j].x, lo[j].y, lo[j].z, lo[j].w, hi[j].x, hi[j].y, hi[j].z, hi[j].w};
		uint32_t r = (ride >> 8) & 0xFFFu;
#pragma unroll
		for (int k = 0; k < V; ++k) {
			if (k == 4) r = ride >> 20;
			const bool keep = (ride >> k) & 1u;
			packed[j][keep ? r : VVM_BLOCK_THREADS * 8u + tid] = v[k];
			r += keep ? 1u : 0u;
		}
	}
	__syncthreads();
#pragma unroll
	for (uint32_t j = 0; j < PPB; ++j) {
		const bool used = p0 + j < num_pairs;
		const rsrc_t rd = make_rsrc(dst(used ? p0 + j : p0) + 4ull * base, used ? 4u * total : 0u);
		const u32x4 x = *(const u32x4*)&packed[j][tid * 4u];
		__builtin_amdgcn_raw_buffer_store_b128(x, rd, (int)(tid * 16u), 0, AUX_STREAM_STORE);
		if constexpr (V == 8) {
			const u32x4 y = *(const u32x4*)&packed[j][VVM_HALF_TILE + tid * 4u];
			__builtin_amdgcn_raw_buffer_store_b128(y, rd, (int)(VVM_HALF_TILE * 4u + tid * 16u), 0, AUX_STREAM_STORE);
		}
	}
}
template <int V>
__global__ void __launch_bounds__(VVM_BLOCK_THREADS) vvm_through_kernel(const VvmThroughArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t packed[VVM_THROUGH_PAIRS_PER_BLOCK][VVM_BLOCK_THREADS * 8u + VVM_BLOCK_THREADS];
	through_tile<V>(a.status, a.ride, a.num_instances, a.num_pairs, blockIdx.x, blockIdx.y * VVM_THROUGH_PAIRS_PER_BLOCK, packed,
	             [&](uint32_t i) { return a.src[i]; }, [&](uint32_t i) { return a.dst[i]; });
}
// The same for the members of a batch: one block of VvmThroughArgs per Exec in device memory (pad = the number of its first tile
// within this launch), workgroup x takes tile x of the launch: the last Exec whose first tile is not beyond it.
__global__ void __launch_bounds__(VVM_BLOCK_THREADS) vvm_through_batch_kernel(const VvmThroughArgs* execs, uint32_t num_execs)
{
	__shared__ __attribute__((aligned(16))) uint32_t packed[VVM_THROUGH_PAIRS_PER_BLOCK][VVM_BLOCK_THREADS * 8u + VVM_BLOCK_THREADS];
	uint32_t lo = 0, hi = num_execs;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (sload32(execs + mid, offsetof(VvmThroughArgs, pad) / 4) <= blockIdx.x) lo = mid; else hi = mid;
	}
	const VvmThroughArgs* e = execs + lo;
	const uint32_t tile = blockIdx.x - sload32(e, offsetof(VvmThroughArgs, pad) / 4);
	if (tile >= sload32(e, offsetof(VvmThroughArgs, num_tiles) / 4)) return;
	through_tile<8>((const unsigned long long*)sload64(e, offsetof(VvmThroughArgs, status) / 8), (const uint32_t*)sload64(e, offsetof(VvmThroughArgs, ride) / 8),
	             sload32(e, offsetof(VvmThroughArgs, num_instances) / 4), sload32(e, offsetof(VvmThroughArgs, num_pairs) / 4), tile, blockIdx.y * VVM_THROUGH_PAIRS_PER_BLOCK, packed,
	             [&](uint32_t i) { return sload64(e, offsetof(VvmThroughArgs, src) / 8 + i); }, [&](uint32_t i) { return sload64(e, offsetof(VvmThroughArgs, dst) / 8 + i); });
}
extern "C" int vvm_launch_through_kernel(const VvmThroughArgs* args, hipStream_t stream)
{
	if (args->num_pairs == 0 || args->num_tiles == 0) return 0;
	const dim3 grid(args->num_tiles, (args->num_pairs + VVM_THROUGH_PAIRS_PER_BLOCK - 1u) / VVM_THROUGH_PAIRS_PER_BLOCK);
	if (args->instances_per_lane == 4) hipLaunchKernelGGL(vvm_through_kernel<4>, grid, dim3(VVM_BLOCK_THREADS), 0, stream, *args);
	else hipLaunchKernelGGL(vvm_through_kernel<8>, grid, dim3(VVM_BLOCK_THREADS), 0, stream, *args);
	return (int)hipGetLastError();
}
extern "C" int vvm_launch_through_batch_kernel(const VvmThroughArgs* device_execs, uint32_t num_execs, uint32_t total_tiles, uint32_t max_pairs, hipStream_t stream)
{
	if (num_execs == 0 || total_tiles == 0 || max_pairs == 0) return 0;
	const dim3 grid(total_tiles, (max_pairs + VVM_THROUGH_PAIRS_PER_BLOCK - 1u) / VVM_THROUGH_PAIRS_PER_BLOCK);
	hipLaunchKernelGGL(vvm_through_batch_kernel, grid, dim3(VVM_BLOCK_THREADS), 0, stream, device_execs, num_execs);
	return (int)hipGetLastError();
}

// update_id's MaxUsedID (VectorVMRuntime.cpp:1141-1174, 1383-1399), run after the Exec: per chunk of the reference's chunking,
// the maximum over the first NumOutput entries of the chunk's ID register, NumOutput = the chunk's survivors.  (Not the IDs of
// the survivors: the register is not compacted.  Reproduced as it is.)
__global__ void __launch_bounds__(256) vvm_max_id_kernel(const VvmIdDesc* idp)
{
	const VvmIdDesc d = *idp;
	const uint32_t total = (uint32_t)*d.survivors;
	int32_t m = -1;
	for (uint32_t ci = blockIdx.x; ci < d.num_chunks; ci += gridDim.x) {
		const uint32_t start = d.chunk_prefix[ci];
		const uint32_t end = (ci + 1u < d.num_chunks) ? d.chunk_prefix[ci + 1u] : total;
		const uint32_t first = ci * d.chunk_instances;
		for (uint32_t j = threadIdx.x; j < end - start; j += blockDim.x) m = max(m, d.id_scratch[first + j]);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
	if ((threadIdx.x & 63u) == 0 && m != -1) atomicMax(d.max_id, m);
}
extern "C" int vvm_launch_max_id_kernel(const VvmIdDesc* idp, uint32_t num_chunks, hipStream_t stream)
{
	const uint32_t grid = num_chunks < 1024u ? (num_chunks ? num_chunks : 1u) : 1024u;
	hipLaunchKernelGGL(vvm_max_id_kernel, dim3(grid), dim3(256), 0, stream, idp);
	return (int)hipGetLastError();
}

// ---- the reference's xorwow, laid out per draw and instance --------------------------------------------
// One step of the four generators (VectorVMRuntime.cpp:468-483); x[w][l] = lane l of State[w], ctr[l] = lane l of Counters.
DEV void xorwow_step(uint32_t (&x)[5][4], uint32_t (&ctr)[4], uint32_t (&out)[4])
{
#pragma unroll
	for (int l = 0; l < 4; ++l) {
		uint32_t t = x[4][l];
		const uint32_t s0 = x[0][l];
		x[4][l] = x[3][l]; x[3][l] = x[2][l]; x[2][l] = x[1][l]; x[1][l] = s0;
		t ^= t >> 2;
		t ^= t << 1;
		t ^= s0 ^ (s0 ^ (s0 << 4));      // as written at :478
		x[0][l] = t;
		ctr[l] += VVM_XORWOW_WEYL;
		out[l] = t + ctr[l];
	}
}
// Thread (s, i): random instruction i, loops [64 s, 64 s + 64) of the Exec.  Step number of (chunk c, instruction with first
// draw f and width w, loop l of the chunk, draw j of the loop) = D * c * chunk_loops + f * loops(c) + l * w + j, D = draws per
// loop of the whole script: the chunks before it ran every draw over all their loops, the instructions before it over this
// chunk's loops.
__global__ void __launch_bounds__(256) vvm_xorwow_fill_kernel(const VvmXorwowFill p, const uint32_t* __restrict__ jump, uint32_t* __restrict__ words)
{
	const uint32_t ins = blockIdx.y;
	const uint32_t first = p.first[ins], width = p.width[ins];
	uint32_t g = (blockIdx.x * 256u + threadIdx.x) * VVM_XORWOW_SEGMENT_LOOPS;
	const uint32_t g_end = min(g + VVM_XORWOW_SEGMENT_LOOPS, p.total_loops);
	while (g < g_end) {
		const uint32_t chunk = g / p.chunk_loops, l = g - chunk * p.chunk_loops;
		const uint32_t loops_here = min(p.chunk_loops, p.total_loops - chunk * p.chunk_loops);
		const uint64_t n = (uint64_t)p.num_draws * chunk * p.chunk_loops + (uint64_t)first * loops_here + (uint64_t)l * width;
		// the state after n steps
		uint32_t x[5][4], ctr[4];
#pragma unroll
		for (int w = 0; w < 5; ++w)
#pragma unroll
			for (int k = 0; k < 4; ++k) x[w][k] = p.state[4 * w + k];
#pragma unroll
		for (int k = 0; k < 4; ++k) ctr[k] = p.state[20 + k] + VVM_XORWOW_WEYL * (uint32_t)n;
		for (uint32_t b = 0; b < p.jump_powers; ++b) {
			if (!((n >> b) & 1u)) continue;
			const uint32_t* col = jump + (size_t)b * VVM_XORWOW_STATE_BITS * 5u;
			uint32_t y[5][4] = {};
#pragma unroll
			for (int w = 0; w < 5; ++w)
				for (uint32_t bit = 0; bit < 32u; ++bit) {
					const uint32_t* cj = col + ((uint32_t)w * 32u + bit) * 5u;
					const uint32_t c0 = cj[0], c1 = cj[1], c2 = cj[2], c3 = cj[3], c4 = cj[4];
#pragma unroll
					for (int k = 0; k < 4; ++k) {
						const uint32_t m = 0u - ((x[w][k] >> bit) & 1u);
						y[0][k] ^= c0 & m; y[1][k] ^= c1 & m; y[2][k] ^= c2 & m; y[3][k] ^= c3 & m; y[4][k] ^= c4 & m;
					}
				}
#pragma unroll
			for (int w = 0; w < 5; ++w)
#pragma unroll
				for (int k = 0; k < 4; ++k) x[w][k] = y[w][k];
		}
		// ... and on through the loops of this segment that lie in the chunk
		const uint32_t stop = min(g_end, (chunk + 1u) * p.chunk_loops);
		for (; g < stop; ++g)
			for (uint32_t j = 0; j < width; ++j) {
				uint32_t out[4];
				xorwow_step(x, ctr, out);
				uint4 v; v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
				*(uint4*)(words + (size_t)(first + j) * p.stride + (size_t)g * 4u) = v;
			}
	}
}
extern "C" int vvm_launch_xorwow_fill(const VvmXorwowFill* params, const uint32_t* jump, uint32_t* words, hipStream_t stream)
{
	if (!params || !jump || !words || params->num_instructions == 0 || params->num_instructions > 255u || params->chunk_loops == 0 ||
	    params->total_loops == 0 || params->jump_powers > VVM_XORWOW_JUMP_POWERS || (params->stride & 3u) || (uint64_t)params->total_loops * 4u > params->stride)
		return (int)hipErrorInvalidValue;
	const uint32_t segments = (params->total_loops + VVM_XORWOW_SEGMENT_LOOPS - 1u) / VVM_XORWOW_SEGMENT_LOOPS;
	hipLaunchKernelGGL(vvm_xorwow_fill_kernel, dim3((segments + 255u) / 256u, params->num_instructions), dim3(256), 0, stream, *params, jump, words);
	return (int)hipGetLastError();
}

// ---- host-callable launcher --------------------------------------------------------------------------
extern "C" int vvm_launch_exec_kernel(int instances_per_lane, const VvmExecDesc* descs, uint32_t num_execs, uint32_t total_tiles,
                                      VvmLaunchState* ls, int many_registers, uint32_t lds_spill_slots, int full, uint32_t grid, hipStream_t stream,
                                      const void* inline_block, uint32_t inline_bytes, uint32_t tag, int fused)
{
	// instances_per_lane 8 with many_registers set names the light eight-wide kernel (eight resident registers, six staging slots)
	const bool light = instances_per_lane == 8 && many_registers != 0;
	if (tag > VVM_MAX_LAUNCH_TAG) return (int)hipErrorInvalidValue;
	KernelEnv env = {};      // (the unused part of the block travels too: no indeterminate bytes in the kernel arguments)
	if (inline_block) {
		if (descs || num_execs != 1 || inline_bytes > VVM_INLINE_PARAM_BYTES) return (int)hipErrorInvalidValue;
		memcpy(env.block, inline_block, inline_bytes);
	} else if (!descs) {
		return (int)hipErrorInvalidValue;
	}
	if (fused && (!full || !descs || instances_per_lane == 8)) return (int)hipErrorInvalidValue;      // (the FULL kernels run fused passes)
	env.descs = descs; env.num_execs = num_execs; env.total_tiles = total_tiles; env.ls = ls; env.lds_spill_slots = lds_spill_slots; env.tag = tag; env.fused = fused ? 1u : 0u;
	env.run_shift = (total_tiles >= 4u * grid) ? 1u : 0u;
	{ const char* dbg = getenv("VVM_DEBUG_SKIP"); env.debug_skip = dbg ? (uint32_t)strtoul(dbg, nullptr, 0) : 0u; }
	const size_t lds_bytes = (instances_per_lane == 8 ? LDS_SPILL_OFF(8, full != 0, light) : instances_per_lane == 4 ? LDS_SPILL_OFF(4, full != 0, false) : LDS_SPILL_OFF(1, full != 0, false)) +
	                         (size_t)lds_spill_slots * VVM_BLOCK_THREADS * 4u * (size_t)instances_per_lane;
	const bool spill = many_registers != 0;      // the kernels with 32 resident registers and LDS register slots
#ifdef VVM_EXPERIMENT_ONE_KERNEL
	(void)spill; (void)full; hipLaunchKernelGGL((vvm_exec_kernel<VVM_EXPERIMENT_ONE_KERNEL>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env);
	return (int)hipGetLastError();
#else
#define VVM_LAUNCH(V, S, F) hipLaunchKernelGGL((vvm_exec_kernel<V, S, F>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env)
	if (instances_per_lane == 8) {
		if (full || lds_spill_slots) return (int)hipErrorInvalidValue;      // (eight instances per lane: plain scripts only)
		if (light) { if (env.debug_skip) hipLaunchKernelGGL((vvm_exec_kernel<8, true, false, true>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env); else VVM_LAUNCH(8, true, false); }
		else if (env.debug_skip) hipLaunchKernelGGL((vvm_exec_kernel<8, false, false, true>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env);
		else VVM_LAUNCH(8, false, false);
	} else if (instances_per_lane == 4) {
		if (full) {
			if (spill && full == 2) hipLaunchKernelGGL((vvm_exec_kernel<4, true, true, false, true>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env);      // (leaves ride words)
			else if (spill) VVM_LAUNCH(4, true, true);
			else if (full == 2) hipLaunchKernelGGL((vvm_exec_kernel<4, false, true, false, true>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env);      // (leaves ride words)
			else VVM_LAUNCH(4, false, true);
		}
		else if (spill) VVM_LAUNCH(4, true, false);
		else if (env.debug_skip) hipLaunchKernelGGL((vvm_exec_kernel<4, false, false, true>), dim3(grid), dim3(VVM_BLOCK_THREADS), lds_bytes, stream, env);
		else VVM_LAUNCH(4, false, false);
	} else {
		if (full) { if (spill) VVM_LAUNCH(1, true, true); else VVM_LAUNCH(1, false, true); }
		else      { if (spill) VVM_LAUNCH(1, true, false); else VVM_LAUNCH(1, false, false); }
	}
#undef VVM_LAUNCH
#endif
	return (int)hipGetLastError();
}

extern "C" int vvm_kernel_max_dynamic_lds(void)
{
	// raises the dynamic-LDS limit of the kernels that use more than the default 64 KB, on the current device
#ifndef VVM_EXPERIMENT_ONE_KERNEL
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<4, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<4, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<8, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<8, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	(void)hipFuncSetAttribute((const void*)vvm_exec_kernel<8, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
#endif
	return 160 * 1024 - 64;
}

extern "C" uint32_t vvm_kernel_resident_registers(int instances_per_lane, int many_registers)
{
	if (instances_per_lane == 1) return RegTraits<1, true>::RV;
	return many_registers ? RegTraits<4, true>::RV : RegTraits<4, false>::RV;
}
extern "C" uint32_t vvm_kernel_threaded_layout(void) { return VVM_T_LAYOUT; }
extern "C" uint32_t vvm_kernel_lds_fixed_bytes(int instances_per_lane, int full) { return instances_per_lane == 4 ? LDS_SPILL_OFF(4, full != 0, false) : LDS_SPILL_OFF(1, full != 0, false); }
