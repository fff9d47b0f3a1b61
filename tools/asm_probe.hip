// asm_probe.hip -- feasibility probe for a threaded-code interpreter core written in gfx950 assembly.
//
// What it tries: the per-lane register file in fixed VGPRs, laid out register-major (the eight instance slots of virtual register i
// are the eight consecutive VGPRs FILE + 8 i ..), so that one GPR-index setting reaches a whole register as four 64-bit pairs and
// packed fp32 instructions work on it; instructions of 32 bytes whose second half holds what the assembly needs ready-made (byte
// offsets of an operand-fetch routine and an operation routine, operand bytes pre-multiplied); dispatch by s_setpc_b64 into routines
// of a fixed size, no compare chains.  It runs a program of dependent mad / mul / add instructions ITERS times per workgroup, with no
// memory traffic, checks the result against the same arithmetic on the host, and prints the time per interpreted instruction.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/asm_probe.hip -o tools/asm_probe && tools/asm_probe [ops] [iters]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float vec32f __attribute__((ext_vector_type(32)));
typedef float vec16f __attribute__((ext_vector_type(16)));

#define NREG 10                 // virtual registers (the light kernel's file)
#define SLOT_BYTES 128          // size of one routine slot
#define TABLE_SLOTS 32          // power of two


struct Ins {
	uint32_t w[8];              // w[4]: fetch routine offset | operation routine offset << 16;  w[5]: operand bytes a | b << 8 | c << 16 | d << 24;
	                            // w[6]: e | dst << 8 (register operands: index * 8; uniform operands: lane of the uniform register)
};

#define STR2(x) #x
#define STR(x) STR2(x)
// VGPR / SGPR assignments inside the assembly.  The file: v40..v119; one group of eight temporaries.
// SGPRs: two instruction buffers X = s[64:71], Y = s[72:79] (an instruction executes from one while the next arrives in the other:
// every routine exists twice, once per buffer, so nothing is copied), table bases s[80:81] (X routines) and s[82:83] (Y routines),
// jump target s[84:85], a uniform operand as a pair s[86:87].
#define FB "40"
#define T "120"
#define X "64"
#define Y "72"
// word k of the instruction in buffer B
#define W(B, k) "s[" B "+" #k "]"
// the end of every routine: wait for the next instruction (in flight into buffer N), start the load of the one after it into buffer C (this
// routine's own, which it is done with), jump to the next instruction's routine in the table of N's routines (base register pair NB)
#define NEXT(C, N, NB) \
	"s_waitcnt lgkmcnt(0)\n s_load_dwordx8 s[" C ":" C "+7], %[prog], %[pc]\n s_add_u32 %[pc], %[pc], 32\n" \
	"s_add_u32 s84, s[" NB "], " W(N, 4) "\n s_addc_u32 s85, s[" NB "+1], 0\n s_setpc_b64 s[84:85]\n"
#define MOV4(D, S) \
	"v_mov_b64 v[" D "+0:" D "+1], v[" S "+0:" S "+1]\n v_mov_b64 v[" D "+2:" D "+3], v[" S "+2:" S "+3]\n" \
	"v_mov_b64 v[" D "+4:" D "+5], v[" S "+4:" S "+5]\n v_mov_b64 v[" D "+6:" D "+7], v[" S "+6:" S "+7]\n"
#define PK4(op, D, A, B) \
	op " v[" D "+0:" D "+1], v[" A "+0:" A "+1], " B(0) "\n" op " v[" D "+2:" D "+3], v[" A "+2:" A "+3], " B(2) "\n" \
	op " v[" D "+4:" D "+5], v[" A "+4:" A "+5], " B(4) "\n" op " v[" D "+6:" D "+7], v[" A "+6:" A "+7], " B(6) "\n"
#define VT(k) "v[" T "+" #k ":" T "+" #k "+1]"
#define VF(k) "v[" FB "+" #k ":" FB "+" #k "+1]"
#define SU(k) "s[86:87]"
// word 5: a | c << 8 | d << 16 | e << 24;  word 6: b;  word 7: dst  (register operands: index * 8; uniform operands: lane of the uniform register)
#define UNIFORM(B, k, shift) "s_lshr_b32 s88, " W(B, k) ", " #shift "\n v_readlane_b32 s86, %[uni], s88\n s_mov_b32 s87, s86\n s_nop 0\n"
// the routines of one buffer C (next instruction in N, whose routines start at the address in register pair NB)
#define ROUTINES(C, N, NB, EXIT) \
	/* 0 exit */ "s_branch " EXIT "\n" \
	".p2align 7\n" /* 1 mul r r -> anywhere */ \
		"s_set_gpr_idx_on " W(C, 5) ", gpr_idx(SRC0)\n" MOV4(T, FB) "s_set_gpr_idx_idx " W(C, 6) "\n" PK4("v_pk_mul_f32", T, FB, VT) \
		"s_set_gpr_idx_on " W(C, 7) ", gpr_idx(DST)\n" MOV4(FB, T) "s_set_gpr_idx_off\n" NEXT(C, N, NB) \
	".p2align 7\n" /* 2 mul r r -> a */ \
		"s_set_gpr_idx_on " W(C, 6) ", gpr_idx(SRC0)\n" MOV4(T, FB) "s_set_gpr_idx_on " W(C, 7) ", gpr_idx(SRC0,DST)\n" PK4("v_pk_mul_f32", FB, FB, VT) "s_set_gpr_idx_off\n" NEXT(C, N, NB) \
	".p2align 7\n" /* 3 add r u -> a */ \
		UNIFORM(C, 6, 0) "s_set_gpr_idx_on " W(C, 7) ", gpr_idx(SRC0,DST)\n" PK4("v_pk_add_f32", FB, FB, SU) "s_set_gpr_idx_off\n" NEXT(C, N, NB) \
	".p2align 7\n" /* 4 mad r u r -> a : (a * u) + c */ \
		UNIFORM(C, 6, 0) "s_set_gpr_idx_on " W(C, 5) ", gpr_idx(SRC0)\n" PK4("v_pk_mul_f32", T, FB, SU) "s_lshr_b32 s88, " W(C, 5) ", 8\n s_set_gpr_idx_idx s88\n" \
		PK4("v_pk_add_f32", T, FB, VT) "s_set_gpr_idx_on " W(C, 7) ", gpr_idx(DST)\n" MOV4(FB, T) "s_set_gpr_idx_off\n" NEXT(C, N, NB) \
	".p2align 7\n s_branch " EXIT "\n .p2align 7\n s_branch " EXIT "\n .p2align 7\n s_branch " EXIT "\n .p2align 7\n"

enum { R_EXIT = 0, H_MUL_RR = 1, H_MUL_RR_A = 2, H_ADD_RU_A = 3, H_MAD_RUR = 4 };

__device__ __forceinline__ void interpret(const Ins* prog, uint32_t& pc, uint32_t uni, vec32f& F0, vec32f& F1, vec16f& F2, uint32_t& x0, uint32_t& x1)
{
	asm volatile(
		"s_getpc_b64 s[80:81]\n"
		"1:\n"
		"s_add_u32 s82, s80, 8f - 1b\n s_addc_u32 s83, s81, 0\n"
		"s_add_u32 s80, s80, 7f - 1b\n s_addc_u32 s81, s81, 0\n"
		"s_load_dwordx8 s[" X ":" X "+7], %[prog], %[pc]\n s_add_u32 %[pc], %[pc], 32\n"
		"s_load_dwordx8 s[" Y ":" Y "+7], %[prog], %[pc]\n s_add_u32 %[pc], %[pc], 32\n"
		"s_waitcnt lgkmcnt(0)\n"
		"s_add_u32 s84, s80, " W(X, 4) "\n s_addc_u32 s85, s81, 0\n s_setpc_b64 s[84:85]\n"
		".p2align 7\n"
		"7:\n" ROUTINES(X, Y, "82", "5f")
		"8:\n" ROUTINES(Y, X, "80", "6f")
		"5:\n s_mov_b32 %[x0], " W(X, 0) "\n s_mov_b32 %[x1], " W(X, 1) "\n s_branch 4f\n"
		"6:\n s_mov_b32 %[x0], " W(Y, 0) "\n s_mov_b32 %[x1], " W(Y, 1) "\n"
		"4:\n s_waitcnt lgkmcnt(0)\n s_sub_u32 %[pc], %[pc], 32\n"
		: [pc] "+s"(pc), [x0] "=s"(x0), [x1] "=s"(x1), "+{v[40:71]}"(F0), "+{v[72:103]}"(F1), "+{v[104:119]}"(F2)
		: [prog] "s"(prog), [uni] "v"(uni)
		: "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",
		  "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85",
		  "s86", "s87", "s88", "scc", "memory");
}

__global__ void __launch_bounds__(256, 3) probe_kernel(const Ins* prog, uint32_t iters, const float* consts, float* out)
{
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	uint32_t uni = __float_as_uint(consts[lane & 7u]);       // lane i of `uni` holds constant i (i < 8)
	vec32f F0, F1;
	vec16f F2;
	for (int i = 0; i < 32; ++i) { F0[i] = 1.0f + 0.001f * (float)((tid * 32u + i) % 977u); F1[i] = 0.5f + 0.002f * (float)((tid * 32u + i) % 313u); }
	for (int i = 0; i < 16; ++i) F2[i] = 2.0f;
	uint32_t x0 = 0, x1 = 0;
	for (uint32_t it = 0; it < iters; ++it) {
		uint32_t pc = 0;
		interpret(prog, pc, uni, F0, F1, F2, x0, x1);
	}
	float* o = out + ((size_t)blockIdx.x * 256u + tid) * 80u;
	for (int i = 0; i < 32; ++i) { o[i] = F0[i]; o[32 + i] = F1[i]; }
	for (int i = 0; i < 16; ++i) o[64 + i] = F2[i];
	if (tid == 0 && blockIdx.x == 0) { out[(size_t)gridDim.x * 256u * 80u] = __uint_as_float(x0); }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
	const int n_ops = argc > 1 ? atoi(argv[1]) : 64;
	const uint32_t iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 40;
	const float consts[8] = { 0.0f, 1.0009765625f, 0.125f, -0.5f, 2.0f, 0.75f, 1.5f, -1.0f };
	// the dependent chain of the arithmetic benchmark: mad(a, c1, b), mul(a, b), add(a, c2) over four running values (registers 0..3)
	std::vector<Ins> prog;
	auto emit = [&](uint32_t routine, uint32_t a, uint32_t b, uint32_t c, uint32_t dst) {
		Ins in; memset(&in, 0, sizeof in);
		in.w[4] = routine * SLOT_BYTES;
		in.w[5] = a | c << 8;
		in.w[6] = b;
		in.w[7] = dst;
		prog.push_back(in);
	};
	const bool in_place = !(argc > 3 && atoi(argv[3]) == 0);
	for (int k = 0; k < n_ops; ++k) {
		const uint32_t a = (k % 4) * 8, b = ((k + 1) % 4) * 8;
		if (k % 3 == 0) emit(H_MAD_RUR, a, 1, b, a);
		else if (k % 3 == 1) emit(in_place ? H_MUL_RR_A : H_MUL_RR, a, b, 0, a);
		else emit(H_ADD_RU_A, a, 2, 0, a);
	}
	{ Ins done; memset(&done, 0, sizeof done); done.w[0] = 0xD0D0D0D0u; prog.push_back(done); prog.push_back(done); prog.push_back(done); }

	int dev_cus = 256;
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	dev_cus = prop.multiProcessorCount;
	const uint32_t grid = (uint32_t)dev_cus * 3u;
	Ins* d_prog; float* d_consts; float* d_out;
	const size_t out_floats = (size_t)grid * 256u * 80u + 1;
	CHECK(hipMalloc(&d_prog, prog.size() * sizeof(Ins)));
	CHECK(hipMalloc(&d_consts, sizeof consts));
	CHECK(hipMalloc(&d_out, out_floats * 4));
	CHECK(hipMemcpy(d_prog, prog.data(), prog.size() * sizeof(Ins), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_consts, consts, sizeof consts, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), 0, 0, d_prog, 1u, d_consts, d_out);      // one iteration: the check
	CHECK(hipDeviceSynchronize());
	std::vector<float> out(out_floats);
	CHECK(hipMemcpy(out.data(), d_out, out_floats * 4, hipMemcpyDeviceToHost));
	// host reference for workgroup 0 and the last one
	size_t bad = 0;
	for (uint32_t wg : { 0u, grid - 1 })
		for (uint32_t tid = 0; tid < 256; ++tid) {
			float f[80];
			for (int i = 0; i < 32; ++i) { f[i] = 1.0f + 0.001f * (float)((tid * 32u + i) % 977u); f[32 + i] = 0.5f + 0.002f * (float)((tid * 32u + i) % 313u); }
			for (int i = 0; i < 16; ++i) f[64 + i] = 2.0f;
			for (int k = 0; k < n_ops; ++k) {
				const int a = (k % 4) * 8, b = ((k + 1) % 4) * 8;
				for (int s = 0; s < 8; ++s) {
					volatile float t;
					if (k % 3 == 0) { t = f[a + s] * consts[1]; t = t + f[b + s]; }
					else if (k % 3 == 1) t = f[a + s] * f[b + s];
					else t = f[a + s] + consts[2];
					f[a + s] = t;
				}
			}
			const float* o = out.data() + ((size_t)wg * 256u + tid) * 80u;
			for (int i = 0; i < 80; ++i) if (memcmp(&o[i], &f[i], 4) != 0) { if (bad < 5) printf("mismatch wg %u tid %u element %d: got %g want %g\n", wg, tid, i, o[i], f[i]); ++bad; }
		}
	uint32_t x0; memcpy(&x0, &out[out_floats - 1], 4);
	printf("check: %zu mismatches; the instruction the assembly left at: %08x\n", bad, x0);
	if (bad) return 1;
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), 0, 0, d_prog, iters, d_consts, d_out);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipDeviceSynchronize());
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		// one "tile" = one pass of a workgroup over the program with eight instances per lane (2048 instances)
		printf("%d ops x %u passes on %u workgroups: %.3f ms = %.1f ns per interpreted instruction and workgroup pass\n", n_ops, iters, grid, ms, ms * 1e6 / ((double)iters * n_ops));
	}
	return 0;
}
