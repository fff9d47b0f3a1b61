// Does it pay to have the whole chip on ONE stream at a time?  Eight streams in, eight out, 2 GiB each way, 8 KB per stream and
// trip (the interpreter's tile), persistent workgroups that all reside.  Every trip has sixteen phases (eight of loads, eight of
// stores) and the workgroups pass a grid-wide barrier between phases, so the chip moves in step:
//   mode 0  stream-major: phase s loads (stores) the trip's 8 KB of stream s only -- all resident workgroups together read one
//           contiguous run of one stream (grid x 8 KB), then the next stream's;
//   mode 1  mixed: phase j loads (stores) the j-th eighth of the trip's piece of EVERY stream -- same barriers, same bytes per
//           phase and thread, but all sixteen streams stay in flight as they do in the interpreter;
//   mode 2  no barriers, all loads then all stores (what tools/stream_probe.hip measures), for the barriers' own cost;
//   mode 3  no barriers, one stream in flight per workgroup at a time (a wait behind every stream's loads and stores).
// mode 0 against mode 1 is the effect of the ordering alone.  The barrier gives up after a bounded number of polls.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { const f4* in[8]; f4* out[8]; };
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target)
{
	__syncthreads();
	bool ok = true;
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		unsigned polls = 0;
		while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
			__builtin_amdgcn_s_sleep(1);
			if (++polls > (1u << 22)) { ok = false; break; }
		}
	}
	__syncthreads();
	return ok;
}
template <int MODE>
__global__ void __launch_bounds__(256) probe(Ptrs p, unsigned trips, unsigned* counter, unsigned* failed)
{
	extern __shared__ unsigned char lds[];
	unsigned phase = 0;
	const unsigned rounds = trips / gridDim.x;      // (the host makes trips a multiple of the grid)
	for (unsigned r = 0; r < rounds; ++r) {
		const size_t i = ((size_t)r * gridDim.x + blockIdx.x) * 512;      // first f4 of the trip's piece of every stream
		f4 v[8][2];
		if (MODE == 2) {
#pragma unroll
			for (int s = 0; s < 8; ++s) { v[s][0] = __builtin_nontemporal_load(&p.in[s][i + threadIdx.x]); v[s][1] = __builtin_nontemporal_load(&p.in[s][i + 256 + threadIdx.x]); }
#pragma unroll
			for (int s = 0; s < 8; ++s) { __builtin_nontemporal_store(v[s][0] * 1.0001f, &p.out[s][i + threadIdx.x]); __builtin_nontemporal_store(v[s][1] * 1.0001f, &p.out[s][i + 256 + threadIdx.x]); }
			continue;
		}
		if (MODE == 3) {      // no barriers, but a workgroup has one stream in flight at a time: the workgroups start together and do the same work
#pragma unroll
			for (int s = 0; s < 8; ++s) { v[s][0] = __builtin_nontemporal_load(&p.in[s][i + threadIdx.x]); v[s][1] = __builtin_nontemporal_load(&p.in[s][i + 256 + threadIdx.x]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
			for (int s = 0; s < 8; ++s) { __builtin_nontemporal_store(v[s][0] * 1.0001f, &p.out[s][i + threadIdx.x]); __builtin_nontemporal_store(v[s][1] * 1.0001f, &p.out[s][i + 256 + threadIdx.x]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
			continue;
		}
		// mixed: in phase j thread t handles f4 (t % 64) + 64 j of streams t / 64 and t / 64 + 4
		const unsigned ms = threadIdx.x >> 6, ml = threadIdx.x & 63u;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			if (!grid_barrier(counter, ++phase * gridDim.x)) { *failed = 1; return; }
			if (MODE == 0) { v[j][0] = __builtin_nontemporal_load(&p.in[j][i + threadIdx.x]); v[j][1] = __builtin_nontemporal_load(&p.in[j][i + 256 + threadIdx.x]); }
			else {
#pragma unroll
				for (int h = 0; h < 2; ++h) {      // (the stream index is not a compile-time constant here: pick the pointer by selects)
					const unsigned s = ms + 4u * h;
					const f4* q = p.in[0];
#pragma unroll
					for (int c = 1; c < 8; ++c) q = (s == (unsigned)c) ? p.in[c] : q;
					v[j][h] = __builtin_nontemporal_load(&q[i + ml + 64u * j]);
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			if (!grid_barrier(counter, ++phase * gridDim.x)) { *failed = 1; return; }
			if (MODE == 0) { __builtin_nontemporal_store(v[j][0] * 1.0001f, &p.out[j][i + threadIdx.x]); __builtin_nontemporal_store(v[j][1] * 1.0001f, &p.out[j][i + 256 + threadIdx.x]); }
			else {
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const unsigned s = ms + 4u * h;
					f4* q = p.out[0];
#pragma unroll
					for (int c = 1; c < 8; ++c) q = (s == (unsigned)c) ? p.out[c] : q;
					__builtin_nontemporal_store(v[j][h] * 1.0001f, &q[i + ml + 64u * j]);
				}
			}
		}
	}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static char* base;
static unsigned* words;
static hipEvent_t ea, eb;
static const size_t total = (size_t)2 << 30;
template <int MODE> void run(int wg_per_cu)
{
	const size_t bytes = total / 8;
	Ptrs p;
	for (int s = 0; s < 8; ++s) { p.in[s] = (const f4*)(base + s * bytes); p.out[s] = (f4*)(base + (8 + s) * bytes); }
	const int grid = 256 * wg_per_cu;
	const unsigned trips = (unsigned)(bytes / 8192 / grid * grid);
	const size_t lds = (size_t)(160 * 1024 / wg_per_cu) / 1280 * 1280;      // (LDS is handed out in units of 1280 bytes: anything above a multiple costs a whole workgroup per CU)
	auto k = probe<MODE>;
	if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	int resident = 0;
	CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k, 256, lds));
	if (resident < wg_per_cu) { printf("mode %d: only %d workgroups per CU would reside, %d asked for: skipped\n", MODE, resident, wg_per_cu); return; }
	float best = 1e30f;
	for (int rep = 0; rep < 6; ++rep) {
		CK(hipMemset(words, 0, 8));
		CK(hipEventRecord(ea));
		hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, p, trips, words, words + 1);
		CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
		unsigned h[2]; CK(hipMemcpy(h, words, 8, hipMemcpyDeviceToHost));
		if (h[1]) { printf("mode %d: a barrier gave up\n", MODE); exit(1); }
		float ms; CK(hipEventElapsedTime(&ms, ea, eb));
		if (rep && ms < best) best = ms;
	}
	const double moved = 2.0 * trips * 8192.0 * 8;
	printf("mode %d wg/CU %d: %.3f ms %5.0f GB/s (%u rounds of %d trips)\n", MODE, wg_per_cu, best, moved / best / 1e6, trips / grid, grid);
	fflush(stdout);
}
int main()
{
	CK(hipMalloc((void**)&base, 2 * total));
	CK(hipMemset(base, 0, 2 * total));
	CK(hipMalloc((void**)&words, 8));
	CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
	for (int w : {3, 4}) { run<2>(w); run<1>(w); run<0>(w); }
	for (int w : {3, 4, 6, 8}) { run<2>(w); run<3>(w); }
	return 0;
}
