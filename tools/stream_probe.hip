// HBM ceiling probe, second form: a fixed 2 GiB read + 2 GiB written as NS streams in and NS out, with the knobs that could
// explain why sixteen concurrent streams copy slower than two:
//   wg/CU     workgroups resident per CU (capped through dynamic LDS)
//   CH        consecutive 4 KB pieces of every stream per trip
//   mode      0 copy, 1 read only (sum kept in a register, one store per workgroup), 2 write only
//   nt        0 default cache policy, 1 non-temporal loads and stores, 2 nt stores only, 3 nt loads only
//   map       0 trips in block order, 1 XCD-chunked: the blocks of one XCD (b % 8) walk one contiguous eighth of the array,
//             2 the same within every round of gridDim.x trips (what a kernel whose tiles must start in index order can do)
//   order     0 all loads then all stores, 1 stream by stream (load s, store s)
// Prints one line per configuration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { const f4* in[8]; f4* out[8]; };
template <int NT> __device__ __forceinline__ f4 ld(const f4* p) { if (NT == 1 || NT == 3) return __builtin_nontemporal_load(p); return *p; }
template <int NT> __device__ __forceinline__ void st(f4 v, f4* p) { if (NT == 1 || NT == 2) __builtin_nontemporal_store(v, p); else *p = v; }
template <int NS, int CH, int MODE, int NT, int ORDER>
__global__ void __launch_bounds__(256) probe(Ptrs p, size_t trips, int map)
{
	extern __shared__ unsigned char lds[];
	f4 acc = 0.0f;
	const size_t per_xcd = trips / 8;
	for (size_t t = blockIdx.x; t < trips; t += gridDim.x) {
		size_t trip = t;
		if (map == 1) trip = (t % 8) * per_xcd + t / 8;      // block b and b + 8 share an XCD: that XCD's blocks take one eighth
		if (map == 2) {                                      // per round of gridDim.x trips, the blocks of an XCD take a contiguous eighth of the round
			const size_t round = t / gridDim.x, b = t % gridDim.x;
			trip = round * gridDim.x + (b % 8) * (gridDim.x / 8) + b / 8;
			if (trip >= trips) trip = t;
		}
		const size_t i = trip * (size_t)(256 * CH) + threadIdx.x;
		if (ORDER == 0) {
			f4 v[NS][CH];
#pragma unroll
			for (int s = 0; s < NS; ++s)
#pragma unroll
				for (int k = 0; k < CH; ++k) v[s][k] = (MODE == 2) ? f4(1.0f) : ld<NT>(&p.in[s][i + k * 256]);
#pragma unroll
			for (int s = 0; s < NS; ++s)
#pragma unroll
				for (int k = 0; k < CH; ++k) { if (MODE == 1) acc += v[s][k]; else st<NT>(v[s][k] * 1.0001f, &p.out[s][i + k * 256]); }
		} else {
#pragma unroll
			for (int s = 0; s < NS; ++s) {
				f4 v[CH];
#pragma unroll
				for (int k = 0; k < CH; ++k) v[k] = ld<NT>(&p.in[s][i + k * 256]);
#pragma unroll
				for (int k = 0; k < CH; ++k) st<NT>(v[k] * 1.0001f, &p.out[s][i + k * 256]);
			}
		}
	}
	if (MODE == 1 && acc.x == 123.456f) p.out[0][threadIdx.x] = acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static char* base;
static hipEvent_t ea, eb;
static const size_t total = (size_t)2 << 30;
template <int NS, int CH, int MODE, int NT, int ORDER> void run(int wg_per_cu, int map, size_t stagger)
{
	const size_t bytes = total / NS, n4 = bytes / 16;
	Ptrs p;
	for (int s = 0; s < NS; ++s) { p.in[s] = (const f4*)(base + s * (bytes + stagger)); p.out[s] = (f4*)(base + (NS + s) * (bytes + stagger)); }
	const size_t trips = n4 / 256 / CH;
	const int grid = wg_per_cu > 0 ? 256 * wg_per_cu : (int)trips;
	const size_t lds = wg_per_cu > 0 ? (160 * 1024 / wg_per_cu - 512) & ~(size_t)255 : 0;
	auto k = probe<NS, CH, MODE, NT, ORDER>;
	if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, p, trips, map);
	CK(hipEventRecord(ea));
	for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, p, trips, map);
	CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
	float ms; CK(hipEventElapsedTime(&ms, ea, eb)); ms /= 10;
	const double moved = (MODE == 0 ? 2.0 : 1.0) * total;
	printf("NS %d CH %d mode %d nt %d order %d wg/CU %d map %d stagger %6zu: %.3f ms %5.0f GB/s\n", NS, CH, MODE, NT, ORDER, wg_per_cu, map, stagger, ms, moved / ms / 1e6);
	fflush(stdout);
}
int main()
{
	CK(hipMalloc((void**)&base, 2 * total + 32 * (1 << 20)));
	CK(hipMemset(base, 0, 2 * total + 32 * (1 << 20)));
	CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
	run<1, 1, 0, 1, 0>(0, 0, 0);
	for (int map : {0, 1, 2}) {
		run<8, 1, 0, 1, 0>(4, map, 0); run<8, 1, 0, 1, 0>(8, map, 0); run<8, 2, 0, 1, 0>(2, map, 0); run<8, 2, 0, 1, 0>(4, map, 0);
		run<8, 1, 0, 1, 1>(4, map, 0); run<8, 1, 0, 1, 1>(8, map, 0); run<8, 2, 0, 1, 1>(2, map, 0); run<8, 2, 0, 1, 1>(4, map, 0); run<8, 4, 0, 1, 1>(2, map, 0);
		run<8, 2, 0, 0, 1>(2, map, 0); run<8, 2, 0, 2, 1>(2, map, 0); run<8, 2, 0, 3, 1>(2, map, 0);
		run<8, 2, 1, 1, 0>(2, map, 0); run<8, 2, 2, 1, 0>(2, map, 0);
	}
	return 0;
}
