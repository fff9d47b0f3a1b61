// HBM ceiling probe: how the copy rate of a fixed 4.29 GB of traffic depends on the number of concurrent streams (NS in, NS
// out), on the spacing of their base addresses and on the launch shape (one 4 KB trip per workgroup, or persistent workgroups
// striding over the array, or persistent workgroups that each take CH consecutive 4 KB pieces of every stream).
// All streams are carved from one allocation; stream s starts at s * (bytes per stream + stagger).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { const f4* in[8]; f4* out[8]; };
template <int NS, int CH>
__global__ void __launch_bounds__(256) copyn(Ptrs p, size_t n4)
{
	for (size_t i = (blockIdx.x * (size_t)blockDim.x) * CH + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x * CH) {
		f4 v[NS][CH];
#pragma unroll
		for (int s = 0; s < NS; ++s)
#pragma unroll
			for (int k = 0; k < CH; ++k) v[s][k] = __builtin_nontemporal_load(&p.in[s][i + k * 256]);
#pragma unroll
		for (int s = 0; s < NS; ++s)
#pragma unroll
			for (int k = 0; k < CH; ++k) __builtin_nontemporal_store(v[s][k] * 1.0001f, &p.out[s][i + k * 256]);
	}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static char* base;
static hipEvent_t ea, eb;
static const size_t total = (size_t)2 << 30;	// bytes read (and as many written)
template <int NS, int CH> void run(size_t stagger, int grid)
{
	const size_t bytes = total / NS, n4 = bytes / 16;
	Ptrs p;
	for (int s = 0; s < NS; ++s) { p.in[s] = (const f4*)(base + s * (bytes + stagger)); p.out[s] = (f4*)(base + (NS + s) * (bytes + stagger)); }
	const int g = grid > 0 ? grid : (int)(n4 / 256 / CH);
	for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((copyn<NS, CH>), dim3(g), dim3(256), 0, 0, p, n4);
	CK(hipEventRecord(ea));
	for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((copyn<NS, CH>), dim3(g), dim3(256), 0, 0, p, n4);
	CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
	float ms; CK(hipEventElapsedTime(&ms, ea, eb)); ms /= 10;
	printf("%d+%d streams, %d x 4 KB per stream and trip, stagger %7zu, grid %6d%s: %.3f ms  %.0f GB/s\n", NS, NS, CH, stagger, g,
		grid > 0 ? " (persistent)" : " (one trip)  ", ms, 2.0 * total / ms / 1e6);
}
int main(int argc, char**)
{
	const size_t maxStagger = 1 << 20;
	CK(hipMalloc((void**)&base, 2 * total + 16 * maxStagger));
	CK(hipMemset(base, 0, 2 * total + 16 * maxStagger));
	CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
	const bool wide = argc > 1;	// any argument: the larger pieces per trip only
	const size_t st[] = {0, 4352, 36864};
	if (wide) {
		for (size_t s : st) {
			run<8, 2>(s, 0); run<8, 4>(s, 0); run<8, 2>(s, 512); run<8, 4>(s, 256); run<8, 4>(s, 512); run<8, 4>(s, 768); run<8, 4>(s, 1024);
			run<4, 4>(s, 0); run<4, 8>(s, 0); run<4, 8>(s, 512);
		}
		return 0;
	}
	for (size_t s : st) {
		run<1, 1>(s, 0); run<2, 1>(s, 0); run<4, 1>(s, 0); run<8, 1>(s, 0);
		run<8, 2>(s, 0); run<8, 1>(s, 1024); run<8, 2>(s, 512); run<8, 2>(s, 1024);
		run<1, 8>(s, 0); run<1, 8>(s, 512); run<1, 8>(s, 1024); run<2, 4>(s, 0); run<2, 8>(s, 512);
	}
	const size_t more[] = {256, 512, 1024, 2048, 8192, 8192 + 256, 16384, 16384 + 4096, 32768, 65536, 131072 + 4096, 262144 + 36864, 524288 + 36864};
	for (size_t s : more) { run<8, 1>(s, 0); run<8, 2>(s, 512); }
	return 0;
}
