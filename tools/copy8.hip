// Practical HBM ceiling for the fountain access pattern: 8 float streams in, 8 float streams out, 128 bits per lane.
// Variants: plain / non-temporal stores / non-temporal loads and stores; one pass per workgroup or persistent workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { const f4* in[8]; f4* out[8]; };
template <int MODE>
__global__ void __launch_bounds__(256) copy8(Ptrs p, size_t n4)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		f4 v[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) v[s] = (MODE >= 2) ? __builtin_nontemporal_load(&p.in[s][i]) : p.in[s][i];
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const f4 x = v[s] * 1.0001f;
			if (MODE >= 1) __builtin_nontemporal_store(x, &p.out[s][i]); else p.out[s][i] = x;
		}
	}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE> int run(const Ptrs& p, size_t n)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const int grids[] = {256 * 4, 256 * 8, 65536};
	for (int g : grids) {
		for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(copy8<MODE>, dim3(g), dim3(256), 0, 0, p, n / 4);
		CK(hipEventRecord(a));
		for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(copy8<MODE>, dim3(g), dim3(256), 0, 0, p, n / 4);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
		printf("mode %d grid %6d: %.3f ms  %.0f GB/s\n", MODE, g, ms, 16.0 * n * 4 / ms / 1e6);
	}
	return 0;
}
int main()
{
	const size_t n = (size_t)64 << 20;
	Ptrs p;
	for (int s = 0; s < 8; ++s) { CK(hipMalloc((void**)&p.in[s], n * 4)); CK(hipMalloc((void**)&p.out[s], n * 4)); CK(hipMemset((void*)p.in[s], 0, n * 4)); }
	return run<0>(p, n) || run<1>(p, n) || run<2>(p, n);
}
