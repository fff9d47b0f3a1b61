// Evaluates sin / cos / exp / log of the "mathfun" family on the float32 bit patterns it reads from stdin and writes the results'
// bit patterns to stdout (four arrays, one per function, each as long as the input).  The functions come from a header that ships
// with this container's PyTorch (ATen/native/cpu/avx_mathfun.h: G. Garberoglio's AVX port of J. Pommier's sse_mathfun.h, zlib
// licence): the eight-lane twin of the four-lane header the reference's VectorSin / VectorCos / VectorExp / VectorLog call
// (Core/Private/Math/UnrealMathSSE.cpp:8-38), which is not part of the reference tree.  Same constants, same operation order,
// no fused multiply-add (explicit intrinsics).  Built and run by make_mathfun_fixtures.py only; tests read the JSON it writes.
#define CPU_CAPABILITY_AVX2 1
#include <immintrin.h>
#include <ATen/native/cpu/avx_mathfun.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

int main()
{
	std::vector<uint32_t> in;
	uint32_t w;
	while (fread(&w, 4, 1, stdin) == 1) in.push_back(w);
	const size_t n = in.size();
	const size_t padded = (n + 7) & ~(size_t)7;
	in.resize(padded, 0x3f800000u);
	std::vector<uint32_t> out[4];
	for (auto& o : out) o.resize(padded);
	for (size_t i = 0; i < padded; i += 8) {
		__m256 x = _mm256_loadu_ps(reinterpret_cast<const float*>(&in[i]));
		_mm256_storeu_ps(reinterpret_cast<float*>(&out[0][i]), sin256_ps(x));
		_mm256_storeu_ps(reinterpret_cast<float*>(&out[1][i]), cos256_ps(x));
		_mm256_storeu_ps(reinterpret_cast<float*>(&out[2][i]), exp256_ps(x));
		_mm256_storeu_ps(reinterpret_cast<float*>(&out[3][i]), log256_ps(x));
	}
	for (auto& o : out) fwrite(o.data(), 4, n, stdout);
	return 0;
}
