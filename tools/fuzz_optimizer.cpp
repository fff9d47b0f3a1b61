// Feeds vvm_optimize_script malformed bytecode (random bytes, and mutated or truncated copies of the scripts given on the
// command line) under the host sanitizers.  Build and run:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -x c++ \
//       renderer_amd/csrc/vvm_optimizer.cpp tools/fuzz_optimizer.cpp -o /tmp/fuzz_optimizer && /tmp/fuzz_optimizer script.bin ...
// (a script file: the bytes of renderer_amd.original.OriginalScript.done()).  Every call must return a status; none may trip a sanitizer.
#include "../include/vvm_abi.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main(int argc, char** argv) {
	std::mt19937 rng(7);
	int ok = 0, bad = 0;
	for (int f = 1; f < argc; ++f) {
		FILE* fp = fopen(argv[f], "rb"); std::vector<uint8_t> base(4096); base.resize(fread(base.data(), 1, 4096, fp)); fclose(fp);
		for (int it = 0; it < 60000; ++it) {
			std::vector<uint8_t> b = base;
			if (rng() % 10 < 3) { b.resize(2 + rng() % 80); for (auto& x : b) x = rng(); }
			else { int m = 1 + rng() % 6; for (int k = 0; k < m; ++k) b[rng() % b.size()] = rng(); if (rng() % 5 == 0) b.resize(2 + rng() % (b.size() - 2)); }
			vvm_ext_fn_io ext[2] = {{3, 3}, {3, 1}};
			uint8_t* blob = nullptr; size_t n = 0; uint32_t err = 0;
			int rc = vvm_optimize_script(b.data(), (int)b.size(), ext, 2, 0, 0, &blob, &n, &err);
			if (rc == 0) ++ok; else ++bad;
			vvm_free_blob(blob);
		}
	}
	printf("ok %d rejected %d\n", ok, bad);
}
