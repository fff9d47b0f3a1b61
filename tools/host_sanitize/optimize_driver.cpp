#include "vvm_abi.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv)
{
	int ok = 0, bad = 0;
	for (int i = 1; i < argc; ++i) {
		std::string base = argv[i];
		std::vector<uint8_t> bc = slurp(base), ex = slurp(base + ".ext");
		const vvm_ext_fn_io* io = ex.empty() ? nullptr : (const vvm_ext_fn_io*)ex.data();
		const int ne = (int)(ex.size() / sizeof(vvm_ext_fn_io));
		uint8_t* blob = nullptr; size_t n = 0; uint32_t err = 0;
		int rc = vvm_optimize_script(bc.data(), (int32_t)bc.size(), io, ne, 7, 0, &blob, &n, &err);
		if (rc == 0) ++ok; else ++bad;
		vvm_free_blob(blob);
		char* text = nullptr; size_t tn = 0;
		vvm_generate_human_readable_script(bc.data(), (int32_t)bc.size(), io, ne, 0, &text, &tn, &err);
		vvm_free_text(text);
	}
	printf("optimized %d, refused %d\n", ok, bad);
	return 0;
}
