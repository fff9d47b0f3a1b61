// host-only sanitizer driver: thaws and links every blob given on the command line in all link modes
#include "vvm_program.h"
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
int main(int argc, char** argv)
{
	int bad = 0;
	for (int i = 1; i < argc; ++i) {
		std::ifstream f(argv[i], std::ios::binary);
		std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
		vvm::ThawedScript s;
		std::string err;
		if (vvm::thaw_blob(b.data(), b.size(), s, err) != 0) { ++bad; continue; }
		for (int mode = 0; mode < 3; ++mode) {
			vvm::LinkedProgram lp;
			int rc = vvm::link_program(s, mode == 1, lp, err, 0, mode == 2);
			if (rc == 0)
				for (uint32_t k = 1; k < lp.num_passes; ++k) { vvm::LinkedProgram more; vvm::link_program(s, mode == 1, more, err, k, false, nullptr, lp.paired ? 1 : 0); }
			// the program without the outputs that hand a stream through (the lean form), and one with every other output only
			if (rc == 0 && mode != 1 && !lp.through_input.empty()) {
				std::vector<uint8_t> select(lp.through_input.size());
				for (size_t o = 0; o < select.size(); ++o) select[o] = lp.through_input[o] < 0;
				vvm::LinkedProgram lean;
				vvm::link_program(s, false, lean, err, 0, mode == 2, &select);
				for (size_t o = 0; o < select.size(); ++o) select[o] = (o & 1u) != 0;
				vvm::LinkedProgram some;
				vvm::link_program(s, false, some, err, 0, mode == 2, &select);
			}
		}
	}
	printf("done, %d blobs did not thaw\n", bad);
	return 0;
}
