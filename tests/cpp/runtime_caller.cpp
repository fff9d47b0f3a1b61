// runtime_caller.cpp -- a C++ caller of the executor written against the reference's interface names
// (VectorVM::Runtime::AllocVectorVMState / ExecVectorVMState / FVectorVMExecContext / FDataSetMeta, include/vvm_runtime.hpp), the
// way a Niagara-side harness would call the reference (VectorVMRuntime.h:10-33).  No Python, no HIP headers: device memory comes
// from the library's own helpers.  tests/test_gpu_cpp_caller.py writes the case files, runs this program and compares what it
// leaves with the CPU oracle.
//
//   runtime_caller <dir>      reads  <dir>/blob.bin  <dir>/consts.bin  <dir>/case.txt ("instances offset inputs outputs" then the
//                                    byte sizes of the constant tables consts.bin is cut into)  <dir>/in_<k>.f32
//                             writes <dir>/out_<k>.f32 (offset + instances elements each)  <dir>/result.txt ("survivors status")
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "vvm_runtime.hpp"

using namespace VectorVM::Runtime;

static std::vector<uint8> read_file(const std::string& path)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
	return std::vector<uint8>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
	if (argc != 2) { std::fprintf(stderr, "usage: runtime_caller <dir>\n"); return 2; }
	const std::string dir = argv[1];
	int instances = 0, offset = 0, num_in = 0, num_out = 0;
	std::vector<int> table_bytes;
	{
		std::ifstream f(dir + "/case.txt");
		f >> instances >> offset >> num_in >> num_out;
		for (int b; f >> b;) table_bytes.push_back(b);
		if (!f.eof() || num_in < 0 || num_out < 0 || instances < 0 || offset < 0) { std::fprintf(stderr, "bad case.txt\n"); return 2; }
	}
	// an empty blob gives no state (VectorVMRuntime.cpp:2313-2316)
	if (AllocVectorVMState(TConstArrayView<uint8>()) != nullptr) { std::fprintf(stderr, "a state from an empty blob\n"); return 1; }

	const std::vector<uint8> blob = read_file(dir + "/blob.bin");
	const std::vector<uint8> consts = read_file(dir + "/consts.bin");
	FVectorVMState* State = AllocVectorVMState(TConstArrayView<uint8>(blob.data(), (int32)blob.size()));
	if (!State) { std::fprintf(stderr, "AllocVectorVMState: %s\n", vvm_last_error()); return 1; }

	const size_t elements = (size_t)offset + (size_t)instances;
	const size_t bytes = elements * sizeof(float) + 64;
	std::vector<const uint8*> in_streams;
	std::vector<uint8*> out_streams;
	for (int k = 0; k < num_in; ++k) {
		const std::vector<uint8> host = read_file(dir + "/in_" + std::to_string(k) + ".f32");
		if (host.size() != elements * sizeof(float)) { std::fprintf(stderr, "in_%d.f32: wrong size\n", k); return 2; }
		void* p = vvm_device_alloc(bytes);
		if (!p || vvm_device_upload(p, host.data(), host.size()) != VVM_OK) { std::fprintf(stderr, "device memory: %s\n", vvm_last_error()); return 1; }
		in_streams.push_back((const uint8*)p);
	}
	for (int k = 0; k < num_out; ++k) {
		void* p = vvm_device_alloc(bytes);
		if (!p || vvm_device_memset(p, 0xCD, bytes) != VVM_OK) { std::fprintf(stderr, "device memory: %s\n", vvm_last_error()); return 1; }
		out_streams.push_back((uint8*)p);
	}

	FDataSetMeta DataSet;      // all streams are 32-bit floats: the int32 and half groups start behind them
	DataSet.Init(FDataSetMeta::FInputRegisterView(in_streams.data(), num_in), FDataSetMeta::FOutputRegisterView(out_streams.data(), num_out),
	             offset, nullptr, nullptr, nullptr, nullptr, nullptr, INDEX_NONE, nullptr);
	DataSet.InputRegisterTypeOffsets[0] = 0; DataSet.InputRegisterTypeOffsets[1] = DataSet.InputRegisterTypeOffsets[2] = (uint32)num_in;
	DataSet.OutputRegisterTypeOffsets[0] = 0; DataSet.OutputRegisterTypeOffsets[1] = DataSet.OutputRegisterTypeOffsets[2] = (uint32)num_out;

	std::vector<const uint8*> table_ptrs;
	{
		size_t at = 0;
		for (int b : table_bytes) { table_ptrs.push_back(consts.data() + at); at += (size_t)b; }
		if (at != consts.size()) { std::fprintf(stderr, "the constant tables do not add up to consts.bin\n"); return 2; }
	}

	FVectorVMExecContext ExecCtx;
	ExecCtx.VVMState = State;
	ExecCtx.DataSets = TArrayView<FDataSetMeta>(&DataSet, 1);
	ExecCtx.NumInstances = instances;
	ExecCtx.ConstantTableData = table_ptrs.data();
	ExecCtx.ConstantTableNumBytes = table_bytes.data();
	ExecCtx.ConstantTableCount = (int32)table_bytes.size();
	ExecVectorVMState(&ExecCtx);
	const int status = LastExecStatus();
	if (status != VVM_OK) std::fprintf(stderr, "ExecVectorVMState: %s\n", vvm_last_error());
	// a second Exec on the same state and context gives the same result (the state is reusable, as in the reference)
	const int32 first_index = DataSet.DataSetAccessIndex;
	ExecVectorVMState(&ExecCtx);
	if (LastExecStatus() != status || DataSet.DataSetAccessIndex != first_index) { std::fprintf(stderr, "the second Exec differs\n"); return 1; }

	for (int k = 0; k < num_out; ++k) {
		std::vector<uint8> host(elements * sizeof(float));
		if (vvm_device_download(host.data(), out_streams[(size_t)k], host.size()) != VVM_OK) { std::fprintf(stderr, "download: %s\n", vvm_last_error()); return 1; }
		std::ofstream f(dir + "/out_" + std::to_string(k) + ".f32", std::ios::binary);
		f.write((const char*)host.data(), (std::streamsize)host.size());
	}
	{
		std::ofstream f(dir + "/result.txt");
		f << (DataSet.DataSetAccessIndex + 1) << " " << status << " " << ExecCtx.Internal.MaxInstancesPerChunk << "\n";      // (survivors: VectorVMRuntime.cpp:2305-2308)
	}
	FreeVectorVMState(State);
	for (const uint8* p : in_streams) vvm_device_free((void*)p);
	for (uint8* p : out_streams) vvm_device_free(p);
	return 0;
}
