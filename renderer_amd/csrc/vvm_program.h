// vvm_program.h -- host side: frozen-blob thaw, bytecode decoding and linking into the device program.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vvm_abi.h"
#include "vvm_device.h"

namespace vvm {

// Everything ThawRuntimeContext recovers from a blob (VectorVMBridge.cpp:181-199), copied out of it.
struct ThawedScript {
	std::vector<uint8_t>  bytecode;
	std::vector<uint16_t> const_remap;            // ConstRemap[1]: remapped const -> original dword index
	std::vector<uint16_t> input_remap;            // InputRemapTable
	std::vector<uint16_t> input_dataset_offsets;  // 8 per input dataset
	std::vector<uint8_t>  out_dataset;            // OutputRemapDataSetIdx
	std::vector<uint16_t> out_type;               // OutputRemapDataType: 0 f32, 1 i32, 2 f16
	std::vector<uint16_t> out_dst;                // OutputRemapDst (0xFFFF = none)
	std::vector<int32_t>  ext_num_inputs, ext_num_outputs;
	uint32_t num_temps = 0, num_consts = 0, num_inputs = 0, num_noadvance = 0, num_input_datasets = 0;
	uint32_t num_outputs_remapped = 0, num_output_instructions = 0, max_output_dataset = 0;
	uint32_t num_ext_fns = 0, max_ext_fn_registers = 0, num_dummy_regs = 0, flags = 0;
	uint64_t hash_id = 0;
};

// One remapped input, classified from InputDataSetOffsets the way VVMBuildMapTableCaches does (VectorVMRuntime.cpp:510-552).
struct InputInfo {
	uint8_t  dataset;
	uint8_t  type;        // 0 f32, 1 i32, 2 f16
	bool     noadvance;
	uint16_t reg;         // per-type register index inside the dataset (InputRemapTable entry)
};

struct LinkedProgram {
	std::vector<VvmIns> code;          // device instructions, terminated by a `done`
	std::vector<InputInfo> inputs;
	uint32_t num_vregs = 0;            // per-lane registers used
	uint32_t instances_per_lane = 4;   // 4 or 1
	uint32_t resident_regs = 16;       // registers in VGPRs on the kernel the program is linked for: 16, or (many_registers) 44 on the four-wide
	                                   // kernel of plain scripts, 32 on the four-wide kernel with the transcendental / noise / ID instructions
	                                   // and on the one-instance-per-lane kernels
	uint32_t lds_spill_slots = 0;      // ... and the registers beyond them, in LDS slots
	bool     many_registers = false;   // runs on the kernels with more than 16 resident registers (and LDS register slots)
	// A plain four-wide program can be sent to the kernel with the transcendental functions as it is (an Exec that reads the reference
	// generator's words): there 32 registers are resident.
	uint32_t lds_slots_with_32_resident() const { return num_vregs > 32 ? num_vregs - 32 : 0; }
	bool     light = false;            // linked for the light eight-wide kernel (eight resident registers, six staged outputs and up to
	                                   // two that wait in registers VVM_LIGHT_RIDE_REG, +1)
	bool     full_kernel = false;      // needs the transcendental / noise kernel
	bool     prefers_routines = false; // a four-wide program of at most sixteen registers that is expected to run faster on the four-wide kernel
	                                   // of many registers (ordinary instructions through hand-written routines, two workgroups per CU) than
	                                   // on the one with the compiled loop (more workgroups per CU): enough ordinary instructions, and many
	                                   // of them per transcendental / noise instruction
	bool     uses_perlin = false;
	// update_id / acquire_id: one entry per dataset whose ID tables this pass works on (its index is the slot of the dataset's
	// VvmIdDesc behind the Exec descriptor); a dataset is either updated (at most one update_id, in the dataset's own pass) or
	// acquired from
	// (updates and acquires both: a script that takes IDs from the dataset's free list and returns the IDs of its dead instances to it in
	// one pass.  The reference then hands IDs from chunk to chunk, so such an Exec runs chunk after chunk: serial_chunks)
	struct IdUse { uint32_t dataset; bool updates; bool acquires; bool serial = false; };      // serial: the script (not necessarily this pass) has both on the dataset
	std::vector<IdUse> id_uses;
	bool     uses_ids = false;         // id_uses is not empty
	bool     updates_ids = false;      // ... and has an update_id
	bool     serial_chunks = false;    // ... and some dataset has both: the Exec is run as one launch per chunk of the reference's chunking
	std::vector<uint8_t> acquires_dataset;   // per output dataset: the script runs acquireindex for it
	std::vector<uint32_t> acquire_order;     // those datasets in program order
	uint32_t num_passes = 1;                 // one pass per acquired dataset (see link_program)
	uint32_t pass_dataset = 0;               // the dataset this program compacts and writes
	// A script that acquires two or more datasets and needs neither the transcendental / noise / ID instructions nor the
	// one-instance-per-lane kernel is linked with the datasets two to a pass (`paired`): a pass of two is ONE program with two
	// chains (num_chains = 2; two datasets in all: num_passes = 1): both acquireindex run in it, each with its own keep mask,
	// count and look-back words, and every output belongs to the chain of its dataset.  chain_dataset[c] = the dataset of chain c (chain 0 = pass_dataset); out_chain[k] = the chain of output-list entry k.
	// Derived constants: an add / sub / mul (float or integer) whose operands are all constants of the script yields the same value
	// for every instance; the linker takes such instructions out of the program and the host computes them when it fills an Exec's
	// constants (the arithmetic is IEEE single precision / 32-bit wrap-around on both sides).  Uniform index num_consts + j names
	// derived[j]; a and b are uniform indices (constants or earlier derived ones).
	struct Derived { uint8_t op; uint16_t a, b; };
	std::vector<Derived> derived;
	uint32_t num_chains = 1;
	bool     paired = false;                 // the script's acquired datasets are taken two to a pass: pass p has those at places 2 p and 2 p + 1
	                                         // of acquire_order (the last pass of an odd number has one), num_passes = ceil(datasets / 2)
	uint32_t chain_dataset[2] = {0, 0};
	std::vector<uint8_t> out_chain;
	// The script's random draws in program order (a random_2x makes two), whether or not this program keeps the instruction:
	// draws[o] = {ordinal of the first draw of o's instruction, draws of that instruction}.  The reference's generator is stepped
	// once per draw and loop in execution order (VectorVMRuntime.cpp:990-1111), which these say.
	struct RandDraw { uint8_t first, width; };
	std::vector<RandDraw> rand_draws;
	uint32_t bytes_in_per_instance = 0, bytes_out_per_instance = 0;
	uint32_t num_prefetched_inputs = 0;
	std::vector<uint8_t> input_used;   // per remapped input: some instruction of this program reads it
	std::vector<int32_t> load_list;    // remapped input index per load slot (8 per load instruction), -1 = unused slot
	std::vector<int32_t> out_list;     // output serial index per output-list entry (staging order), -1 = nothing is written,
	                                   // -2 = the ID register of update_id
	std::vector<uint8_t> out_is16;     // per entry: the destination stream has 16-bit elements
	// Per output serial index (the output's register minus the first output register): the remapped input whose value an output
	// instruction hands through to it unchanged (an advancing 32-bit stream written by outputdata_float / _int32, or an advancing
	// 16-bit stream written by outputdata_half_from_half, behind the dataset's acquireindex), or -1.  -2: the serial index is written by more than one pair.  See `out_select` of link_program.
	std::vector<int32_t> through_input;
};

// Returns VVM_OK or an error code with `err` filled.
int thaw_blob(const uint8_t* blob, size_t n, ThawedScript& out, std::string& err);
// `pass` selects the dataset of a script with several acquireindex datasets (0 .. out.num_passes - 1 of a first link).
// light: link for the light eight-wide kernel; fails with VVM_E_UNSUPPORTED_OP when the script does not fit it.
// out_select: null, or one flag per output serial index: the program writes only the outputs whose flag is set (and computes and
// loads only what those and the acquireindex need).  A script whose record is wider than the register file of the fast kernels is
// linked a second time without the outputs that hand an input stream through (through_input); a copy kernel moves those.
// pairing: for pass 0 leave at -1 (the linker decides whether the datasets go two to a pass, LinkedProgram::paired); for a later
// pass, what pass 0 decided (1 or 0).
int link_program(const ThawedScript& s, bool force_narrow, LinkedProgram& out, std::string& err, uint32_t pass = 0, bool light = false,
                 const std::vector<uint8_t>* out_select = nullptr, int pairing = -1);

}  // namespace vvm
