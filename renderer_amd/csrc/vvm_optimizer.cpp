// vvm_optimizer.cpp -- compiler-emitted ("original") VectorVM bytecode -> optimized bytecode + side tables -> frozen blob.
//
// Host-only code (no device work).  It restates, pass by pass, what the reference does at script-compile time:
//   OptimizeVectorVMScript      VectorVMOptimizer.cpp:700-3177   (the 18 steps; each pass below cites its lines)
//   FreezeOptimizerContext      VectorVMBridge.cpp:151-178       (layout :81-148)
// so that a script in the Niagara compiler's format can be handed to vvm_alloc_state unchanged.  The passes keep the
// reference's scan orders, tie-breaks and known oddities (each one is noted where it happens), because register numbering,
// output grouping and the choice of merged opcodes all follow from them.  The reference's tests hold no optimizer vectors
// and its sources may not be built here, so byte-identity with the reference's output is unpinned; what the tests pin is
// that the optimized script computes what the original one says.
//
// Like the reference, the passes work on three flat operand arrays (register, SSA name, operand class) that instructions
// point into, and on an instruction list that is reordered in place.  Several reference scans compare SSA names without
// looking at the operand class, or read past an instruction's own operands; the flat arrays reproduce those too.
#include "../../include/vvm_abi.h"

#include <algorithm>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum Cat : uint8_t { C_INPUT, C_OUTPUT, C_OP, C_EXTFN, C_INDEXGEN, C_RWBUFFER, C_STAT, C_OTHER };   // VectorVMEditor.h:7-15
enum RegClass : uint8_t { RT_TEMP, RT_CONST, RT_INPUT, RT_OUTPUT, RT_INVALID };                      // VectorVMEditor.h:35-42
const uint16_t NONE16 = 0xFFFF;

// category of every opcode (VectorVM.h:50-233, second column)
Cat category(uint8_t op)
{
	if (op == VVM_OP_done || op == VVM_OP_noise2D || op == VVM_OP_noise3D) return C_OTHER;
	if (op >= VVM_OP_inputdata_float && op <= VVM_OP_inputdata_noadvance_half) return C_INPUT;
	if ((op >= VVM_OP_outputdata_float && op <= VVM_OP_outputdata_half) || op == VVM_OP_outputdata_float_from_half ||
	    op == VVM_OP_outputdata_half_from_half)
		return C_OUTPUT;
	if (op == VVM_OP_acquireindex) return C_INDEXGEN;
	if (op == VVM_OP_external_func_call) return C_EXTFN;
	if (op == VVM_OP_enter_stat_scope || op == VVM_OP_exit_stat_scope) return C_STAT;
	if (op == VVM_OP_update_id || op == VVM_OP_acquire_id) return C_RWBUFFER;
	return C_OP;
}

// inputs of the ops the compiler emits (VectorVMOptimizer.cpp:761-838); 0 = not an op of the original format
int original_op_inputs(uint8_t op)
{
	switch (op) {
	case VVM_OP_mad: case VVM_OP_lerp: case VVM_OP_clamp: case VVM_OP_select: case VVM_OP_clampi:
		return 3;
	case VVM_OP_add: case VVM_OP_sub: case VVM_OP_mul: case VVM_OP_div: case VVM_OP_atan2: case VVM_OP_fmod: case VVM_OP_min:
	case VVM_OP_max: case VVM_OP_pow: case VVM_OP_step: case VVM_OP_cmplt: case VVM_OP_cmple: case VVM_OP_cmpgt: case VVM_OP_cmpge:
	case VVM_OP_cmpeq: case VVM_OP_cmpneq: case VVM_OP_addi: case VVM_OP_subi: case VVM_OP_muli: case VVM_OP_divi: case VVM_OP_mini:
	case VVM_OP_maxi: case VVM_OP_cmplti: case VVM_OP_cmplei: case VVM_OP_cmpgti: case VVM_OP_cmpgei: case VVM_OP_cmpeqi:
	case VVM_OP_cmpneqi: case VVM_OP_bit_and: case VVM_OP_bit_or: case VVM_OP_bit_xor: case VVM_OP_bit_lshift: case VVM_OP_bit_rshift:
	case VVM_OP_logic_and: case VVM_OP_logic_or: case VVM_OP_logic_xor:
		return 2;
	case VVM_OP_rcp: case VVM_OP_rsq: case VVM_OP_sqrt: case VVM_OP_neg: case VVM_OP_abs: case VVM_OP_exp: case VVM_OP_exp2:
	case VVM_OP_log: case VVM_OP_log2: case VVM_OP_sin: case VVM_OP_cos: case VVM_OP_tan: case VVM_OP_asin: case VVM_OP_acos:
	case VVM_OP_atan: case VVM_OP_ceil: case VVM_OP_floor: case VVM_OP_frac: case VVM_OP_trunc: case VVM_OP_round: case VVM_OP_sign:
	case VVM_OP_random: case VVM_OP_absi: case VVM_OP_negi: case VVM_OP_signi: case VVM_OP_randomi: case VVM_OP_bit_not:
	case VVM_OP_logic_not: case VVM_OP_f2i: case VVM_OP_i2f: case VVM_OP_f2b: case VVM_OP_b2f: case VVM_OP_i2b: case VVM_OP_b2i:
	case VVM_OP_fasi: case VVM_OP_iasf:
		return 1;
	default:
		return 0;
	}
}

struct Ins {                     // FVectorVMOptimizeInstruction, VectorVMEditor.h:50-102
	uint8_t  op = 0;
	Cat      cat = C_OTHER;
	int      index = 0;          // position at creation; inputs are named by it once they are folded into operands
	int      merged_into = -1;   // != -1: folded into that instruction, not emitted
	uint32_t regs = 0;           // first operand in the flat arrays; inputs first, then outputs
	int      nin = 0, nout = 0;
	uint16_t dataset = 0;        // input / output / acquireindex / id instructions
	uint16_t input_idx = 0;      // input instructions
	uint16_t dst_reg = 0;        // output instructions: register index within its type
	uint16_t serial = 0;         // output instructions: order of appearance
	uint16_t fn = 0;             // external calls
};

struct Optimizer {
	std::vector<Ins>      ins;
	std::vector<uint16_t> reg, ssa;          // RegisterUsageBuffer, SSARegisterUsageBuffer
	std::vector<uint8_t>  cls;               // RegisterUsageType
	std::vector<uint16_t> parent;            // SSA name -> position of the instruction that writes it
	uint32_t created = 0;                    // instructions created so far (gives Ins::index)
	uint16_t num_ssa = 0;

	// the runtime context being filled (FVectorVMRuntimeContext, VectorVMTypes.h:77-105)
	std::vector<uint16_t> const_slot;        // ConstRemap[0]: original dword index -> remapped index
	std::vector<uint16_t> const_remap;       // ConstRemap[1]: remapped index -> original dword index
	uint16_t num_consts_alloced = 0;
	uint32_t max_output_dataset = 0, num_temps = 0, max_ext_fn_registers = 0, num_dummy_regs = 0, flags = 0;
	int32_t  max_ext_fn_used = -1;
	uint16_t num_noadvance = 0, num_input_datasets = 0, num_outputs_remapped = 0, num_output_instructions = 0;
	std::vector<uint16_t> input_remap, input_dataset_offsets, out_type, out_dst;
	std::vector<uint8_t>  out_dataset;
	std::vector<uint8_t>  bytecode;
	std::vector<uint16_t> final_temp;        // FinalTempRegRemap
	uint32_t error = 0;

	// ---- helpers ----
	uint16_t remap_const(uint16_t byte_offset)      // VectorVMOptimizeRemapConst, :327-367
	{
		if (byte_offset & 3) { error |= VVM_OPT_ERR_CONST_REMAP; return 0; }
		const uint16_t idx = byte_offset >> 2;      // the compiler addresses constants by byte, the runtime by dword
		if (idx >= num_consts_alloced) {
			num_consts_alloced = (uint16_t)((idx + 1 + 511) & ~511);
			const_slot.resize(num_consts_alloced, NONE16);
		}
		if (const_slot[idx] == NONE16) {
			const_slot[idx] = (uint16_t)const_remap.size();
			const_remap.push_back(idx);
		}
		return const_slot[idx];
	}
	Ins& push_ins(uint8_t op)                       // VVMPushNewInstruction, :412-430
	{
		Ins n;
		n.op = op; n.cat = category(op); n.index = (int)created++; n.regs = (uint32_t)reg.size();
		ins.push_back(n);
		return ins.back();
	}
	void push_reg(Ins& in, uint16_t r, RegClass c, bool is_output)      // VVMPushRegUsage_, :492-519
	{
		reg.push_back(c == RT_CONST ? remap_const(r) : r);
		ssa.push_back(0);
		cls.push_back(c);
		(is_output ? in.nout : in.nin)++;
	}
	uint32_t new_regs(int n)
	{
		const uint32_t base = (uint32_t)reg.size();
		reg.resize(base + n, 0); ssa.resize(base + n, 0); cls.resize(base + n, RT_TEMP);
		return base;
	}
	uint16_t ssa_at(uint32_t i) const { return i < ssa.size() ? ssa[i] : NONE16; }
	uint8_t  cls_at(uint32_t i) const { return i < cls.size() ? cls[i] : (uint8_t)RT_INVALID; }
	bool has_parent(uint16_t s) const { return s != NONE16 && s < parent.size() && parent[s] != NONE16; }
	void set_parents()                              // VVMSetParentInstructionIndices, :566-580
	{
		for (size_t i = 0; i < ins.size(); ++i)
			for (int j = 0; j < ins[i].nout; ++j) {
				const uint16_t s = ssa[ins[i].regs + ins[i].nin + j];
				if (s != NONE16 && s < parent.size()) parent[s] = (uint16_t)i;
			}
	}
	void move_ins(size_t from, size_t to)           // the reference's Memmove + store pairs
	{
		if (from > to) std::rotate(ins.begin() + to, ins.begin() + from, ins.begin() + from + 1);
		else if (from < to) std::rotate(ins.begin() + from, ins.begin() + from + 1, ins.begin() + to + 1);
	}
	// every instruction `at` depends on, directly or not, ascending (VVMGetInstructionDependencyChain, :620-671)
	std::vector<uint16_t> dependency_chain(size_t at) const
	{
		std::vector<uint16_t> found, sorted;
		size_t checked = 0;
		const Ins* in = &ins[at];
		for (;;) {
			for (int i = 0; i < in->nin; ++i) {
				const uint16_t s = ssa[in->regs + i];
				const uint8_t c = cls[in->regs + i];
				if (!(c == RT_TEMP || c == RT_OUTPUT) || !has_parent(s)) continue;
				const uint16_t p = parent[s];
				if (std::find(found.begin(), found.end(), p) != found.end()) continue;
				found.push_back(p);
				sorted.insert(std::upper_bound(sorted.begin(), sorted.end(), p), p);
			}
			if (checked >= found.size()) break;
			in = &ins[found[checked++]];
		}
		return sorted;
	}
	bool depends_on(size_t at, size_t on) const     // VVMIsInstructionDependentOnInstruction, :582-618
	{
		std::vector<uint16_t> found;
		size_t checked = 0;
		const Ins* in = &ins[at];
		for (;;) {
			for (int i = 0; i < in->nin; ++i) {
				const uint16_t s = ssa[in->regs + i];
				const uint8_t c = cls[in->regs + i];
				if (!(c == RT_TEMP || c == RT_OUTPUT) || !has_parent(s)) continue;
				const uint16_t p = parent[s];
				if (p == on) return true;
				if (std::find(found.begin(), found.end(), p) == found.end()) found.push_back(p);
			}
			if (checked >= found.size()) break;
			in = &ins[found[checked++]];
		}
		return false;
	}
	bool has_dependencies(const Ins& in) const      // VVMDoesInstructionHaveDependencies, :673-685
	{
		for (int i = 0; i < in.nin; ++i)
			if (ssa[in.regs + i] != NONE16 && (cls[in.regs + i] == RT_TEMP || cls[in.regs + i] == RT_OUTPUT)) return true;
		return false;
	}

	void dump(const char* when) const
	{
		if (!getenv("VVM_OPT_DEBUG")) return;
		fprintf(stderr, "-- %s\n", when);
		for (size_t i = 0; i < ins.size(); ++i) {
			const Ins& in = ins[i];
			fprintf(stderr, "%3zu: op %3d idx %3d merged %3d dst %d |", i, in.op, in.index, in.merged_into, in.dst_reg);
			for (int j = 0; j < in.nin + in.nout; ++j) fprintf(stderr, "%s %c%d(s%d)", j == in.nin ? " ->" : "", "tcio-"[cls[in.regs + j]], reg[in.regs + j], ssa[in.regs + j]);
			fprintf(stderr, "\n");
		}
	}
	int  decode(const uint8_t* bc, int n, const vvm_ext_fn_io* ext, int num_ext);
	void rename_ssa();
	bool remove_unused_ops();
	void remove_casts();
	int  fold_inputs();
	void reorder(int one_past_last_input);
	void sort_output_runs();
	void merge_ops();
	bool apply_merge_rule(uint32_t first, uint32_t second, int found);
	void build_io_tables();
	void minimise_temps();
	void alias_outputs();
	void emit(const vvm_ext_fn_io* ext);
};

uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

// Step 1 (:744-1025): one Ins per original instruction, operands into the flat arrays, constants remapped in order of first use.
int Optimizer::decode(const uint8_t* bc, int n, const vvm_ext_fn_io* ext, int num_ext)
{
	const uint8_t* p = bc;
	const uint8_t* end = bc + n;
	uint16_t num_outputs = 0;
	auto need = [&](int bytes) { return p + bytes <= end; };
	while (p < end) {
		const uint8_t op = *p;
		Ins& in = push_ins(op);
		++p;
		const int op_inputs = original_op_inputs(op);
		if (op_inputs) {
			// [constFlags:u8][in0..][out]: four u16 are read whatever the count (:995-998)
			if (!need(1 + 2 * (op_inputs + 1))) return VVM_OPT_ERR_BYTECODE;
			const uint8_t const_flags = *p;
			for (int k = 0; k < op_inputs; ++k) {
				const bool is_const = (const_flags >> k) & 1;
				push_reg(in, rd16(p + 1 + 2 * k), is_const ? RT_CONST : RT_TEMP, false);
			}
			push_reg(in, rd16(p + 1 + 2 * op_inputs), RT_TEMP, true);
			p += 1 + 2 * (op_inputs + 1);
			continue;
		}
		switch (op) {
		case VVM_OP_done: case VVM_OP_exit_stat_scope:
			break;
		case VVM_OP_inputdata_float: case VVM_OP_inputdata_int32: case VVM_OP_inputdata_half:
		case VVM_OP_inputdata_noadvance_float: case VVM_OP_inputdata_noadvance_int32: case VVM_OP_inputdata_noadvance_half:
			if (!need(6)) return VVM_OPT_ERR_BYTECODE;           // [dataset][input][dst temp]  (:845-866)
			in.dataset = rd16(p); in.input_idx = rd16(p + 2);
			push_reg(in, rd16(p + 4), RT_TEMP, true);
			if (op >= VVM_OP_inputdata_noadvance_float) ++num_noadvance;
			p += 6;
			break;
		case VVM_OP_outputdata_float: case VVM_OP_outputdata_int32: case VVM_OP_outputdata_half: {
			if (!need(9)) return VVM_OPT_ERR_BYTECODE;           // [srcIsConst][dataset][index temp][src][dst]  (:871-891)
			const bool src_const = *p & 1;
			in.dataset = rd16(p + 1);
			if (in.dataset >= 0xFF) return VVM_OPT_ERR_BYTECODE;
			max_output_dataset = std::max<uint32_t>(max_output_dataset, in.dataset + 1u);
			in.dst_reg = rd16(p + 7);
			in.serial = num_outputs++;
			push_reg(in, rd16(p + 3), RT_TEMP, false);
			push_reg(in, rd16(p + 5), src_const ? RT_CONST : RT_TEMP, false);
			p += 9;
			break;
		}
		case VVM_OP_acquireindex: {
			if (!need(7)) return VVM_OPT_ERR_BYTECODE;           // [srcIsConst][dataset][predicate][index temp out]  (:894-911)
			const bool src_const = *p & 1;
			in.dataset = rd16(p + 1);
			push_reg(in, rd16(p + 3), src_const ? RT_CONST : RT_TEMP, false);
			push_reg(in, rd16(p + 5), RT_TEMP, true);
			p += 7;
			break;
		}
		case VVM_OP_external_func_call: {
			if (!need(1)) return VVM_OPT_ERR_BYTECODE;           // [fn:u8][inputs][outputs]  (:914-950)
			const uint8_t fn = *p;
			if ((int)fn >= num_ext) return VVM_OPT_ERR_EXTERNAL_FUNCTION;
			const int ni = ext[fn].NumInputs, no = ext[fn].NumOutputs;
			if (ni < 0 || no < 0 || !need(1 + 2 * (ni + no))) return VVM_OPT_ERR_BYTECODE;
			in.fn = fn;
			for (int i = 0; i < ni; ++i) {
				const uint16_t v = rd16(p + 1 + 2 * i);
				if (v == NONE16) push_reg(in, NONE16, RT_INVALID, false);
				else if (v & 0x8000) push_reg(in, v & 0x7FFF, RT_TEMP, false);      // high bit: a temp register
				else push_reg(in, v, RT_CONST, false);
			}
			for (int i = 0; i < no; ++i) {
				const uint16_t v = rd16(p + 1 + 2 * (ni + i));
				if ((v & 0x8000) && v != NONE16) return VVM_OPT_ERR_BYTECODE;         // an output cannot be a constant
				push_reg(in, v, RT_TEMP, true);
			}
			max_ext_fn_used = std::max<int32_t>(max_ext_fn_used, fn);
			max_ext_fn_registers = std::max<uint32_t>(max_ext_fn_registers, (uint32_t)(ni + no));
			p += 1 + 2 * (ni + no);
			break;
		}
		case VVM_OP_exec_index:
			if (!need(2)) return VVM_OPT_ERR_BYTECODE;           // [dst temp:u8][pad]  (:952-956)
			push_reg(in, *p, RT_TEMP, true);
			p += 2;
			break;
		case VVM_OP_enter_stat_scope:
			if (!need(2)) return VVM_OPT_ERR_BYTECODE;
			p += 2;
			break;
		case VVM_OP_update_id: case VVM_OP_acquire_id: {
			if (!need(6)) return VVM_OPT_ERR_BYTECODE;           // [dataset][id temp][tag temp]  (:962-981)
			in.dataset = rd16(p);
			const bool out = op == VVM_OP_acquire_id;
			push_reg(in, rd16(p + 2), RT_TEMP, out);
			push_reg(in, rd16(p + 4), RT_TEMP, out);
			p += 6;
			break;
		}
		default:
			return VVM_OPT_ERR_BYTECODE;       // noise*, merged opcodes and half_to_float are not part of the original format
		}
	}
	if (reg.size() >= 0xFFFF) return VVM_OPT_ERR_REGISTER_USAGE | VVM_OPT_ERR_OVERFLOW;      // 16-bit operand positions (:1072)
	return error;
}

// Step 5 (:1089-1183): every value gets a name of its own.  A register's value runs from the instruction that writes it to
// the next one that writes the same register; the readers in between get the writer's name.  Known oddities, kept: the scan
// compares register numbers without looking at the operand class, and it stops at an overwriting instruction before looking
// at that instruction's own reads (so an in-place update is not linked to the value it reads).
void Optimizer::rename_ssa()
{
	ssa = reg;
	uint16_t count = 0;
	for (size_t oi = 0; oi < ins.size(); ++oi) {
		const Ins& o = ins[oi];
		for (int j = 0; j < o.nout; ++j) {
			const uint32_t at = o.regs + o.nin + j;
			const uint16_t out = reg[at];
			if (out == NONE16 || cls[at] != RT_TEMP) { ssa[at] = NONE16; continue; }
			ssa[at] = count;
			int last_use = -1;
			bool overwritten = false;
			for (size_t ui = oi + 1; ui < ins.size() && !overwritten; ++ui) {
				const Ins& u = ins[ui];
				for (int k = 0; k < u.nout; ++k)
					if (reg[u.regs + u.nin + k] == out) { overwritten = true; break; }
				if (overwritten) break;
				for (int k = 0; k < u.nin; ++k) {
					if (reg[u.regs + k] != out) continue;
					if (u.cat == C_OUTPUT) {
						// operand 0 of an output is the index register, operand 1 its source
						if (o.op == VVM_OP_acquireindex) {
							if (j == k) { ssa[u.regs + j] = count; last_use = (int)ui; }
						} else {
							ssa[u.regs + 1] = count; last_use = (int)ui;
						}
					} else {
						ssa[u.regs + k] = count; last_use = (int)ui;
					}
				}
			}
			if (overwritten || last_use != -1) ++count;
			else ssa[at] = NONE16;          // never read: the instruction goes away in the next step
		}
	}
	num_ssa = count;
}

// Step 6 (:1185-1241): ops none of whose results is read are dropped, repeatedly.  (After a removal the reference restarts its
// scan at position 1, not 0.)
bool Optimizer::remove_unused_ops()
{
	bool removed_any = false;
	int sanity = 0;
	for (;;) {
		int removed = 0;
		for (size_t i = 0; i < ins.size(); ++i) {
			const Ins& in = ins[i];
			if (in.cat != C_OP) continue;
			bool required = false;
			for (int o = 0; o < in.nout && !required; ++o) {
				const uint16_t name = ssa[in.regs + in.nin + o];
				for (size_t j = i + 1; j < ins.size() && !required; ++j)
					for (int k = 0; k < ins[j].nin; ++k)
						if (cls[ins[j].regs + k] == RT_TEMP && ssa[ins[j].regs + k] == name) { required = true; break; }
			}
			if (!required) {
				ins.erase(ins.begin() + i);
				++removed;
				removed_any = true;
				i = 0;
			}
		}
		if (++sanity >= 16384) { error |= VVM_OPT_ERR_REDUNDANT_INSTRUCTION; break; }
		if (!removed) break;
	}
	return removed_any;
}

// Step 7 (:1243-1278): fasi / iasf are bit casts; their readers read the cast's source instead.
void Optimizer::remove_casts()
{
	for (size_t ci = 0; ci < ins.size(); ++ci) {
		const Ins c = ins[ci];
		if (c.op != VVM_OP_iasf && c.op != VVM_OP_fasi) continue;
		const uint32_t src = c.regs, dst = c.regs + 1;
		for (size_t i = ci + 1; i < ins.size(); ++i)
			for (int k = 0; k < ins[i].nin; ++k) {
				const uint32_t at = ins[i].regs + k;
				if (ssa[at] == ssa[dst] && cls[at] == cls[dst]) { ssa[at] = ssa[src]; reg[at] = reg[src]; cls[at] = cls[src]; }
			}
		ins.erase(ins.begin() + ci);
		--ci;
	}
}

// Step 8 (:1280-1398): a temp that comes straight from an input instruction becomes an input operand (named by the input
// instruction's index until the tables are built).  A half input read by an op goes through an inserted half_to_float; an
// output that copies a half input becomes one of the *_from_half outputs.  Returns the position after the last input
// instruction (and the stat instructions that follow it).
int Optimizer::fold_inputs()
{
	int one_past_last_input = -1;
	for (size_t i = 0; i < ins.size(); ++i) {
		if (ins[i].cat != C_INPUT) continue;
		one_past_last_input = (int)i + 1;
		const Ins input = ins[i];
		const uint16_t name = ssa[input.regs];
		const bool half = input.op == VVM_OP_inputdata_half || input.op == VVM_OP_inputdata_noadvance_half;
		for (size_t j = i + 1; j < ins.size(); ++j) {
			Ins& u = ins[j];
			if (u.cat == C_INPUT || u.cat == C_OTHER || u.cat == C_STAT) continue;
			if (u.cat == C_OUTPUT) {
				const uint32_t src = u.regs + 1;
				if (cls[src] == RT_TEMP && ssa[src] == name) {
					ssa[src] = (uint16_t)input.index;
					cls[src] = RT_INPUT;
					if (input.op == VVM_OP_inputdata_half && u.op == VVM_OP_outputdata_float) u.op = VVM_OP_outputdata_float_from_half;
					else if (input.op == VVM_OP_inputdata_half && u.op == VVM_OP_outputdata_half) u.op = VVM_OP_outputdata_half_from_half;
				}
				continue;
			}
			for (int k = 0; k < ins[j].nin; ++k) {
				const uint32_t at = ins[j].regs + k;
				if (ssa[at] != name || cls[at] != RT_TEMP) continue;
				if (!half) { ssa[at] = (uint16_t)input.index; cls[at] = RT_INPUT; continue; }
				bool have_conversion = false;
				for (int b = (int)j - 1; b >= 0 && !have_conversion; --b)
					have_conversion = ins[b].op == VVM_OP_half_to_float && ssa[ins[b].regs + 1] == name;
				if (!have_conversion) {
					const uint16_t dst_reg = reg[at];
					Ins conv = push_ins(VVM_OP_half_to_float);
					ins.pop_back();
					conv.regs = new_regs(2);
					conv.nin = 1; conv.nout = 1;
					reg[conv.regs] = (uint16_t)input.index; cls[conv.regs] = RT_INPUT; ssa[conv.regs] = (uint16_t)input.index;
					reg[conv.regs + 1] = dst_reg; cls[conv.regs + 1] = RT_TEMP; ssa[conv.regs + 1] = name;
					ins.insert(ins.begin() + j, conv);
					++j;                 // the reader is one further down now; its remaining operands read the same temp, if at all
					break;
				}
			}
		}
	}
	if (one_past_last_input == -1) one_past_last_input = 0;
	while (!ins.empty() && (size_t)one_past_last_input < ins.size() - 1 && ins[one_past_last_input].cat == C_STAT) ++one_past_last_input;
	return one_past_last_input;
}

// Steps 9-11 (:1400-1529): acquireindex and what it depends on moves to the top, each output moves up to just below the
// nearest instruction that mentions its source, and ops that depend on nothing move down to just above their first reader.
void Optimizer::reorder(int one_past_last_input)
{
	parent.assign(std::max<size_t>(num_ssa, reg.size()) + 4, NONE16);
	set_parents();
	size_t lowest = (size_t)one_past_last_input;
	for (size_t i = 0; i < ins.size(); ++i) {              // step 9
		if (ins[i].op != VVM_OP_acquireindex) continue;
		const size_t acquire_at = i;
		const std::vector<uint16_t> chain = dependency_chain(acquire_at);
		for (uint16_t dep : chain) {
			if (dep > lowest) {
				if (ins[dep].op == VVM_OP_external_func_call)          // external calls keep their order
					for (size_t k = lowest; k < dep; ++k)
						if (ins[k].op == VVM_OP_external_func_call) lowest = k + 1;
				move_ins(dep, lowest);
				set_parents();
			}
			++lowest;
		}
		if (lowest < acquire_at) {
			move_ins(acquire_at, lowest++);
			set_parents();
		}
	}
	for (size_t oi = 0; oi < ins.size(); ++oi) {           // step 10
		if (ins[oi].cat != C_OUTPUT) continue;
		const uint16_t index_name = ssa[ins[oi].regs], src_name = ssa[ins[oi].regs + 1];
		const int acquire_at = index_name < parent.size() ? (int)parent[index_name] : (int)NONE16;
		size_t insert_at = SIZE_MAX;
		for (int i = (int)oi - 1; i >= acquire_at && insert_at == SIZE_MAX; --i)
			for (int j = 0; j < ins[i].nin + ins[i].nout; ++j)
				if (ssa_at(ins[i].regs + j) == src_name) { insert_at = (size_t)i + 1; break; }     // (the class is not compared)
		if (insert_at != SIZE_MAX && insert_at < ins.size() - 1 && oi > insert_at) {
			move_ins(oi, insert_at);
			set_parents();
		}
	}
	size_t last_swap = NONE16;                              // step 11
	for (size_t i = 0; i < ins.size(); ++i) {
		const size_t skip = last_swap;
		last_swap = NONE16;
		if (ins[i].cat != C_OP || has_dependencies(ins[i])) continue;
		for (size_t j = i + 1; j < ins.size(); ++j) {
			if (!depends_on(j, i)) continue;
			if (j > i + 1 && j != skip) {
				move_ins(i, j - 1);
				last_swap = j;
				--i;
				set_parents();
			}
			break;
		}
	}
}

// Step 12 (:1531-1578): every run of two or more adjacent outputs is sorted by (kind, dataset, index register, destination).
void Optimizer::sort_output_runs()
{
	auto key = [&](const Ins& o) -> uint64_t {                      // VVMOutputInsGetSortKey, :370-380
		return ((uint64_t)(o.op - VVM_OP_outputdata_float) << 62) + ((uint64_t)o.dataset << 48) + ((uint64_t)ssa[o.regs] << 16) + o.dst_reg;
	};
	for (size_t i = 0; i < ins.size(); ++i) {
		if (ins[i].cat != C_OUTPUT) continue;
		size_t j = i + 1;
		while (j < ins.size() && ins[j].cat == C_OUTPUT) ++j;
		if (j == ins.size()) break;             // a run that reaches the end is left alone (the stream ends in `done`)
		if (j - i > 1) {
			std::stable_sort(ins.begin() + i, ins.begin() + j, [&](const Ins& a, const Ins& b) { return key(a) < key(b); });   // what the bubble sort computes
			i = j;
		}
	}
}

// ---- the merged opcodes as data -------------------------------------------------------------------------------------------------
// One row per way two instructions fold into a merged opcode of VectorVM.h:153-231 (the decisions of VectorVMOptimizer.cpp:
// 1815-2518).  `first` makes a value that `second` reads; the row says which pairs it covers, what vetoes it, and how the merged
// instruction's operands are taken from the two.  Rows for one (first, second) are tried in table order and the first whose
// conditions hold decides; a veto on that row means the pair stays as it is.
namespace merge {
enum Reads : uint8_t {       // how `second` reads `first` (what the pair search found)
	R_ANY,                   // no condition
	R_OPERAND,               // through exactly one of its operands
	R_OPERAND0, R_OPERAND1,  // ... through that operand
	R_EXEC_INDEX,            // `first` is exec_index and `second` its only reader
	R_TWIN                   // the two have the same inputs (the *_2x forms and sin_cos): the reference's search for these never
	                         // yields a pair (see merge_ops), so the rows are inert; they are listed because the reference has them
};
enum At : uint8_t {          // where first's result sits among second's operands, compared by name and register class
	AT_ANY, AT_0, AT_1, AT_2, AT_0_OR_1
};
enum : uint8_t {
	F_END_PASS = 1,          // no merge, and the whole pass ends here (the reference returns from the pass: :2095-2098)
	F_EXEC_ADDI = 2          // the veto of exec_index + addi (below)
};
constexpr uint8_t NO_VETO = 0xFF;
// `take`: the merged instruction's registers, inputs then outputs, two characters each:
//   a<i> / b<i>   register i of first / second (inputs first, then outputs)
//   bs            of a two-input `second`, the input that is not first's result: input 1 if input 0 equals the result, else input 0
//   bk            the same choice made from the pair search's finding instead of a comparison
struct Rule {
	uint8_t first, second;
	uint8_t reads, at;
	uint8_t merged, nin, nout;
	const char* take;
	uint8_t dup_lo, dup_hi;  // veto when first's result appears twice among second's operands [dup_lo, dup_hi] (NO_VETO: no such check)
	uint8_t flags;
};
#define CHAIN1 "a0bsb2"              /* a one-input first, a two-input second */
#define CHAIN2 "a0a1bsb2"            /* two-input first */
#define CHAIN3 "a0a1a2bsb2"          /* three-input first */
static const Rule kRules[] = {
	{VVM_OP_exec_index, VVM_OP_i2f,  R_EXEC_INDEX, AT_ANY, VVM_OP_exec_indexf,     0, 1, "b1",   NO_VETO, 0, 0},
	{VVM_OP_exec_index, VVM_OP_addi, R_EXEC_INDEX, AT_ANY, VVM_OP_exec_index_addi, 1, 1, "bsb2", NO_VETO, 0, F_EXEC_ADDI},
	// compare + select: the "greater" and "not equal" forms use the opposite compare with the two selected values swapped
	{VVM_OP_cmplt,   VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmplt_select,  4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_cmple,   VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmple_select,  4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_cmpgt,   VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmple_select,  4, 1, "a0a1b2b1b3", NO_VETO, 0, 0},
	{VVM_OP_cmpge,   VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmplt_select,  4, 1, "a0a1b2b1b3", NO_VETO, 0, 0},
	{VVM_OP_cmpeq,   VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmpeq_select,  4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_cmpneq,  VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmpeq_select,  4, 1, "a0a1b2b1b3", NO_VETO, 0, 0},
	{VVM_OP_cmplti,  VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmplti_select, 4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_cmplei,  VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmplei_select, 4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_cmpgti,  VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmplei_select, 4, 1, "a0a1b2b1b3", NO_VETO, 0, 0},
	{VVM_OP_cmpgei,  VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmplti_select, 4, 1, "a0a1b2b1b3", NO_VETO, 0, 0},
	{VVM_OP_cmpeqi,  VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmpeqi_select, 4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_cmpneqi, VVM_OP_select, R_OPERAND0, AT_ANY, VVM_OP_cmpeqi_select, 4, 1, "a0a1b2b1b3", NO_VETO, 0, 0},
	// (compare + logic_and / logic_or: twenty-four rows, generated: see rules())
	{VVM_OP_mad, VVM_OP_mad,  R_OPERAND, AT_2,   VVM_OP_mad_mad0, 5, 1, "a0a1a2b0b1b3", 0, 3, 0},      // first is second's addend
	{VVM_OP_mad, VVM_OP_mad,  R_OPERAND, AT_ANY, VVM_OP_mad_mad1, 5, 1, "a0a1a2bsb2b3", 0, 3, 0},      // first is a factor of second
	{VVM_OP_mad, VVM_OP_add,  R_OPERAND, AT_ANY, VVM_OP_mad_add,  4, 1, CHAIN3, NO_VETO, 0, 0},
	{VVM_OP_mad, VVM_OP_mul,  R_OPERAND, AT_ANY, VVM_OP_mad_mul,  4, 1, CHAIN3, NO_VETO, 0, 0},
	{VVM_OP_mad, VVM_OP_sub,  R_OPERAND, AT_0,   VVM_OP_mad_sub0, 4, 1, CHAIN3, NO_VETO, 0, 0},
	{VVM_OP_mad, VVM_OP_sub,  R_OPERAND, AT_ANY, VVM_OP_mad_sub1, 4, 1, CHAIN3, NO_VETO, 0, 0},
	{VVM_OP_mad, VVM_OP_sqrt, R_OPERAND, AT_ANY, VVM_OP_mad_sqrt, 3, 1, "a0a1a2b1", NO_VETO, 0, 0},
	{VVM_OP_mul, VVM_OP_mad,  R_OPERAND, AT_0,   VVM_OP_mul_mad0, 4, 1, "a0a1b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_mul, VVM_OP_mad,  R_OPERAND, AT_1,   VVM_OP_mul_mad0, 4, 1, "a0a1b0b2b3", NO_VETO, 0, 0},
	{VVM_OP_mul, VVM_OP_mad,  R_OPERAND, AT_2,   VVM_OP_mul_mad1, 4, 1, "a0a1b0b1b3", NO_VETO, 0, 0},
	{VVM_OP_mul, VVM_OP_add,  R_OPERAND, AT_ANY, VVM_OP_mul_add,  3, 1, CHAIN2, 0, 1, 0},
	{VVM_OP_mul, VVM_OP_sub,  R_OPERAND, AT_0,   VVM_OP_mul_sub0, 3, 1, CHAIN2, 0, 1, 0},
	{VVM_OP_mul, VVM_OP_sub,  R_OPERAND, AT_ANY, VVM_OP_mul_sub1, 3, 1, CHAIN2, 0, 1, 0},
	{VVM_OP_mul, VVM_OP_mul,  R_OPERAND, AT_ANY, VVM_OP_mul_mul,  3, 1, CHAIN2, 0, 1, 0},
	{VVM_OP_mul, VVM_OP_max,  R_OPERAND, AT_ANY, VVM_OP_mul_max,  3, 1, CHAIN2, 0, 1, 0},
	{VVM_OP_mul, VVM_OP_mul,  R_TWIN,    AT_ANY, VVM_OP_mul_2x,   2, 2, "a0a1a2b2", NO_VETO, 0, 0},
	// add + mad merges only when the sum is the mad's addend; any other add + mad ends the pass
	{VVM_OP_add, VVM_OP_mad,  R_OPERAND, AT_2,   VVM_OP_add_mad1, 4, 1, "a0a1b0b1b3", 0, 2, 0},
	{VVM_OP_add, VVM_OP_mad,  R_OPERAND, AT_ANY, 0,               0, 0, "", NO_VETO, 0, F_END_PASS},
	{VVM_OP_add, VVM_OP_add,  R_OPERAND, AT_ANY, VVM_OP_add_add,  3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_sub, VVM_OP_cmplt, R_OPERAND1, AT_ANY, VVM_OP_sub_cmplt1, 3, 1, "a0a1b0b2", NO_VETO, 0, 0},
	{VVM_OP_sub, VVM_OP_neg,  R_OPERAND, AT_ANY, VVM_OP_sub_neg,  2, 1, "a0a1b1", NO_VETO, 0, 0},
	{VVM_OP_sub, VVM_OP_mul,  R_OPERAND, AT_ANY, VVM_OP_sub_mul,  3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_div, VVM_OP_mad,  R_OPERAND, AT_0_OR_1, VVM_OP_div_mad0, 4, 1, "a0a1bsb2b3", NO_VETO, 0, 0},      // the quotient must be a factor of the mad
	{VVM_OP_div, VVM_OP_f2i,  R_OPERAND, AT_ANY, VVM_OP_div_f2i,  2, 1, "a0a1b1", NO_VETO, 0, 0},
	{VVM_OP_div, VVM_OP_mul,  R_OPERAND, AT_ANY, VVM_OP_div_mul,  3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_muli, VVM_OP_addi, R_ANY, AT_ANY, VVM_OP_muli_addi, 3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_addi, VVM_OP_bit_rshift, R_ANY, AT_ANY, VVM_OP_addi_bit_rshift, 3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_addi, VVM_OP_muli, R_ANY, AT_ANY, VVM_OP_addi_muli, 3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_b2i, VVM_OP_b2i,  R_TWIN, AT_ANY, VVM_OP_b2i_2x, 1, 2, "a0a1b1", NO_VETO, 0, 0},
	{VVM_OP_i2f, VVM_OP_div,  R_OPERAND, AT_0,   VVM_OP_i2f_div0, 2, 1, "a0b1b2", NO_VETO, 0, 0},          // the converted value is the numerator
	{VVM_OP_i2f, VVM_OP_div,  R_OPERAND, AT_ANY, VVM_OP_i2f_div1, 2, 1, "a0b0b2", NO_VETO, 0, 0},
	{VVM_OP_i2f, VVM_OP_mul,  R_OPERAND, AT_ANY, VVM_OP_i2f_mul,  2, 1, CHAIN1, NO_VETO, 0, 0},
	{VVM_OP_i2f, VVM_OP_mad,  R_OPERAND, AT_0,   VVM_OP_i2f_mad0, 3, 1, "a0b1b2b3", NO_VETO, 0, 0},
	{VVM_OP_i2f, VVM_OP_mad,  R_OPERAND, AT_1,   VVM_OP_i2f_mad0, 3, 1, "a0b0b2b3", NO_VETO, 0, 0},
	{VVM_OP_i2f, VVM_OP_mad,  R_OPERAND, AT_ANY, VVM_OP_i2f_mad1, 3, 1, "b0b1a0b3", NO_VETO, 0, 0},
	{VVM_OP_f2i, VVM_OP_select, R_OPERAND1, AT_ANY, VVM_OP_f2i_select1, 3, 1, "b0a0b2b3", NO_VETO, 0, 0},
	{VVM_OP_f2i, VVM_OP_maxi, R_OPERAND, AT_ANY, VVM_OP_f2i_maxi, 2, 1, CHAIN1, NO_VETO, 0, 0},
	{VVM_OP_f2i, VVM_OP_addi, R_OPERAND, AT_ANY, VVM_OP_f2i_addi, 2, 1, CHAIN1, NO_VETO, 0, 0},
	{VVM_OP_bit_and, VVM_OP_i2f, R_ANY, AT_ANY, VVM_OP_bit_and_i2f, 2, 1, "a0a1b1", NO_VETO, 0, 0},
	{VVM_OP_random, VVM_OP_random, R_TWIN, AT_ANY, VVM_OP_random_2x, 1, 2, "a0a1b1", NO_VETO, 0, 0},
	{VVM_OP_select, VVM_OP_mul, R_ANY, AT_ANY, VVM_OP_select_mul, 4, 1, CHAIN3, NO_VETO, 0, 0},
	{VVM_OP_select, VVM_OP_add, R_ANY, AT_ANY, VVM_OP_select_add, 4, 1, CHAIN3, NO_VETO, 0, 0},
	{VVM_OP_sin, VVM_OP_cos, R_TWIN, AT_ANY, VVM_OP_sin_cos, 1, 2, "a0a1b1", NO_VETO, 0, 0},                 // (sine first in the merged form)
	{VVM_OP_cos, VVM_OP_sin, R_TWIN, AT_ANY, VVM_OP_sin_cos, 1, 2, "a0b1a1", NO_VETO, 0, 0},
	{VVM_OP_neg, VVM_OP_cmplt, R_OPERAND, AT_ANY, VVM_OP_neg_cmplt, 2, 1, "a0bkb2", NO_VETO, 0, 0},
	{VVM_OP_random, VVM_OP_add, R_OPERAND, AT_ANY, VVM_OP_random_add, 2, 1, "a0bkb2", NO_VETO, 0, 0},
	{VVM_OP_max, VVM_OP_f2i, R_OPERAND0, AT_ANY, VVM_OP_max_f2i, 2, 1, "a0a1b1", NO_VETO, 0, 0},
	{VVM_OP_bit_rshift, VVM_OP_bit_and, R_ANY, AT_ANY, VVM_OP_bit_rshift_bit_and, 3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_fmod, VVM_OP_add, R_ANY, AT_ANY, VVM_OP_fmod_add, 3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_bit_or, VVM_OP_muli, R_ANY, AT_ANY, VVM_OP_bit_or_muli, 3, 1, CHAIN2, NO_VETO, 0, 0},
	{VVM_OP_bit_lshift, VVM_OP_bit_or, R_ANY, AT_ANY, VVM_OP_bit_lshift_bit_or, 3, 1, CHAIN2, NO_VETO, 0, 0},
};
#undef CHAIN1
#undef CHAIN2
#undef CHAIN3
// the table above plus compare + logic_and / logic_or (the merged opcodes follow the compares' order, VectorVM.h:165-188)
static const std::vector<Rule>& rules()
{
	static const std::vector<Rule> all = [] {
		std::vector<Rule> v(std::begin(kRules), std::end(kRules));
		static const uint8_t compares[12] = {VVM_OP_cmplt, VVM_OP_cmple, VVM_OP_cmpgt, VVM_OP_cmpge, VVM_OP_cmpeq, VVM_OP_cmpneq,
		                                     VVM_OP_cmplti, VVM_OP_cmplei, VVM_OP_cmpgti, VVM_OP_cmpgei, VVM_OP_cmpeqi, VVM_OP_cmpneqi};
		for (int x = 0; x < 12; ++x) {
			v.push_back(Rule{compares[x], VVM_OP_logic_and, R_OPERAND, AT_ANY, (uint8_t)(VVM_OP_cmplt_logic_and + x), 3, 1, "a0a1bsb2", NO_VETO, 0, 0});
			v.push_back(Rule{compares[x], VVM_OP_logic_or, R_OPERAND, AT_ANY, (uint8_t)(VVM_OP_cmplt_logic_or + x), 3, 1, "a0a1bsb2", NO_VETO, 0, 0});
		}
		return v;
	}();
	return all;
}
}  // namespace merge

// Folds instruction `second` and the instruction `first` it reads into one, if a rule covers the pair.  `found`: what the pair
// search recorded (0: exec_index and its reader; 1 << x: operand x of second reads first; -1: twins).  Returns false when the
// pass has to end.
bool Optimizer::apply_merge_rule(uint32_t first, uint32_t second, int found)
{
	using namespace merge;
	Ins& a = ins[first];
	Ins& b = ins[second];
	const uint32_t ar = a.regs, br = b.regs;
	const uint32_t result = a.nin;                      // first's (first) result among its registers
	auto equal = [&](uint32_t ai, uint32_t bi) { return ssa[ar + ai] == ssa[br + bi] && cls[ar + ai] == cls[br + bi]; };
	for (const Rule& r : rules()) {
		if (r.first != a.op || r.second != b.op) continue;
		const bool reads_ok = r.reads == R_ANY || (r.reads == R_OPERAND && found >= 1) || (r.reads == R_OPERAND0 && found == 1) ||
		                      (r.reads == R_OPERAND1 && found == 2) || (r.reads == R_EXEC_INDEX && found == 0) || (r.reads == R_TWIN && found == -1);
		if (!reads_ok) continue;
		const bool at_ok = r.at == AT_ANY || (r.at == AT_0 && equal(result, 0)) || (r.at == AT_1 && equal(result, 1)) || (r.at == AT_2 && equal(result, 2)) ||
		                   (r.at == AT_0_OR_1 && (equal(result, 0) || equal(result, 1)));
		if (!at_ok) continue;
		if (r.flags & F_END_PASS) return false;
		if (r.dup_lo != NO_VETO) {
			// first's result twice among second's operands (e.g. mul(t, t)): no merged form (RegDupeCheck, :432-452)
			for (uint32_t x = r.dup_lo; x < r.dup_hi; ++x)
				if (ssa[br + x] == ssa[ar + result] && cls[br + x] == RT_TEMP)
					for (uint32_t y = x + 1; y <= r.dup_hi; ++y)
						if (ssa[br + x] == ssa[br + y] && cls[br + x] == cls[br + y]) return true;
		}
		const uint32_t count = (uint32_t)strlen(r.take) / 2;
		const uint32_t nr = new_regs((int)count);
		// exec_index + addi: not when both addends match the register that follows exec_index's result in the register list (the
		// reference compares register 1 of an instruction that has one, :1840-1846; kept, registers allocated first as there)
		if ((r.flags & F_EXEC_ADDI) && equal(1, 0) && equal(1, 1)) return true;
		for (uint32_t t = 0; t < count; ++t) {
			const char who = r.take[2 * t], which = r.take[2 * t + 1];
			uint32_t from;
			if (who == 'a') from = ar + (uint32_t)(which - '0');
			else if (which == 's') from = br + (equal(result, 0) ? 1u : 0u);
			else if (which == 'k') from = br + (found == 1 ? 1u : 0u);
			else from = br + (uint32_t)(which - '0');
			reg[nr + t] = reg[from]; ssa[nr + t] = ssa[from]; cls[nr + t] = cls[from];
		}
		a.merged_into = (int)second;
		b.op = r.merged; b.nin = r.nin; b.nout = r.nout; b.regs = nr;
		return true;
	}
	return true;
}

// Steps 13-14 (:1582-2518): pairs of ops where the second reads the first one's result (and nothing else does) are folded into
// the merged opcodes of VectorVM.h:153-231.
void Optimizer::merge_ops()
{
	struct Pair { uint32_t first, second; int kind; };       // kind 0: exec_index + reader; else 1 << (operand of `second` that reads `first`)
	std::vector<Pair> pairs;
	for (size_t i = 0; i < ins.size(); ++i) {
		const Ins& a = ins[i];
		if (a.op == VVM_OP_exec_index) {
			// merges with its one reader if that is an i2f or addi
			const uint16_t name = ssa[a.regs];
			int reader = -1;
			bool ok = true;
			for (size_t j = i + 1; j < ins.size() && ok; ++j)
				for (int k = 0; k < ins[j].nin; ++k) {
					if (cls[ins[j].regs + k] != RT_TEMP || ssa[ins[j].regs + k] != name) continue;
					if ((ins[j].op == VVM_OP_i2f || ins[j].op == VVM_OP_addi) && reader == -1) { reader = (int)j; break; }
					ok = false;
					break;
				}
			if (ok && reader != -1) pairs.push_back(Pair{(uint32_t)i, (uint32_t)reader, 0});
			continue;
		}
		if (a.cat != C_OP) continue;
		// The reference's search for two ops with the same inputs (the *_2x, sin_cos forms) compares against instruction 1
		// of the script whatever `i` is, and on a match leaves the whole search (:1644-1665): it never yields a pair, and when
		// instruction 1 is an op it can end the search early.  Kept as it is.
		if (ins.size() > 1 && ins[1].cat == C_OP && a.nin == ins[1].nin) {
			bool differs = false;
			for (int k = 0; k < a.nin && !differs; ++k)
				differs = cls[a.regs + k] == RT_TEMP && cls[ins[1].regs + k] == RT_TEMP && ssa[a.regs + k] != ssa[ins[1].regs + k];
			if (!differs) break;
		}
		for (size_t j = i + 1; j < ins.size(); ++j) {
			const Ins& b = ins[j];
			if (b.cat != C_OP || b.nin >= 4) continue;
			int kind = 0;
			bool ok = true;
			for (int k = 0; k < a.nout && ok; ++k) {
				if (cls[a.regs + a.nin + k] != RT_TEMP) continue;
				const uint16_t name = ssa[a.regs + a.nin + k];
				for (int x = 0; x < b.nin; ++x) {
					if (cls[b.regs + x] != RT_TEMP || ssa[b.regs + x] != name) continue;
					if (kind) { ok = false; break; }          // read twice, e.g. mul(a, a): no merged form
					kind = 1 << x;
				}
			}
			if (!ok || !kind) continue;
			// the result must not be read again further down
			for (size_t k = j + 1; k < ins.size() && ok; ++k)
				for (int o = 0; o < a.nout && ok; ++o)
					for (int x = 0; x < ins[k].nin; ++x)
						if (cls[ins[k].regs + x] == RT_TEMP && ssa[ins[k].regs + x] == ssa[a.regs + a.nin + o]) { ok = false; break; }
			// nor by anything in between (compared by name only, for as many results as the in-between instruction has)
			for (size_t k = i + 1; k < j && ok; ++k)
				for (int o = 0; o < ins[k].nout && ok; ++o)
					for (int x = 0; x < ins[k].nin; ++x)
						if (ssa_at(ins[k].regs + x) == ssa_at(a.regs + a.nin + o)) { ok = false; break; }
			if (ok) pairs.push_back(Pair{(uint32_t)i, (uint32_t)j, kind});
		}
	}

	for (const Pair& pr : pairs) {
		if (ins[pr.first].merged_into != -1 || ins[pr.second].merged_into != -1) continue;
		if (!apply_merge_rule(pr.first, pr.second, pr.kind)) break;
	}
}

// Step 15 (:2525-2774): the distinct inputs, sorted by (dataset, kind, index), become the remapped inputs, with per-dataset start
// offsets of each kind; the distinct outputs fill the tables indexed by output serial number.
void Optimizer::build_io_tables()
{
	std::vector<const Ins*> inputs, outputs;
	std::vector<std::pair<int, const Ins*>> input_alias;     // a repeated read of one input: its index -> the first read
	for (const Ins& in : ins) {
		if (in.cat == C_INPUT) {
			auto it = std::find_if(inputs.begin(), inputs.end(), [&](const Ins* o) { return o->op == in.op && o->dataset == in.dataset && o->input_idx == in.input_idx; });
			if (it == inputs.end()) inputs.push_back(&in);
			else input_alias.push_back({in.index, *it});      // (the reference leaves this case's mapping unset)
		} else if (in.cat == C_OUTPUT) {
			auto it = std::find_if(outputs.begin(), outputs.end(), [&](const Ins* o) { return o->op == in.op && o->dataset == in.dataset && o->dst_reg == in.dst_reg; });
			if (it == outputs.end()) outputs.push_back(&in);
		}
	}
	if (!inputs.empty()) {
		auto key = [](const Ins* i) { return ((uint32_t)i->dataset << 24) | ((uint32_t)(i->op - VVM_OP_inputdata_float) << 20) | i->input_idx; };
		std::sort(inputs.begin(), inputs.end(), [&](const Ins* a, const Ins* b) { return key(a) < key(b); });
		int max_dataset = 0, max_index = 0;
		for (const Ins* i : inputs) { max_dataset = std::max<int>(max_dataset, i->dataset); max_index = std::max(max_index, i->index); }
		for (auto& al : input_alias) max_index = std::max(max_index, al.first);
		std::vector<uint16_t> slot_of(max_index + 1, 0);
		num_input_datasets = (uint16_t)(max_dataset + 1);
		// per dataset: [0..5] first slot of float, int32, half, no-advance float, int32, half; [6] end; [7] number of no-advance inputs
		input_dataset_offsets.assign((size_t)num_input_datasets * 8, 0);
		int prev_kind = 0, prev_dataset = 0;
		const int last_kind = VVM_OP_inputdata_noadvance_half - VVM_OP_inputdata_float;
		for (size_t i = 0; i < inputs.size(); ++i) {
			const Ins* in = inputs[i];
			const int kind = in->op - VVM_OP_inputdata_float;
			slot_of[in->index] = (uint16_t)i;
			if (in->dataset != prev_dataset) {
				for (int j = prev_kind + 1; j <= last_kind + 1; ++j) input_dataset_offsets[prev_dataset * 8 + j] = (uint16_t)i;
				for (int d = prev_dataset + 1; d < in->dataset; ++d)
					for (int k = 0; k < 8; ++k) input_dataset_offsets[d * 8 + k] = (uint16_t)i;
				for (int j = 0; j <= kind; ++j) input_dataset_offsets[in->dataset * 8 + j] = (uint16_t)i;
				prev_kind = kind; prev_dataset = in->dataset;
			} else if (kind != prev_kind) {
				for (int j = prev_kind + 1; j <= kind; ++j) input_dataset_offsets[in->dataset * 8 + j] = (uint16_t)i;
				prev_kind = kind;
			}
		}
		for (int j = prev_kind + 1; j <= last_kind + 1; ++j) input_dataset_offsets[prev_dataset * 8 + j] = (uint16_t)inputs.size();
		for (int d = 0; d < num_input_datasets; ++d) input_dataset_offsets[d * 8 + 7] = input_dataset_offsets[d * 8 + 6] - input_dataset_offsets[d * 8 + 3];
		for (auto& al : input_alias) slot_of[al.first] = slot_of[al.second->index];
		input_remap.resize(inputs.size());
		for (size_t i = 0; i < inputs.size(); ++i) input_remap[i] = inputs[i]->input_idx;
		for (size_t i = 0; i < reg.size(); ++i)
			if (cls[i] == RT_INPUT) reg[i] = ssa[i] < slot_of.size() ? slot_of[ssa[i]] : 0;
	}
	if (!outputs.empty()) {
		int largest = 0;
		for (const Ins* o : outputs) largest = std::max<int>(largest, o->serial);
		out_dataset.assign(largest + 1, 0);
		out_type.assign(largest + 1, 0);
		out_dst.assign(largest + 1, NONE16);           // a repeated output leaves its later serial numbers without a destination
		for (const Ins* o : outputs) {
			out_dataset[o->serial] = (uint8_t)o->dataset;
			out_type[o->serial] = (o->op == VVM_OP_outputdata_int32) ? 1 : (o->op == VVM_OP_outputdata_half || o->op == VVM_OP_outputdata_half_from_half) ? 2 : 0;
			out_dst[o->serial] = o->dst_reg;
		}
		num_outputs_remapped = (uint16_t)(largest + 1);
		num_output_instructions = (uint16_t)outputs.size();
	}
}

// Step 16 (:2776-2853): temp registers.  Every value lives from the instruction that writes it to the last one that reads it; walking
// the program from its end, a value's register is claimed at its last reader and handed back at its writer, and a claim always takes
// the lowest register that is free (the numbering the reference's output has, so it is not a free choice).  The most registers
// claimed at once is the script's temp count.
namespace {
// Registers 0, 1, 2, ... with "lowest free one first": those handed back wait in a min-heap, the ones never claimed lie above `fresh`.
class LowestFreeRegister {
public:
	uint16_t claim()
	{
		uint16_t r;
		if (!returned.empty() && returned.front() < fresh) { std::pop_heap(returned.begin(), returned.end(), std::greater<uint16_t>()); r = returned.back(); returned.pop_back(); }
		else r = fresh++;
		peak = std::max<uint32_t>(peak, ++claimed);
		return r;
	}
	void hand_back(uint16_t r) { returned.push_back(r); std::push_heap(returned.begin(), returned.end(), std::greater<uint16_t>()); --claimed; }
	uint32_t most_claimed() const { return peak; }
private:
	std::vector<uint16_t> returned;
	uint16_t fresh = 0;
	uint32_t claimed = 0, peak = 0;
};
}  // namespace

void Optimizer::minimise_temps()
{
	LowestFreeRegister pool;
	std::vector<uint16_t> home(num_ssa, NONE16);      // per value: the register it occupies between its writer and its last reader
	for (size_t back = ins.size(); back-- > 0;) {
		const Ins& in = ins[back];
		const bool takes_part = in.cat != C_INPUT && in.merged_into == -1 && in.nin + in.nout != 0;
		if (!takes_part) continue;
		// results first: the value's life starts here, so its register is free above this instruction -- also for this instruction's own
		// operands, which is how `a = f(a)` keeps one register
		for (uint32_t at = in.regs + in.nin, end = at + in.nout; at < end; ++at) {
			if (cls[at] != RT_TEMP) continue;
			const uint16_t value = ssa[at];
			const bool read_later = value < num_ssa && home[value] != NONE16;      // (NONE16 itself is no value: it fails the first test)
			reg[at] = read_later ? home[value] : NONE16;      // never read: e.g. an external function's unused output
			if (read_later) { pool.hand_back(home[value]); home[value] = NONE16; }
		}
		for (uint32_t at = in.regs, end = at + in.nin; at < end; ++at) {
			if (cls[at] != RT_TEMP || ssa[at] >= num_ssa) continue;
			uint16_t& r = home[ssa[at]];
			if (r == NONE16) r = pool.claim();      // the value's last reader
			reg[at] = r;
		}
	}
	num_temps = pool.most_claimed();
}

// Step 17 (:2855-2978): output buffers instead of temp registers.  Going through the program from its end, every temp REGISTER (not
// value: the decision holds for everything the register carries higher up, as in the reference) ends in one of three states:
//   - it feeds a float / int32 output instruction: from there upwards the register IS that output's buffer;
//   - it is met in another instruction while the buffer of an output that is not fed by a temp is unclaimed: it becomes that buffer
//     (the buffers are claimed last-found-first);
//   - otherwise it stays a temp; the temps that stay are renumbered densely.
// One walk, with the decisions in a table that each instruction's operands are translated through on the way up.
void Optimizer::alias_outputs()
{
	const size_t n = std::max<size_t>(num_outputs_remapped, num_temps);
	final_temp.resize(n);
	for (size_t i = 0; i < n; ++i) final_temp[i] = (uint16_t)i;
	if (num_outputs_remapped == 0) return;

	enum : uint8_t { UNDECIDED, STAYS, IS_OUTPUT };
	struct Fate { uint8_t what = UNDECIDED; uint16_t serial = 0; };
	std::vector<Fate> fate(num_temps);
	std::vector<uint16_t> unclaimed;      // serials of outputs whose buffers may serve as scratch, in the order found
	auto temp_at = [&](uint32_t at) -> Fate* { return (at < reg.size() && cls[at] == RT_TEMP && reg[at] < num_temps) ? &fate[reg[at]] : nullptr; };
	auto translate = [&](uint32_t at) {      // an operand whose register has become an output buffer names that buffer
		if (Fate* f = temp_at(at); f && f->what == IS_OUTPUT) { reg[at] = f->serial; cls[at] = RT_OUTPUT; }
	};

	for (size_t back = ins.size(); back-- > 0;) {
		const Ins& in = ins[back];
		const uint32_t first = in.regs, count = (uint32_t)(in.nin + in.nout);
		for (uint32_t k = 0; k < count; ++k) translate(first + k);      // decisions made further down the program
		if (in.merged_into != -1 || in.cat == C_INPUT) continue;
		if (in.op == VVM_OP_outputdata_float || in.op == VVM_OP_outputdata_int32) {
			const uint32_t src = first + 1;
			if (cls[src] != RT_TEMP) {      // a constant, an input, or a register that already is some output's buffer
				if (std::find(unclaimed.begin(), unclaimed.end(), in.serial) == unclaimed.end()) unclaimed.push_back(in.serial);
			} else if (Fate* f = temp_at(src); f && f->what == UNDECIDED) {
				*f = Fate{IS_OUTPUT, in.serial};
				translate(src);      // (this operand only: the instruction's other operand was looked at before the decision)
			}
			continue;
		}
		// acquireindex: its result only orders instructions, nothing reads it at run time: it takes no part in the decisions
		const uint32_t deciding = in.op == VVM_OP_acquireindex ? count - 1 : count;
		for (uint32_t k = 0; k < deciding; ++k) {
			Fate* f = temp_at(first + k);
			if (!f || f->what != UNDECIDED) { translate(first + k); continue; }
			if (unclaimed.empty()) { f->what = STAYS; continue; }
			*f = Fate{IS_OUTPUT, unclaimed.back()};
			unclaimed.pop_back();
			translate(first + k);
		}
		for (uint32_t k = deciding; k < count; ++k) translate(first + k);
	}
	uint16_t kept = 0;
	for (uint32_t t = 0; t < num_temps; ++t)
		if (fate[t].what == STAYS) final_temp[t] = kept++;
	num_temps = kept;
}

// Step 18 (:2980-3160): the optimized stream.  Register operands are indices into [temps | constants | inputs | input copies |
// outputs] (VVMOptWriteReg_, :534-564).
void Optimizer::emit(const vvm_ext_fn_io* ext)
{
	const uint16_t C = (uint16_t)const_remap.size(), I = (uint16_t)input_remap.size();
	auto put8 = [&](uint32_t v) { bytecode.push_back((uint8_t)v); };
	auto put16 = [&](uint32_t v) { bytecode.push_back((uint8_t)(v & 0xFF)); bytecode.push_back((uint8_t)(v >> 8)); };
	auto put_reg = [&](uint32_t at) {
		uint16_t r = reg[at];
		const uint8_t c = cls[at];
		if (r == NONE16 || c == RT_INVALID) { put16(NONE16); return; }
		if (c == RT_CONST) r += (uint16_t)num_temps;
		else if (c == RT_INPUT) r += (uint16_t)(num_temps + C);
		else if (c == RT_OUTPUT) r += (uint16_t)(num_temps + C + 2 * I);
		else r = r < final_temp.size() ? final_temp[r] : r;
		put16(r);
	};
	for (size_t i = 0; i < ins.size(); ++i) {
		const Ins& in = ins[i];
		if (in.merged_into != -1) continue;
		if (in.op == VVM_OP_random || in.op == VVM_OP_randomi || in.op == VVM_OP_random_add || in.op == VVM_OP_random_2x)
			flags |= VVM_FLAG_HAS_RAND_INSTRUCTION;
		switch (in.cat) {
		case C_INPUT: case C_STAT:
			break;
		case C_OUTPUT: {
			// adjacent outputs of one kind, dataset and index register share an instruction, up to 255
			int count = 1;
			for (size_t j = i + 1; j < ins.size(); ++j) {
				const Ins& nx = ins[j];
				if (!(nx.op == in.op && nx.dataset == in.dataset && ssa[nx.regs] == ssa[in.regs])) break;
				if (++count >= 0xFF) break;
			}
			put8(in.op); put8(count); put8(in.dataset);
			for (int j = 0; j < count; ++j) put_reg(ins[i + j].regs + 1);
			for (int j = 0; j < count; ++j) put16(num_temps + C + 2 * I + ins[i + j].serial);
			i += count - 1;
			break;
		}
		case C_OP:
			put8(in.op);
			for (int j = 0; j < in.nin + in.nout; ++j) put_reg(in.regs + j);
			break;
		case C_INDEXGEN:
			put8(in.op); put_reg(in.regs); put8(in.dataset);
			break;
		case C_EXTFN: {
			uint32_t dummies = 0;
			put8(in.op); put16(in.fn);
			for (int j = 0; j < ext[in.fn].NumInputs + ext[in.fn].NumOutputs; ++j) {
				if (reg[in.regs + j] == NONE16) ++dummies;
				put_reg(in.regs + j);
			}
			num_dummy_regs = std::max(num_dummy_regs, dummies);
			break;
		}
		case C_RWBUFFER:
			put8(in.op); put_reg(in.regs); put_reg(in.regs + 1); put8(in.dataset);
			break;
		case C_OTHER:
			if (in.op == VVM_OP_done) {
				for (int k = 0; k < 8; ++k) put8(VVM_OP_done);       // the dispatch loop reads ahead of the instruction it executes
				i = ins.size();
			}
			break;
		}
	}
	if (!bytecode.empty()) put8(0);
}

size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }

}  // namespace

// names of the opcodes in the order of the enum in include/vvm_abi.h (VectorVM.h:50-233)
static const char* const kOpNames[179] = {
	"done", "add", "sub", "mul", "div", "mad", "lerp", "rcp",
	"rsq", "sqrt", "neg", "abs", "exp", "exp2", "log", "log2",
	"sin", "cos", "tan", "asin", "acos", "atan", "atan2", "ceil",
	"floor", "fmod", "frac", "trunc", "clamp", "min", "max", "pow",
	"round", "sign", "step", "random", "noise", "cmplt", "cmple", "cmpgt",
	"cmpge", "cmpeq", "cmpneq", "select", "addi", "subi", "muli", "divi",
	"clampi", "mini", "maxi", "absi", "negi", "signi", "randomi", "cmplti",
	"cmplei", "cmpgti", "cmpgei", "cmpeqi", "cmpneqi", "bit_and", "bit_or", "bit_xor",
	"bit_not", "bit_lshift", "bit_rshift", "logic_and", "logic_or", "logic_xor", "logic_not", "f2i",
	"i2f", "f2b", "b2f", "i2b", "b2i", "inputdata_float", "inputdata_int32", "inputdata_half",
	"inputdata_noadvance_float", "inputdata_noadvance_int32", "inputdata_noadvance_half", "outputdata_float", "outputdata_int32", "outputdata_half", "acquireindex", "external_func_call",
	"exec_index", "noise2D", "noise3D", "enter_stat_scope", "exit_stat_scope", "update_id", "acquire_id", "half_to_float",
	"fasi", "iasf", "exec_indexf", "exec_index_addi", "cmplt_select", "cmple_select", "cmpeq_select", "cmplti_select",
	"cmplei_select", "cmpeqi_select", "cmplt_logic_and", "cmple_logic_and", "cmpgt_logic_and", "cmpge_logic_and", "cmpeq_logic_and", "cmpne_logic_and",
	"cmplti_logic_and", "cmplei_logic_and", "cmpgti_logic_and", "cmpgei_logic_and", "cmpeqi_logic_and", "cmpnei_logic_and", "cmplt_logic_or", "cmple_logic_or",
	"cmpgt_logic_or", "cmpge_logic_or", "cmpeq_logic_or", "cmpne_logic_or", "cmplti_logic_or", "cmplei_logic_or", "cmpgti_logic_or", "cmpgei_logic_or",
	"cmpeqi_logic_or", "cmpnei_logic_or", "mad_add", "mad_sub0", "mad_sub1", "mad_mul", "mad_sqrt", "mad_mad0",
	"mad_mad1", "mul_mad0", "mul_mad1", "mul_add", "mul_sub0", "mul_sub1", "mul_mul", "mul_max",
	"mul_2x", "add_mad1", "add_add", "sub_cmplt1", "sub_neg", "sub_mul", "div_mad0", "div_f2i",
	"div_mul", "muli_addi", "addi_bit_rshift", "addi_muli", "b2i_2x", "i2f_div0", "i2f_div1", "i2f_mul",
	"i2f_mad0", "i2f_mad1", "f2i_select1", "f2i_maxi", "f2i_addi", "fmod_add", "bit_and_i2f", "bit_rshift_bit_and",
	"neg_cmplt", "bit_or_muli", "bit_lshift_bit_or", "random_add", "random_2x", "max_f2i", "select_mul", "select_add",
	"sin_cos", "outputdata_float_from_half", "outputdata_half_from_half",
};

// the reference's eighteen steps in order; returns its error bits
static uint32_t run_passes(Optimizer& o, const uint8_t* bytecode, int32_t num_bytes, const vvm_ext_fn_io* ext_fns, int32_t num_ext_fns)
{
	uint32_t err = (uint32_t)o.decode(bytecode, num_bytes, ext_fns, num_ext_fns);       // steps 1-3
	if (err) return err;
	o.rename_ssa();                                                                   // steps 4-5
	if (o.remove_unused_ops()) o.rename_ssa();                                        // step 6
	o.remove_casts();                                                                 // step 7
	const int one_past_last_input = o.fold_inputs();                                  // step 8
	o.dump("before reorder");
	o.reorder(one_past_last_input);                                                   // steps 9-11
	o.dump("after reorder");
	o.sort_output_runs();                                                             // step 12
	o.dump("after output sort");
	o.merge_ops();                                                                    // steps 13-14
	o.build_io_tables();                                                              // step 15
	o.minimise_temps();                                                               // step 16
	o.alias_outputs();                                                                // step 17
	o.dump("before emit");
	o.emit(ext_fns);                                                                  // step 18
	return o.error;
}

// GenerateHumanReadableScript (VectorVMOptimizer.h:18, VectorVMOptimizer.cpp:104-253): the optimizer's instruction list after all
// passes, one line per instruction that was not folded into another one, in the reference's layout: a summary, the (empty)
// constant table section, then "<position>\t| <text>;" with registers written as R / C / I / O [index] by their class.  The
// reference prints op names in editor builds and __OP__<number> otherwise; names here.  It needs the optimizer's intermediate
// state (there: VVMFlag_OptSaveIntermediateState keeps it), so the script is optimized again for the listing.
extern "C" VVM_API int vvm_generate_human_readable_script(const uint8_t* bytecode, int32_t num_bytes, const vvm_ext_fn_io* ext_fns, int32_t num_ext_fns,
                                                          uint32_t flags, char** text, size_t* text_bytes, uint32_t* error_flags)
{
	if (text) *text = nullptr;
	if (text_bytes) *text_bytes = 0;
	if (error_flags) *error_flags = 0;
	if (!text || num_ext_fns < 0 || (num_ext_fns > 0 && !ext_fns)) return VVM_E_INVALID_ARG;
	Optimizer o;
	o.flags = flags;
	if (bytecode && num_bytes > 1) {
		const uint32_t err = run_passes(o, bytecode, num_bytes, ext_fns, num_ext_fns);
		if (err) {
			if (error_flags) *error_flags = err;
			return VVM_E_BAD_BLOB;
		}
	}
	std::string out;
	char line[96];
	auto registers = [&](uint32_t first, int count) {
		for (int k = 0; k < count; ++k) {
			if (k) out += ", ";
			static const char letter[4] = {'R', 'C', 'I', 'O'};
			const uint8_t c = o.cls[first + (uint32_t)k];
			if (c < 4) out += letter[c];
			snprintf(line, sizeof line, "[%u]", (unsigned)o.reg[first + (uint32_t)k]);
			out += line;
		}
	};
	out += "\n-------------------------------\nSummary\n-------------------------------\n";
	snprintf(line, sizeof line, "Num Byte Code Ops: %zu\n", o.ins.size()); out += line;
	snprintf(line, sizeof line, "Num Constants: %zu\n", o.const_remap.size()); out += line;
	out += "\n-------------------------------\nConstant Table\n-------------------------------\n";
	out += "-------------------------------\n";
	snprintf(line, sizeof line, "Byte Code (%zu Ops)\n", o.ins.size()); out += line;
	out += "-------------------------------\n";
	for (size_t i = 0; i < o.ins.size(); ++i) {
		const Ins& in = o.ins[i];
		if (in.merged_into != -1) continue;
		snprintf(line, sizeof line, "%zu\t| ", i); out += line;
		const char* name = in.op < 179 ? kOpNames[in.op] : "?";
		switch (in.cat) {
		case C_INPUT:
			registers(in.regs + (uint32_t)in.nin, 1); out += " = "; out += name; out += "()";
			break;
		case C_OUTPUT:
			out += name;
			snprintf(line, sizeof line, "(%u, %u, ", (unsigned)in.dataset, (unsigned)in.dst_reg); out += line;
			registers(in.regs, in.nin); out += ")";
			break;
		case C_OP: case C_INDEXGEN:
			registers(in.regs + (uint32_t)in.nin, 1); out += " = "; out += name; out += "(";
			registers(in.regs, in.nin); out += ")";
			break;
		case C_EXTFN:
			out += name; out += "(";
			registers(in.regs, in.nin);
			registers(in.regs + (uint32_t)in.nin, in.nout);      // (no separator between the two lists, as in the reference: :182-183)
			out += ")";
			break;
		default:
			out += name; out += "(";
			registers(in.regs, in.nin); out += ")";
			break;
		}
		out += ";\n";
	}
	out += "-------------------------------\n";
	char* copy = (char*)malloc(out.size() + 1);
	if (!copy) { if (error_flags) *error_flags = VVM_OPT_ERR_OUT_OF_MEMORY; return VVM_E_TOO_LARGE; }
	memcpy(copy, out.c_str(), out.size() + 1);
	*text = copy;
	if (text_bytes) *text_bytes = out.size();
	return VVM_OK;
}

extern "C" VVM_API int vvm_optimize_script(const uint8_t* bytecode, int32_t num_bytes, const vvm_ext_fn_io* ext_fns, int32_t num_ext_fns,
                                           uint64_t hash_id, uint32_t flags, uint8_t** blob, size_t* blob_bytes, uint32_t* error_flags)
{
	if (blob) *blob = nullptr;
	if (blob_bytes) *blob_bytes = 0;
	if (error_flags) *error_flags = 0;
	if (!blob || !blob_bytes || num_ext_fns < 0 || (num_ext_fns > 0 && !ext_fns)) return VVM_E_INVALID_ARG;
	if (!bytecode || num_bytes <= 1) return VVM_OK;                 // an empty script optimizes to nothing (:703-712)

	Optimizer o;
	o.flags = flags;
	const uint32_t err = run_passes(o, bytecode, num_bytes, ext_fns, num_ext_fns);
	if (err) {
		if (error_flags) *error_flags = err;
		return VVM_E_BAD_BLOB;
	}

	// FreezeOptimizerContext: 54 packed header bytes, then 16-byte aligned sections (VectorVMBridge.cpp:81-178)
	const size_t C = o.const_remap.size(), I = o.input_remap.size(), O = o.num_outputs_remapped, E = (size_t)num_ext_fns;
	const size_t off_bc  = VVM_BLOB_FIRST_SECTION;
	const size_t off_cr  = align16(off_bc + o.bytecode.size());
	const size_t off_ir  = align16(off_cr + 2 * C);
	const size_t off_ido = align16(off_ir + 2 * I);
	const size_t off_ods = align16(off_ido + 16 * (size_t)o.num_input_datasets);
	const size_t off_ot  = align16(off_ods + O);
	const size_t off_od  = align16(off_ot + 2 * O);
	const size_t off_ext = align16(off_od + 2 * O);
	const size_t total   = align16(off_ext + (size_t)VVM_BLOB_EXTFN_ENTRY_BYTES * E);
	uint8_t* out = (uint8_t*)calloc(total, 1);
	if (!out) { if (error_flags) *error_flags = VVM_OPT_ERR_OUT_OF_MEMORY; return VVM_E_TOO_LARGE; }
	uint8_t* h = out;
	auto w16 = [&](uint16_t v) { memcpy(h, &v, 2); h += 2; };
	auto w32 = [&](uint32_t v) { memcpy(h, &v, 4); h += 4; };
	w32((uint32_t)o.bytecode.size()); w32(o.max_output_dataset); w16(o.num_consts_alloced); w32(o.num_temps);
	w16((uint16_t)C); w16((uint16_t)I); w16(o.num_noadvance); w16(o.num_input_datasets); w16(o.num_outputs_remapped);
	w16(o.num_output_instructions); w32((uint32_t)E); w32(E ? o.max_ext_fn_registers : 0); w32(o.num_dummy_regs);
	w32((uint32_t)o.max_ext_fn_used); w32(o.flags);
	memcpy(h, &hash_id, 8);
	if (!o.bytecode.empty()) memcpy(out + off_bc, o.bytecode.data(), o.bytecode.size());
	if (C) memcpy(out + off_cr, o.const_remap.data(), 2 * C);
	if (I) memcpy(out + off_ir, o.input_remap.data(), 2 * I);
	if (o.num_input_datasets) memcpy(out + off_ido, o.input_dataset_offsets.data(), 16 * (size_t)o.num_input_datasets);
	if (O) {
		memcpy(out + off_ods, o.out_dataset.data(), O);
		memcpy(out + off_ot, o.out_type.data(), 2 * O);
		memcpy(out + off_od, o.out_dst.data(), 2 * O);
	}
	for (size_t i = 0; i < E; ++i) {       // {function pointer (null in a frozen blob), NumInputs, NumOutputs}
		memcpy(out + off_ext + VVM_BLOB_EXTFN_ENTRY_BYTES * i + 8, &ext_fns[i].NumInputs, 4);
		memcpy(out + off_ext + VVM_BLOB_EXTFN_ENTRY_BYTES * i + 12, &ext_fns[i].NumOutputs, 4);
	}
	*blob = out;
	*blob_bytes = total;
	return VVM_OK;
}

extern "C" VVM_API void vvm_free_blob(uint8_t* blob) { free(blob); }
extern "C" VVM_API void vvm_free_text(char* text) { free(text); }
