// vvm_program.cpp -- blob thaw, optimized-bytecode decoding and linking into the device program.
//
// Reference behaviour reproduced here:
//   ThawRuntimeContext                         VectorVMBridge.cpp:181-199 (layout :81-148)
//   register index space of the bytecode       VectorVMRuntime.cpp:345-349 (temps | consts | inputs | input copies | outputs)
//   instruction encodings                      VectorVMRuntime.cpp:555-569 (outputs), :826-828 (acquireindex), :1444-1450
//                                              (external call), :1131-1134 / :1406 (ids), :1558-1904 (ops)
//   input classification                       VectorVMRuntime.cpp:510-552 (VVMBuildMapTableCaches)
#include "vvm_program.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace vvm {

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static size_t al16(size_t n) { return (n + 15) & ~(size_t)15; }

int thaw_blob(const uint8_t* blob, size_t n, ThawedScript& s, std::string& err)
{
	if (!blob || n == 0) { err = "empty blob"; return VVM_E_BAD_BLOB; }   // RT.cpp:2313-2316
	if (n < VVM_BLOB_FIRST_SECTION) { err = "blob shorter than its header"; return VVM_E_BAD_BLOB; }
	const uint8_t* h = blob;
	// packed archive order, VectorVMBridge.cpp:81-101
	const uint32_t num_bytecode = rd32(h + 0);
	s.max_output_dataset = rd32(h + 4);
	/* NumConstsAlloced = rd16(h + 8) */
	s.num_temps = rd32(h + 10);
	s.num_consts = rd16(h + 14);
	s.num_inputs = rd16(h + 16);
	s.num_noadvance = rd16(h + 18);
	s.num_input_datasets = rd16(h + 20);
	s.num_outputs_remapped = rd16(h + 22);
	s.num_output_instructions = rd16(h + 24);
	s.num_ext_fns = rd32(h + 26);
	s.max_ext_fn_registers = rd32(h + 30);
	s.num_dummy_regs = rd32(h + 34);
	/* MaxExtFnUsed = rd32(h + 38) */
	s.flags = rd32(h + 42) & ~(uint32_t)VVM_FLAG_DATA_MAP_CACHE_SETUP;   // RT.cpp:2384
	s.hash_id = rd64(h + 46);

	const size_t off_bc  = VVM_BLOB_FIRST_SECTION;
	const size_t off_cr  = al16(off_bc + num_bytecode);
	const size_t off_ir  = al16(off_cr + 2u * (size_t)s.num_consts);
	const size_t off_ido = al16(off_ir + 2u * (size_t)s.num_inputs);
	const size_t off_ods = al16(off_ido + 16u * (size_t)s.num_input_datasets);
	const size_t off_ot  = al16(off_ods + (size_t)s.num_outputs_remapped);
	const size_t off_od  = al16(off_ot + 2u * (size_t)s.num_outputs_remapped);
	const size_t off_ext = al16(off_od + 2u * (size_t)s.num_outputs_remapped);
	const size_t total   = al16(off_ext + (size_t)VVM_BLOB_EXTFN_ENTRY_BYTES * s.num_ext_fns);
	if (total > n) { err = "blob sections run past its end"; return VVM_E_BAD_BLOB; }
	if (s.max_output_dataset > 255 || s.num_ext_fns > 4096) { err = "implausible header"; return VVM_E_BAD_BLOB; }

	s.bytecode.assign(blob + off_bc, blob + off_bc + num_bytecode);
	auto copy16 = [&](std::vector<uint16_t>& v, size_t off, size_t count) {
		v.resize(count);
		if (count) memcpy(v.data(), blob + off, 2 * count);
	};
	copy16(s.const_remap, off_cr, s.num_consts);
	copy16(s.input_remap, off_ir, s.num_inputs);
	copy16(s.input_dataset_offsets, off_ido, 8u * (size_t)s.num_input_datasets);
	s.out_dataset.assign(blob + off_ods, blob + off_ods + s.num_outputs_remapped);
	copy16(s.out_type, off_ot, s.num_outputs_remapped);
	copy16(s.out_dst, off_od, s.num_outputs_remapped);
	s.ext_num_inputs.resize(s.num_ext_fns);
	s.ext_num_outputs.resize(s.num_ext_fns);
	for (uint32_t i = 0; i < s.num_ext_fns; ++i) {
		const uint8_t* e = blob + off_ext + (size_t)VVM_BLOB_EXTFN_ENTRY_BYTES * i;
		s.ext_num_inputs[i] = (int32_t)rd32(e + 8);
		s.ext_num_outputs[i] = (int32_t)rd32(e + 12);
	}
	return VVM_OK;
}

// ---- opcode table ------------------------------------------------------------------------------------
enum OpKind : uint8_t { K_INVALID = 0, K_ORD, K_DST_ONLY, K_ACQUIRE, K_OUTPUT, K_EXTFN, K_IDS, K_DONE };
struct OpInfo { uint8_t kind, nin, nout; bool heavy; };

static OpInfo g_ops[256];
static std::once_flag g_ops_once;

static void fill_op_table()
{
	memset(g_ops, 0, sizeof(g_ops));
	auto set = [](std::initializer_list<int> ops, uint8_t nin, uint8_t nout) {
		for (int o : ops) g_ops[o] = OpInfo{K_ORD, nin, nout, false};
	};
	set({VVM_OP_rcp, VVM_OP_rsq, VVM_OP_sqrt, VVM_OP_neg, VVM_OP_abs, VVM_OP_exp, VVM_OP_exp2, VVM_OP_log, VVM_OP_log2, VVM_OP_sin,
	     VVM_OP_cos, VVM_OP_tan, VVM_OP_asin, VVM_OP_acos, VVM_OP_atan, VVM_OP_ceil, VVM_OP_floor, VVM_OP_frac, VVM_OP_trunc,
	     VVM_OP_round, VVM_OP_sign, VVM_OP_absi, VVM_OP_negi, VVM_OP_signi, VVM_OP_bit_not, VVM_OP_logic_not, VVM_OP_f2i, VVM_OP_i2f,
	     VVM_OP_f2b, VVM_OP_b2f, VVM_OP_i2b, VVM_OP_b2i, VVM_OP_random, VVM_OP_randomi, VVM_OP_half_to_float, VVM_OP_exec_index_addi}, 1, 1);
	set({VVM_OP_add, VVM_OP_sub, VVM_OP_mul, VVM_OP_div, VVM_OP_atan2, VVM_OP_fmod, VVM_OP_min, VVM_OP_max, VVM_OP_pow, VVM_OP_step,
	     VVM_OP_cmplt, VVM_OP_cmple, VVM_OP_cmpgt, VVM_OP_cmpge, VVM_OP_cmpeq, VVM_OP_cmpneq, VVM_OP_addi, VVM_OP_subi, VVM_OP_muli,
	     VVM_OP_divi, VVM_OP_mini, VVM_OP_maxi, VVM_OP_cmplti, VVM_OP_cmplei, VVM_OP_cmpgti, VVM_OP_cmpgei, VVM_OP_cmpeqi, VVM_OP_cmpneqi,
	     VVM_OP_bit_and, VVM_OP_bit_or, VVM_OP_bit_xor, VVM_OP_bit_lshift, VVM_OP_bit_rshift, VVM_OP_logic_and, VVM_OP_logic_or,
	     VVM_OP_logic_xor, VVM_OP_sub_neg, VVM_OP_div_f2i, VVM_OP_i2f_div0, VVM_OP_i2f_div1, VVM_OP_i2f_mul, VVM_OP_f2i_maxi,
	     VVM_OP_f2i_addi, VVM_OP_bit_and_i2f, VVM_OP_neg_cmplt, VVM_OP_max_f2i, VVM_OP_random_add}, 2, 1);
	set({VVM_OP_mad, VVM_OP_lerp, VVM_OP_clamp, VVM_OP_select, VVM_OP_clampi, VVM_OP_mad_sqrt, VVM_OP_mul_add, VVM_OP_mul_sub0,
	     VVM_OP_mul_sub1, VVM_OP_mul_mul, VVM_OP_mul_max, VVM_OP_add_add, VVM_OP_sub_cmplt1, VVM_OP_sub_mul, VVM_OP_div_mul,
	     VVM_OP_muli_addi, VVM_OP_addi_bit_rshift, VVM_OP_addi_muli, VVM_OP_i2f_mad0, VVM_OP_i2f_mad1, VVM_OP_f2i_select1,
	     VVM_OP_fmod_add, VVM_OP_bit_rshift_bit_and, VVM_OP_bit_or_muli, VVM_OP_bit_lshift_bit_or}, 3, 1);
	for (int o = VVM_OP_cmplt_logic_and; o <= VVM_OP_cmpnei_logic_or; ++o) g_ops[o] = OpInfo{K_ORD, 3, 1, false};
	set({VVM_OP_cmplt_select, VVM_OP_cmple_select, VVM_OP_cmpeq_select, VVM_OP_cmplti_select, VVM_OP_cmplei_select,
	     VVM_OP_cmpeqi_select, VVM_OP_mad_add, VVM_OP_mad_sub0, VVM_OP_mad_sub1, VVM_OP_mad_mul, VVM_OP_mul_mad0, VVM_OP_mul_mad1,
	     VVM_OP_add_mad1, VVM_OP_div_mad0, VVM_OP_select_mul, VVM_OP_select_add}, 4, 1);
	set({VVM_OP_mad_mad0, VVM_OP_mad_mad1}, 5, 1);
	set({VVM_OP_mul_2x}, 2, 2);
	set({VVM_OP_b2i_2x, VVM_OP_random_2x, VVM_OP_sin_cos}, 1, 2);
	for (int o : {VVM_OP_exp, VVM_OP_exp2, VVM_OP_log, VVM_OP_log2, VVM_OP_sin, VVM_OP_cos, VVM_OP_tan, VVM_OP_asin, VVM_OP_acos,
	              VVM_OP_atan, VVM_OP_atan2, VVM_OP_fmod, VVM_OP_pow, VVM_OP_fmod_add, VVM_OP_sin_cos})
		g_ops[o].heavy = true;
	g_ops[VVM_OP_done] = OpInfo{K_DONE, 0, 0, false};
	g_ops[VVM_OP_exec_index] = OpInfo{K_DST_ONLY, 0, 1, false};
	g_ops[VVM_OP_exec_indexf] = OpInfo{K_DST_ONLY, 0, 1, false};
	g_ops[VVM_OP_acquireindex] = OpInfo{K_ACQUIRE, 1, 0, false};
	for (int o : {VVM_OP_outputdata_float, VVM_OP_outputdata_int32, VVM_OP_outputdata_half, VVM_OP_outputdata_float_from_half,
	              VVM_OP_outputdata_half_from_half})
		g_ops[o] = OpInfo{K_OUTPUT, 0, 0, false};
	g_ops[VVM_OP_external_func_call] = OpInfo{K_EXTFN, 0, 0, true};
	g_ops[VVM_OP_update_id] = OpInfo{K_IDS, 2, 0, false};
	g_ops[VVM_OP_acquire_id] = OpInfo{K_IDS, 0, 2, false};
}
static void build_op_table() { std::call_once(g_ops_once, fill_op_table); }

// ---- decoded instruction -----------------------------------------------------------------------------
struct Decoded {
	uint8_t op = 0, kind = 0;
	std::vector<uint16_t> in, out;                 // bytecode register indices
	std::vector<std::pair<uint16_t, uint16_t>> pairs;   // output instructions: (src, dst)
	uint8_t dataset = 0;
	uint16_t fn = 0;
};

static int decode(const ThawedScript& s, std::vector<Decoded>& out, std::string& err)
{
	build_op_table();
	const std::vector<uint8_t>& bc = s.bytecode;
	size_t ip = 0;
	char msg[160];
	auto need = [&](size_t nbytes) { return ip + nbytes <= bc.size(); };
	for (;;) {
		if (!need(1)) { err = "bytecode ends without a done"; return VVM_E_BAD_BLOB; }
		const uint8_t op = bc[ip++];
		const OpInfo& info = g_ops[op];
		Decoded d;
		d.op = op; d.kind = info.kind;
		if (info.kind == K_DONE) break;
		switch (info.kind) {
		case K_ORD: {
			const size_t n = (size_t)info.nin + info.nout;
			if (!need(2 * n)) { err = "truncated op"; return VVM_E_BAD_BLOB; }
			for (size_t i = 0; i < info.nin; ++i) d.in.push_back(rd16(&bc[ip + 2 * i]));
			for (size_t i = 0; i < info.nout; ++i) d.out.push_back(rd16(&bc[ip + 2 * (info.nin + i)]));
			ip += 2 * n;
			break;
		}
		case K_DST_ONLY:
			if (!need(2)) { err = "truncated exec_index"; return VVM_E_BAD_BLOB; }
			d.out.push_back(rd16(&bc[ip])); ip += 2;
			break;
		case K_ACQUIRE:
			if (!need(3)) { err = "truncated acquireindex"; return VVM_E_BAD_BLOB; }
			d.in.push_back(rd16(&bc[ip])); d.dataset = bc[ip + 2]; ip += 3;
			break;
		case K_OUTPUT: {
			if (!need(2)) { err = "truncated output"; return VVM_E_BAD_BLOB; }
			const size_t n = bc[ip];
			d.dataset = bc[ip + 1];
			if (!need(2 + 4 * n)) { err = "truncated output"; return VVM_E_BAD_BLOB; }
			for (size_t j = 0; j < n; ++j)
				d.pairs.emplace_back(rd16(&bc[ip + 2 + 2 * j]), rd16(&bc[ip + 2 + 2 * (n + j)]));
			ip += 2 + 4 * n;
			break;
		}
		case K_EXTFN: {
			if (!need(2)) { err = "truncated external call"; return VVM_E_BAD_BLOB; }
			d.fn = rd16(&bc[ip]);
			if (d.fn >= s.num_ext_fns) { err = "external function index out of range"; return VVM_E_BAD_BLOB; }
			const size_t ni = (size_t)s.ext_num_inputs[d.fn], no = (size_t)s.ext_num_outputs[d.fn];
			if (!need(2 + 2 * (ni + no))) { err = "truncated external call"; return VVM_E_BAD_BLOB; }
			for (size_t i = 0; i < ni; ++i) d.in.push_back(rd16(&bc[ip + 2 + 2 * i]));
			for (size_t i = 0; i < no; ++i) d.out.push_back(rd16(&bc[ip + 2 + 2 * (ni + i)]));
			ip += 2 + 2 * (ni + no);
			break;
		}
		case K_IDS:
			if (!need(5)) { err = "truncated id op"; return VVM_E_BAD_BLOB; }
			if (op == VVM_OP_update_id) { d.in.push_back(rd16(&bc[ip])); d.in.push_back(rd16(&bc[ip + 2])); }
			else { d.out.push_back(rd16(&bc[ip])); d.out.push_back(rd16(&bc[ip + 2])); }
			d.dataset = bc[ip + 4]; ip += 5;
			break;
		default:
			// inputs, stat scopes, fasi/iasf and the noise ops are never emitted in the optimized format
			// (VectorVMOptimizer.cpp:3044-3046, 3112-3113); the reference's handlers for them are empty stubs
			snprintf(msg, sizeof(msg), "opcode %u at byte %zu has no optimized-format encoding", (unsigned)op, ip - 1);
			err = msg;
			return VVM_E_UNSUPPORTED_OP;
		}
		out.push_back(std::move(d));
	}
	return VVM_OK;
}

// ---- linker ------------------------------------------------------------------------------------------
static const uint32_t kMaxSpillWide = 11;      // LDS register slots of the 4-instances-per-lane kernel (4 KB each per workgroup; two
                                               // workgroups per CU must fit: 2 x (36 KB + 11 x 4 KB) = 160 KB; exactly: fixed part 35 712 bytes, 36 480 with the noise tables)
static const uint32_t kFileWidePlain = 44;     // registers in VGPRs on the four-wide kernel of plain scripts (the file of its hand-written routines:
                                               // VVM_FILE4_REGS of the kernels); the four-wide kernel with the transcendental functions keeps 32
static const uint32_t kMaxSpillNarrow = 128;   // 1 KB each

// Packs the per-lane registers of a linked program by live range.  The program is straight-line code that every tile runs from
// the top, so a register's value lives from the instruction that writes it to the last one that reads it, and two values whose
// ranges do not overlap can share a register.  An instruction reads all its sources before it writes its destination, so the
// destination may take over a register whose value dies at that very instruction.  The loaded inputs keep registers
// 0..num_loaded-1 (the load instructions address consecutive numbers) and give them up after their last use; a register that
// is read before anything wrote it keeps a register of its own from the top of the program (its content is whatever the last
// tile left, as the reference's is whatever its batch memory held).  Returns the number of registers in use after packing.
// Values written to the provisional registers from `first_ride` on keep fixed registers outside the packed ones:
// VVM_LIGHT_RIDE_REG + (provisional number - first_ride).
static uint32_t reallocate_registers(std::vector<VvmIns>& code, uint32_t num_loaded, uint32_t provisional, uint32_t first_ride = 0xFFFFFFFFu)
{
	struct Value { int32_t def, last; uint32_t reg; bool fixed; };
	struct Ref { uint32_t ins, field, value; };
	std::vector<Value> values;
	std::vector<Ref> refs;
	std::vector<int32_t> holds(provisional, -1);      // value each provisional register holds at the current instruction
	auto field_of = [](VvmIns& k, uint32_t f) -> uint32_t { const uint32_t w = (&k.w1)[f >> 1]; return (f & 1) ? w >> 16 : w & 0xFFFFu; };
	auto set_field = [](VvmIns& k, uint32_t f, uint32_t v) {
		uint32_t& w = (&k.w1)[f >> 1];
		w = (f & 1) ? (w & 0xFFFFu) | (v << 16) : (w & 0xFFFF0000u) | v;
	};
	for (uint32_t i = 0; i < num_loaded; ++i) { values.push_back(Value{-1, -1, i, false}); holds[i] = (int32_t)i; }
	for (uint32_t p = 0; p < code.size(); ++p) {
		VvmIns& k = code[p];
		const uint32_t op = k.w0 & 0xFFu;
		uint32_t reads = 0;
		bool writes = false;
		if (op == VVM_DOP_ACQUIRE) reads = (k.w0 >> 24) != VVM_ACQ_SIGN ? 2 : 1;
		else if (op == VVM_DOP_UPDATE_ID) reads = 1;
		else if (op == VVM_DOP_OUTPUT32 || op == VVM_DOP_OUTPUT16 || op == VVM_DOP_OUTPUT16_FROM16 || op == VVM_DOP_OUTPUT32_FROM16) reads = (k.w0 >> 16) & 0xFFu;
		else if (op == VVM_OP_done || op >= VVM_DOP_LOADN16) reads = 0;
		else { reads = (k.w0 >> 8) & 7u; writes = true; }
		for (uint32_t f = 0; f < reads; ++f) {
			const uint32_t o = field_of(k, f);
			if (VVM_OPND_CLS(o) != VVM_OPC_VREG) continue;
			const uint32_t r = VVM_OPND_IDX(o);
			if (holds[r] < 0) { holds[r] = (int32_t)values.size(); values.push_back(Value{-1, -1, 0, false}); }
			values[(size_t)holds[r]].last = (int32_t)p;
			refs.push_back(Ref{p, f, (uint32_t)holds[r]});
		}
		if (writes) {
			const uint32_t o = field_of(k, 5);
			if (VVM_OPND_CLS(o) == VVM_OPC_VREG) {
				const uint32_t r = VVM_OPND_IDX(o);
				holds[r] = (int32_t)values.size();
				values.push_back(Value{(int32_t)p, (int32_t)p, r >= first_ride ? VVM_LIGHT_RIDE_REG + (r - first_ride) : 0u, r >= first_ride});
				refs.push_back(Ref{p, 5, (uint32_t)holds[r]});
			}
		}
	}
	// linear scan in order of definition; the lowest free register first
	std::vector<uint32_t> order;
	for (uint32_t v = num_loaded; v < values.size(); ++v) if (!values[v].fixed) order.push_back(v);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return values[a].def < values[b].def; });
	std::vector<int32_t> busy_until(num_loaded, 0);      // per register: last instruction that reads the value it holds
	for (uint32_t i = 0; i < num_loaded; ++i) busy_until[i] = values[i].last;
	for (uint32_t v : order) {
		Value& val = values[v];
		uint32_t reg = 0;
		// a value defined at instruction p may take a register whose value is last read at p or earlier; a value live from the top
		// (def -1) only a register of its own
		while (reg < busy_until.size() && !(val.def >= 0 && busy_until[reg] <= val.def)) ++reg;
		if (reg == busy_until.size()) busy_until.push_back(0);
		busy_until[reg] = val.last;
		val.reg = reg;
	}
	for (const Ref& r : refs) set_field(code[r.ins], r.field, VVM_OPND(VVM_OPC_VREG, values[r.value].reg));
	return (uint32_t)busy_until.size();
}

// acquireindex whose predicate is the mask of a compare that nothing else reads: the compare is folded into the acquireindex
// (keep = a <cmp> b, which is the sign bit of the compare's all-ones / all-zeros mask, VectorVMRuntime.cpp:838-841), one
// interpreted instruction and one register less.  `holds_value` says which bytecode registers are per-lane values.
static uint32_t acquire_compare_code(uint8_t op)
{
	switch (op) {
	case VVM_OP_cmplt: return VVM_ACQ_LT;   case VVM_OP_cmple: return VVM_ACQ_LE;   case VVM_OP_cmpgt: return VVM_ACQ_GT;
	case VVM_OP_cmpge: return VVM_ACQ_GE;   case VVM_OP_cmpeq: return VVM_ACQ_EQ;   case VVM_OP_cmpneq: return VVM_ACQ_NE;
	case VVM_OP_cmplti: return VVM_ACQ_LTI; case VVM_OP_cmplei: return VVM_ACQ_LEI; case VVM_OP_cmpgti: return VVM_ACQ_GTI;
	case VVM_OP_cmpgei: return VVM_ACQ_GEI; case VVM_OP_cmpeqi: return VVM_ACQ_EQI; case VVM_OP_cmpneqi: return VVM_ACQ_NEI;
	default: return VVM_ACQ_SIGN;
	}
}
template <typename F> static void fold_compares_into_acquires(std::vector<Decoded>& ins, F holds_value)
{
	for (size_t k = 0; k < ins.size(); ++k) {
		if (ins[k].kind != K_ACQUIRE || ins[k].in.size() != 1 || !holds_value(ins[k].in[0])) continue;
		const uint16_t r = ins[k].in[0];
		// the instruction that made the predicate: the last one before k that writes r
		size_t j = k;
		for (size_t q = k; q-- > 0;)
			if (std::find(ins[q].out.begin(), ins[q].out.end(), r) != ins[q].out.end()) { j = q; break; }
		if (j == k) continue;
		const uint32_t code = acquire_compare_code(ins[j].op);
		if (code == VVM_ACQ_SIGN || ins[j].kind != K_ORD || ins[j].out.size() != 1 || ins[j].in.size() != 2) continue;
		// nobody else reads the mask: no instruction from the compare up to the one that next writes the register (or the end)
		bool other_reader = false;
		for (size_t q = j + 1; q < ins.size() && !other_reader; ++q) {
			if (q != k) {
				other_reader |= std::find(ins[q].in.begin(), ins[q].in.end(), r) != ins[q].in.end();
				for (auto& pr : ins[q].pairs) other_reader |= pr.first == r;
			}
			if (q != k && std::find(ins[q].out.begin(), ins[q].out.end(), r) != ins[q].out.end()) break;      // (reads its inputs before it writes)
		}
		if (other_reader) continue;
		// the compare's operands still hold the same values at the acquireindex
		bool clobbered = false;
		for (size_t q = j + 1; q < k && !clobbered; ++q)
			for (uint16_t w : ins[q].out) clobbered |= (w == ins[j].in[0] || w == ins[j].in[1]);
		if (clobbered) continue;
		ins[k].in = ins[j].in;
		ins[k].fn = (uint16_t)code;
		ins.erase(ins.begin() + (ptrdiff_t)j);
		--k;
	}
}

static int link_program_as(const ThawedScript& s, bool force_narrow, LinkedProgram& lp, std::string& err, uint32_t pass, bool light, bool allow_chains, bool& chains_failed,
                           const std::vector<uint8_t>* out_select);

int link_program(const ThawedScript& s, bool force_narrow, LinkedProgram& lp, std::string& err, uint32_t pass, bool light, const std::vector<uint8_t>* out_select, int pairing)
{
	// A script with several acquired datasets: first with the datasets taken two to a pass (each pass a program with two chains, the
	// last one of an odd number alone); if one of those programs needs a kernel that has no second chain, one pass per dataset.
	// What pass k means follows from that choice, so it is made once, when pass 0 is linked, for all passes of the script
	// (LinkedProgram::paired of pass 0 says which; the caller hands it back for the passes behind it).
	const bool may_pair = !light && !force_narrow && !getenv("VVM_NO_CHAINS");
	bool chains_failed = false;
	if (pairing >= 0 || pass != 0) return link_program_as(s, force_narrow, lp, err, pass, light, may_pair && pairing == 1, chains_failed, out_select);
	if (light) {
		// a caller that asks for the script's light program without saying how the passes are numbered means the whole script: a script
		// with several acquired datasets has none (its passes may have one each: pairing = 0 and the pass)
		const int rc = link_program_as(s, force_narrow, lp, err, 0, true, false, chains_failed, out_select);
		if (rc == VVM_OK && lp.num_passes != 1) { err = "script does not fit the light eight-wide kernel"; return VVM_E_UNSUPPORTED_OP; }
		return rc;
	}
	if (may_pair) {
		int rc = link_program_as(s, force_narrow, lp, err, 0, light, true, chains_failed, out_select);
		if (!chains_failed) {
			if (rc != VVM_OK || !lp.paired || lp.num_passes == 1) return rc;
			bool all = true;
			for (uint32_t k = 1; k < lp.num_passes && all; ++k) {
				LinkedProgram trial;
				std::string why;
				bool failed = false;
				all = link_program_as(s, force_narrow, trial, why, k, light, true, failed, nullptr) == VVM_OK;
			}
			if (all) return rc;
		}
		lp = LinkedProgram();
		err.clear();
	}
	return link_program_as(s, force_narrow, lp, err, 0, light, false, chains_failed, out_select);
}

static int link_program_as(const ThawedScript& s, bool force_narrow, LinkedProgram& lp, std::string& err, uint32_t pass, bool light, bool allow_chains, bool& chains_failed,
                           const std::vector<uint8_t>* out_select)
{
	chains_failed = false;
	std::vector<Decoded> ins;
	int rc = decode(s, ins, err);
	if (rc != VVM_OK) return rc;
	char msg[200];

	const uint32_t T = s.num_temps, C = s.num_consts, I = s.num_inputs;
	const uint32_t O = std::max<uint32_t>(s.num_outputs_remapped, s.num_output_instructions);
	const uint32_t out0 = T + C + 2 * I;
	fold_compares_into_acquires(ins, [&](uint16_t r) { return r != VVM_INVALID_REG && (r < T || (r >= out0 && r < out0 + O)); });
	if (C > VVM_MAX_OPND_INDEX || I > 0xFFF || O > VVM_MAX_OPND_INDEX) { err = "script exceeds the device operand index range"; return VVM_E_TOO_LARGE; }

	// classify inputs from InputDataSetOffsets: per dataset float,int,half then no-advance float,int,half (RT.cpp:515-550)
	lp.inputs.assign(I, InputInfo{0, 0, false, 0});
	{
		uint32_t counter = 0;
		for (uint32_t ds = 0; ds < s.num_input_datasets; ++ds) {
			const uint16_t* off = &s.input_dataset_offsets[ds * 8];
			for (int j = 0; j < 3; ++j)
				for (uint32_t k = off[j]; k < off[j + 1]; ++k) {
					if (counter >= I || k >= I) { err = "InputDataSetOffsets inconsistent with NumInputsRemapped"; return VVM_E_BAD_BLOB; }
					lp.inputs[counter++] = InputInfo{(uint8_t)ds, (uint8_t)j, false, s.input_remap[k]};
				}
			if (off[7] > 0)
				for (int j = 0; j < 3; ++j)
					for (uint32_t k = off[j + 3]; k < off[j + 4]; ++k) {
						if (counter >= I || k >= I) { err = "InputDataSetOffsets inconsistent with NumInputsRemapped"; return VVM_E_BAD_BLOB; }
						lp.inputs[counter++] = InputInfo{(uint8_t)ds, (uint8_t)j, true, s.input_remap[k]};
					}
		}
		if (counter != I) { err = "InputDataSetOffsets do not cover every remapped input"; return VVM_E_BAD_BLOB; }   // RT.cpp:551
	}

	// which output registers are used as op operands (temps aliased onto output buffers, VectorVMOptimizer.cpp:2855-2978)
	std::vector<int32_t> out_vreg(O, -1);
	uint32_t nv = T;
	auto reg_class_ok = [&](uint16_t r) { return r == VVM_INVALID_REG || r < out0 + O; };
	auto touch = [&](uint16_t r) {
		if (r == VVM_INVALID_REG) return;
		if (r < T) return;
		if (r >= out0) {
			const uint32_t o = r - out0;
			if (out_vreg[o] < 0) out_vreg[o] = (int32_t)nv++;
		}
	};
	lp.acquires_dataset.assign(std::max<uint32_t>(s.max_output_dataset, 1), 0);
	bool heavy = false, uses_perlin = false;
	for (const Decoded& d : ins) {
		for (uint16_t r : d.in) if (!reg_class_ok(r)) { err = "register index out of range"; return VVM_E_BAD_BLOB; }
		for (uint16_t r : d.out) if (!reg_class_ok(r)) { err = "register index out of range"; return VVM_E_BAD_BLOB; }
		if (g_ops[d.op].heavy) heavy = true;
		if (d.kind == K_EXTFN) uses_perlin = true;
		if (d.kind == K_IDS) {
			// Any number of datasets.  update_id returns IDs to the free list that acquire_id takes them from, chunk by chunk in the
			// reference (VectorVMRuntime.cpp:1175, 1411-1423): a script that does both to ONE dataset hands IDs from a chunk to the
			// next, and its Exec is run chunk after chunk (serial_chunks).  Supported in the order a spawn script with a first update has
			// them: acquire_id, then update_id.
			const bool upd = d.op == VVM_OP_update_id;
			bool known = false;
			for (LinkedProgram::IdUse& u : lp.id_uses) {
				if (u.dataset != d.dataset) continue;
				known = true;
				if (upd) {
					if (u.updates) { err = "more than one update_id for a dataset is not supported by the GPU executor"; return VVM_E_UNSUPPORTED_OP; }
					u.updates = true;
				} else {
					if (u.updates) { err = "acquire_id behind update_id on one dataset is not supported by the GPU executor (the chunk's own freed IDs would be taken again)"; return VVM_E_UNSUPPORTED_OP; }
					u.acquires = true;
				}
			}
			if (!known) lp.id_uses.push_back(LinkedProgram::IdUse{d.dataset, upd, !upd});
			if (lp.id_uses.size() > VVM_MAX_ID_USES) { err = "update_id / acquire_id on too many datasets"; return VVM_E_TOO_LARGE; }
		}
		if (d.kind == K_OUTPUT) {
			for (auto& p : d.pairs) {
				if (!reg_class_ok(p.first) || p.second < out0 || p.second >= out0 + O) { err = "output operand out of range"; return VVM_E_BAD_BLOB; }
				// an output register as a SOURCE is a value some op stored there
				if (p.first >= out0 || p.first < T) touch(p.first);
			}
			continue;
		}
		if (d.kind == K_ACQUIRE) {
			if (d.dataset >= lp.acquires_dataset.size()) { err = "acquireindex names a dataset past MaxOutputDataSet"; return VVM_E_BAD_BLOB; }
			if (lp.acquires_dataset[d.dataset]) { err = "more than one acquireindex for a dataset"; return VVM_E_UNSUPPORTED_OP; }
			lp.acquires_dataset[d.dataset] = 1;
			lp.acquire_order.push_back(d.dataset);
		}
		for (uint16_t r : d.in) touch(r);
		for (uint16_t r : d.out) {
			if (r != VVM_INVALID_REG && r >= T && r < out0) { err = "an op writes a constant or input register"; return VVM_E_BAD_BLOB; }
			touch(r);
		}
	}
	// Outputs that hand an input stream through unchanged (see LinkedProgram::through_input), then the selection of outputs this
	// program writes (out_select): the pairs that are not selected leave the decoded script here, so that everything below (what
	// is computed, what is loaded, the output list) follows from the ones that stay.
	lp.through_input.assign(O, -1);
	{
		std::vector<uint8_t> writers(O, 0);
		std::vector<uint8_t> acquired(lp.acquires_dataset.size(), 0);
		for (const Decoded& d : ins) {
			if (d.kind == K_ACQUIRE) acquired[d.dataset] = 1;
			if (d.kind != K_OUTPUT) continue;
			for (auto& p : d.pairs) {
				const uint32_t serial = p.second - out0;
				if (writers[serial]++) { lp.through_input[serial] = -2; continue; }
				const bool word_copy = d.op == VVM_OP_outputdata_float || d.op == VVM_OP_outputdata_int32;
				const bool half_copy = d.op == VVM_OP_outputdata_half_from_half;      // (16 bits as they are, VectorVMRuntime.cpp:663-742)
				// (of a script with two acquired datasets only the first one's: the copy kernel works from the first chain's results)
				const bool behind_acquire = d.dataset < acquired.size() && acquired[d.dataset] && d.dataset == lp.acquire_order.front();
				if ((word_copy || half_copy) && behind_acquire && p.first >= T + C && p.first < T + C + I) {
					const uint32_t i = p.first - (T + C);
					if (!lp.inputs[i].noadvance && (lp.inputs[i].type == 2) == half_copy) lp.through_input[serial] = (int32_t)i;
				}
			}
		}
		for (uint32_t o = 0; o < O; ++o) if (writers[o] > 1) lp.through_input[o] = -2;
	}
	if (out_select) {
		if (out_select->size() != O) { err = "internal: output selection of the wrong size"; return VVM_E_INVALID_ARG; }
		for (size_t k = 0; k < ins.size(); ++k) {
			Decoded& d = ins[k];
			if (d.kind != K_OUTPUT) continue;
			d.pairs.erase(std::remove_if(d.pairs.begin(), d.pairs.end(), [&](const std::pair<uint16_t, uint16_t>& p) { return !(*out_select)[p.second - out0]; }), d.pairs.end());
			if (d.pairs.empty()) { ins.erase(ins.begin() + (ptrdiff_t)k); --k; }
		}
	}
	// Derived constants (see LinkedProgram::derived): registers [out0 + O, ...) stand for them in the decoded instructions from
	// here on.  A register holds a derived constant from the folded instruction that wrote it to the next write of the register.
	const uint32_t derived0 = out0 + O;
	if (!getenv("VVM_NO_CONSTANT_FOLDING")) {
		std::vector<int32_t> holds_derived((size_t)out0 + O, -1);
		auto is_value_reg = [&](uint16_t r) { return r != VVM_INVALID_REG && (r < T || (r >= out0 && r < out0 + O)); };
		auto uniform_index = [&](uint16_t r) -> int32_t {      // the uniform a register operand names, or -1
			if (r == VVM_INVALID_REG) return -1;
			if (r >= T && r < T + C) return (int32_t)(r - T);
			if (r >= derived0) return (int32_t)(C + (r - derived0));
			return -1;
		};
		for (size_t k = 0; k < ins.size(); ++k) {
			Decoded& d = ins[k];
			auto rewrite = [&](uint16_t& r) { if (is_value_reg(r) && holds_derived[r] >= 0) r = (uint16_t)(derived0 + (uint32_t)holds_derived[r]); };
			for (uint16_t& r : d.in) rewrite(r);
			for (auto& pr : d.pairs) rewrite(pr.first);
			bool folded = false;
			const bool foldable_op = d.op == VVM_OP_add || d.op == VVM_OP_sub || d.op == VVM_OP_mul || d.op == VVM_OP_addi || d.op == VVM_OP_subi || d.op == VVM_OP_muli;
			if (d.kind == K_ORD && foldable_op && d.in.size() == 2 && d.out.size() == 1 && is_value_reg(d.out[0])) {
				const int32_t a = uniform_index(d.in[0]), b = uniform_index(d.in[1]);
				if (a >= 0 && b >= 0 && lp.derived.size() < 255 && derived0 + lp.derived.size() + 1 < VVM_INVALID_REG && C + lp.derived.size() + 1 <= VVM_MAX_OPND_INDEX) {
					holds_derived[d.out[0]] = (int32_t)lp.derived.size();
					lp.derived.push_back(LinkedProgram::Derived{d.op, (uint16_t)a, (uint16_t)b});
					folded = true;
				}
			}
			if (folded) { ins.erase(ins.begin() + (ptrdiff_t)k); --k; continue; }
			for (uint16_t r : d.out) if (is_value_reg(r)) holds_derived[r] = -1;
		}
	}

	// A device instruction has one destination, so the reference's multi-output instructions are split.  The reference reads
	// all inputs before it writes any output (per 4 lanes); when more than one destination is also an input of the same
	// instruction, all but the last of those are computed into scratch registers and copied afterwards.
	uint32_t max_scratch = 0;
	for (const Decoded& d : ins) {
		if (d.out.size() < 2 || (d.kind != K_ORD && d.kind != K_EXTFN)) continue;
		uint32_t aliasing = 0;
		for (size_t i = 0; i < d.out.size(); ++i) {
			if (d.out[i] == VVM_INVALID_REG) continue;
			bool later_same = false;
			for (size_t j = i + 1; j < d.out.size(); ++j) later_same |= d.out[j] == d.out[i];
			if (later_same) continue;
			if (std::find(d.in.begin(), d.in.end(), d.out[i]) != d.in.end()) ++aliasing;
		}
		if (aliasing > 1) max_scratch = std::max(max_scratch, aliasing - 1);
	}
	const uint32_t scratch0 = nv;
	nv += max_scratch;
	for (const LinkedProgram::IdUse& u : lp.id_uses)
		if (u.updates && !(u.dataset < lp.acquires_dataset.size() && lp.acquires_dataset[u.dataset])) {
			err = "update_id on a dataset the script has no acquireindex for";       // it walks that dataset's keep mask (VectorVMRuntime.cpp:1178)
			return VVM_E_UNSUPPORTED_OP;
		}
	// A script that runs acquireindex for several datasets is executed in one pass per dataset: every pass runs the whole
	// script, but only the pass's dataset is compacted and written (the others' acquireindex and outputs are left out of the
	// pass's program).  The instructions are pure functions of the inputs, so every pass computes the same values.
	// (allow_chains: the datasets two to a pass, both acquireindex of a pair in one program, each with its own chain)
	const uint32_t acquired = (uint32_t)lp.acquire_order.size();
	lp.paired = allow_chains && acquired >= 2 && lp.id_uses.empty();
	lp.num_passes = lp.paired ? (acquired + 1) / 2 : std::max<uint32_t>(1, acquired);
	if (pass >= lp.num_passes) { err = "pass index out of range"; return VVM_E_INVALID_ARG; }
	const uint32_t first_of_pass = lp.paired ? 2 * pass : pass;
	const bool chains = lp.paired && first_of_pass + 1 < acquired;
	lp.num_chains = chains ? 2 : 1;
	for (LinkedProgram::IdUse& u : lp.id_uses)
		if (u.updates && u.acquires) {
			u.serial = true;
			lp.serial_chunks = true;
		}
	const bool multi_pass = lp.num_passes > 1;
	const uint32_t pass_dataset = lp.acquire_order.empty() ? 0 : lp.acquire_order[first_of_pass];
	const uint32_t second_dataset = chains ? lp.acquire_order[first_of_pass + 1] : pass_dataset;
	lp.pass_dataset = pass_dataset;
	lp.chain_dataset[0] = pass_dataset;
	lp.chain_dataset[1] = second_dataset;
	auto chain_of = [&](uint32_t dataset) -> uint32_t { return (chains && dataset == second_dataset) ? 1u : 0u; };
	auto in_pass = [&](uint32_t dataset) { return dataset == pass_dataset || dataset == second_dataset; };
	// update_id runs in its dataset's pass; acquire_id in every pass (its register results may feed any dataset's outputs; what it
	// writes to the ID tables depends on the instance index alone, so every pass writes the same)
	if (multi_pass) {
		for (LinkedProgram::IdUse& u : lp.id_uses)
			if (u.updates && u.acquires && !in_pass(u.dataset)) u.updates = false;      // (the dataset's acquire_id stays: `serial` says where it reads)
		lp.id_uses.erase(std::remove_if(lp.id_uses.begin(), lp.id_uses.end(), [&](const LinkedProgram::IdUse& u) { return u.updates && !in_pass(u.dataset); }), lp.id_uses.end());
	}
	lp.uses_ids = !lp.id_uses.empty();
	lp.updates_ids = false;
	for (const LinkedProgram::IdUse& u : lp.id_uses) lp.updates_ids = lp.updates_ids || u.updates;
	auto id_slot = [&](uint32_t dataset) -> uint32_t {
		for (size_t i = 0; i < lp.id_uses.size(); ++i) if (lp.id_uses[i].dataset == dataset) return (uint32_t)i;
		return 0;
	};

	// Instructions whose results reach neither an output, an acquireindex nor an ID table of this pass are left out (one
	// backward sweep over the straight-line script).  A pass of a script with several acquired datasets then only computes what
	// its own dataset needs, and only loads the streams that feeds on.  Nothing observable changes: such results live in
	// per-lane registers only, also where the script keeps them in an output register.
	std::vector<uint8_t> needed(ins.size(), 1);
	{
		std::vector<uint8_t> live((size_t)out0 + O, 0);
		auto holds_value = [&](uint16_t r) { return r != VVM_INVALID_REG && (r < T || (r >= out0 && r < derived0)); };      // (not the derived constants)
		heavy = false; uses_perlin = false;
		for (size_t k = ins.size(); k-- > 0;) {
			const Decoded& d = ins[k];
			const bool other_pass = multi_pass && !in_pass(d.dataset);
			bool keep = true;
			switch (d.kind) {
			case K_OUTPUT:
				keep = !other_pass;
				if (keep) for (auto& p : d.pairs) if (holds_value(p.first)) live[p.first] = 1;
				break;
			case K_ACQUIRE:
				keep = !other_pass;
				if (keep) for (uint16_t r : d.in) if (holds_value(r)) live[r] = 1;
				break;
			case K_IDS:
				if (d.op == VVM_OP_update_id) {
					keep = !other_pass;
					if (keep && holds_value(d.in[0])) live[d.in[0]] = 1;
				} else {
					for (uint16_t r : d.out) if (holds_value(r)) live[r] = 0;      // (kept whatever reads it: it takes IDs off the free list)
				}
				break;
			case K_ORD: case K_DST_ONLY: case K_EXTFN: {
				keep = false;
				for (uint16_t r : d.out) keep |= holds_value(r) && live[r];
				if (keep) {
					for (uint16_t r : d.out) if (holds_value(r)) live[r] = 0;
					for (uint16_t r : d.in) if (holds_value(r)) live[r] = 1;
					if (g_ops[d.op].heavy) heavy = true;
					if (d.kind == K_EXTFN) uses_perlin = true;
				}
				break;
			}
			default: break;
			}
			needed[k] = keep ? 1 : 0;
		}
	}

	// Every advancing input stream the script reads is loaded once, at the start of a tile, into a per-lane register of its own
	// (VVM_DOP_LOADN / VVM_DOP_LOADN16, eight streams per instruction, in order of first use; 32-bit streams first).  The
	// interpreter loop itself then never touches stream memory.
	std::vector<uint32_t> prefetch32, prefetch16;
	{
		std::vector<uint8_t> seen(I, 0);
		for (size_t k = 0; k < ins.size(); ++k) {
			if (!needed[k]) continue;
			const Decoded& d = ins[k];
			auto consider = [&](uint16_t r) {
				if (r == VVM_INVALID_REG || r < T + C || r >= T + C + I) return;
				const uint32_t i = r - (T + C);
				if (seen[i] || lp.inputs[i].noadvance) return;
				seen[i] = 1;
				(lp.inputs[i].type == 2 ? prefetch16 : prefetch32).push_back(i);
			};
			for (uint16_t r : d.in) consider(r);
			for (auto& p : d.pairs) consider(p.first);
		}
	}
	const uint32_t num_loaded = (uint32_t)(prefetch32.size() + prefetch16.size());

	// Provisional register numbers for the emission below: the loaded inputs take 0..num_loaded-1 (consecutive numbers, in load
	// order, which is what the load instructions address), the temps, aliased output registers and scratch registers follow.
	// reallocate_registers() then packs the latter by live range, and the kernel shape is chosen from the packed count.
	if (num_loaded > 160 || num_loaded + nv > VVM_MAX_OPND_INDEX) {
		snprintf(msg, sizeof(msg), "script needs %u per-lane registers; the executor holds at most %u", num_loaded + nv, 32 + kMaxSpillNarrow);
		err = msg;
		return VVM_E_TOO_LARGE;
	}
	std::vector<uint32_t> place(nv);
	for (uint32_t i = 0; i < nv; ++i) place[i] = num_loaded + i;
	std::vector<int32_t> input_vreg(I, -1);
	{
		uint32_t next_vreg = 0;
		for (uint32_t i : prefetch32) input_vreg[i] = (int32_t)next_vreg++;
		for (uint32_t i : prefetch16) input_vreg[i] = (int32_t)next_vreg++;
	}
	lp.num_prefetched_inputs = num_loaded;
	lp.full_kernel = heavy || lp.uses_ids;      // (the ID instructions are compiled into the full kernels only)
	lp.uses_perlin = uses_perlin;

	lp.input_used.assign(I, 0);
	auto operand = [&](uint16_t r) -> uint32_t {
		if (r == VVM_INVALID_REG) return VVM_OPND(VVM_OPC_NONE, 0);
		if (r < T) return VVM_OPND(VVM_OPC_VREG, place[r]);
		if (r < T + C) return (r - T < VVM_MAX_UNIFORMS) ? VVM_OPND(VVM_OPC_UNIF, r - T) : VVM_OPND(VVM_OPC_CONST, r - T);
		if (r < T + C + I) {
			const uint32_t i = r - (T + C);
			lp.input_used[i] = 1;
			if (input_vreg[i] >= 0) return VVM_OPND(VVM_OPC_VREG, (uint32_t)input_vreg[i]);
			return VVM_OPND(lp.inputs[i].type == 2 ? VVM_OPC_NOADV_HALF : VVM_OPC_NOADV, i);
		}
		if (r < out0) return VVM_OPND(VVM_OPC_NONE, 0);   // input base copies: never operands
		if (r >= derived0) {                              // a derived constant: the uniforms behind the script's constants
			const uint32_t u = C + (r - derived0);
			return (u < VVM_MAX_UNIFORMS) ? VVM_OPND(VVM_OPC_UNIF, u) : VVM_OPND(VVM_OPC_CONST, u);
		}
		return VVM_OPND(VVM_OPC_VREG, place[(size_t)out_vreg[r - out0]]);
	};

	// ---- emit ----
	lp.code.clear();
	struct SplitOut { uint16_t reg; uint32_t w0; };
	// Emits one device instruction per destination of `d` (all share the input operands o[0..4]) in an order that gives the
	// reference's result: all inputs are read before any output is written, and of two writes to one register the later wins.
	auto emit_split = [&](const Decoded& d, std::vector<SplitOut>& outs, const uint32_t* o) {
		std::vector<SplitOut> plain, aliasing;
		for (size_t i = 0; i < outs.size(); ++i) {
			bool later_same = false;
			for (size_t j = i + 1; j < outs.size(); ++j) later_same |= outs[j].reg == outs[i].reg;
			if (later_same) continue;
			const bool is_input = std::find(d.in.begin(), d.in.end(), outs[i].reg) != d.in.end();
			(is_input ? aliasing : plain).push_back(outs[i]);
		}
		auto put = [&](uint32_t w0, uint32_t dst) {
			VvmIns k;
			k.w0 = w0; k.w1 = o[0] | (o[1] << 16); k.w2 = o[2] | (o[3] << 16); k.w3 = o[4] | (dst << 16);
			lp.code.push_back(k);
		};
		for (const SplitOut& so : plain) put(so.w0, operand(so.reg));
		for (size_t i = 0; i + 1 < aliasing.size(); ++i) put(aliasing[i].w0, VVM_OPND(VVM_OPC_VREG, place[scratch0 + i]));
		if (!aliasing.empty()) put(aliasing.back().w0, operand(aliasing.back().reg));
		for (size_t i = 0; i + 1 < aliasing.size(); ++i) {   // copy: x | x
			const uint32_t src = VVM_OPND(VVM_OPC_VREG, place[scratch0 + i]);
			VvmIns k;
			k.w0 = VVM_INS_W0(VVM_OP_bit_or, 2, 1, 0, 0);
			k.w1 = src | (src << 16); k.w2 = 0; k.w3 = operand(aliasing[i].reg) << 16;
			lp.code.push_back(k);
		}
	};
	// A load instruction names eight consecutive slots of the load list (one stream address each, in the order the loads are
	// issued), so the kernel fetches all eight addresses with one scalar load.  -1 pads a partly used instruction.
	lp.load_list.clear();
	lp.out_list.clear();
	lp.out_is16.clear();
	auto emit_loads = [&](const std::vector<uint32_t>& prefetch, uint32_t dop) {
		for (size_t i = 0; i < prefetch.size(); i += 8) {
			const size_t n = std::min<size_t>(8, prefetch.size() - i);
			VvmIns k;
			k.w0 = VVM_INS_W0(dop, 0, 0, n, (uint32_t)input_vreg[prefetch[i]]);
			k.w1 = (uint32_t)lp.load_list.size(); k.w2 = 0; k.w3 = 0;
			for (size_t j = 0; j < 8; ++j) lp.load_list.push_back(j < n ? (int32_t)prefetch[i + j] : -1);
			lp.code.push_back(k);
		}
	};
	emit_loads(prefetch32, VVM_DOP_LOADN);
	emit_loads(prefetch16, VVM_DOP_LOADN16);
	uint32_t rand_ordinal = 0;
	int acquired_dataset = -1;       // dataset of the acquireindex seen so far (a pass has one; with two chains: the first chain's)
	int acquired_second = -1;        // ... the second chain's
	// The output list is cut into groups of VVM_STAGE_SLOTS entries (one staging slot each).  An output instruction never
	// straddles two groups, and a flush instruction precedes the first instruction that stages into a new group.  When the
	// script has an update_id, the last entry of every group is kept for it (-1 where it is not used).
	const size_t group_capacity = light ? VVM_LIGHT_GROUP_SLOTS : lp.updates_ids ? VVM_STAGE_SLOTS - 1 : VVM_STAGE_SLOTS;
	size_t out_cursor = 0;           // next free entry
	// light programs: the 32-bit outputs that fall on entries six and seven of the list wait in registers instead of staging slots
	// (VVM_LIGHT_RIDE_REG); what each of them outputs, as an operand
	// Which outputs: as many as the script has beyond six, by preference those whose value an instruction after the acquireindex
	// makes (that instruction then writes the ride register directly); any other value gets there by a copy.
	const uint32_t first_ride = num_loaded + nv;      // provisional registers of the two
	struct Ride { uint32_t src; size_t at; bool direct; };      // source operand, the program position of the output, value made after the acquireindex
	std::vector<Ride> rides_;
	size_t rides_wanted = 0, acquire_at = (size_t)-1;
	if (light) {
		size_t total_pairs = 0;
		for (size_t q = 0; q < ins.size(); ++q) if (needed[q] && ins[q].kind == K_OUTPUT) total_pairs += ins[q].pairs.size();
		rides_wanted = total_pairs > VVM_LIGHT_GROUP_SLOTS ? std::min<size_t>(2, total_pairs - VVM_LIGHT_GROUP_SLOTS) : 0;
	}
	auto ins_field = [](VvmIns& k, uint32_t f) -> uint32_t { const uint32_t w = (&k.w1)[f >> 1]; return (f & 1) ? w >> 16 : w & 0xFFFFu; };
	auto ins_set_field = [](VvmIns& k, uint32_t f, uint32_t v) {
		uint32_t& w = (&k.w1)[f >> 1];
		w = (f & 1) ? (w & 0xFFFFu) | (v << 16) : (w & 0xFFFF0000u) | v;
	};
	auto ordinary = [](const VvmIns& k) { const uint32_t op = k.w0 & 0xFFu; return op != VVM_OP_done && op < VVM_DOP_LOADN16; };
	// the instruction before position `before` that last wrote per-lane register operand `reg` (lp.code.size(): none)
	auto last_def = [&](uint32_t reg, size_t before) -> size_t {
		for (size_t q = before; q-- > 0;)
			if (ordinary(lp.code[q]) && ins_field(lp.code[q], 5) == reg) return q;
		return lp.code.size();
	};
	if (light && first_ride + 2 > VVM_MAX_OPND_INDEX) { err = "script does not fit the light eight-wide kernel"; return VVM_E_UNSUPPORTED_OP; }
	size_t open_group = 0;           // group the instructions emitted so far stage into
	auto open_slot = [&]() -> size_t {
		if (out_cursor % VVM_STAGE_SLOTS == group_capacity) out_cursor = (out_cursor / VVM_STAGE_SLOTS + 1) * VVM_STAGE_SLOTS;
		const size_t group = out_cursor / VVM_STAGE_SLOTS;
		if (group != open_group) {
			lp.code.push_back(VvmIns{VVM_INS_W0(VVM_DOP_FLUSH, 0, 0, 0, 0), 0, 0, 0});
			open_group = group;
		}
		if (lp.out_list.size() < (group + 1) * VVM_STAGE_SLOTS) {
			lp.out_list.resize((group + 1) * VVM_STAGE_SLOTS, -1);
			lp.out_is16.resize((group + 1) * VVM_STAGE_SLOTS, 0);
		}
		return out_cursor;
	};
	for (size_t at_ins = 0; at_ins < ins.size(); ++at_ins) {
		const Decoded& d = ins[at_ins];
		if (!needed[at_ins]) {
			// a draw that is left out keeps its place in the numbering: the draws that stay must not change
			if (d.op == VVM_OP_random || d.op == VVM_OP_randomi || d.op == VVM_OP_random_add) { lp.rand_draws.push_back({(uint8_t)rand_ordinal, 1}); rand_ordinal += 1; }
			else if (d.op == VVM_OP_random_2x) { lp.rand_draws.push_back({(uint8_t)rand_ordinal, 2}); lp.rand_draws.push_back({(uint8_t)rand_ordinal, 2}); rand_ordinal += 2; }
			if (rand_ordinal > 255) { err = "more than 255 random draws in one script"; return VVM_E_TOO_LARGE; }
			continue;
		}
		VvmIns k{0, 0, 0, 0};
		switch (d.kind) {
		case K_ORD: case K_DST_ONLY: {
			uint32_t o[6] = {0, 0, 0, 0, 0, 0};
			for (size_t i = 0; i < d.in.size(); ++i) o[i] = operand(d.in[i]);
			for (size_t i = 0; i < d.out.size(); ++i)
				if (VVM_OPND_CLS(operand(d.out[i])) != VVM_OPC_VREG) { err = "an op destination is not a temp or output register"; return VVM_E_BAD_BLOB; }
			// per destination: the single-output device op and its aux / aux2
			uint32_t op1 = d.op, aux[2] = {0, 0}, aux2[2] = {0, 0};
			if (d.op == VVM_OP_random || d.op == VVM_OP_randomi || d.op == VVM_OP_random_add) { lp.rand_draws.push_back({(uint8_t)rand_ordinal, 1}); aux2[0] = rand_ordinal++; }
			else if (d.op == VVM_OP_random_2x) {                                                  // RT.cpp:1079-1111
				op1 = VVM_OP_random; aux2[0] = rand_ordinal; aux2[1] = rand_ordinal + 1;
				lp.rand_draws.push_back({(uint8_t)rand_ordinal, 2}); lp.rand_draws.push_back({(uint8_t)rand_ordinal, 2});
				rand_ordinal += 2;
			}
			else if (d.op == VVM_OP_mul_2x) op1 = VVM_OP_mul;                                     // RT.cpp:1743-1777
			else if (d.op == VVM_OP_b2i_2x) op1 = VVM_OP_b2i;                                     // RT.cpp:1875-1904
			else if (d.op == VVM_OP_sin_cos) aux[1] = 1;                                          // RT.cpp:1491-1516: sine first, cosine second
			if (rand_ordinal > 255) { err = "more than 255 random draws in one script"; return VVM_E_TOO_LARGE; }
			std::vector<SplitOut> outs;
			for (size_t i = 0; i < d.out.size(); ++i) outs.push_back(SplitOut{d.out[i], VVM_INS_W0(op1, d.in.size(), 1, aux[i], aux2[i])});
			emit_split(d, outs, o);
			break;
		}
		case K_ACQUIRE:
			if (multi_pass && !in_pass(d.dataset)) break;       // another pass's dataset
			if (chain_of(d.dataset)) acquired_second = (int)d.dataset; else acquired_dataset = (int)d.dataset;
			if (acquire_at == (size_t)-1) acquire_at = lp.code.size();
			k.w0 = VVM_INS_W0(VVM_DOP_ACQUIRE, 1, 0, chain_of(d.dataset), d.in.size() == 2 ? d.fn : VVM_ACQ_SIGN);      // (fn: the folded compare, see fold_compares_into_acquires)
			k.w1 = operand(d.in[0]) | (d.in.size() == 2 ? operand(d.in[1]) << 16 : 0u);
			lp.code.push_back(k);
			break;
		case K_OUTPUT: {
			uint32_t dop = VVM_DOP_OUTPUT32;
			if (d.op == VVM_OP_outputdata_half) dop = VVM_DOP_OUTPUT16;
			else if (d.op == VVM_OP_outputdata_half_from_half) dop = VVM_DOP_OUTPUT16_FROM16;
			else if (d.op == VVM_OP_outputdata_float_from_half) dop = VVM_DOP_OUTPUT32_FROM16;
			if (multi_pass && !in_pass(d.dataset)) break;       // another pass's dataset
			// An output that no acquireindex precedes, or whose dataset is not the acquired one, writes nothing
			// (VectorVMRuntime.cpp:565, 769): its list entry stays empty.
			const bool live = (acquired_dataset >= 0 && (uint32_t)acquired_dataset == d.dataset) || (acquired_second >= 0 && (uint32_t)acquired_second == d.dataset);
			const uint32_t chain = chain_of(d.dataset);
			for (size_t i = 0; i < d.pairs.size();) {
				if (light && dop == VVM_DOP_OUTPUT32 && live && rides_.size() < 2) {
					const uint32_t src = operand(d.pairs[i].first);
					bool direct = false;
					if (VVM_OPND_CLS(src) == VVM_OPC_VREG && rides_.size() < rides_wanted) {
						const size_t def = last_def(src, lp.code.size());
						direct = def < lp.code.size() && def > acquire_at;
						for (const Ride& r : rides_) if (r.src == src && last_def(src, r.at) == def) direct = false;      // (the same value twice: the second one is copied)
					}
					if (direct || out_cursor % VVM_STAGE_SLOTS >= VVM_LIGHT_GROUP_SLOTS) {
						if (lp.out_list.size() < VVM_STAGE_SLOTS) { lp.out_list.resize(VVM_STAGE_SLOTS, -1); lp.out_is16.resize(VVM_STAGE_SLOTS, 0); }
						lp.out_list[VVM_LIGHT_GROUP_SLOTS + rides_.size()] = (int32_t)(d.pairs[i].second - out0);
						rides_.push_back(Ride{src, lp.code.size(), direct});
						++i;
						continue;
					}
				}
				// sources in o0..o2; the destinations are consecutive entries of the output list, first one in w3
				const size_t at = open_slot();
				const size_t n = std::min<size_t>(std::min<size_t>(3, d.pairs.size() - i), group_capacity - at % VVM_STAGE_SLOTS);
				uint32_t o[3] = {0, 0, 0};
				if (at + n > 0xFFFFu) { err = "too many outputs"; return VVM_E_TOO_LARGE; }
				k.w3 = (uint32_t)at;
				for (size_t j = 0; j < n; ++j) {
					o[j] = operand(d.pairs[i + j].first);
					lp.out_list[at + j] = live ? (int32_t)(d.pairs[i + j].second - out0) : -1;      // output serial index
					lp.out_is16[at + j] = (dop == VVM_DOP_OUTPUT16 || dop == VVM_DOP_OUTPUT16_FROM16);
					if (lp.out_chain.size() < lp.out_list.size()) lp.out_chain.resize(lp.out_list.size(), 0);
					lp.out_chain[at + j] = (uint8_t)chain;
				}
				out_cursor = at + n;
				k.w0 = VVM_INS_W0(dop, 0, 0, n, chain);
				k.w1 = o[0] | (o[1] << 16); k.w2 = o[2];
				lp.code.push_back(k);
				i += n;
			}
			break;
		}
		case K_EXTFN: {
			// the built-in is bound at Exec time through ExtFunctionTable; only the shapes of the stated built-in set link
			const int ni = s.ext_num_inputs[d.fn], no = s.ext_num_outputs[d.fn];
			if (!(ni == 3 && (no == 1 || no == 3))) {
				snprintf(msg, sizeof(msg), "external function %u has %d inputs / %d outputs; the GPU built-ins are perlin3 (3/1) and curl_noise3 (3/3)", (unsigned)d.fn, ni, no);
				err = msg;
				return VVM_E_UNSUPPORTED_OP;
			}
			// one device instruction per output in use; aux = built-in, aux2 = output number (the operands are the inputs followed
			// by the outputs, RT.cpp:1450-1464; an unused output is the dummy register 0xFFFF)
			uint32_t o[6] = {0, 0, 0, 0, 0, 0};
			for (int i = 0; i < 3; ++i) o[i] = operand(d.in[(size_t)i]);
			std::vector<SplitOut> outs;
			for (int i = 0; i < no; ++i) {
				if (d.out[(size_t)i] == VVM_INVALID_REG) continue;
				if (VVM_OPND_CLS(operand(d.out[(size_t)i])) != VVM_OPC_VREG) { err = "an external function output is not a temp or output register"; return VVM_E_BAD_BLOB; }
				outs.push_back(SplitOut{d.out[(size_t)i], VVM_INS_W0(VVM_OP_external_func_call, 3, 1, no == 1 ? VVM_EXTFN_PERLIN3 : VVM_EXTFN_CURL_NOISE3, i)});
			}
			emit_split(d, outs, o);
			break;
		}
		case K_IDS: {
			if (d.op == VVM_OP_update_id && multi_pass && !in_pass(d.dataset)) break;       // another pass's dataset
			if (d.op == VVM_OP_update_id) {
				// staged like an output: one entry of the output list, one staging slot (in[1], the tag register, is not read:
				// VectorVMRuntime.cpp:1133 only loads its address)
				if (acquired_dataset < 0 || (uint32_t)acquired_dataset != d.dataset) { err = "update_id before the dataset's acquireindex"; return VVM_E_UNSUPPORTED_OP; }
				// its entry is always the last one of the group that is open (the kernel handles it in that slot only)
				const size_t at = open_group * VVM_STAGE_SLOTS + (VVM_STAGE_SLOTS - 1);
				if (at + 1 > 0xFFFFu) { err = "too many outputs"; return VVM_E_TOO_LARGE; }
				if (lp.out_list.size() < at + 1) { lp.out_list.resize(at + 1, -1); lp.out_is16.resize(at + 1, 0); }
				lp.out_list[at] = -2 - (int32_t)id_slot(d.dataset);      // (-2 - slot of the dataset's VvmIdDesc)
				k.w0 = VVM_INS_W0(VVM_DOP_UPDATE_ID, 0, 0, 1, d.dataset);
				k.w1 = operand(d.in[0]); k.w2 = 0; k.w3 = (uint32_t)at;
				lp.code.push_back(k);
			} else {
				// acquire_id: two destinations (ID, tag), no inputs; aux selects which
				uint32_t o[6] = {0, 0, 0, 0, 0, 0};
				std::vector<SplitOut> outs;
				for (size_t i = 0; i < 2; ++i) {
					if (d.out[i] == VVM_INVALID_REG) continue;
					if (VVM_OPND_CLS(operand(d.out[i])) != VVM_OPC_VREG) { err = "an acquire_id destination is not a temp or output register"; return VVM_E_BAD_BLOB; }
					outs.push_back(SplitOut{d.out[i], VVM_INS_W0(VVM_OP_acquire_id, 0, 1, i, id_slot(d.dataset))});
				}
				emit_split(d, outs, o);
			}
			break;
		}
		default: break;
		}
	}
	// the riding outputs, last one first (a copy that is inserted moves what follows it)
	for (size_t n = rides_.size(); n-- > 0;) {
		const Ride& r = rides_[n];
		const uint32_t ride_reg = VVM_OPND(VVM_OPC_VREG, first_ride + (uint32_t)n);
		if (!r.direct) {
			lp.code.insert(lp.code.begin() + (ptrdiff_t)r.at, VvmIns{VVM_INS_W0(VVM_DOP_COPY, 1, 1, 0, 0), r.src, 0, ride_reg << 16});
			continue;
		}
		// the instruction that makes the value writes the ride register, and so reads everyone who reads the value: up to the
		// instruction that puts the next value into the register it had
		const size_t def = last_def(r.src, r.at);
		ins_set_field(lp.code[def], 5, ride_reg);
		for (size_t q = def + 1; q < lp.code.size(); ++q) {
			VvmIns& k = lp.code[q];
			const uint32_t op = k.w0 & 0xFFu;
			const uint32_t reads = ordinary(k) ? (k.w0 >> 8) & 7u : (op >= VVM_DOP_OUTPUT32 && op <= VVM_DOP_OUTPUT32_FROM16) ? (k.w0 >> 16) & 0xFFu : 0u;
			for (uint32_t f = 0; f < reads; ++f) if (ins_field(k, f) == r.src) ins_set_field(k, f, ride_reg);
			if (ordinary(k) && ins_field(k, 5) == r.src) break;
		}
	}
	lp.code.push_back(VvmIns{VVM_INS_W0(VVM_OP_done, 0, 0, 0, 0), 0, 0, 0});
	lp.code.push_back(VvmIns{VVM_INS_W0(VVM_OP_done, 0, 0, 0, 0), 0, 0, 0});   // the kernel fetches one instruction ahead
	// pack the per-lane registers, then choose the kernel shape: up to 16 registers: everything in VGPRs at full occupancy
	// (four or eight instances per lane); up to 44 (plain scripts) or 32 resident registers + LDS slots: the four-wide kernels with
	// many registers; beyond: one instance per lane.  Registers past the resident ones live in LDS slots.
	const uint32_t packed = reallocate_registers(lp.code, num_loaded, num_loaded + nv + 2, light ? first_ride : 0xFFFFFFFFu);
	uint32_t V = 4, RV = 16;
	if (packed > 16) RV = lp.full_kernel ? 32 : kFileWidePlain;
	if (force_narrow || packed > RV + kMaxSpillWide) { V = 1; RV = 32; }
	if (packed > RV + (V == 4 ? kMaxSpillWide : kMaxSpillNarrow)) {
		snprintf(msg, sizeof(msg), "script needs %u per-lane registers; the executor holds at most %u", packed, RV + kMaxSpillNarrow);
		err = msg;
		return VVM_E_TOO_LARGE;
	}
	if (chains && (lp.full_kernel || V != 4)) { chains_failed = true; err = "two chains need a kernel that has them"; return VVM_E_UNSUPPORTED_OP; }
	lp.out_chain.resize(lp.out_list.size(), 0);
	if (light) {
		// the light eight-wide kernel: eight resident registers, one load instruction of 32-bit streams into registers 0.., ops of at
		// most four operands, none of the transcendental / noise / ID instructions, one acquired dataset (of a script with several: the pass's),
		// and at most eight outputs: with
		// three workgroups per CU the LDS has room for six staging slots each, two more 32-bit outputs wait in registers, and a second
		// group of slots would have to resolve the tile's look-back on the spot (measured: 1.04 ms against 0.84 ms of the general
		// kernel on eight outputs of 64 Mi instances; so was parking two outputs in global memory: 0.85 ms)
		bool ok = packed <= 8 && V == 4 && !lp.full_kernel && !lp.uses_ids && lp.num_chains == 1 && prefetch16.empty() && !prefetch32.empty() && prefetch32.size() <= 8;
		for (const VvmIns& k : lp.code) {
			const uint32_t op = k.w0 & 0xFFu;
			if (op != VVM_OP_done && op < VVM_DOP_LOADN16 && ((k.w0 >> 8) & 7u) > 4u) ok = false;
			if (op == VVM_DOP_FLUSH) ok = false;
		}
		if (ok) ok = (lp.code[0].w0 & 0xFFu) == VVM_DOP_LOADN && (lp.code[0].w0 >> 24) == 0;      // (the loads address registers 0..)
		if (!ok) { err = "script does not fit the light eight-wide kernel"; return VVM_E_UNSUPPORTED_OP; }
		lp.light = true;
	}
	lp.num_vregs = packed;
	lp.instances_per_lane = V;
	lp.resident_regs = RV;
	lp.many_registers = (RV > 16 && V == 4) || packed > RV;
	lp.lds_spill_slots = packed > RV ? packed - RV : 0;
	{
		// Which of the two four-wide kernels a program of at most sixteen registers runs on.  Measured (tools/full_kernel_choice.py,
		// tools/reg_cliff.py --four-wide): the kernel of many registers saves about 4 us per ordinary instruction at 16 Mi instances
		// (its hand-written routines; 0.1 us at 64 Ki instances, where the 9 instructions of a short script run 0.4 us slower on it)
		// and, with two workgroups per CU instead of three, loses 10 to 16 us per transcendental function or perlin3 call and over
		// 100 us per curl_noise3 call.  In ordinary instructions:
		uint32_t ordinary_ins = 0, heavy_weight = 0;
		for (const VvmIns& k : lp.code) {
			const uint32_t op = k.w0 & 0xFFu;
			if (op == VVM_OP_done || op >= VVM_DOP_LOADN16) continue;
			if (op == VVM_OP_external_func_call) heavy_weight += ((k.w0 >> 16) & 0xFFu) == VVM_EXTFN_CURL_NOISE3 ? 16u : 5u;      // (one instruction per output of the call)
			else if (op == VVM_OP_fmod || op == VVM_OP_fmod_add) heavy_weight += 2u;
			else if (op == VVM_OP_pow || op == VVM_OP_atan2 || op == VVM_OP_asin || op == VVM_OP_acos || op == VVM_OP_atan || op == VVM_OP_tan) heavy_weight += 4u;
			else if (g_ops[op].heavy) heavy_weight += 3u;
			else ++ordinary_ins;
		}
		lp.prefers_routines = V == 4 && ordinary_ins >= heavy_weight + 12u;
	}
	for (VvmIns& k : lp.code) {
		const uint32_t op = k.w0 & 0xFFu;
		if (op == VVM_OP_done || op >= VVM_DOP_LOADN16) k.w0 = (k.w0 & ~(7u << 8)) | (VVM_NIN_SPECIAL << 8);
	}

	lp.bytes_in_per_instance += 4u * (uint32_t)prefetch32.size() + 2u * (uint32_t)prefetch16.size();      // the streams this pass loads
	for (uint32_t i = 0; i < s.num_outputs_remapped; ++i) if (s.out_dst[i] != 0xFFFF) lp.bytes_out_per_instance += (s.out_type[i] == 2) ? 2 : 4;
	return VVM_OK;
}

}  // namespace vvm
