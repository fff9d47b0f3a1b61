// vvm_host.cpp -- C ABI of the executor (include/vvm_abi.h): state allocation and the Exec launcher.
//
//   vvm_alloc_state / vvm_free_state   <- AllocVectorVMState / FreeVectorVMState   VectorVMRuntime.cpp:2311-2425
//   vvm_exec*                          <- ExecVectorVMState                        VectorVMRuntime.cpp:2182-2309
// What the reference does per Exec on the CPU and what happens to it here:
//   ext-fn table check, zero NumOutputPerDataSet   :2184-2195   -> host, before the launch
//   VVMBuildMapTableCaches (first Exec)            :2198-2202   -> host, resolve_maps()
//   constants broadcast                            :2204-2207   -> host packs one dword per constant, the kernel broadcasts
//   chunking                                       :2210-2261   -> the kernel does not chunk; Internal.* is computed as the reference does, and the
//                                                                   chunk size goes to the ID instructions, whose results depend on it
//   SetupBatchStatePtrs                            :315-443     -> host packs stream addresses (InstanceOffset applied)
//   chunk loop                                     :2276-2295   -> ONE kernel launch (vvm_kernels.hip)
//   DataSetAccessIndex = count - 1                 :2305-2308   -> host, after the 4-byte-per-dataset read-back
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vvm_abi.h"
#include "vvm_device.h"
#include "vvm_kernels.h"
#include "vvm_threaded.h"
#include "vvm_program.h"

#include "vvm_perlin_table.inc"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& what)
{
	g_last_error = what;
	return code;
}
int hip_fail(hipError_t e, const char* where)
{
	g_last_error = std::string(where) + ": " + hipGetErrorString(e);
	return VVM_E_HIP;
}
#define HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

constexpr int kParamSlots = 8;

// hipMemset on device memory returns before the fill has run (it is enqueued on the null stream), and the Execs run on
// non-blocking streams, which do not wait for the null stream: a kernel launched right after could run ahead of the fill and have
// its look-back words wiped under it.  Every fill the library makes is therefore waited for.
hipError_t memset_and_wait(void* dst, int value, size_t bytes)
{
	const hipError_t e = hipMemset(dst, value, bytes);
	return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

// Pinned staging + device copy of the per-Exec parameter block, used round-robin so that several Execs can be queued.
struct ParamRing {
	uint8_t* host[kParamSlots] = {};
	uint8_t* dev[kParamSlots] = {};
	hipEvent_t done[kParamSlots] = {};
	bool pending[kParamSlots] = {};
	size_t slot_bytes = 0;
	int next = 0;

	int reserve(size_t bytes)
	{
		if (bytes <= slot_bytes) return VVM_OK;
		HIP_TRY(hipDeviceSynchronize());
		release();
		slot_bytes = std::max<size_t>(4096, (bytes + 4095) & ~(size_t)4095);
		for (int i = 0; i < kParamSlots; ++i) {
			HIP_TRY(hipHostMalloc((void**)&host[i], slot_bytes, hipHostMallocDefault));
			HIP_TRY(hipMalloc((void**)&dev[i], slot_bytes));
			HIP_TRY(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
			pending[i] = false;
		}
		return VVM_OK;
	}
	int acquire(int& slot)
	{
		slot = next;
		next = (next + 1) % kParamSlots;
		if (pending[slot]) { HIP_TRY(hipEventSynchronize(done[slot])); pending[slot] = false; }
		return VVM_OK;
	}
	void release()
	{
		for (int i = 0; i < kParamSlots; ++i) {
			if (host[i]) (void)hipHostFree(host[i]);
			if (dev[i]) (void)hipFree(dev[i]);
			if (done[i]) (void)hipEventDestroy(done[i]);
			host[i] = dev[i] = nullptr; done[i] = nullptr; pending[i] = false;
		}
		slot_bytes = 0;
	}
};

// Working memory of vvm_exec_noalloc, the entry point for callers that must not allocate: the launch words (with the survivor
// counts in their spare words), one look-back word per tile and, for Execs that run the lean form of their script (vvm_state::lean),
// one ride word per lane and tile up to VVM_NOALLOC_RIDE_TILES tiles (larger Execs run the whole script there).  The array is part of the code object, so every device the
// library runs on holds a copy of its own from the moment the code object is loaded there until the process ends.  Its name is
// only a handle on the host: the address on a device comes from hipGetSymbolAddress with that device current (ensure_device_info),
// and that address is what host code passes on.
#define VVM_NOALLOC_MAX_TILES 131072u
#define VVM_NOALLOC_STATUS_OFF ((sizeof(VvmLaunchState) + 255u) & ~(size_t)255u)
#define VVM_NOALLOC_RIDE_TILES 4096u
#define VVM_NOALLOC_RIDE_OFF (VVM_NOALLOC_STATUS_OFF + 8u * (size_t)VVM_NOALLOC_MAX_TILES)
#define VVM_NOALLOC_SCRATCH_BYTES (VVM_NOALLOC_RIDE_OFF + 4u * (size_t)VVM_BLOCK_THREADS * VVM_NOALLOC_RIDE_TILES)
static __device__ __align__(256) unsigned char scratch[VVM_NOALLOC_SCRATCH_BYTES];

// Host-side twin of the epoch word of one set of launch words (VvmLaunchState::epoch): every launch through the set tags its
// look-back words with the epoch and leaves the next one behind, 1 .. 2^30 - 1 and round again.  Once the count has gone round,
// a word written 2^30 launches ago (by an Exec with more tiles than any since) could carry the live tag again, so the launch
// that follows the wrap clears the look-back words first, on its stream.
struct EpochMirror {
	uint32_t next = 1;        // the epoch the next launch through the set will use
	bool wrapped = false;     // the count has just gone round: clear the look-back words ahead of the next launch
	void launched() { next = (next + 1u) & 0x3FFFFFFFu; if (next == 0) { next = 1; wrapped = true; } }
	hipError_t before_launch(unsigned long long* status, size_t words, hipStream_t stream)
	{
		if (!wrapped) return hipSuccess;
		wrapped = false;
		return status && words ? hipMemsetAsync(status, 0, words * 8, stream) : hipSuccess;
	}
};

struct DeviceInfo {
	int num_cus = 256;
	EpochMirror scratch_epoch[64];      // per device ordinal: of the launch words in that device's scratch array
	std::mutex noalloc_mutex[64];       // per device ordinal: one vvm_exec_noalloc at a time (they share the scratch array)
	bool ready[64] = {};      // per device ordinal: the kernels' LDS limit has been raised there
	uint8_t* scratch_ptr[64] = {};      // per device ordinal: where that device's copy of `scratch` lies
	uint32_t* xorwow_jump[64] = {};     // per device ordinal: the jump table of the reference's random generator (built on first use)
	// per device ordinal: the streams that states run on when the caller names none.  A few, shared: a process that holds streams
	// by the thousand (one per emitter's state) uses up what the device has for ALL processes, and a process that starts then fails
	// inside the runtime while it creates its first queue (seen with 1024 states of one stream each).
	hipStream_t state_streams[64][4] = {};
	uint32_t next_state_stream[64] = {};
};
DeviceInfo g_dev;
std::mutex g_dev_mutex;
// counts the states made and released so far: what is remembered about a set of states (the plan of a batch) holds only while it
// stands still, since an address may come back as another state
std::atomic<uint64_t> g_state_generation{0};
hipError_t reset_launch_state(VvmLaunchState* d);

// The stream of a new state on the current device: the next of the device's few, made on first use and kept for the process.
int state_stream(int dev, hipStream_t* out)
{
	std::lock_guard<std::mutex> lock(g_dev_mutex);
	if (dev < 0 || dev >= 64) return fail(VVM_E_INVALID_ARG, "device ordinal out of range");
	const uint32_t k = g_dev.next_state_stream[dev]++ & 3u;
	if (!g_dev.state_streams[dev][k]) HIP_TRY(hipStreamCreateWithFlags(&g_dev.state_streams[dev][k], hipStreamNonBlocking));
	*out = g_dev.state_streams[dev][k];
	return VVM_OK;
}

int ensure_device_info()
{
	std::lock_guard<std::mutex> lock(g_dev_mutex);
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (dev < 0 || dev >= 64) return fail(VVM_E_INVALID_ARG, "device ordinal out of range");
	if (g_dev.ready[dev]) return VVM_OK;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, dev));
	g_dev.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	(void)vvm_kernel_max_dynamic_lds();      // (a function attribute belongs to the device that is current)
	// the kernels jump to the routine offsets that this file writes into the programs: both must come from the same table
	if (vvm_kernel_threaded_layout() != VVM_T_LAYOUT) return fail(VVM_E_HIP, "vvm_kernels.o and vvm_host.o were built from different routine tables (vvm_threaded_gen.h): rebuild");
	uint8_t* ptr = nullptr;
	HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
	if (!ptr) return fail(VVM_E_HIP, "hipGetSymbolAddress(scratch) returned no address");
	HIP_TRY(memset_and_wait(ptr, 0, VVM_NOALLOC_SCRATCH_BYTES));      // tag 0 is never valid
	HIP_TRY(reset_launch_state((VvmLaunchState*)ptr));
	g_dev.scratch_ptr[dev] = ptr;
	g_dev.ready[dev] = true;
	return VVM_OK;
}

}  // namespace

struct vvm_state {
	vvm::ThawedScript script;
	vvm::LinkedProgram wide, narrow;       // narrow is linked on demand (VVM_EXEC_NO_VGPR_WIDE)
	bool narrow_ready = false;
	vvm::LinkedProgram light;              // the script linked for the light eight-wide kernel, when it fits (linked at vvm_alloc_state)
	bool light_ready = false;
	VvmIns* d_code_light = nullptr;
	// ... and of the later passes of a script with several acquired datasets, where a pass has one dataset and fits (entry pass - 1; no
	// code: the pass has no light program)
	std::vector<vvm::LinkedProgram> light_more;
	std::vector<VvmIns*> d_code_light_more;
	// a script with acquireindex on several datasets runs one pass per dataset: the programs of passes 1.. (pass 0 = wide / narrow)
	std::vector<vvm::LinkedProgram> wide_more, narrow_more;
	std::vector<VvmIns*> d_code_wide_more, d_code_narrow_more;
	// ... and, where `wide` takes the datasets two to a pass (LinkedProgram::paired): the script as one program per dataset, linked on
	// demand for the Execs that need kernels without a second chain (those that read the reference generator's words)
	vvm::LinkedProgram solo;
	std::vector<vvm::LinkedProgram> solo_more;
	VvmIns* d_code_solo = nullptr;
	std::vector<VvmIns*> d_code_solo_more;
	bool solo_ready = false;

	// A script whose record is too wide for the register file of the light eight-wide kernel, but whose extra width is streams that
	// it hands through unchanged, also has this form: a program with everything the script computes (`lean`: for the light kernel,
	// else the general eight-wide one, else the four-wide one with the transcendental functions), and the list of handed-through
	// outputs, which vvm_through_kernel moves behind that program's launch from the acquireindex results the program leaves per
	// tile and lane (`d_ride`).  Which Execs run that way: prepare().
	vvm::LinkedProgram lean;
	VvmIns* d_code_lean = nullptr;
	bool lean_ready = false;
	std::vector<uint32_t> through;         // output serial indices that are handed through
	uint32_t* d_ride = nullptr;
	uint32_t ride_capacity = 0;            // tiles

	VvmIns* d_code_wide = nullptr;
	VvmIns* d_code_narrow = nullptr;
	uint8_t* d_perlin = nullptr;
	VvmLaunchState* d_launch = nullptr;
	EpochMirror epoch;                     // of *d_launch
	unsigned long long* d_status = nullptr;
	uint32_t status_capacity = 0;          // tiles
	int32_t* d_counts = nullptr;
	int32_t* counts_src = nullptr;         // where the last Exec left its counts: d_counts, the caller's vvm_exec_options::DeviceCounts, or the device's scratch
	// what the last enqueued Exec ran on: the state's own launch words, look-back words and counts, or (vvm_exec_noalloc) the device's scratch
	VvmLaunchState* cur_launch = nullptr;
	unsigned long long* cur_status = nullptr;
	size_t cur_status_words = 0;
	int32_t* cur_counts = nullptr;
	bool cur_counts_in_launch = false;
	bool any_ids = false;                  // some pass of the script has update_id / acquire_id
	bool counts_in_launch = false;         // d_counts lies in the spare words of *d_launch: counts and launch words come back in one copy
	int32_t* h_counts = nullptr;           // pinned, nc = number of datasets: [0, nc) survivor counts and [nc, 2 nc) MaxUsedID per dataset as read
	                                       // back from d_counts (same layout there), [2 nc, 3 nc) MaxUsedID per dataset on its way to the
	                                       // device, from word mirror_word() on the launch words as read back
	ParamRing ring;
	hipStream_t stream = nullptr;          // one of the device's shared streams (state_stream)
	hipStream_t last_stream = nullptr;
	bool in_flight = false;                // an asynchronous Exec has been enqueued and not finished yet
	uint64_t batch_stamp = 0;              // the last vvm_exec_batch call this state took part in
	bool results_pending = false;          // a kernel has run whose counts have not been copied back yet

	// map caches, VectorVMRuntime.cpp:485-553 (constants only: inputs are classified at link time)
	bool maps_ready = false;
	std::vector<uint8_t>  const_table;     // ConstMapCacheIdx
	std::vector<uint16_t> const_dword;     // ConstMapCacheSrc

	// kernel timing: one event pair per launch since the last collect, on the stream the kernel runs on
	bool timing = false;
	bool timing_span = false;              // ... with ONE event pair around all Execs between two collects instead of a pair per Exec
	bool span_open = false;                // the span's first event has been recorded
	bool span_closed = false;              // ... and its second one, by a vvm_exec_finish behind the last Exec so far
	uint32_t span_launches = 0;
	std::vector<hipEvent_t> ev_start, ev_stop;
	size_t ev_used = 0;
	uint32_t execs_since_status_reset = 0;

	// persistent IDs (update_id / acquire_id): scratch of the MaxUsedID pass, and what the write-back after the Exec needs
	uint32_t* d_chunk_prefix = nullptr; uint32_t chunk_capacity = 0;
	int32_t*  d_id_scratch = nullptr;   uint32_t id_scratch_capacity = 0;
	std::vector<int32_t> id_num_free_start, id_num_spawned_start;      // per dataset: the counters when the Exec was enqueued
	uint32_t id_buffer_pieces = 0;         // pieces of d_chunk_prefix / d_id_scratch (chunk_capacity / id_scratch_capacity entries each)
	uint32_t id_datasets = 0;              // one more than the highest dataset an update_id / acquire_id names
	size_t num_datasets() const { return std::max<uint32_t>(std::max<uint32_t>(script.max_output_dataset, 1), id_datasets); }
	size_t mirror_word() const { return 3 * num_datasets() + 4; }
	uint32_t* d_rand_words = nullptr;  // the reference's generator: 32 bits per draw and instance of the Exec in flight (vvm_exec_options::RandXorwowState)
	size_t    rand_words_capacity = 0; // words
	uint32_t  rand_stride = 0;         // words from one draw's to the next draw's
	uint32_t  last_width = 0;          // instances per lane of the kernel the last Exec ran on
	int32_t*  d_id_snapshot = nullptr; // scripts with acquire_id and update_id on one dataset: the FreeIDTable entries the chunk's acquire_id reads
	uint32_t  id_snapshot_capacity = 0;
	bool      serial_done = false;     // ... and: the last Exec was run chunk after chunk, to the end, by the enqueue call
	int       device = -1;             // ordinal of the device that holds the state's memory (the one current at vvm_alloc_state)
};

namespace {

// The program as it lies in device memory (VvmDevIns): the linker's instructions, each followed by what the hand-written core of the
// eight-wide kernels executes from (vvm_threaded.h): the routine for the instruction's op and operand kinds, and its operands as bytes.
// An instruction gets a routine when it is an ordinary one whose op has routines, whose operands are per-lane registers of the file
// (light program: eight registers and the two riding ones; a program of many registers: 44 or 32; otherwise sixteen) or uniform registers, not all of them uniform, and
// whose destination is a register of the file.  The other kernels read the linker's four words only.
void encode_device_program(const vvm::LinkedProgram& lp, std::vector<VvmDevIns>& out)
{
	// (a program of more than sixteen registers runs four instances per lane: its registers are four VGPRs each, VVM_DESC_REGS_OF_4.
	// The kernel of plain scripts has 44 of them in its file, the one with the transcendental / noise / ID instructions 32: a plain
	// program that can be sent there too -- as one of several passes, or for the reference generator's words -- gets routines for
	// what both have in their files; the rest of its instructions run without one on either.)
	const bool both_kernels = lp.full_kernel || lp.num_passes > 1 || !lp.rand_draws.empty();
	const uint32_t file_regs = lp.light ? VVM_LIGHT_RIDE_REG + 2u : lp.many_registers ? (both_kernels ? 32u : 44u) : 16u;
	const uint32_t vgprs_per_reg = lp.many_registers ? 4u : 8u;
	out.clear();
	out.reserve(lp.code.size() + 1);
	for (const VvmIns& k : lp.code) {
		VvmDevIns d = {k, 0, 0, 0, 0};
		const uint32_t op = k.w0 & 0xFFu, nin = (k.w0 >> 8) & 7u;
		if (nin >= 1 && nin <= 5 && op != VVM_OP_done) {
			const uint32_t o[6] = {k.w1 & 0xFFFFu, k.w1 >> 16, k.w2 & 0xFFFFu, k.w2 >> 16, k.w3 & 0xFFFFu, k.w3 >> 16};
			uint32_t byte[5] = {0, 0, 0, 0, 0}, uniform_mask = 0;
			bool ok = VVM_OPND_CLS(o[5]) == VVM_OPC_VREG && VVM_OPND_IDX(o[5]) < file_regs;
			for (uint32_t i = 0; i < nin && ok; ++i) {
				const uint32_t cls = VVM_OPND_CLS(o[i]), idx = VVM_OPND_IDX(o[i]);
				if (cls == VVM_OPC_VREG && idx < file_regs) byte[i] = idx * vgprs_per_reg;
				else if (cls == VVM_OPC_UNIF && idx < VVM_MAX_UNIFORMS) { byte[i] = idx; uniform_mask |= 1u << i; }
				else ok = false;
			}
			const uint32_t routine = ok ? vvm_threaded_routine(op, nin, uniform_mask) : 0u;
			if (routine != 0) {
				d.routine = routine;
				d.opnds = byte[0] | byte[2] << 8 | byte[3] << 16 | byte[4] << 24;
				d.opnd_b = byte[1];
				d.dst = VVM_OPND_IDX(o[5]) * vgprs_per_reg;
			}
		}
		out.push_back(d);
	}
	out.push_back(VvmDevIns{VvmIns{VVM_INS_W0(VVM_OP_done, VVM_NIN_SPECIAL, 0, 0, 0), 0, 0, 0}, 0, 0, 0, 0});      // (the core fetches two instructions ahead)
}

int upload_code(const vvm::LinkedProgram& lp, VvmIns** d_code)
{
	std::vector<VvmDevIns> dev;
	encode_device_program(lp, dev);
	HIP_TRY(hipMalloc((void**)d_code, dev.size() * sizeof(VvmDevIns)));
	HIP_TRY(hipMemcpy(*d_code, dev.data(), dev.size() * sizeof(VvmDevIns), hipMemcpyHostToDevice));
	return VVM_OK;
}

int ensure_status(vvm_state* st, uint32_t tiles)
{
	if (tiles <= st->status_capacity) return VVM_OK;
	if (st->last_stream) HIP_TRY(hipStreamSynchronize(st->last_stream));
	if (st->d_status) HIP_TRY(hipFree(st->d_status));
	const uint32_t cap = std::max<uint32_t>(1024, tiles + tiles / 4);
	HIP_TRY(hipMalloc((void**)&st->d_status, (size_t)cap * 8));
	HIP_TRY(memset_and_wait(st->d_status, 0, (size_t)cap * 8));     // tag 0 is never valid
	st->status_capacity = cap;
	return VVM_OK;
}

int resolve_maps(vvm_state* st, const vvm_exec_context* ctx)
{
	const vvm::ThawedScript& s = st->script;
	if (ctx->ConstantTableCount > 0xFF) return fail(VVM_E_INVALID_ARG, "more than 255 constant tables");     // RT.cpp:488
	st->const_table.assign(s.num_consts, 0);
	st->const_dword.assign(s.num_consts, 0);
	for (uint32_t i = 0; i < s.num_consts; ++i) {
		const uint32_t remapped = s.const_remap[i];
		uint32_t acc = 0;
		bool found = false;
		for (int j = 0; j < ctx->ConstantTableCount; ++j) {
			const uint32_t ndw = (uint32_t)ctx->ConstantTableNumBytes[j] >> 2;
			if (acc + ndw > remapped) {
				st->const_table[i] = (uint8_t)j;
				st->const_dword[i] = (uint16_t)(remapped - acc);
				found = true;
				break;
			}
			acc += ndw;
		}
		if (!found) return fail(VVM_E_INVALID_ARG, "a constant the script reads lies past the end of the constant tables");
	}
	st->maps_ready = true;
	return VVM_OK;
}

// A four-wide program of at most sixteen registers has two kernels to choose from: the one with the compiled interpreter loop (four
// workgroups per CU; with the transcendental / noise / ID instructions three) and the one of many registers, which runs ordinary
// instructions through the hand-written routines (two per CU): see LinkedProgram::prefers_routines.
// VVM_FOUR_WIDE_ROUTINES=0 / 1: never / always the latter (to measure the choice).
bool four_wide_prefers_routines(const vvm::LinkedProgram& lp)
{
	static const char* forced = getenv("VVM_FOUR_WIDE_ROUTINES");
	if (forced && (forced[0] == '0' || forced[0] == '1')) return forced[0] == '1';
	return lp.prefers_routines;
}

struct PreparedExec {
	const vvm::LinkedProgram* lp = nullptr;
	const VvmIns* d_code = nullptr;
	uint32_t tiles = 0;
	size_t bytes = 0;           // parameter block: desc + consts + in_ptrs + out_ptrs
	size_t off_consts = 0, off_in = 0, off_load = 0, off_out = 0, off_ids = 0;
	uint32_t chunk_instances = 0, num_chunks = 0;      // the reference's chunking of this Exec
	uint32_t instances_per_lane = 4;                   // of the kernel this Exec runs on (8, 4 or 1)
	bool light = false;                                // ... the light eight-wide kernel
	bool in_batch = false;                             // set by the caller: the Exec is a member of a batch (decides the kernel of small Execs, see prepare)
	uint32_t lean_max_tiles = 0;                       // set by the caller: the lean form of the script only up to so many tiles (vvm_exec_noalloc; 0: no limit)
	bool lean = false;                                 // the Exec runs the state's lean program; vvm_through_kernel follows it
	uint64_t ride_addr = 0;                            // where the program leaves its per-lane acquireindex results for that kernel (0: nowhere)
	uint32_t id_buffers = 0;                           // which piece of the state's update_id buffers this pass works in (fused passes run together)
	bool reference_random = false;                     // the Exec's random ops read the reference generator's words (then it runs on a FULL kernel)
	// One chunk of an Exec that is run chunk after chunk (LinkedProgram::serial_chunks): set by the caller.  The chunk's first instance
	// within the whole Exec (advancing inputs, exec_index and the random generator's key start there) and the number of survivors the
	// chunks before it have written (outputs and IDTable positions start there).
	uint32_t chunk_first = 0, out_base = 0;
	const int32_t* out_bases = nullptr;      // ... per dataset (a script that writes several), num_out_bases of them; else out_base for every output
	uint32_t num_out_bases = 0;
	bool     serial_chunk = false;
	const int32_t* free_read_snapshot = nullptr;       // ... and where the copy of the FreeIDTable entries lies that the chunk's acquire_id reads
	bool full() const { return lp->full_kernel || reference_random; }
	// what vvm_launch_exec_kernel takes as many_registers (a program with two chains runs four-wide on the kernel with 32 registers: it has the second chain)
	bool launch_many() const { return light || lp->many_registers || (instances_per_lane == 4 && (lp->num_chains > 1 || four_wide_prefers_routines(*lp))); }
	// registers in LDS slots on the kernel the Exec runs on (a plain program of many registers on a kernel with the transcendental
	// functions, for the reference generator's words or as one of several fused passes: 32 resident ones there, not 44)
	uint32_t lds_slots() const { return full() ? lp->lds_slots_with_32_resident() : lp->lds_spill_slots; }
};

int g_page_kb = 64;      // vm.PageSizeInKB, VectorVMRuntime.cpp:44-50
// VVM_NO_INLINE_PARAMS: every parameter block goes through device memory (to measure what the kernel-argument route saves)
const bool g_no_inline_params = getenv("VVM_NO_INLINE_PARAMS") != nullptr;

// The reference's chunking (VectorVMRuntime.cpp:2210-2261), which also fills ctx->Internal.
void reference_chunking(const vvm::ThawedScript& s, vvm_exec_context* ctx, uint32_t& chunk_instances, uint32_t& num_chunks)
{
	// fixed per-batch overhead, VectorVMRuntime.cpp:2391-2414 (sizeof(FVectorVMBatchState) = 224 on a 64-bit target, rounded up to 256)
	const size_t num_ptr_regs = (size_t)s.num_temps + s.num_consts + 2u * (size_t)s.num_inputs + s.num_output_instructions;
	const size_t overhead = 256 + 16u * (size_t)s.max_output_dataset + 9u * (size_t)s.max_ext_fn_registers + 16u * (size_t)s.num_dummy_regs +
	                        9u * num_ptr_regs + 512u;
	const size_t page = (size_t)g_page_kb << 10;
	uint32_t total_loops = ((uint32_t)ctx->NumInstances + 3u) >> 2;
	if (total_loops < 1) total_loops = 1;
	const size_t per_loop = 16u * (size_t)s.num_temps + s.max_output_dataset;
	// (unsigned as in the reference: a page smaller than the overhead wraps to "everything fits"; a page exactly as large as
	// the overhead would divide by zero there and counts as one byte here)
	const size_t avail = page != overhead ? page - overhead : 1;
	const size_t need = (((size_t)total_loops + 3) & ~(size_t)3) * per_loop;
	uint32_t chunks = 1, max_loops = total_loops;
	if (avail < need) {
		chunks = (uint32_t)((need + avail - 1) / avail);
		max_loops = (total_loops + chunks - 1) / chunks;
	}
	ctx->Internal.PerBatchRegisterDataBytesRequired = (uint32_t)(max_loops * per_loop);
	ctx->Internal.NumBytesRequiredPerBatch = (uint32_t)overhead + ctx->Internal.PerBatchRegisterDataBytesRequired;
	ctx->Internal.MaxChunksPerBatch = chunks;
	ctx->Internal.MaxInstancesPerChunk = max_loops << 2;
	chunk_instances = max_loops << 2;
	num_chunks = ((uint32_t)ctx->NumInstances + chunk_instances - 1) / chunk_instances;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }

// Validates the context and sizes its parameter block.
int link_more_passes(const vvm::ThawedScript& s, bool narrow, const vvm::LinkedProgram& first, std::vector<vvm::LinkedProgram>& more, std::vector<VvmIns*>& d_code)
{
	more.resize(first.num_passes - 1);
	d_code.assign(first.num_passes - 1, nullptr);
	for (uint32_t k = 1; k < first.num_passes; ++k) {
		std::string err;
		int rc = vvm::link_program(s, narrow, more[k - 1], err, k, false, nullptr, first.paired ? 1 : 0);
		if (rc != VVM_OK) return fail(rc, err);
		rc = upload_code(more[k - 1], &d_code[k - 1]);
		if (rc != VVM_OK) return rc;
	}
	return VVM_OK;
}

// The lean form of a script (vvm_state::lean), where it can pay: the whole script does not fit the light eight-wide kernel (every
// stream it reads sits in a per-lane register from the top of a tile), at least two of its outputs hand a 32-bit input stream
// through unchanged, and without those it does fit.  From which size on an Exec takes the form: prepare().  Not built is not an error.
// (host only: vvm_inspect_blob plans the same way without a device)
bool plan_through_programs(const vvm::ThawedScript& script, const vvm::LinkedProgram& w, bool fits_light, vvm::LinkedProgram& lean, std::vector<uint32_t>& through)
{
	through.clear();
	if (getenv("VVM_NO_THROUGH")) return false;
	// (one acquired dataset, or two in one pass: then the first one's outputs, see LinkedProgram::through_input)
	if (fits_light || w.num_passes != 1 || w.uses_ids || w.acquire_order.empty() || w.acquire_order.size() > 2) return false;
	const size_t O = w.through_input.size();
	for (size_t o = 0; o < O; ++o) {
		if (w.through_input[o] == -2) { through.clear(); return false; }
		if (w.through_input[o] >= 0) through.push_back((uint32_t)o);
	}
	if (through.size() < 2) { through.clear(); return false; }
	std::string why;
	std::vector<uint8_t> select(O, 0);
	for (size_t o = 0; o < O; ++o) select[o] = w.through_input[o] < 0;
	lean = vvm::LinkedProgram();
	if (vvm::link_program(script, false, lean, why, 0, true, &select) == VVM_OK) return true;
	// what the script computes does not fit the light kernel: the general eight-wide kernel leaves the ride words too, when the
	// whole script is beyond it (more than sixteen registers) and the rest is not
	lean = vvm::LinkedProgram();
	bool ok = (w.many_registers || w.instances_per_lane == 1) && vvm::link_program(script, false, lean, why, 0, false, &select) == VVM_OK;
	// (the general eight-wide kernel has the second chain; a lean program with transcendental or noise functions runs four instances
	// per lane on the kernel that has those, with sixteen registers, and has one chain)
	ok = ok && lean.instances_per_lane == 4 && !lean.many_registers && !lean.uses_ids && lean.num_passes == 1 && (!lean.full_kernel || lean.num_chains == 1);
	if (!ok) { lean = vvm::LinkedProgram(); through.clear(); }
	return ok;
}

int plan_through(vvm_state* st)
{
	if (!plan_through_programs(st->script, st->wide, st->light_ready, st->lean, st->through)) return VVM_OK;
	const int rc = upload_code(st->lean, &st->d_code_lean);
	if (rc != VVM_OK) return rc;
	st->lean_ready = true;
	return VVM_OK;
}

// The stream addresses of the handed-through pairs of an Exec, as fill_params forms them (element [InstanceOffset] of each stream).
// `first`: the pairs from that one of the state's list on, as many as a block takes (a record with more handed-through streams
// takes several launches of the copy kernel).
int fill_through(vvm_state* st, const vvm_exec_context* ctx, VvmThroughArgs& a, size_t first = 0)
{
	const vvm::ThawedScript& s = st->script;
	a.num_pairs = 0;
	for (size_t k = first; k < st->through.size() && k < first + VVM_THROUGH_MAX_PAIRS; ++k) {
		const uint32_t o = st->through[k];
		const vvm::InputInfo& in = st->wide.inputs[(size_t)st->wide.through_input[o]];
		if (o >= s.num_outputs_remapped || s.out_dst[o] == 0xFFFF) continue;
		const uint8_t dso = s.out_dataset[o];
		const uint16_t ty = s.out_type[o];
		if ((int)in.dataset >= ctx->DataSets.num || (int)dso >= ctx->DataSets.num || ty > 2 || (ty == 2) != (in.type == 2))
			return fail(VVM_E_INVALID_ARG, "script reads or writes a dataset the context does not provide");
		const vvm_dataset_meta& dsi = ctx->DataSets.data[in.dataset];
		const vvm_dataset_meta& dsw = ctx->DataSets.data[dso];
		const uint32_t ii = dsi.InputRegisterTypeOffsets[in.type] + in.reg, oi = dsw.OutputRegisterTypeOffsets[ty] + s.out_dst[o];
		if ((int)ii >= dsi.InputRegisters.num || !dsi.InputRegisters.data) return fail(VVM_E_INVALID_ARG, "input register index past InputRegisters.num");
		if ((int)oi >= dsw.OutputRegisters.num || !dsw.OutputRegisters.data) return fail(VVM_E_INVALID_ARG, "output register index past OutputRegisters.num");
		if (!dsw.OutputRegisters.data[oi]) continue;      // (no destination: nothing is written, as for any output)
		if (!dsi.InputRegisters.data[ii]) return fail(VVM_E_INVALID_ARG, "an input stream the script reads is a null pointer");
		const uint64_t stride = ty == 2 ? 2u : 4u;
		a.src[a.num_pairs] = (uint64_t)(uintptr_t)dsi.InputRegisters.data[ii] + (uint64_t)(int64_t)dsi.InstanceOffset * stride;
		a.dst[a.num_pairs] = ((uint64_t)(uintptr_t)dsw.OutputRegisters.data[oi] + (uint64_t)(int64_t)dsw.InstanceOffset * stride) | (ty == 2 ? VVM_THROUGH_IS16 : 0ull);
		++a.num_pairs;
	}
	return VVM_OK;
}

int ensure_ride(vvm_state* st, uint32_t tiles, hipStream_t stream)
{
	if (tiles <= st->ride_capacity) return VVM_OK;
	HIP_TRY(hipStreamSynchronize(stream));      // (the stream of the state's last Exec: exec_enqueue has waited for any other one)
	if (st->d_ride) { HIP_TRY(hipFree(st->d_ride)); st->d_ride = nullptr; st->ride_capacity = 0; }
	const uint32_t cap = std::max<uint32_t>(1024, tiles + tiles / 4);
	HIP_TRY(hipMalloc((void**)&st->d_ride, (size_t)cap * VVM_BLOCK_THREADS * 4u));
	st->ride_capacity = cap;
	return VVM_OK;
}

int prepare(vvm_state* st, vvm_exec_context* ctx, const vvm_exec_options* opt, PreparedExec& pe, uint32_t pass = 0)
{
	const vvm::ThawedScript& s = st->script;
	if ((uint32_t)ctx->ExtFunctionTable.num != s.num_ext_fns)                                         // RT.cpp:2184-2187
		return fail(VVM_E_EXTFN_MISMATCH, "ExtFunctionTable.num differs from the script's external function count");
	if (s.bytecode.empty()) return fail(VVM_E_NO_BYTECODE, "state has no bytecode");                    // RT.cpp:2264-2267
	if (ctx->NumInstances < 0) return fail(VVM_E_INVALID_ARG, "negative NumInstances");
	if (ctx->DataSets.num < 0 || (ctx->DataSets.num > 0 && !ctx->DataSets.data)) return fail(VVM_E_INVALID_ARG, "DataSets view is null");
	if (s.num_ext_fns > 0 && !ctx->ExtFunctionTable.data) return fail(VVM_E_INVALID_ARG, "ExtFunctionTable view is null");
	if (ctx->ConstantTableCount < 0 || (ctx->ConstantTableCount > 0 && (!ctx->ConstantTableNumBytes || !ctx->ConstantTableData)))
		return fail(VVM_E_INVALID_ARG, "ConstantTableData / ConstantTableNumBytes is null");
	const bool narrow = opt && (opt->Flags & VVM_EXEC_NO_VGPR_WIDE);
	pe.light = false;
	if (narrow && st->wide.instances_per_lane != 1) {
		if (!st->narrow_ready) {
			std::string err;
			int rc = vvm::link_program(s, true, st->narrow, err);
			if (rc != VVM_OK) return fail(rc, err);
			rc = upload_code(st->narrow, &st->d_code_narrow);
			if (rc != VVM_OK) return rc;
			rc = link_more_passes(s, true, st->narrow, st->narrow_more, st->d_code_narrow_more);
			if (rc != VVM_OK) return rc;
			st->narrow_ready = true;
		}
		if (pass > st->narrow_more.size()) return fail(VVM_E_INVALID_ARG, "internal: pass index out of range");
		pe.lp = pass ? &st->narrow_more[pass - 1] : &st->narrow; pe.d_code = pass ? st->d_code_narrow_more[pass - 1] : st->d_code_narrow;
	} else if (opt && opt->RandXorwowState && !st->wide.rand_draws.empty() && st->wide.paired) {
		// the kernels that read the reference generator's words have one chain: one program per dataset
		if (!st->solo_ready) {
			std::string err;
			int rc = vvm::link_program(s, false, st->solo, err, 0, false, nullptr, 0);
			if (rc != VVM_OK) return fail(rc, err);
			rc = upload_code(st->solo, &st->d_code_solo);
			if (rc != VVM_OK) return rc;
			rc = link_more_passes(s, false, st->solo, st->solo_more, st->d_code_solo_more);
			if (rc != VVM_OK) return rc;
			st->solo_ready = true;
		}
		if (pass > st->solo_more.size()) return fail(VVM_E_INVALID_ARG, "internal: pass index out of range");
		pe.lp = pass ? &st->solo_more[pass - 1] : &st->solo; pe.d_code = pass ? st->d_code_solo_more[pass - 1] : st->d_code_solo;
	} else {
		if (pass > st->wide_more.size()) return fail(VVM_E_INVALID_ARG, "internal: pass index out of range");
		pe.lp = pass ? &st->wide_more[pass - 1] : &st->wide; pe.d_code = pass ? st->d_code_wide_more[pass - 1] : st->d_code_wide;
	}
	for (uint32_t i = 0; i < s.num_ext_fns; ++i) {
		const vvm_ext_function* f = ctx->ExtFunctionTable.data[i];
		const int want = (s.ext_num_outputs[i] == 1) ? VVM_EXTFN_PERLIN3 : VVM_EXTFN_CURL_NOISE3;
		if (!f || f->builtin_id != want)
			return fail(VVM_E_UNSUPPORTED_OP, "external function is not one of the GPU built-ins (perlin3: 3 in/1 out, curl_noise3: 3 in/3 out)");
	}
	if (!st->maps_ready) {                                                                             // RT.cpp:2198-2202
		int rc = resolve_maps(st, ctx);
		if (rc != VVM_OK) return rc;
	}
	pe.instances_per_lane = pe.lp->instances_per_lane;
	pe.reference_random = opt && opt->RandXorwowState && !st->wide.rand_draws.empty();
	pe.lean = false;
	{
		// the lean form: for a single Exec large enough that a second launch pays (VVM_THROUGH_MIN_INSTANCES)
		const uint32_t flags = opt ? opt->Flags : 0;
		// (two launches instead of one: against the four-wide kernels with 32 registers that pays at every size measured, against the
		// general eight-wide kernel, which a script of nine to sixteen registers runs on whole, from about 2 Mi)
		const bool whole_is_slow = st->wide.many_registers || st->wide.instances_per_lane == 1;
		// (a member of a batch shares its launches with the others: there the form pays from 1 Ki instances on, for scripts that would
		// otherwise run on the kernels with 32 registers and whose lean program fits the light kernel; 4096 emitters of 1 / 4 / 16 Ki
		// instances, 24 streams each way: 0.25 / 0.52 / 1.99 ms against 0.28 / 0.83 / 3.11 ms whole)
		const bool one_block = st->through.size() <= VVM_THROUGH_MAX_PAIRS;      // (a member of a batch has one block for the copy kernel)
		const bool big = (pe.in_batch ? (whole_is_slow && st->lean.light && one_block && ctx->NumInstances >= VVM_BATCH_THROUGH_MIN_INSTANCES)
		                              : ctx->NumInstances >= (whole_is_slow ? VVM_THROUGH_MIN_INSTANCES : VVM_THROUGH_MIN_INSTANCES_EIGHT_WIDE)) ||
		                 ((flags & VVM_EXEC_FORCE_EIGHT_WIDE) && (!pe.in_batch || (st->lean.light && one_block)));
		const uint64_t lean_tile = (uint64_t)VVM_BLOCK_THREADS * (st->lean.full_kernel ? 4u : 8u);
		const bool room = pe.lean_max_tiles == 0 || ((uint64_t)std::max(ctx->NumInstances, 0) + lean_tile - 1) / lean_tile <= pe.lean_max_tiles;
		if (st->lean_ready && pass == 0 && !narrow && ctx->NumInstances > 0 && room && !pe.reference_random && big &&
		    !(flags & VVM_EXEC_NO_THROUGH_KERNEL) && !(!st->lean.full_kernel && (flags & VVM_EXEC_NO_EIGHT_WIDE)) && !(st->lean.light && (flags & VVM_EXEC_NO_LIGHT_KERNEL))) {
			pe.lean = true;
			pe.lp = &st->lean;
			pe.d_code = st->d_code_lean;
			pe.instances_per_lane = st->lean.full_kernel ? 4 : 8;
			pe.light = st->lean.light;
		}
	}
	if (pe.reference_random) {
		if (pe.in_batch) return fail(VVM_E_UNSUPPORTED_OP, "RandXorwowState: not available to the members of a batch");
		if (pe.lp->num_chains > 1) return fail(VVM_E_UNSUPPORTED_OP, "internal: RandXorwowState with a program of two chains");      // (prepare picks the one-chain programs above)
		// (the random ops of the FULL kernels read the words; a program linked for the plain kernels runs on them as it is, given the LDS)
		if ((size_t)vvm_kernel_lds_fixed_bytes((int)pe.instances_per_lane, 1) + (size_t)pe.lds_slots() * VVM_BLOCK_THREADS * 4u * pe.instances_per_lane > (size_t)(160 * 1024 - 64))
			return fail(VVM_E_UNSUPPORTED_OP, "RandXorwowState: the script's registers leave no room in the LDS of the kernels that read the generator's words");
	}
	{
		// eight instances per lane: the same program, twice the work per interpreted instruction; only worth it when there are
		// enough tiles left to keep every CU busy for several rounds
		const uint32_t flags = opt ? opt->Flags : 0;
		const bool can = !pe.lean && pe.lp->instances_per_lane == 4 && !pe.full() && !pe.lp->many_registers && !(flags & VVM_EXEC_NO_EIGHT_WIDE);
		// (the size of the Exec itself decides, also inside a batch: many small Execs measured faster four wide, 0.88 ms against
		// 0.92 ms for 4096 Execs of 16 Ki instances)
		// scripts that get by with eight registers have a program for the light eight-wide kernel (three workgroups per CU)
		const bool has_light = pass == 0 ? st->light_ready : (pass - 1 < st->d_code_light_more.size() && st->d_code_light_more[pass - 1] != nullptr);
		const bool light_ok = can && has_light && pe.lp == (pass ? &st->wide_more[pass - 1] : &st->wide) && !narrow && !(flags & VVM_EXEC_NO_LIGHT_KERNEL);
		// (a member of a batch shares its launch with the other members, so the light kernel pays off from a few tiles on: 4096
		// emitters of 16 Ki instances, all fit for it, 0.735 ms against 0.904 ms four-wide; a single Exec of that size takes the
		// same 13.5 us on either)
		const bool big = ctx->NumInstances >= VVM_EIGHT_WIDE_MIN_INSTANCES || (pe.in_batch && light_ok && ctx->NumInstances >= VVM_BATCH_LIGHT_MIN_INSTANCES);
		if (can && (big || (flags & VVM_EXEC_FORCE_EIGHT_WIDE))) {
			pe.instances_per_lane = 8;
			if (light_ok) { pe.lp = pass ? &st->light_more[pass - 1] : &st->light; pe.d_code = pass ? st->d_code_light_more[pass - 1] : st->d_code_light; pe.light = true; }
		}
	}
	const uint32_t tile = VVM_BLOCK_THREADS * pe.instances_per_lane;
	pe.tiles = ((uint32_t)ctx->NumInstances + tile - 1) / tile;
	const uint32_t O = std::max<uint32_t>(s.num_outputs_remapped, s.num_output_instructions);
	pe.off_consts = align_up(sizeof(VvmExecDesc), 16);
	pe.off_in = align_up(pe.off_consts + 4u * ((size_t)s.num_consts + pe.lp->derived.size()), 16);     // (so the table is padded to a multiple of 4 dwords)
	pe.off_load = align_up(pe.off_in + 8u * (size_t)s.num_inputs, 64);
	pe.off_out = align_up(pe.off_load + 8u * pe.lp->load_list.size(), 64) + 64;      // (the entry before the list is in use too)
	pe.off_ids = align_up(pe.off_out + 8u * align_up(pe.lp->out_list.size(), VVM_STAGE_SLOTS), 16);
	pe.bytes = align_up(pe.off_ids + pe.lp->id_uses.size() * sizeof(VvmIdDesc), 256);      // (the block itself is 256-byte aligned)
	(void)O;
	reference_chunking(s, ctx, pe.chunk_instances, pe.num_chunks);
	if (pe.serial_chunk) {
		// this Exec IS one chunk of the whole Exec's chunking (worked out anew for its own size, the chunking may cut it again: the
		// reference sizes a chunk by the average, not by what fits)
		pe.chunk_instances = std::max<uint32_t>(4u, ((uint32_t)ctx->NumInstances + 3u) & ~3u);
		pe.num_chunks = ctx->NumInstances > 0 ? 1u : 0u;
	}
	for (const vvm::LinkedProgram::IdUse& use : pe.lp->id_uses) {
		// update_id / acquire_id: the tables must be there, in device memory, and small enough for 32-bit byte offsets
		if ((int)use.dataset >= ctx->DataSets.num) return fail(VVM_E_INVALID_ARG, "the script's ID instructions name a dataset the context does not provide");
		const vvm_dataset_meta& ds = ctx->DataSets.data[use.dataset];
		if (!ds.FreeIDTable || !ds.FreeIDTable->data || !ds.NumFreeIDs) return fail(VVM_E_INVALID_ARG, "the script has ID instructions but the dataset has no FreeIDTable / NumFreeIDs");
		if (ds.FreeIDTable->max < 0 || ds.FreeIDTable->max > (1 << 29)) return fail(VVM_E_INVALID_ARG, "FreeIDTable->Max() out of range");
		if (use.updates) {
			if (!ds.IDTable || !ds.IDTable->data || !ds.MaxUsedID) return fail(VVM_E_INVALID_ARG, "update_id needs the dataset's IDTable and MaxUsedID");
			if (ds.IDTable->num < 0 || ds.IDTable->num > (1 << 29)) return fail(VVM_E_INVALID_ARG, "IDTable->Num() out of range");
			// (a script that also takes its IDs from this dataset returns them to the places they came from)
			if (!use.acquires && (int64_t)*ds.NumFreeIDs + ctx->NumInstances > ds.FreeIDTable->max) return fail(VVM_E_INVALID_ARG, "FreeIDTable cannot take the IDs of every instance");
		}
		if (use.acquires) {
			if (!ds.NumSpawnedIDs) return fail(VVM_E_INVALID_ARG, "acquire_id needs the dataset's NumSpawnedIDs");
			if (*ds.NumFreeIDs < ctx->NumInstances) return fail(VVM_E_INVALID_ARG, "fewer free IDs than instances");              // check(), RT.cpp:1415
			if ((int64_t)*ds.NumSpawnedIDs + ctx->NumInstances > ds.FreeIDTable->max) return fail(VVM_E_INVALID_ARG, "spawned IDs do not fit FreeIDTable");   // RT.cpp:1431
			// the reference reads the free IDs chunk by chunk before it logs them at the end of the same table; with the two
			// regions disjoint the order does not matter
			if ((int64_t)*ds.NumFreeIDs > (int64_t)ds.FreeIDTable->max - *ds.NumSpawnedIDs - ctx->NumInstances)
				return fail(VVM_E_INVALID_ARG, "the free IDs and the spawned-ID log overlap in FreeIDTable");
		}
		// two datasets of one Exec must not share ID tables (each is worked on as if alone)
		for (const vvm::LinkedProgram::IdUse& other : pe.lp->id_uses)
			if (other.dataset < use.dataset && (int)other.dataset < ctx->DataSets.num && ctx->DataSets.data[other.dataset].FreeIDTable == ds.FreeIDTable)
				return fail(VVM_E_INVALID_ARG, "two datasets with ID instructions share one FreeIDTable");
	}
	return VVM_OK;
}

// The script's remapped constants, one dword each, from the call's constant tables (VectorVMRuntime.cpp:2204-2207; the kernel
// broadcasts them).
// Behind them the program's derived constants (LinkedProgram::derived): sums, differences and products of constants, computed here
// in the arithmetic the kernel would use per instance (IEEE single precision, unfused; 32-bit wrap-around for the integer ops).
int fill_constants(vvm_state* st, const vvm_exec_context* ctx, uint32_t* consts, const vvm::LinkedProgram& lp)
{
	const vvm::ThawedScript& s = st->script;
	for (uint32_t i = 0; i < s.num_consts; ++i) {
		if ((int)st->const_table[i] >= ctx->ConstantTableCount || !ctx->ConstantTableData || !ctx->ConstantTableData[st->const_table[i]] ||
		    4u * ((size_t)st->const_dword[i] + 1) > (size_t)ctx->ConstantTableNumBytes[st->const_table[i]])
			return fail(VVM_E_INVALID_ARG, "a constant the script reads lies outside the constant tables of this call");
		const uint8_t* table = ctx->ConstantTableData[st->const_table[i]];
		memcpy(&consts[i], table + 4u * (size_t)st->const_dword[i], 4);
	}
	for (size_t j = 0; j < lp.derived.size(); ++j) {
		const vvm::LinkedProgram::Derived& d = lp.derived[j];
		const uint32_t ua = consts[d.a], ub = consts[d.b];
		volatile float fa, fb;      // (volatile: the operation is done at run time, in single precision, as written)
		memcpy((void*)&fa, &ua, 4); memcpy((void*)&fb, &ub, 4);
		uint32_t r = 0;
		float fr = 0.0f;
		switch (d.op) {
		case VVM_OP_add: fr = fa + fb; memcpy(&r, &fr, 4); break;
		case VVM_OP_sub: fr = fa - fb; memcpy(&r, &fr, 4); break;
		case VVM_OP_mul: fr = fa * fb; memcpy(&r, &fr, 4); break;
		case VVM_OP_addi: r = ua + ub; break;
		case VVM_OP_subi: r = ua - ub; break;
		default: r = ua * ub; break;      // muli
		}
		consts[s.num_consts + j] = r;
	}
	return VVM_OK;
}

// Fills one parameter block at `host` that will live at device address `dev`; dev = null: the block goes to the kernel as a
// kernel argument, and the fields that point into the block hold offsets from its start (VVM_DESC_RELATIVE).
int fill_params(vvm_state* st, vvm_exec_context* ctx, const vvm_exec_options* opt, const PreparedExec& pe, uint8_t* host,
                uint8_t* dev_block, unsigned long long* d_status, uint32_t first_tile, int32_t* d_counts = nullptr)
{
	const uintptr_t dev = (uintptr_t)dev_block;
	const vvm::ThawedScript& s = st->script;
	const vvm::LinkedProgram& lp = *pe.lp;
	VvmExecDesc* d = (VvmExecDesc*)host;
	memset(d, 0, sizeof(*d));
	d->prog = pe.d_code;
	d->consts = (const uint32_t*)(dev + pe.off_consts);
	d->in_ptrs = (const uint64_t*)(dev + pe.off_in);
	d->load_ptrs = (const uint64_t*)(dev + pe.off_load);
	d->out_list = (const uint64_t*)(dev + pe.off_out);
	if (!dev_block && lp.uses_ids) return fail(VVM_E_INVALID_ARG, "internal: a script with ID instructions needs its parameter block in device memory");
	d->status = d_status;
	d->counts = d_counts ? d_counts : (opt && opt->DeviceCounts && !st->any_ids) ? opt->DeviceCounts : st->d_counts;
	if (lp.uses_perlin && !st->d_perlin) return fail(VVM_E_INVALID_ARG, "internal: the noise table of this state was not uploaded");
	d->perlin = st->d_perlin;
	d->num_instances = (uint32_t)ctx->NumInstances;
	d->num_tiles = pe.tiles;
	d->first_tile = first_tile;
	d->exec_index_base = (opt ? opt->ExecIndexBase : 0) + (int32_t)pe.chunk_first;
	d->rand_seed_lo = opt ? (uint32_t)opt->RandSeed : 0;
	d->rand_seed_hi = opt ? (uint32_t)(opt->RandSeed >> 32) : 0;
	d->rand_instance_base = (opt ? opt->RandInstanceBase : 0) + (int32_t)pe.chunk_first;
	d->rand_words = pe.reference_random ? st->d_rand_words : nullptr;
	d->rand_stride = pe.reference_random ? st->rand_stride : 0;
	d->flags = (lp.uses_perlin ? VVM_DESC_USES_PERLIN : 0) | (dev_block ? 0 : VVM_DESC_RELATIVE) | (lp.many_registers ? VVM_DESC_REGS_OF_4 : 0);
	d->num_unif = std::min<uint32_t>((s.num_consts + (uint32_t)lp.derived.size() + 3u) & ~3u, VVM_MAX_UNIFORMS);

	uint32_t* consts = (uint32_t*)(host + pe.off_consts);
	memset(consts, 0, pe.off_in - pe.off_consts);
	{
		const int crc = fill_constants(st, ctx, consts, lp);
		if (crc != VVM_OK) return crc;
	}
	uint64_t* in_ptrs = (uint64_t*)(host + pe.off_in);                                                  // RT.cpp:366-416
	for (uint32_t i = 0; i < s.num_inputs; ++i) {
		const vvm::InputInfo& in = lp.inputs[i];
		if ((int)in.dataset >= ctx->DataSets.num) return fail(VVM_E_INVALID_ARG, "script reads a dataset the context does not provide");
		const vvm_dataset_meta& ds = ctx->DataSets.data[in.dataset];
		const uint32_t idx = ds.InputRegisterTypeOffsets[in.type] + in.reg;
		if ((int)idx >= ds.InputRegisters.num || !ds.InputRegisters.data) return fail(VVM_E_INVALID_ARG, "input register index past InputRegisters.num");
		if (i < lp.input_used.size() && lp.input_used[i] && !ds.InputRegisters.data[idx])
			return fail(VVM_E_INVALID_ARG, "an input stream the script reads is a null pointer");
		const uint64_t stride = (in.type == 2) ? 2 : 4;
		in_ptrs[i] = (uint64_t)(uintptr_t)ds.InputRegisters.data[idx] + (uint64_t)((int64_t)ds.InstanceOffset + (in.noadvance ? 0 : (int64_t)pe.chunk_first)) * stride;
	}
	uint64_t* load_ptrs = (uint64_t*)(host + pe.off_load);
	for (size_t i = 0; i < lp.load_list.size(); ++i) load_ptrs[i] = lp.load_list[i] >= 0 ? in_ptrs[lp.load_list[i]] : 0;
	uint64_t* out_list = (uint64_t*)(host + pe.off_out);                                                // RT.cpp:418-441
	memset(out_list, 0, pe.bytes - pe.off_out);
	{   // the entry before the list: where the Exec's last tile writes the survivor count of the acquired dataset
		out_list[-1] = (uint64_t)(uintptr_t)(d->counts + lp.pass_dataset);
		out_list[-3] = pe.ride_addr;      // (the light kernel reads it on every tile: 0 unless vvm_through_kernel follows)
		if (lp.num_chains > 1) out_list[-2] = (uint64_t)(uintptr_t)(d->counts + lp.chain_dataset[1]);      // (the second chain's)
	}
	for (size_t k = 0; k < lp.out_list.size(); ++k) {
		const int32_t i = lp.out_list[k];
		if (i < 0 || (uint32_t)i >= s.num_outputs_remapped || s.out_dst[i] == 0xFFFF) continue;
		const uint8_t dsi = s.out_dataset[i];
		const uint16_t ty = s.out_type[i];
		if ((int)dsi >= ctx->DataSets.num || ty > 2) return fail(VVM_E_INVALID_ARG, "script writes a dataset the context does not provide");
		const vvm_dataset_meta& ds = ctx->DataSets.data[dsi];
		const uint32_t idx = ds.OutputRegisterTypeOffsets[ty] + s.out_dst[i];
		if ((int)idx >= ds.OutputRegisters.num || !ds.OutputRegisters.data) return fail(VVM_E_INVALID_ARG, "output register index past OutputRegisters.num");
		const uint64_t stride = (ty == 2) ? 2 : 4;
		if (!ds.OutputRegisters.data[idx]) continue;
		const int64_t written_so_far = (pe.out_bases && dsi < pe.num_out_bases) ? (int64_t)pe.out_bases[dsi] : (int64_t)pe.out_base;
		const uint64_t addr = (uint64_t)(uintptr_t)ds.OutputRegisters.data[idx] + (uint64_t)((int64_t)ds.InstanceOffset + written_so_far) * stride;
		if (addr & ~VVM_OUT_ADDR_MASK) return fail(VVM_E_INVALID_ARG, "output register address does not fit 48 bits");
		out_list[k] = addr | (lp.out_is16[k] ? VVM_OUT_IS16 : 0) | ((k < lp.out_chain.size() && lp.out_chain[k]) ? VVM_OUT_CHAIN1 : 0);
	}
	for (size_t k = 0; k < lp.out_list.size(); ++k) {
		if (lp.out_is16[k]) out_list[k & ~(size_t)(VVM_STAGE_SLOTS - 1)] |= VVM_OUT_GROUP16;
	}
	if (lp.uses_ids) d->ids = (const VvmIdDesc*)(dev + pe.off_ids);      // (dev is an address here: checked above)
	for (size_t u = 0; u < lp.id_uses.size(); ++u) {
		const vvm::LinkedProgram::IdUse& use = lp.id_uses[u];
		const vvm_dataset_meta& ds = ctx->DataSets.data[use.dataset];
		VvmIdDesc* ids = (VvmIdDesc*)(host + pe.off_ids) + u;
		memset(ids, 0, sizeof(*ids));
		ids->id_table = ds.IDTable ? ds.IDTable->data : nullptr;
		ids->id_table_num = ds.IDTable ? (uint32_t)ds.IDTable->num : 0;
		ids->free_table = ds.FreeIDTable->data;
		ids->free_table_max = (uint32_t)ds.FreeIDTable->max;
		ids->chunk_prefix = st->d_chunk_prefix ? st->d_chunk_prefix + (size_t)pe.id_buffers * st->chunk_capacity : nullptr;
		ids->id_scratch = st->d_id_scratch ? st->d_id_scratch + (size_t)pe.id_buffers * st->id_scratch_capacity : nullptr;
		ids->max_id = d->counts + st->num_datasets() + use.dataset;      // the dataset's word behind the survivor counts
		ids->survivors = d->counts + use.dataset;
		ids->instance_offset = ds.InstanceOffset + ((pe.out_bases && use.dataset < pe.num_out_bases) ? pe.out_bases[use.dataset] : (int32_t)pe.out_base);
		ids->num_free_start = *ds.NumFreeIDs;
		const bool both = use.updates && use.acquires;      // (then this Exec is one chunk: prepare and exec_serial_chunks see to it)
		ids->push_start = both ? *ds.NumFreeIDs - ctx->NumInstances : *ds.NumFreeIDs;
		// (a pass of another dataset runs the dataset's acquire_id too, and reads the same copy: its update_id's pass may have run by then)
		ids->free_read = (use.serial && pe.free_read_snapshot) ? pe.free_read_snapshot - ((int64_t)*ds.NumFreeIDs - ctx->NumInstances) : ds.FreeIDTable->data;
		ids->num_spawned_start = ds.NumSpawnedIDs ? *ds.NumSpawnedIDs : 0;
		ids->acquire_tag = ds.IDAcquireTag;
		ids->chunk_instances = pe.chunk_instances;
		ids->num_instances = (uint32_t)ctx->NumInstances;
		ids->num_chunks = pe.num_chunks;
	}
	for (size_t k = 0; k < lp.out_list.size(); ++k)
		if (lp.out_list[k] <= -2) {      // an update_id: the entry is the address of its dataset's descriptor
			out_list[k] = (uint64_t)(uintptr_t)(d->ids + (size_t)(-2 - lp.out_list[k])) | VVM_OUT_UPDATE_ID;
			out_list[k & ~(size_t)(VVM_STAGE_SLOTS - 1)] |= VVM_OUT_GROUP16;
		}
	return VVM_OK;
}

// fresh launch words: everything zero, epoch 1
hipError_t reset_launch_state(VvmLaunchState* d)
{
	hipError_t e = memset_and_wait(d, 0, sizeof(VvmLaunchState));
	if (e != hipSuccess) return e;
	const uint32_t one = 1;
	return hipMemcpy(&d->epoch, &one, 4, hipMemcpyHostToDevice);
}

// Persistent workgroups: as many as can be resident (workgroups per CU by the kernel's register budget), never more than tiles.
uint32_t grid_for(uint32_t tiles, uint32_t instances_per_lane, bool full, bool many_registers)
{
	// (eight instances per lane with many_registers set: the light kernel, three workgroups per CU)
	uint32_t per_cu = (instances_per_lane == 8 && many_registers) ? 3u : (instances_per_lane == 8 || (instances_per_lane == 4 && many_registers)) ? 2u : full ? 3u : instances_per_lane == 4 ? 4u : 8u;
	static const uint32_t forced = getenv("VVM_DEBUG_WG_PER_CU") ? (uint32_t)atoi(getenv("VVM_DEBUG_WG_PER_CU")) : 0u;      // (measurement aid)
	if (forced) per_cu = std::min(per_cu, forced);
	return std::max<uint32_t>(1, std::min<uint32_t>(tiles, (uint32_t)g_dev.num_cus * per_cu * (256u / VVM_BLOCK_THREADS)));
}

// update_id's buffers (survivors before each of the reference's chunks, every instance's ID): `pieces` of them, each for
// `num_chunks` chunks and `instances` instances.
int ensure_id_buffers(vvm_state* st, uint32_t pieces, uint32_t num_chunks, uint32_t instances, hipStream_t stream)
{
	if (pieces <= st->id_buffer_pieces && num_chunks <= st->chunk_capacity && instances <= st->id_scratch_capacity) return VVM_OK;
	HIP_TRY(hipStreamSynchronize(stream));
	if (st->d_chunk_prefix) HIP_TRY(hipFree(st->d_chunk_prefix));
	if (st->d_id_scratch) HIP_TRY(hipFree(st->d_id_scratch));
	st->d_chunk_prefix = nullptr; st->d_id_scratch = nullptr;
	st->id_buffer_pieces = std::max(pieces, st->id_buffer_pieces);
	st->chunk_capacity = std::max(st->chunk_capacity, num_chunks + num_chunks / 4 + 64);
	st->id_scratch_capacity = std::max(st->id_scratch_capacity, instances + instances / 4 + 1024);
	HIP_TRY(hipMalloc((void**)&st->d_chunk_prefix, (size_t)st->id_buffer_pieces * st->chunk_capacity * 4));
	HIP_TRY(hipMalloc((void**)&st->d_id_scratch, (size_t)st->id_buffer_pieces * st->id_scratch_capacity * 4));
	return VVM_OK;
}

// The ID counters of the datasets a pass works on, as they are when the Exec is enqueued; MaxUsedID goes to its device word
// (the kernel only ever raises it, VectorVMRuntime.cpp:1383-1399).
int stage_id_counters(vvm_state* st, const vvm_exec_context* ctx, const vvm::LinkedProgram& lp, hipStream_t stream)
{
	for (const vvm::LinkedProgram::IdUse& use : lp.id_uses) {
		const vvm_dataset_meta& ds = ctx->DataSets.data[use.dataset];
		st->id_num_free_start[use.dataset] = *ds.NumFreeIDs;
		st->id_num_spawned_start[use.dataset] = ds.NumSpawnedIDs ? *ds.NumSpawnedIDs : 0;
		if (use.updates) {
			const size_t nd = st->num_datasets();
			st->h_counts[2 * nd + use.dataset] = *ds.MaxUsedID;
			HIP_TRY(hipMemcpyAsync(st->d_counts + nd + use.dataset, st->h_counts + 2 * nd + use.dataset, 4, hipMemcpyHostToDevice, stream));
		}
	}
	return VVM_OK;
}

// One step of one of the reference generator's four lanes without the counter (VectorVMRuntime.cpp:470-479): the linear part,
// from which the jump table is built.
void xorwow_linear_step(uint32_t (&x)[5])
{
	uint32_t t = x[4];
	const uint32_t s0 = x[0];
	x[4] = x[3]; x[3] = x[2]; x[2] = x[1]; x[1] = s0;
	t ^= t >> 2;
	t ^= t << 1;
	t ^= s0 ^ (s0 ^ (s0 << 4));      // as written at :478
	x[0] = t;
}

// The jump table of the current device: column j of M^(2^b) for every state bit j and power b (see vvm_device.h).
int ensure_xorwow_jump(int device, const uint32_t** out)
{
	if (!g_dev.xorwow_jump[device]) {
		std::vector<uint32_t> table(VVM_XORWOW_JUMP_WORDS);
		auto column = [&](uint32_t b, uint32_t j) { return &table[((size_t)b * VVM_XORWOW_STATE_BITS + j) * 5u]; };
		for (uint32_t j = 0; j < VVM_XORWOW_STATE_BITS; ++j) {
			uint32_t x[5] = {0, 0, 0, 0, 0};
			x[j / 32] = 1u << (j % 32);
			xorwow_linear_step(x);
			memcpy(column(0, j), x, sizeof x);
		}
		for (uint32_t b = 1; b < VVM_XORWOW_JUMP_POWERS; ++b)      // M^(2^b) = M^(2^(b-1)) squared: its column j = the lower power applied to its own column j
			for (uint32_t j = 0; j < VVM_XORWOW_STATE_BITS; ++j) {
				const uint32_t* v = column(b - 1, j);
				uint32_t y[5] = {0, 0, 0, 0, 0};
				for (uint32_t i = 0; i < VVM_XORWOW_STATE_BITS; ++i)
					if ((v[i / 32] >> (i % 32)) & 1u) { const uint32_t* ci = column(b - 1, i); for (int w = 0; w < 5; ++w) y[w] ^= ci[w]; }
				memcpy(column(b, j), y, sizeof y);
			}
		uint32_t* d = nullptr;
		HIP_TRY(hipMalloc((void**)&d, table.size() * 4));
		const hipError_t e = hipMemcpy(d, table.data(), table.size() * 4, hipMemcpyHostToDevice);
		if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "hipMemcpy(xorwow jump table)"); }
		g_dev.xorwow_jump[device] = d;
	}
	*out = g_dev.xorwow_jump[device];
	return VVM_OK;
}

// vvm_exec_options::RandXorwowState: lays out, ahead of the Exec on its stream, the 32 bits of every draw and instance as the
// reference's generator yields them in its execution order (VectorVMRuntime.cpp:2276-2295 over the chunks of this Exec).
int enqueue_reference_random(vvm_state* st, const vvm_exec_context* ctx, const vvm_exec_options* opt, const PreparedExec& pe, hipStream_t stream)
{
	const std::vector<vvm::LinkedProgram::RandDraw>& draws = st->wide.rand_draws;      // (every program of the script numbers the draws alike)
	const uint32_t total_loops = ((uint32_t)ctx->NumInstances + 3u) >> 2;
	if (total_loops == 0 || draws.empty()) return VVM_OK;
	// whole tiles of the widest kernel: the lanes past the last instance read too (their values go nowhere)
	const uint64_t stride = ((uint64_t)total_loops * 4u + 2047u) & ~(uint64_t)2047u;
	if (stride > 0xFFFFF800ull) return fail(VVM_E_TOO_LARGE, "RandXorwowState: too many instances");
	const size_t need = (size_t)stride * draws.size();
	if (need > st->rand_words_capacity) {
		HIP_TRY(hipStreamSynchronize(stream));
		if (st->d_rand_words) { HIP_TRY(hipFree(st->d_rand_words)); st->d_rand_words = nullptr; st->rand_words_capacity = 0; }
		HIP_TRY(hipMalloc((void**)&st->d_rand_words, need * 4));
		st->rand_words_capacity = need;
	}
	st->rand_stride = (uint32_t)stride;
	const uint32_t* jump = nullptr;
	const int rc = ensure_xorwow_jump(st->device, &jump);
	if (rc != VVM_OK) return rc;
	VvmXorwowFill p;
	memset(&p, 0, sizeof p);
	memcpy(p.state, opt->RandXorwowState, sizeof p.state);
	p.total_loops = total_loops;
	p.chunk_loops = pe.chunk_instances / 4u;
	p.num_draws = (uint32_t)draws.size();
	p.stride = (uint32_t)stride;
	for (size_t o = 0; o < draws.size(); ++o)
		if (draws[o].first == o) { p.first[p.num_instructions] = draws[o].first; p.width[p.num_instructions] = draws[o].width; ++p.num_instructions; }
	const uint64_t steps = (uint64_t)total_loops * draws.size();
	while (p.jump_powers < VVM_XORWOW_JUMP_POWERS && (steps >> p.jump_powers) != 0) ++p.jump_powers;
	const int lrc = vvm_launch_xorwow_fill(&p, jump, st->d_rand_words, stream);
	if (lrc != 0) return hip_fail((hipError_t)lrc, "kernel launch (random generator)");
	return VVM_OK;
}

// A script that acquires several datasets has one program per dataset (a pass).  When all passes run four instances per lane,
// or all one, they go out as ONE launch of a FULL kernel that runs every pass on a tile before it moves to the next tile: one
// launch instead of several, and the passes' look-back chains advance together.  Passes that would run eight instances per
// lane are launched one after the other, on the faster eight-wide kernels (measured on 64 Mi instances, two datasets:
// 1.48 ms against 1.86 ms fused; the later passes do not find the shared streams in the L2 either way, the tiles in flight
// are several times its size).  `handled` stays false when the passes have to be launched one after the other.
int enqueue_fused_passes(vvm_state* st, vvm_exec_context* ctx, const vvm_exec_options* opt, uint32_t K, hipStream_t stream, VvmLaunchState* d_launch, bool timed, bool& handled)
{
	handled = false;
	vvm_exec_options o = opt ? *opt : vvm_exec_options{};
	std::vector<PreparedExec> pes(K);
	bool many = false, ids = false;
	uint32_t spill = 0;
	for (uint32_t k = 0; k < K; ++k) {
		const int rc = prepare(st, ctx, &o, pes[k], k);
		if (rc != VVM_OK) return rc;
		if (pes[k].instances_per_lane != pes[0].instances_per_lane || pes[k].instances_per_lane == 8) return VVM_OK;      // (the FULL kernels run four or one)
		if (pes[k].lp->num_chains > 1) return VVM_OK;      // (... and one chain: passes of two datasets each are launched one after the other)
		many = many || pes[k].launch_many();
		spill = std::max(spill, pes[k].lp->lds_slots_with_32_resident());      // (every pass runs on the FULL kernel)
		ids = ids || pes[k].lp->updates_ids;
		pes[k].id_buffers = k;
	}
	const uint32_t lanes = pes[0].instances_per_lane, tiles = pes[0].tiles;
	if ((size_t)vvm_kernel_lds_fixed_bytes((int)lanes, 1) + (size_t)spill * VVM_BLOCK_THREADS * 4u * lanes > (size_t)(160 * 1024 - 64)) return VVM_OK;      // (the CU's LDS: the passes then go one after the other)
	if ((uint64_t)K * tiles > 0xFFFFFFFFull) return VVM_OK;
	int rc = ensure_status(st, K * tiles);      // every pass has look-back words of its own
	if (rc != VVM_OK) return rc;
	st->cur_status = st->d_status;
	st->cur_status_words = (size_t)st->status_capacity;
	if (ids) {
		rc = ensure_id_buffers(st, K, pes[0].num_chunks, (uint32_t)ctx->NumInstances, stream);
		if (rc != VVM_OK) return rc;
	}
	// the slot: the K descriptors side by side (the kernel indexes them), then every pass's block as fill_params lays it out
	std::vector<size_t> block_off(K);
	size_t total = align_up(K * sizeof(VvmExecDesc), 256);
	for (uint32_t k = 0; k < K; ++k) { block_off[k] = total; total += pes[k].bytes; }
	rc = st->ring.reserve(total);
	if (rc != VVM_OK) return rc;
	int slot = -1;
	rc = st->ring.acquire(slot);
	if (rc != VVM_OK) return rc;
	uint8_t* const host = st->ring.host[slot];
	uint8_t* const dev = st->ring.dev[slot];
	for (uint32_t k = 0; k < K; ++k) {
		rc = stage_id_counters(st, ctx, *pes[k].lp, stream);
		if (rc != VVM_OK) return rc;
		rc = fill_params(st, ctx, &o, pes[k], host + block_off[k], dev + block_off[k], st->d_status + (size_t)k * tiles, 0);
		if (rc != VVM_OK) return rc;
		memcpy(host + k * sizeof(VvmExecDesc), host + block_off[k], sizeof(VvmExecDesc));      // (its table fields are device addresses)
	}
	HIP_TRY(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, stream));
	st->last_width = lanes;
	if (timed) HIP_TRY(hipEventRecord(st->ev_start[st->ev_used], stream));
	HIP_TRY(st->epoch.before_launch(st->d_status, st->status_capacity, stream));      // (fused passes always run on the state's own launch words)
	const int lrc = vvm_launch_exec_kernel((int)lanes, (const VvmExecDesc*)dev, K, tiles, d_launch, many ? 1 : 0, spill, 1, grid_for(tiles, lanes, true, many), stream, nullptr, 0, 0u, 1);
	if (lrc != 0) return hip_fail((hipError_t)lrc, "kernel launch (fused passes)");
	st->epoch.launched();
	if (timed) { HIP_TRY(hipEventRecord(st->ev_stop[st->ev_used], stream)); st->ev_used++; }
	for (uint32_t k = 0; k < K; ++k)
		for (size_t u = 0; u < pes[k].lp->id_uses.size(); ++u) {
			if (!pes[k].lp->id_uses[u].updates) continue;
			const int mrc = vvm_launch_max_id_kernel((const VvmIdDesc*)(dev + block_off[k] + pes[k].off_ids) + u, pes[k].num_chunks, stream);
			if (mrc != 0) return hip_fail((hipError_t)mrc, "MaxUsedID kernel launch");
		}
	HIP_TRY(hipEventRecord(st->ring.done[slot], stream));
	st->ring.pending[slot] = true;
	handled = true;
	return VVM_OK;
}

int write_back_counts(vvm_state* st, vvm_exec_context* ctx, bool ran_kernel)
{
	const vvm::ThawedScript& s = st->script;
	for (uint32_t i = 0; i < s.max_output_dataset && (int)i < ctx->DataSets.num; ++i) {             // RT.cpp:2305-2308
		int32_t n = 0;
		if (ran_kernel && i < st->wide.acquires_dataset.size() && st->wide.acquires_dataset[i]) n = st->h_counts[i];
		ctx->DataSets.data[i].DataSetAccessIndex = n - 1;
	}
	return VVM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" {

VVM_API const char* vvm_last_error(void) { return g_last_error.c_str(); }

VVM_API void vvm_xorwow_seed(uint64_t cycles_first, uint64_t cycles_second, uint64_t batch_state_address, uint32_t* state24)
{
	if (!state24) return;
	// a PCG-like generator seeds the xorwow state, VectorVMRuntime.cpp:447-463
	uint64_t seed_state = cycles_first ^ (cycles_second << 32);
	const uint64_t increment = ((batch_state_address << 32) ^ 0xCAFEF00DD15EA5E5ull) | 1u;
	for (int k = 0; k < 20; ++k) {
		const uint64_t before = seed_state;
		seed_state = before * 6364136223846793005ull + increment;
		const uint32_t mixed = (uint32_t)(((before >> 18) ^ before) >> 27);
		const uint32_t turn = (uint32_t)(before >> 59);
		state24[k] = (mixed >> turn) | (mixed << ((0u - turn) & 31u));
	}
	// Counters = the two 64-bit numbers as four 32-bit lanes (:464)
	state24[20] = (uint32_t)increment;  state24[21] = (uint32_t)(increment >> 32);
	state24[22] = (uint32_t)seed_state; state24[23] = (uint32_t)(seed_state >> 32);
}

VVM_API void vvm_set_page_size_kb(int kb) { g_page_kb = (kb > 0 && kb <= 1 << 20) ? kb : 64; }

VVM_API vvm_state* vvm_alloc_state(const uint8_t* blob, size_t num_bytes)
{
	g_state_generation.fetch_add(1, std::memory_order_relaxed);
	vvm_state* st = new vvm_state();
	std::string err;
	int rc = vvm::thaw_blob(blob, num_bytes, st->script, err);
	if (rc == VVM_OK) rc = vvm::link_program(st->script, false, st->wide, err);
	if (rc != VVM_OK) { fail(rc, err); delete st; return nullptr; }
	auto bail = [&](hipError_t e, const char* where) -> vvm_state* { hip_fail(e, where); vvm_free_state(st); return nullptr; };
	if (ensure_device_info() != VVM_OK) { vvm_free_state(st); return nullptr; }
	if (hipGetDevice(&st->device) != hipSuccess) { fail(VVM_E_HIP, "hipGetDevice"); vvm_free_state(st); return nullptr; }
	if (upload_code(st->wide, &st->d_code_wide) != VVM_OK) { vvm_free_state(st); return nullptr; }
	if (link_more_passes(st->script, false, st->wide, st->wide_more, st->d_code_wide_more) != VVM_OK) { vvm_free_state(st); return nullptr; }
	// the light program of every pass that has one dataset: pass p of `wide` is the dataset of the (2 p)-th acquireindex where the
	// datasets go two to a pass, else of the p-th; that is the pass number of the script linked one dataset to a pass
	st->light_more.resize(st->wide_more.size());
	st->d_code_light_more.assign(st->wide_more.size(), nullptr);
	for (uint32_t p = 0; p < st->wide.num_passes; ++p) {
		const vvm::LinkedProgram& whole = p ? st->wide_more[p - 1] : st->wide;
		if (whole.instances_per_lane != 4 || whole.full_kernel || whole.many_registers || whole.num_chains != 1 || whole.uses_ids) continue;
		std::string why;
		vvm::LinkedProgram& light = p ? st->light_more[p - 1] : st->light;
		if (vvm::link_program(st->script, false, light, why, st->wide.paired ? 2 * p : p, true, nullptr, 0) != VVM_OK) continue;
		if (upload_code(light, p ? &st->d_code_light_more[p - 1] : &st->d_code_light) != VVM_OK) { vvm_free_state(st); return nullptr; }
		if (p == 0) st->light_ready = true;
	}
	if (plan_through(st) != VVM_OK) { vvm_free_state(st); return nullptr; }
	hipError_t e;
	st->any_ids = st->wide.uses_ids;
	st->const_table.reserve(st->script.num_consts);      // (filled by the first Exec; reserved here so that no Exec allocates for them)
	st->const_dword.reserve(st->script.num_consts);
	for (const vvm::LinkedProgram& more : st->wide_more) st->any_ids = st->any_ids || more.uses_ids;
	bool any_perlin = st->wide.uses_perlin;      // (a pass links only what its dataset needs: any of them may be the one with the noise call)
	for (const vvm::LinkedProgram& more : st->wide_more) any_perlin = any_perlin || more.uses_perlin;
	if (any_perlin) {
		if ((e = hipMalloc((void**)&st->d_perlin, 512)) != hipSuccess) return bail(e, "hipMalloc(perlin)");
		if ((e = hipMemcpy(st->d_perlin, k_perlin_perm_host, 512, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(perlin)");
	}
	if ((e = hipMalloc((void**)&st->d_launch, sizeof(VvmLaunchState))) != hipSuccess) return bail(e, "hipMalloc(launch state)");
	if ((e = reset_launch_state(st->d_launch)) != hipSuccess) return bail(e, "reset(launch state)");
	for (const vvm::LinkedProgram::IdUse& u : st->wide.id_uses) st->id_datasets = std::max(st->id_datasets, u.dataset + 1);
	for (const vvm::LinkedProgram& more : st->wide_more)
		for (const vvm::LinkedProgram::IdUse& u : more.id_uses) st->id_datasets = std::max(st->id_datasets, u.dataset + 1);
	const size_t ncounts = 2 * st->num_datasets() + 4;      // (survivor counts, then one MaxUsedID word per dataset)
	st->id_num_free_start.assign(st->num_datasets(), 0);
	st->id_num_spawned_start.assign(st->num_datasets(), 0);
	if (ncounts * 4 <= sizeof(((VvmLaunchState*)nullptr)->reserved)) {
		// (a synchronous Exec of a small emitter is bound by the round trips of the call: one copy back instead of two is 4 of 25 us)
		st->d_counts = (int32_t*)((uint8_t*)st->d_launch + offsetof(VvmLaunchState, reserved));      // zeroed with the launch words
		st->counts_in_launch = true;
	} else {
		if ((e = hipMalloc((void**)&st->d_counts, ncounts * 4)) != hipSuccess) return bail(e, "hipMalloc(counts)");
		if ((e = memset_and_wait(st->d_counts, 0, ncounts * 4)) != hipSuccess) return bail(e, "hipMemset(counts)");
	}
	const size_t host_bytes = st->mirror_word() * 4 + sizeof(VvmLaunchState);
	if ((e = hipHostMalloc((void**)&st->h_counts, host_bytes, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc(counts)");
	memset(st->h_counts, 0, host_bytes);
	if (state_stream(st->device, &st->stream) != VVM_OK) { vvm_free_state(st); return nullptr; }
	return st;
}

VVM_API void vvm_free_state(vvm_state* st)
{
	if (!st) return;
	g_state_generation.fetch_add(1, std::memory_order_relaxed);
	if (st->last_stream) (void)hipStreamSynchronize(st->last_stream);
	st->ring.release();
	if (st->d_code_wide) (void)hipFree(st->d_code_wide);
	if (st->d_code_narrow) (void)hipFree(st->d_code_narrow);
	if (st->d_code_light) (void)hipFree(st->d_code_light);
	for (VvmIns* p : st->d_code_light_more) if (p) (void)hipFree(p);
	if (st->d_code_lean) (void)hipFree(st->d_code_lean);
	if (st->d_ride) (void)hipFree(st->d_ride);
	for (VvmIns* p : st->d_code_wide_more) if (p) (void)hipFree(p);
	for (VvmIns* p : st->d_code_narrow_more) if (p) (void)hipFree(p);
	if (st->d_code_solo) (void)hipFree(st->d_code_solo);
	for (VvmIns* p : st->d_code_solo_more) if (p) (void)hipFree(p);
	if (st->d_perlin) (void)hipFree(st->d_perlin);
	if (st->d_launch) (void)hipFree(st->d_launch);
	if (st->d_status) (void)hipFree(st->d_status);
	if (st->d_counts && !st->counts_in_launch) (void)hipFree(st->d_counts);
	if (st->d_chunk_prefix) (void)hipFree(st->d_chunk_prefix);
	if (st->d_id_scratch) (void)hipFree(st->d_id_scratch);
	if (st->d_rand_words) (void)hipFree(st->d_rand_words);
	if (st->d_id_snapshot) (void)hipFree(st->d_id_snapshot);
	if (st->h_counts) (void)hipHostFree(st->h_counts);
	for (hipEvent_t e : st->ev_start) (void)hipEventDestroy(e);
	for (hipEvent_t e : st->ev_stop) (void)hipEventDestroy(e);
	delete st;
}

// Enqueues one Exec.  no_alloc: the Exec works in the device's scratch array instead of the state's own buffers and refuses
// whatever would need an allocation (vvm_exec_noalloc).
static int exec_enqueue(vvm_exec_context* ctx, const vvm_exec_options* opt, bool no_alloc, const PreparedExec* chunk = nullptr);

// An Exec of a script that takes IDs from a dataset's free list and returns IDs to it (acquire_id and update_id on one dataset).  The
// reference runs such a script chunk after chunk on the one list (VectorVMRuntime.cpp:1175-1181, 1411-1435): the IDs a chunk's dead
// instances return are the first ones the next chunk takes.  That is a dependency from chunk to chunk through the data, so the Exec
// is run the same way: one launch per chunk of the reference's chunking, each finished before the next starts (its counters are
// what the next one starts from), outputs appended behind the survivors so far.  Within a chunk the places the returned IDs go to
// are places the chunk's own IDs are read from, by other workgroups at other times: acquire_id reads a copy made ahead of the launch.
// Correct, not fast: a launch and a host round trip per chunk (about 30 us), for the emitters with persistent IDs whose spawn
// script carries the first update.  Synchronous; vvm_exec_finish then has nothing left to do.
static int exec_serial_chunks(vvm_exec_context* ctx, const vvm_exec_options* opt)
{
	vvm_state* st = ctx->VVMState;
	const vvm::ThawedScript& s = st->script;
	if (opt && opt->RandXorwowState && !st->wide.rand_draws.empty())
		return fail(VVM_E_UNSUPPORTED_OP, "RandXorwowState: not available to a script with update_id and acquire_id on one dataset");
	if (ctx->NumInstances < 0) return fail(VVM_E_INVALID_ARG, "negative NumInstances");
	uint32_t both_dataset = 0;
	for (const vvm::LinkedProgram::IdUse& u : st->wide.id_uses) if (u.serial) both_dataset = u.dataset;      // (every pass keeps the dataset's entry)
	if ((int)both_dataset >= ctx->DataSets.num || !ctx->DataSets.data) return fail(VVM_E_INVALID_ARG, "the script's ID instructions name a dataset the context does not provide");
	const vvm_dataset_meta& ds = ctx->DataSets.data[both_dataset];
	if (!ds.FreeIDTable || !ds.FreeIDTable->data || !ds.NumFreeIDs) return fail(VVM_E_INVALID_ARG, "the script has ID instructions but the dataset has no FreeIDTable / NumFreeIDs");
	uint32_t chunk_instances = 0, num_chunks = 0;
	reference_chunking(s, ctx, chunk_instances, num_chunks);      // (also what the whole Exec leaves in ctx->Internal)
	const uint32_t n = (uint32_t)ctx->NumInstances;
	if (chunk_instances > st->id_snapshot_capacity) {
		if (st->last_stream) HIP_TRY(hipStreamSynchronize(st->last_stream));
		if (st->d_id_snapshot) (void)hipFree(st->d_id_snapshot);
		st->d_id_snapshot = nullptr; st->id_snapshot_capacity = 0;
		HIP_TRY(hipMalloc((void**)&st->d_id_snapshot, (size_t)chunk_instances * 4));
		st->id_snapshot_capacity = chunk_instances;
	}
	hipStream_t stream = (opt && opt->Stream) ? (hipStream_t)opt->Stream : st->stream;
	const size_t nd = std::min<size_t>(s.max_output_dataset, (size_t)ctx->DataSets.num);
	std::vector<int32_t> written(nd, 0);
	vvm_exec_context one = *ctx;      // (shares the dataset descriptions: the chunk's counters and survivor counts land in the caller's)
	uint32_t first = 0;
	do {
		const uint32_t count = std::min(chunk_instances, n - first);
		one.NumInstances = (int32_t)count;
		PreparedExec where;
		where.serial_chunk = true;
		where.chunk_first = first;
		where.out_base = st->wide.pass_dataset < nd ? (uint32_t)written[st->wide.pass_dataset] : 0u;
		where.out_bases = written.data(); where.num_out_bases = (uint32_t)nd;      // (every dataset's outputs go behind its own survivors so far)
		const int32_t num_free = *ds.NumFreeIDs;
		if (num_free < (int32_t)count) return fail(VVM_E_INVALID_ARG, "fewer free IDs than instances");              // check(), RT.cpp:1415
		if (count) {
			if (st->last_stream && st->last_stream != stream) HIP_TRY(hipStreamSynchronize(st->last_stream));
			HIP_TRY(hipMemcpyAsync(st->d_id_snapshot, ds.FreeIDTable->data + (num_free - (int32_t)count), (size_t)count * 4, hipMemcpyDeviceToDevice, stream));
		}
		where.free_read_snapshot = st->d_id_snapshot;
		int rc = exec_enqueue(&one, opt, false, &where);
		if (rc == VVM_OK) rc = vvm_exec_finish(&one);
		if (rc != VVM_OK) return rc;
		for (size_t d = 0; d < nd; ++d) written[d] += ctx->DataSets.data[d].DataSetAccessIndex + 1;
		first += count;
	} while (first < n);
	for (size_t d = 0; d < nd; ++d) ctx->DataSets.data[d].DataSetAccessIndex = written[d] - 1;             // RT.cpp:2305-2308
	st->serial_done = true;
	return VVM_OK;
}

static int exec_enqueue(vvm_exec_context* ctx, const vvm_exec_options* opt, bool no_alloc, const PreparedExec* chunk)
{
	if (!ctx || !ctx->VVMState) return fail(VVM_E_INVALID_ARG, "null context or state");
	vvm_state* st = ctx->VVMState;
	if (!chunk && st->wide.serial_chunks) {
		if (no_alloc) return fail(VVM_E_UNSUPPORTED_OP, "vvm_exec_noalloc: scripts with update_id / acquire_id need buffers sized per Exec");
		{
			int dev = -1;
			HIP_TRY(hipGetDevice(&dev));
			if (dev != st->device) return fail(VVM_E_INVALID_ARG, "the state was allocated on another device than the one that is current");
		}
		return exec_serial_chunks(ctx, opt);
	}
	{   // the state's code, tables and launch words live on one device: the Exec has to be issued with that device current
		int dev = -1;
		HIP_TRY(hipGetDevice(&dev));
		if (dev != st->device) return fail(VVM_E_INVALID_ARG, "the state was allocated on another device than the one that is current");
	}
	if (st->any_ids && st->results_pending)      // (the dataset's counters are read here and written back by the finish: see include/vvm_abi.h)
		return fail(VVM_E_INVALID_ARG, "a script with update_id / acquire_id takes one Exec at a time: finish the queued Exec first");
	if (no_alloc) {
		if (st->any_ids) return fail(VVM_E_UNSUPPORTED_OP, "vvm_exec_noalloc: scripts with update_id / acquire_id need buffers sized per Exec");
		if ((opt && (opt->Flags & VVM_EXEC_NO_VGPR_WIDE)) && st->wide.instances_per_lane != 1 && !st->narrow_ready)
			return fail(VVM_E_UNSUPPORTED_OP, "vvm_exec_noalloc: the one-instance-per-lane program of this state has not been linked yet");
		if ((size_t)std::max<uint32_t>(st->script.max_output_dataset, 1) + 4 > sizeof(((VvmLaunchState*)nullptr)->reserved) / 4)
			return fail(VVM_E_TOO_LARGE, "vvm_exec_noalloc: too many output datasets for the scratch array's count words");
	}
	if (no_alloc && opt && opt->RandXorwowState && !st->wide.rand_draws.empty())
		return fail(VVM_E_UNSUPPORTED_OP, "vvm_exec_noalloc: RandXorwowState needs a buffer sized per Exec");
	PreparedExec pe;
	pe.lean_max_tiles = no_alloc ? VVM_NOALLOC_RIDE_TILES : 0;
	if (chunk) { pe.serial_chunk = true; pe.chunk_first = chunk->chunk_first; pe.out_base = chunk->out_base; pe.out_bases = chunk->out_bases; pe.num_out_bases = chunk->num_out_bases; pe.free_read_snapshot = chunk->free_read_snapshot; }
	int rc = prepare(st, ctx, opt, pe);
	if (rc != VVM_OK) return rc;
	if (no_alloc && pe.reference_random) return fail(VVM_E_UNSUPPORTED_OP, "vvm_exec_noalloc: RandXorwowState needs a buffer sized per Exec");
	uint8_t* const scratch_ptr = no_alloc ? g_dev.scratch_ptr[st->device] : nullptr;
	if (no_alloc) {
		if (!scratch_ptr) return fail(VVM_E_HIP, "vvm_exec_noalloc: the scratch array of this device has no address");
		const uint32_t lanes = pe.instances_per_lane;
		if ((uint64_t)pe.tiles * pe.lp->num_chains > VVM_NOALLOC_MAX_TILES) {      // (every chain has look-back words of its own)
			char msg[192];
			snprintf(msg, sizeof msg, "vvm_exec_noalloc: %d instances are %u tiles at %u instances per lane; the scratch array holds look-back words for %u tiles",
			         ctx->NumInstances, pe.tiles, lanes, VVM_NOALLOC_MAX_TILES);
			return fail(VVM_E_TOO_LARGE, msg);
		}
	}
	hipStream_t stream = (opt && opt->Stream) ? (hipStream_t)opt->Stream : st->stream;
	if (st->last_stream && st->last_stream != stream) HIP_TRY(hipStreamSynchronize(st->last_stream));
	st->last_stream = stream;
	st->in_flight = true;
	st->last_width = pe.instances_per_lane;
	int32_t* const caller_counts = (opt && opt->DeviceCounts && !st->any_ids) ? opt->DeviceCounts : nullptr;
	// the launch words, look-back words and counts this Exec runs on
	VvmLaunchState* const d_launch = no_alloc ? (VvmLaunchState*)scratch_ptr : st->d_launch;
	int32_t* const own_counts = no_alloc ? (int32_t*)(scratch_ptr + offsetof(VvmLaunchState, reserved)) : st->d_counts;
	st->counts_src = caller_counts ? caller_counts : own_counts;
	st->cur_launch = d_launch;
	st->cur_counts = own_counts;
	st->cur_counts_in_launch = no_alloc ? true : st->counts_in_launch;
	if (pe.tiles == 0) {     // nothing to run: zero chunks in the reference too (RT.cpp:2277)
		if (st->results_pending) { HIP_TRY(hipStreamSynchronize(stream)); st->results_pending = false; }
		if (caller_counts) HIP_TRY(hipMemsetAsync(caller_counts, 0, 4 * (size_t)std::max<uint32_t>(st->script.max_output_dataset, 1), stream));
		st->h_counts[0] = 0;
		for (uint32_t i = 0; i < st->script.max_output_dataset; ++i) st->h_counts[i] = 0;
		return VVM_OK;
	}
	if (!no_alloc) {
		if ((uint64_t)pe.tiles * pe.lp->num_chains > 0xFFFFFFFFull) return fail(VVM_E_TOO_LARGE, "too many tiles");
		rc = ensure_status(st, pe.tiles * pe.lp->num_chains);      // (every chain has look-back words of its own)
		if (rc != VVM_OK) return rc;
	}
	unsigned long long* d_status = no_alloc ? (unsigned long long*)(scratch_ptr + VVM_NOALLOC_STATUS_OFF) : st->d_status;
	st->cur_status = d_status;
	st->cur_status_words = no_alloc ? (size_t)VVM_NOALLOC_MAX_TILES : (size_t)st->status_capacity;
	const bool span = st->timing && st->timing_span && !no_alloc;
	const bool timed = st->timing && !st->timing_span && st->ev_used < 4096 && (!no_alloc || st->ev_used < st->ev_start.size());      // (no new events without allocating)
	if (span) {
		if (st->ev_start.empty()) {
			hipEvent_t a = nullptr, b = nullptr;
			HIP_TRY(hipEventCreate(&a));
			HIP_TRY(hipEventCreate(&b));
			st->ev_start.push_back(a); st->ev_stop.push_back(b);
		}
		if (!st->span_open) { HIP_TRY(hipEventRecord(st->ev_start[0], stream)); st->span_open = true; }
		st->span_closed = false;
		st->span_launches++;
	}
	if (timed) {
		if (st->ev_used == st->ev_start.size()) {
			hipEvent_t a = nullptr, b = nullptr;
			HIP_TRY(hipEventCreate(&a));
			HIP_TRY(hipEventCreate(&b));
			st->ev_start.push_back(a); st->ev_stop.push_back(b);
		}
	}
	if (pe.reference_random) {
		rc = enqueue_reference_random(st, ctx, opt, pe, stream);
		if (rc != VVM_OK) return rc;
	}
	const uint32_t num_passes = pe.lp->num_passes;
	if (num_passes > 1 && !no_alloc && !chunk && !(opt && (opt->Flags & VVM_EXEC_SEPARATE_PASSES))) {      // (a chunk's passes carry the chunk's place: one launch each)
		bool handled = false;
		rc = enqueue_fused_passes(st, ctx, opt, num_passes, stream, d_launch, timed, handled);
		if (rc != VVM_OK) return rc;
		if (handled) { st->results_pending = true; return VVM_OK; }
	}
	// Otherwise one launch per pass (one pass unless the script runs acquireindex for several datasets).  The passes share the
	// look-back words: every launch leaves a new epoch behind, which invalidates the words of the one before.
	for (uint32_t pass = 0; pass < num_passes; ++pass) {
		if (pass != 0) {
			rc = prepare(st, ctx, opt, pe, pass);
			if (rc != VVM_OK) return rc;
			// A later pass may run on a kernel with smaller tiles than the first (each pass is linked for what its own dataset needs:
			// one may fit four instances per lane where another takes one), so it can have more tiles, and more look-back words.
			if (no_alloc) {
				if ((uint64_t)pe.tiles * pe.lp->num_chains > VVM_NOALLOC_MAX_TILES)
					return fail(VVM_E_TOO_LARGE, "vvm_exec_noalloc: a pass of the script has more tiles than the scratch array holds look-back words for");
			} else {
				if ((uint64_t)pe.tiles * pe.lp->num_chains > 0xFFFFFFFFull) return fail(VVM_E_TOO_LARGE, "too many tiles");
				rc = ensure_status(st, pe.tiles * pe.lp->num_chains);
				if (rc != VVM_OK) return rc;
				d_status = st->d_status;
				st->last_width = pe.instances_per_lane;
				st->cur_status = d_status;
				st->cur_status_words = (size_t)st->status_capacity;
			}
		}
		VvmThroughArgs through;
		if (pe.lean) {
			uint32_t* ride = no_alloc ? (uint32_t*)(scratch_ptr + VVM_NOALLOC_RIDE_OFF) : nullptr;      // (prepare has kept the Exec within VVM_NOALLOC_RIDE_TILES)
			if (!no_alloc) {
				rc = ensure_ride(st, pe.tiles, stream);
				if (rc != VVM_OK) return rc;
				ride = st->d_ride;
			}
			pe.ride_addr = (uint64_t)(uintptr_t)ride;
			through.status = d_status; through.ride = ride;
			through.num_instances = (uint32_t)ctx->NumInstances; through.num_tiles = pe.tiles; through.pad = 0;
			through.instances_per_lane = pe.instances_per_lane;
			// (every block's streams are validated before anything is launched: last block first, so that the first one is left in place)
			const size_t blocks = (st->through.size() + VVM_THROUGH_MAX_PAIRS - 1) / VVM_THROUGH_MAX_PAIRS;
			for (size_t b = blocks; b-- > 0;) {
				rc = fill_through(st, ctx, through, b * VVM_THROUGH_MAX_PAIRS);
				if (rc != VVM_OK) return rc;
			}
		}
		if (pe.lp->updates_ids) {      // (one update_id per pass, and the passes follow each other on the stream: they share one piece of the buffers)
			rc = ensure_id_buffers(st, 1, pe.num_chunks, (uint32_t)ctx->NumInstances, stream);
			if (rc != VVM_OK) return rc;
		}
		rc = stage_id_counters(st, ctx, *pe.lp, stream);
		if (rc != VVM_OK) return rc;
		// A small block travels as a kernel argument; a larger one, or one whose ID descriptor other kernels read, through a slot
		// of the ring (pinned copy, device copy, one host-to-device copy ahead of the launch).
		const bool fits_inline = pe.bytes <= VVM_INLINE_PARAM_BYTES && !pe.lp->uses_ids;
		if (no_alloc && !fits_inline) return fail(VVM_E_TOO_LARGE, "vvm_exec_noalloc: the script's parameter block (constants, stream addresses) exceeds the 1 KB that travels as a kernel argument");
		const bool inline_block = fits_inline && (no_alloc || (!g_no_inline_params && !(opt && (opt->Flags & VVM_EXEC_PARAMS_IN_MEMORY))));
		int slot = -1;
		alignas(64) uint8_t local_block[VVM_INLINE_PARAM_BYTES];
		if (inline_block) {
			rc = fill_params(st, ctx, opt, pe, local_block, nullptr, d_status, 0, no_alloc ? st->counts_src : nullptr);
			if (rc != VVM_OK) return rc;
		} else {
			rc = st->ring.reserve(pe.bytes);
			if (rc != VVM_OK) return rc;
			rc = st->ring.acquire(slot);
			if (rc != VVM_OK) return rc;
			rc = fill_params(st, ctx, opt, pe, st->ring.host[slot], st->ring.dev[slot], d_status, 0);
			if (rc != VVM_OK) return rc;
			HIP_TRY(hipMemcpyAsync(st->ring.dev[slot], st->ring.host[slot], pe.bytes, hipMemcpyHostToDevice, stream));
		}
		if (timed && pass == 0) HIP_TRY(hipEventRecord(st->ev_start[st->ev_used], stream));
		EpochMirror& epoch = no_alloc ? g_dev.scratch_epoch[st->device] : st->epoch;
		HIP_TRY(epoch.before_launch(d_status, st->cur_status_words, stream));
		const int lrc = vvm_launch_exec_kernel((int)pe.instances_per_lane, inline_block ? nullptr : (const VvmExecDesc*)st->ring.dev[slot], 1, pe.tiles, d_launch,
		                                       pe.launch_many() ? 1 : 0, pe.lds_slots(), pe.full() ? ((pe.lean && pe.instances_per_lane == 4) ? 2 : 1) : 0,
		                                       grid_for(pe.tiles, pe.instances_per_lane, pe.full(), pe.launch_many()), stream,
		                                       inline_block ? local_block : nullptr, inline_block ? (uint32_t)pe.bytes : 0u, 0u, 0);
		if (lrc != 0) return hip_fail((hipError_t)lrc, "kernel launch");
		epoch.launched();
		if (pe.lean) {
			for (size_t first = 0; first < st->through.size(); first += VVM_THROUGH_MAX_PAIRS) {
				if (first != 0) {
					rc = fill_through(st, ctx, through, first);
					if (rc != VVM_OK) return rc;
				}
				const int trc = vvm_launch_through_kernel(&through, stream);
				if (trc != 0) return hip_fail((hipError_t)trc, "kernel launch (handed-through streams)");
			}
		}
		if (timed && pass + 1 == num_passes) { HIP_TRY(hipEventRecord(st->ev_stop[st->ev_used], stream)); st->ev_used++; }
		for (size_t u = 0; u < pe.lp->id_uses.size(); ++u) {
			if (!pe.lp->id_uses[u].updates) continue;
			const int mrc = vvm_launch_max_id_kernel((const VvmIdDesc*)(st->ring.dev[slot] + pe.off_ids) + u, pe.num_chunks, stream);
			if (mrc != 0) return hip_fail((hipError_t)mrc, "MaxUsedID kernel launch");
		}
		if (slot >= 0) {
			HIP_TRY(hipEventRecord(st->ring.done[slot], stream));
			st->ring.pending[slot] = true;
		}
	}
	// (the survivor counts and the launch words come back in vvm_exec_finish: of several Execs queued on a state only the last
	// one's are reported, and two copies per Exec were a fifth of the host's time per call)
	st->results_pending = true;
	return VVM_OK;
}

VVM_API int vvm_exec_finish(vvm_exec_context* ctx)
{
	if (!ctx || !ctx->VVMState) return fail(VVM_E_INVALID_ARG, "null context or state");
	vvm_state* st = ctx->VVMState;
	if (st->serial_done) { st->serial_done = false; return VVM_OK; }      // (exec_serial_chunks has finished every chunk and written the results)
	const bool ran = ctx->NumInstances > 0;
	if (st->last_stream) {
		const size_t nc = st->num_datasets(), mirror = st->mirror_word();
		const size_t with_ids = st->any_ids ? 2 * nc : nc;      // words to read back: the survivor counts, and the MaxUsedID words behind them
		const bool pending = st->results_pending;
		const bool with_caller = st->counts_src && st->counts_src != st->cur_counts;      // counts left with the caller
		// (timing over a span of Execs: the span ends behind the last kernel, ahead of this call's read-back; a later finish moves it)
		if (pending && st->timing_span && st->span_open) { HIP_TRY(hipEventRecord(st->ev_stop[0], st->last_stream)); st->span_closed = true; }
		if (pending) {
			if (with_caller) {
				HIP_TRY(hipMemcpyAsync(st->h_counts, st->counts_src, nc * 4, hipMemcpyDeviceToHost, st->last_stream));
				HIP_TRY(hipMemcpyAsync(st->h_counts + mirror, st->cur_launch, 80, hipMemcpyDeviceToHost, st->last_stream));
			} else if (st->cur_counts_in_launch) {      // the launch words up to and including stats[], then the counts and the MaxUsedID words behind them
				HIP_TRY(hipMemcpyAsync(st->h_counts + mirror, st->cur_launch, offsetof(VvmLaunchState, reserved) + with_ids * 4, hipMemcpyDeviceToHost, st->last_stream));
			} else {
				HIP_TRY(hipMemcpyAsync(st->h_counts, st->cur_counts, with_ids * 4, hipMemcpyDeviceToHost, st->last_stream));     // (and the MaxUsedID words)
				HIP_TRY(hipMemcpyAsync(st->h_counts + mirror, st->cur_launch, 80, hipMemcpyDeviceToHost, st->last_stream));     // the words up to and including stats[]
			}
		}
		HIP_TRY(hipStreamSynchronize(st->last_stream));
		if (pending && st->cur_counts_in_launch && !with_caller) memcpy(st->h_counts, (const uint8_t*)(st->h_counts + mirror) + offsetof(VvmLaunchState, reserved), with_ids * 4);
		if (pending && ran && st->cur_launch) {
			// The epoch word as the device holds it now.  It equals the host's twin unless launches went through these words that the
			// host did not issue itself: replays of a graph that an enqueue was captured into.  The twin follows the device; a value
			// below the twin's means that the count has gone round, and the next launch clears the look-back words first.
			const uint32_t on_device = ((const VvmLaunchState*)(st->h_counts + mirror))->epoch & 0x3FFFFFFFu;
			EpochMirror& twin = st->cur_launch == st->d_launch ? st->epoch : g_dev.scratch_epoch[st->device];
			if (on_device != 0 && on_device != twin.next) {
				if (on_device < twin.next) twin.wrapped = true;
				twin.next = on_device;
			}
		}
	}
	st->results_pending = false;
	st->in_flight = false;
	if (ran) {
		const VvmLaunchState* ls = (const VvmLaunchState*)(st->h_counts + st->mirror_word());
		if (getenv("VVM_DEBUG_STATS"))
			fprintf(stderr, "vvm stats (cumulative): lookbacks %llu rounds %llu polls %llu clocks/16 %llu | ticket %llu load %llu interp %llu tiles %llu\n", ls->stats[0], ls->stats[1], ls->stats[2], ls->stats[3], ls->stats[4], ls->stats[5], ls->stats[6], ls->stats[7]);
		if (ls->pad != 0) {
			const bool lds_base = ls->pad == 2;
			(void)reset_launch_state(st->cur_launch);
			(st->cur_launch == st->d_launch ? st->epoch : g_dev.scratch_epoch[st->device]) = EpochMirror();
			if (st->cur_status) (void)memset_and_wait(st->cur_status, 0, st->cur_status_words * 8);
			if (lds_base) return fail(VVM_E_HIP, "the kernel's dynamic LDS does not start at address 0 (the noise tables are addressed absolutely)");
			return fail(VVM_E_HIP, "look-back timed out inside the kernel (a tile never published its count)");
		}
	}
	if (ran && st->any_ids) {
		// the datasets' ID counters, as the reference leaves them (VectorVMRuntime.cpp:1175, 1383-1399, 1411-1429); a dataset's
		// update_id sits in the program of its own pass, acquire_id in every pass's
		struct Use { bool updates = false, acquires = false; };
		std::vector<Use> uses(st->num_datasets());
		auto gather = [&](const vvm::LinkedProgram& lp) {
			for (const vvm::LinkedProgram::IdUse& use : lp.id_uses) {
				uses[use.dataset].updates = uses[use.dataset].updates || use.updates;
				uses[use.dataset].acquires = uses[use.dataset].acquires || use.acquires;
			}
		};
		gather(st->wide);
		for (const vvm::LinkedProgram& more : st->wide_more) gather(more);
		for (size_t dataset = 0; dataset < uses.size(); ++dataset) {
			const Use use = uses[dataset];
			if (!use.updates && !use.acquires) continue;
			vvm_dataset_meta& ds = ctx->DataSets.data[dataset];
			int32_t num_free = st->id_num_free_start[dataset];
			if (use.acquires) {
				num_free -= ctx->NumInstances;
				*ds.NumSpawnedIDs = st->id_num_spawned_start[dataset] + ctx->NumInstances;
			}
			if (use.updates) {
				num_free += ctx->NumInstances - st->h_counts[dataset];
				*ds.MaxUsedID = st->h_counts[st->num_datasets() + dataset];
			}
			*ds.NumFreeIDs = num_free;
		}
	}
	return write_back_counts(st, ctx, ran);
}

VVM_API int vvm_exec_async(vvm_exec_context* ctx, const vvm_exec_options* opt) { return exec_enqueue(ctx, opt, false); }

VVM_API int vvm_exec_noalloc(vvm_exec_context* ctx, const vvm_exec_options* opt)
{
	if (!ctx || !ctx->VVMState) return fail(VVM_E_INVALID_ARG, "null context or state");
	const int dev = ctx->VVMState->device;
	if (dev < 0 || dev >= 64) return fail(VVM_E_INVALID_ARG, "device ordinal out of range");
	// every state on a device works in that device's one scratch array (launch words, counts, look-back words, ride words): the
	// calls are synchronous, so concurrent callers simply take turns
	std::lock_guard<std::mutex> lock(g_dev.noalloc_mutex[dev]);
	int rc = exec_enqueue(ctx, opt, true);
	if (rc != VVM_OK) return rc;
	return vvm_exec_finish(ctx);
}

// Test hook: sets the epoch (1 .. 2^30 - 1) that the next launch through the state's own launch words (scratch = 0) or through the
// scratch array of the state's device (scratch = 1) tags its look-back words with, as if that many launches had gone before.
VVM_API int vvm_debug_set_launch_epoch(vvm_state* st, uint32_t epoch, int scratch)
{
	if (!st || epoch == 0 || epoch > 0x3FFFFFFFu) return fail(VVM_E_INVALID_ARG, "bad state or epoch");
	if (st->last_stream) HIP_TRY(hipStreamSynchronize(st->last_stream));
	VvmLaunchState* ls = scratch ? (VvmLaunchState*)g_dev.scratch_ptr[st->device] : st->d_launch;
	if (!ls) return fail(VVM_E_INVALID_ARG, "no launch words");
	HIP_TRY(hipMemcpy(&ls->epoch, &epoch, 4, hipMemcpyHostToDevice));
	EpochMirror& m = scratch ? g_dev.scratch_epoch[st->device] : st->epoch;
	m.next = epoch;
	m.wrapped = false;
	return VVM_OK;
}

VVM_API int vvm_exec_ex(vvm_exec_context* ctx, const vvm_exec_options* opt)
{
	int rc = vvm_exec_async(ctx, opt);
	if (rc != VVM_OK) return rc;
	return vvm_exec_finish(ctx);
}

VVM_API int vvm_exec(vvm_exec_context* ctx) { return vvm_exec_ex(ctx, nullptr); }

// --- batched launch ---------------------------------------------------------------------------------
namespace {
struct BatchWorkspace {
	std::mutex mutex;
	uint8_t* host = nullptr; uint8_t* dev = nullptr; size_t bytes = 0;
	unsigned long long* d_status = nullptr; uint32_t status_capacity = 0;
	int32_t* d_counts = nullptr; int32_t* h_counts = nullptr; size_t counts_capacity = 0;   // survivor counts of all Execs of a batch
	VvmLaunchState* d_launch[8] = {};
	VvmLaunchState* h_launch = nullptr;
	uint32_t* d_ride = nullptr; uint32_t ride_capacity = 0;      // one word per lane and tile for the lean members (see vvm_state::lean), same tile numbering as d_status
	// Every launch of a batch tags its look-back words with a number drawn from this counter.  The launches share d_status, at
	// offsets that move from call to call, through one set of launch words per kernel shape: with each set counting its own
	// epochs, a word written through one set could carry the tag another set reaches later, and a tile that polls before its
	// predecessor has published would take that stale count.  One counter per array rules it out: a tag is never used twice
	// between two clears of the array.
	uint32_t next_tag = 1;
	hipStream_t stream = nullptr;
	// The parameter blocks of a slice are uploaded on a stream of their own, so that the copy runs beside the kernels of the slice
	// before; the kernels' stream waits for the slice's event.  (On one stream every upload sat between two kernels.)
	hipStream_t upload_stream = nullptr;
	hipEvent_t uploaded[8] = {};
	hipEvent_t finished[8] = {};      // per slice, behind its kernels and the read-back of its counts
	int device = -1;      // the workspace belongs to the device of its first batch

	void release()
	{
		if (stream) (void)hipStreamSynchronize(stream);
		if (upload_stream) (void)hipStreamSynchronize(upload_stream);
		if (host) (void)hipHostFree(host);
		if (dev) (void)hipFree(dev);
		if (d_status) (void)hipFree(d_status);
		if (d_ride) (void)hipFree(d_ride);
		if (d_counts) (void)hipFree(d_counts);
		if (h_counts) (void)hipHostFree(h_counts);
		if (h_launch) (void)hipHostFree(h_launch);
		for (VvmLaunchState*& p : d_launch) { if (p) (void)hipFree(p); p = nullptr; }
		for (int i = 0; i < 8; ++i) {
			if (uploaded[i]) (void)hipEventDestroy(uploaded[i]);
			if (finished[i]) (void)hipEventDestroy(finished[i]);
			uploaded[i] = finished[i] = nullptr;
		}
		if (upload_stream) (void)hipStreamDestroy(upload_stream);
		if (stream) (void)hipStreamDestroy(stream);
		host = dev = nullptr; d_status = nullptr; d_counts = h_counts = nullptr; h_launch = nullptr; stream = upload_stream = nullptr; d_ride = nullptr;
		bytes = 0; status_capacity = 0; counts_capacity = 0; ride_capacity = 0;
	}
};
BatchWorkspace g_batch;      // the workspace of vvm_exec_batch; a vvm_batch handle owns one of its own
uint64_t g_batch_stamp = 0;
std::mutex g_batch_stamp_mutex;

// A batch is cut into slices of consecutive Execs; within a slice the Execs are grouped by kernel shape (instances per lane,
// register tier, full), one launch per group.  Everything here is decided from the contexts' sizes and the states' programs;
// a vvm_batch handle keeps it from call to call.
struct BatchGroup { int key = 0; std::vector<int> members; uint32_t tiles = 0, spill = 0; size_t desc_off = 0; };
struct BatchSlice {
	int first = 0, count = 0; std::vector<BatchGroup> groups; size_t region_off = 0, desc_bytes = 0, head_bytes = 0, used_bytes = 0;
	// the lean members of the slice (vvm_state::lean): their VvmThroughArgs blocks lie side by side at through_off of the parameter
	// buffer, in the order of `through_members`, and one launch of vvm_through_batch_kernel behind the slice's kernels serves them all
	std::vector<int> through_members; size_t through_off = 0; uint32_t through_tiles = 0, through_max_pairs = 0;
};
struct BatchPlan {
	int n = 0;
	std::vector<PreparedExec> pes;
	std::vector<uint8_t> single;            // Execs that take launches of their own (ID instructions, several acquired datasets)
	std::vector<BatchSlice> slices;
	std::vector<size_t> counts_off;         // [i] first word of Exec i in the batch's counts array
	std::vector<size_t> desc_off, const_off, table_off;   // [i] where Exec i's descriptor, constants and other tables lie in the parameter buffer
	std::vector<uint32_t> status_off, first_tile;   // [i] first look-back word; ticket number of the Exec's first tile within its launch
	std::vector<size_t> through_off;                // [i] lean members: where the Exec's VvmThroughArgs block lies in the parameter buffer
	std::vector<uint32_t> through_first_tile;       // [i] ... and the number of its first tile within the slice's copy-kernel launch
	bool any_lean = false;
	size_t total_counts = 0, total_bytes = 0;
	uint32_t total_tiles = 0;
};

int kernel_key(const PreparedExec& pe) { return (int)pe.instances_per_lane * 4 + (pe.launch_many() ? 2 : 0) + (pe.full() ? 1 : 0); }

// vvm_exec_batch keeps the plan of its last call: a caller that runs the same emitters tick after tick (same states, sizes and
// options; stream addresses and constants are read anew on every call) pays for sorting them into launches once.  What the plan was
// made from, per Exec:
struct BatchCallKey {
	const vvm_state* state; int32_t num_instances, num_datasets, num_ext_fns, num_const_tables; uint32_t flags; bool xorwow;
	bool operator==(const BatchCallKey& o) const
	{
		return state == o.state && num_instances == o.num_instances && num_datasets == o.num_datasets && num_ext_fns == o.num_ext_fns &&
		       num_const_tables == o.num_const_tables && flags == o.flags && xorwow == o.xorwow;
	}
};
BatchCallKey batch_call_key(const vvm_exec_context& c, const vvm_exec_options* o)
{
	return BatchCallKey{c.VVMState, c.NumInstances, c.DataSets.num, c.ExtFunctionTable.num, c.ConstantTableCount, o ? o->Flags : 0u,
	                    o && o->RandXorwowState != nullptr};
}
struct BatchPlanCache {
	bool valid = false;
	uint64_t generation = 0;
	int num_slices = 0;
	bool have_opts = false;
	std::vector<BatchCallKey> keys;
	BatchPlan plan;
	uint64_t plans_made = 0;      // (test hook: vvm_debug_batch_plans_made)
};
BatchPlanCache g_batch_cache;      // (guarded by g_batch.mutex)

// Tiles, ticket numbers, look-back words and buffer offsets of every Exec, from the groups' member lists and the prepared sizes.
void batch_layout(BatchPlan& plan, const vvm_exec_context* ctxs)
{
	plan.total_tiles = 0;
	plan.total_counts = 0;
	plan.any_lean = false;
	size_t needed = 0;
	for (int i = 0; i < plan.n; ++i) {
		plan.counts_off[(size_t)i] = plan.total_counts;
		if (!plan.single[(size_t)i]) plan.total_counts += std::max<uint32_t>(ctxs[i].VVMState->script.max_output_dataset, 1);
	}
	uint32_t status_off = 0;
	for (BatchSlice& slice : plan.slices) {
		// a slice's region of the parameter buffer: the descriptors of its groups, contiguous per group (the kernel indexes them),
		// then the constants of its Execs (descriptors and constants are what changes from tick to tick: they are uploaded
		// together, without the stream tables behind them), then the other tables
		slice.region_off = needed;
		slice.desc_bytes = align_up((size_t)slice.count * sizeof(VvmExecDesc), 256);
		size_t const_bytes = 0;
		for (const BatchGroup& grp : slice.groups)
			for (int i : grp.members) const_bytes += plan.pes[(size_t)i].off_in - plan.pes[(size_t)i].off_consts;
		slice.head_bytes = slice.desc_bytes + align_up(const_bytes, 256);
		size_t desc_off = slice.region_off, const_off = slice.region_off + slice.desc_bytes, table_off = slice.region_off + slice.head_bytes;
		for (BatchGroup& grp : slice.groups) {
			grp.desc_off = desc_off;
			uint32_t first_tile = 0;
			grp.spill = 0;
			for (int i : grp.members) {
				const PreparedExec& pe = plan.pes[(size_t)i];
				plan.desc_off[(size_t)i] = desc_off;
				plan.const_off[(size_t)i] = const_off;
				plan.table_off[(size_t)i] = table_off;
				const_off += pe.off_in - pe.off_consts;
				plan.status_off[(size_t)i] = status_off;
				plan.first_tile[(size_t)i] = first_tile;
				grp.spill = std::max(grp.spill, pe.lds_slots());
				desc_off += sizeof(VvmExecDesc);
				table_off += pe.bytes;
				status_off += pe.tiles;
				first_tile += pe.tiles;
			}
			grp.tiles = first_tile;
		}
		slice.through_members.clear();
		slice.through_tiles = 0; slice.through_max_pairs = 0;
		slice.through_off = align_up(table_off, 256);
		for (const BatchGroup& grp : slice.groups)
			for (int i : grp.members) {
				if (!plan.pes[(size_t)i].lean) continue;
				plan.through_off[(size_t)i] = slice.through_off + slice.through_members.size() * sizeof(VvmThroughArgs);
				plan.through_first_tile[(size_t)i] = slice.through_tiles;
				slice.through_members.push_back(i);
				slice.through_tiles += plan.pes[(size_t)i].tiles;
				slice.through_max_pairs = std::max<uint32_t>(slice.through_max_pairs, (uint32_t)ctxs[i].VVMState->through.size());
				plan.any_lean = true;
			}
		if (!slice.through_members.empty()) table_off = slice.through_off + slice.through_members.size() * sizeof(VvmThroughArgs);
		slice.used_bytes = table_off - slice.region_off;
		needed += align_up(slice.used_bytes, 256);
	}
	plan.total_tiles = status_off;
	plan.total_bytes = needed;
}

int batch_plan(BatchWorkspace& ws, vvm_exec_context* ctxs, int n, const vvm_exec_options* opts, int num_slices, BatchPlan& plan)
{
	uint64_t stamp;
	{ std::lock_guard<std::mutex> lock(g_batch_stamp_mutex); stamp = ++g_batch_stamp; }
	plan = BatchPlan();
	plan.n = n;
	plan.pes.assign((size_t)n, PreparedExec());
	plan.single.assign((size_t)n, 0);
	plan.counts_off.assign((size_t)n, 0); plan.desc_off.assign((size_t)n, 0); plan.const_off.assign((size_t)n, 0); plan.table_off.assign((size_t)n, 0);
	plan.status_off.assign((size_t)n, 0); plan.first_tile.assign((size_t)n, 0);
	plan.through_off.assign((size_t)n, 0); plan.through_first_tile.assign((size_t)n, 0);
	plan.slices.assign((size_t)num_slices, BatchSlice());
	// The first slice is the smallest (the GPU idles while it is filled), and each later one is as large as all before it together:
	// filling an Exec's block takes the host about half the time the GPU needs for the Exec, so the fill stays ahead.
	// (five slices: n/16, n/4, n/2, 3n/4, n.  The host fills an Exec's block in about 0.11 us and needs about 30 us to upload and launch
	// a slice; the GPU takes about 0.18 us per 16 Ki-instance Exec.  The small first slice gets the GPU going early; the equal later ones
	// keep the wait for the last slice's fill short, which with n/8, n/4, n/2, n was 0.15 ms of idle GPU.)
	auto slice_end = [&](int sl) {
		if (sl + 1 == num_slices) return n;
		if (num_slices == 5) return (int)((int64_t)n * (sl == 0 ? 1 : 4 * sl) / 16);
		return (int)(((int64_t)n << sl) >> (num_slices - 1));      // n/8, n/4, n/2, n
	};
	for (int sl = 0; sl < num_slices; ++sl) {
		BatchSlice& slice = plan.slices[(size_t)sl];
		slice.first = sl == 0 ? 0 : slice_end(sl - 1);
		slice.count = slice_end(sl) - slice.first;
		for (int i = slice.first; i < slice.first + slice.count; ++i) {
			vvm_state* st = ctxs[i].VVMState;
			if (!st) return fail(VVM_E_INVALID_ARG, "null state in batch");
			if (st->device != ws.device) return fail(VVM_E_INVALID_ARG, "a state of the batch was allocated on another device than the one that is current");
			if (st->batch_stamp == stamp) return fail(VVM_E_INVALID_ARG, "a state appears twice in one batch (one Exec at a time per state)");
			st->batch_stamp = stamp;
			plan.pes[(size_t)i].in_batch = true;
			int rc = prepare(st, &ctxs[i], opts ? &opts[i] : nullptr, plan.pes[(size_t)i]);
			if (rc != VVM_OK) return rc;
			const vvm::LinkedProgram& lp = *plan.pes[(size_t)i].lp;
			// Scripts with ID instructions (they need the MaxUsedID pass and the dataset's counters) and scripts that take one kernel
			// pass per acquired dataset do not share a launch: they are enqueued one by one behind the batched kernels, on the same stream.
			if (lp.uses_ids || lp.num_passes > 1 || lp.num_chains > 1) { plan.single[(size_t)i] = 1; continue; }
			const int key = kernel_key(plan.pes[(size_t)i]);
			BatchGroup* g = nullptr;
			for (BatchGroup& c : slice.groups) if (c.key == key) g = &c;
			if (!g) { slice.groups.push_back(BatchGroup()); g = &slice.groups.back(); g->key = key; }
			g->members.push_back(i);
		}
		if (slice.groups.size() > 8) return fail(VVM_E_TOO_LARGE, "too many kernel shapes in one batch");
	}
	batch_layout(plan, ctxs);
	return VVM_OK;
}

// Whether the remembered plan is the plan of this call; if so, what prepare() leaves in every context is written again (the
// reference's chunking figures, which follow vm.PageSizeInKB) and its checks of the context's views are repeated.
bool batch_plan_still_holds(BatchPlanCache& cache, vvm_exec_context* ctxs, int n, const vvm_exec_options* opts, int num_slices)
{
	if (!cache.valid || cache.plan.n != n || cache.num_slices != num_slices || cache.have_opts != (opts != nullptr) ||
	    cache.generation != g_state_generation.load(std::memory_order_relaxed))
		return false;
	for (int i = 0; i < n; ++i) {
		const vvm_exec_context& c = ctxs[i];
		if (!(cache.keys[(size_t)i] == batch_call_key(c, opts ? &opts[i] : nullptr))) return false;
		if ((c.DataSets.num > 0 && !c.DataSets.data) || (c.ConstantTableCount > 0 && (!c.ConstantTableNumBytes || !c.ConstantTableData))) return false;
		const vvm::ThawedScript& s = c.VVMState->script;
		if (s.num_ext_fns > 0 && !c.ExtFunctionTable.data) return false;
		for (uint32_t f = 0; f < s.num_ext_fns; ++f) {
			const vvm_ext_function* fn = c.ExtFunctionTable.data[f];
			if (!fn || fn->builtin_id != ((s.ext_num_outputs[f] == 1) ? VVM_EXTFN_PERLIN3 : VVM_EXTFN_CURL_NOISE3)) return false;
		}
	}
	for (int i = 0; i < n; ++i) {
		PreparedExec& pe = cache.plan.pes[(size_t)i];
		reference_chunking(ctxs[i].VVMState->script, &ctxs[i], pe.chunk_instances, pe.num_chunks);
	}
	return true;
}

// Streams, events and buffers large enough for the plan (buffers only ever grow).
int batch_ensure_buffers(BatchWorkspace& ws, const BatchPlan& plan)
{
	if (!ws.stream) HIP_TRY(hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
	if (!ws.upload_stream) HIP_TRY(hipStreamCreateWithFlags(&ws.upload_stream, hipStreamNonBlocking));
	for (size_t sl = 0; sl < plan.slices.size(); ++sl)
		if (!ws.uploaded[sl]) {
			HIP_TRY(hipEventCreateWithFlags(&ws.uploaded[sl], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ws.finished[sl], hipEventDisableTiming));
		}
	if (plan.total_tiles > ws.status_capacity) {
		if (ws.d_status) (void)hipFree(ws.d_status);
		ws.status_capacity = plan.total_tiles + plan.total_tiles / 4 + 1024;
		HIP_TRY(hipMalloc((void**)&ws.d_status, (size_t)ws.status_capacity * 8));
		HIP_TRY(memset_and_wait(ws.d_status, 0, (size_t)ws.status_capacity * 8));      // tag 0 is never valid
	}
	if (plan.any_lean && plan.total_tiles > ws.ride_capacity) {
		if (ws.d_ride) (void)hipFree(ws.d_ride);
		ws.ride_capacity = plan.total_tiles + plan.total_tiles / 4 + 1024;
		HIP_TRY(hipMalloc((void**)&ws.d_ride, (size_t)ws.ride_capacity * VVM_BLOCK_THREADS * 4u));
	}
	const size_t launch_slots = 8 * 8;      // (slices x groups)
	if (!ws.h_launch) HIP_TRY(hipHostMalloc((void**)&ws.h_launch, launch_slots * sizeof(VvmLaunchState), hipHostMallocDefault));
	for (const BatchSlice& slice : plan.slices)
		for (size_t g = 0; g < slice.groups.size(); ++g)
			if (!ws.d_launch[g]) {
				HIP_TRY(hipMalloc((void**)&ws.d_launch[g], sizeof(VvmLaunchState)));
				HIP_TRY(reset_launch_state(ws.d_launch[g]));
			}
	// one counts array for the whole batch (a device-to-host copy per state would cost more than the kernel)
	if (plan.total_counts > ws.counts_capacity) {
		if (ws.d_counts) (void)hipFree(ws.d_counts);
		if (ws.h_counts) (void)hipHostFree(ws.h_counts);
		ws.counts_capacity = plan.total_counts + plan.total_counts / 4 + 64;
		HIP_TRY(hipMalloc((void**)&ws.d_counts, ws.counts_capacity * 4));
		HIP_TRY(hipHostMalloc((void**)&ws.h_counts, ws.counts_capacity * 4, hipHostMallocDefault));
	}
	if (plan.total_bytes > ws.bytes) {
		if (ws.host) (void)hipHostFree(ws.host);
		if (ws.dev) (void)hipFree(ws.dev);
		ws.bytes = align_up(plan.total_bytes + plan.total_bytes / 4, 4096);
		HIP_TRY(hipHostMalloc((void**)&ws.host, ws.bytes, hipHostMallocDefault));
		HIP_TRY(hipMalloc((void**)&ws.dev, ws.bytes));
	}
	return VVM_OK;
}

// Exec i's descriptor and tables into the pinned parameter buffer.  Per-exec parameter blocks are laid out as [desc | tables] by
// fill_params; here desc and tables are split, so fill into a scratch block and scatter (filling the pinned buffer in place
// measured slower: 0.58 ms against 0.42 ms for 4096 Execs).
int batch_fill_exec(BatchWorkspace& ws, const BatchPlan& plan, vvm_exec_context* ctxs, const vvm_exec_options* opts, int i, std::vector<uint8_t>& scratch)
{
	PreparedExec pe = plan.pes[(size_t)i];
	vvm_state* st = ctxs[i].VVMState;
	if (pe.lean) {
		// the program leaves its ride words in the batch's array, at the tiles its look-back words have; the Exec's block for the
		// copy kernel goes into the slice's row of such blocks
		uint32_t* ride = ws.d_ride + (size_t)plan.status_off[(size_t)i] * VVM_BLOCK_THREADS;
		pe.ride_addr = (uint64_t)(uintptr_t)ride;
		VvmThroughArgs* t = (VvmThroughArgs*)(ws.host + plan.through_off[(size_t)i]);
		memset(t, 0, sizeof(*t));
		t->status = ws.d_status + plan.status_off[(size_t)i]; t->ride = ride;
		t->num_instances = (uint32_t)ctxs[i].NumInstances; t->num_tiles = pe.tiles; t->pad = plan.through_first_tile[(size_t)i];
		t->instances_per_lane = 8;      // (lean members of a batch run on the light kernel)
		const int trc = fill_through(st, &ctxs[i], *t);
		if (trc != VVM_OK) return trc;
	}
	// tables live at table_off in the device buffer; fill_params computes table addresses as dev + pe.off_*
	uint8_t* dev_block = ws.dev + plan.table_off[(size_t)i];
	scratch.resize(pe.bytes);
	int rc = fill_params(st, &ctxs[i], opts ? &opts[i] : nullptr, pe, scratch.data(), dev_block, ws.d_status + plan.status_off[(size_t)i], plan.first_tile[(size_t)i],
	                     ws.d_counts + plan.counts_off[(size_t)i]);
	if (rc != VVM_OK) return rc;
	st->last_width = pe.instances_per_lane;
	((VvmExecDesc*)scratch.data())->consts = (const uint32_t*)(ws.dev + plan.const_off[(size_t)i]);      // the batch keeps the constants beside the descriptors
	memcpy(ws.host + plan.desc_off[(size_t)i], scratch.data(), sizeof(VvmExecDesc));
	memcpy(ws.host + plan.const_off[(size_t)i], scratch.data() + pe.off_consts, pe.off_in - pe.off_consts);
	memcpy(ws.host + plan.table_off[(size_t)i], scratch.data(), pe.bytes);
	return VVM_OK;
}

// Uploads `upload_bytes` of the slice's region (0: nothing has changed since the last upload), launches its groups and enqueues
// the read-back of its launch words and counts.
int batch_launch_slice(BatchWorkspace& ws, const BatchPlan& plan, int sl, size_t upload_bytes)
{
	const BatchSlice& slice = plan.slices[(size_t)sl];
	if (slice.count == 0) return VVM_OK;
	if (upload_bytes) {
		HIP_TRY(hipMemcpyAsync(ws.dev + slice.region_off, ws.host + slice.region_off, upload_bytes, hipMemcpyHostToDevice, ws.upload_stream));
		HIP_TRY(hipEventRecord(ws.uploaded[sl], ws.upload_stream));
		HIP_TRY(hipStreamWaitEvent(ws.stream, ws.uploaded[sl], 0));
	}
	for (size_t g = 0; g < slice.groups.size(); ++g) {
		const BatchGroup& grp = slice.groups[g];
		if (grp.tiles == 0) continue;
		const PreparedExec& pe0 = plan.pes[(size_t)grp.members[0]];
		const vvm::LinkedProgram& lp = *pe0.lp;
		if (ws.next_tag > VVM_MAX_LAUNCH_TAG) {      // the tags start over: behind the launches so far, no word may keep an old one
			HIP_TRY(hipMemsetAsync(ws.d_status, 0, (size_t)ws.status_capacity * 8, ws.stream));
			ws.next_tag = 1;
		}
		const uint32_t launch_tag = ws.next_tag++;
		const int lrc = vvm_launch_exec_kernel((int)pe0.instances_per_lane, (const VvmExecDesc*)(ws.dev + grp.desc_off), (uint32_t)grp.members.size(), grp.tiles,
		                                       ws.d_launch[g], pe0.launch_many() ? 1 : 0, grp.spill, lp.full_kernel ? 1 : 0,
		                                       grid_for(grp.tiles, pe0.instances_per_lane, lp.full_kernel, pe0.launch_many()), ws.stream, nullptr, 0, launch_tag, 0);
		if (lrc != 0) { (void)hipStreamSynchronize(ws.stream); return hip_fail((hipError_t)lrc, "batched kernel launch"); }
		HIP_TRY(hipMemcpyAsync(&ws.h_launch[(size_t)sl * 8 + g], ws.d_launch[g], 80, hipMemcpyDeviceToHost, ws.stream));
	}
	if (!slice.through_members.empty()) {
		const int trc = vvm_launch_through_batch_kernel((const VvmThroughArgs*)(ws.dev + slice.through_off), (uint32_t)slice.through_members.size(), slice.through_tiles,
		                                                slice.through_max_pairs, ws.stream);
		if (trc != 0) { (void)hipStreamSynchronize(ws.stream); return hip_fail((hipError_t)trc, "batched kernel launch (handed-through streams)"); }
	}
	// the slice's counts come back behind its kernels, so that the host can hand them to the contexts while later slices run
	const size_t c0 = plan.counts_off[(size_t)slice.first];
	const size_t c1 = slice.first + slice.count < plan.n ? plan.counts_off[(size_t)(slice.first + slice.count)] : plan.total_counts;
	if (c1 > c0) HIP_TRY(hipMemcpyAsync(ws.h_counts + c0, ws.d_counts + c0, (c1 - c0) * 4, hipMemcpyDeviceToHost, ws.stream));
	HIP_TRY(hipEventRecord(ws.finished[sl], ws.stream));
	return VVM_OK;
}

// The Execs that take launches of their own: enqueued behind the batched kernels.
int batch_enqueue_singles(BatchWorkspace& ws, const BatchPlan& plan, vvm_exec_context* ctxs, const vvm_exec_options* opts)
{
	for (int i = 0; i < plan.n; ++i) {
		if (!plan.single[(size_t)i]) continue;
		vvm_exec_options o = opts ? opts[i] : vvm_exec_options{};
		o.Stream = ws.stream;
		o.DeviceCounts = nullptr;      // (a batch keeps every Exec's counts in its own array)
		int rc = vvm_exec_async(&ctxs[i], &o);
		if (rc != VVM_OK) { (void)hipStreamSynchronize(ws.stream); return rc; }
	}
	return VVM_OK;
}

// Waits slice by slice, checks the launch words and hands the counts to the contexts; then finishes the single Execs.
int batch_collect(BatchWorkspace& ws, const BatchPlan& plan, vvm_exec_context* ctxs, double* wait_us)
{
	for (size_t sl = 0; sl < plan.slices.size(); ++sl) {
		const BatchSlice& slice = plan.slices[sl];
		if (slice.count == 0) continue;
		const auto t_wait = std::chrono::steady_clock::now();
		HIP_TRY(hipEventSynchronize(ws.finished[sl]));
		if (wait_us) *wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_wait).count();
		for (size_t g = 0; g < slice.groups.size(); ++g)
			if (slice.groups[g].tiles && ws.h_launch[sl * 8 + g].pad != 0) {
				(void)hipStreamSynchronize(ws.stream);
				for (size_t k = 0; k < 8; ++k) if (ws.d_launch[k]) (void)reset_launch_state(ws.d_launch[k]);
				(void)memset_and_wait(ws.d_status, 0, (size_t)ws.status_capacity * 8);
				return fail(VVM_E_HIP, "look-back timed out inside the batched kernel");
			}
		for (int i = slice.first; i < slice.first + slice.count; ++i) {
			if (plan.single[(size_t)i]) continue;
			vvm_state* st = ctxs[i].VVMState;
			const size_t ncounts = std::max<uint32_t>(st->script.max_output_dataset, 1);
			memcpy(st->h_counts, ws.h_counts + plan.counts_off[(size_t)i], ncounts * 4);
			write_back_counts(st, &ctxs[i], ctxs[i].NumInstances > 0);
		}
	}
	HIP_TRY(hipStreamSynchronize(ws.stream));      // (the Execs that took launches of their own)
	for (int i = 0; i < plan.n; ++i) {
		if (!plan.single[(size_t)i]) continue;
		int rc = vvm_exec_finish(&ctxs[i]);
		if (rc != VVM_OK) return rc;
	}
	return VVM_OK;
}

int batch_bind_device(BatchWorkspace& ws)
{
	int rc = ensure_device_info();
	if (rc != VVM_OK) return rc;
	int dev = -1;
	HIP_TRY(hipGetDevice(&dev));
	if (ws.device < 0) ws.device = dev;
	if (dev != ws.device) return fail(VVM_E_INVALID_ARG, "a batch serves the device that was current at its first use; another device is current now");
	return VVM_OK;
}

// An Exec of a state that is still in flight elsewhere has to finish before the batch touches the state.
int batch_settle_states(const BatchPlan& plan, vvm_exec_context* ctxs)
{
	for (int i = 0; i < plan.n; ++i) {
		vvm_state* st = ctxs[i].VVMState;
		if (st->in_flight && st->last_stream) { HIP_TRY(hipStreamSynchronize(st->last_stream)); st->in_flight = false; }
	}
	return VVM_OK;
}
}  // namespace

VVM_API int vvm_exec_batch(vvm_exec_context* ctxs, int32_t n, const vvm_exec_options* opts)
{
	if (n < 0 || (n > 0 && !ctxs)) return fail(VVM_E_INVALID_ARG, "bad batch");
	if (n == 0) return VVM_OK;
	BatchWorkspace& ws = g_batch;
	std::lock_guard<std::mutex> lock(ws.mutex);
	int rc = batch_bind_device(ws);
	if (rc != VVM_OK) return rc;
	const bool report = getenv("VVM_DEBUG_STATS") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
	const auto t_begin = now();
	// A slice is filled, uploaded and launched while the kernels of the slices before it run, so that most of the host's work on
	// the parameter blocks is hidden behind the GPU.
	const int num_slices = n >= 4096 ? 5 : n >= 1024 ? 3 : 1;
	BatchPlanCache& cache = g_batch_cache;
	BatchPlan& plan = cache.plan;
	if (!batch_plan_still_holds(cache, ctxs, n, opts, num_slices)) {
		cache.valid = false;
		++cache.plans_made;
		rc = batch_plan(ws, ctxs, n, opts, num_slices, plan);
		if (rc != VVM_OK) return rc;
		cache.generation = g_state_generation.load(std::memory_order_relaxed);
		cache.num_slices = num_slices;
		cache.have_opts = opts != nullptr;
		cache.keys.resize((size_t)n);
		for (int i = 0; i < n; ++i) cache.keys[(size_t)i] = batch_call_key(ctxs[i], opts ? &opts[i] : nullptr);
		// (a batch with Execs that take launches of their own is planned every time: what those need checked -- the ID tables'
		// counters -- changes from call to call)
		cache.valid = std::find(plan.single.begin(), plan.single.end(), (uint8_t)1) == plan.single.end();
	}
	rc = batch_settle_states(plan, ctxs);
	if (rc != VVM_OK) return rc;
	const auto t_prepared = now();
	rc = batch_ensure_buffers(ws, plan);
	if (rc != VVM_OK) return rc;
	if (plan.total_counts) HIP_TRY(hipMemsetAsync(ws.d_counts, 0, plan.total_counts * 4, ws.stream));
	std::vector<uint8_t> scratch;
	double fill_us = 0.0;
	for (int sl = 0; sl < num_slices; ++sl) {
		const BatchSlice& slice = plan.slices[(size_t)sl];
		const auto t_slice = now();
		for (const BatchGroup& grp : slice.groups)
			for (int i : grp.members) {
				rc = batch_fill_exec(ws, plan, ctxs, opts, i, scratch);
				if (rc != VVM_OK) { (void)hipStreamSynchronize(ws.upload_stream); (void)hipStreamSynchronize(ws.stream); return rc; }
			}
		fill_us += us(t_slice, now());
		rc = batch_launch_slice(ws, plan, sl, slice.used_bytes);
		if (rc != VVM_OK) return rc;
	}
	rc = batch_enqueue_singles(ws, plan, ctxs, opts);
	if (rc != VVM_OK) return rc;
	const auto t_enqueued = now();
	double wait_us = 0.0;
	rc = batch_collect(ws, plan, ctxs, &wait_us);
	if (report)
		fprintf(stderr, "vvm batch host phases (us): prepare %.0f  fill %.0f (of %.0f in %d slices, launches in between)  wait %.0f, write-back %.0f  (%d Execs, %zu parameter bytes)\n",
		        us(t_begin, t_prepared), fill_us, us(t_prepared, t_enqueued), num_slices, wait_us, us(t_enqueued, now()) - wait_us, n, plan.total_bytes);
	return rc;
}

VVM_API uint64_t vvm_debug_batch_plans_made(void)
{
	std::lock_guard<std::mutex> lock(g_batch.mutex);
	return g_batch_cache.plans_made;
}

// --- a batch that is kept from tick to tick ------------------------------------------------------------
struct vvm_batch {
	BatchWorkspace ws;
	BatchPlan plan;
	vvm_exec_context* ctxs = nullptr;
	std::vector<vvm_exec_options> opts;
	bool have_opts = false;
	std::vector<int32_t> last_instances;
	bool uploaded = false;
};

namespace {
int batch_handle_rebuild(vvm_batch* b)
{
	BatchWorkspace& ws = b->ws;
	const vvm_exec_options* opts = b->have_opts ? b->opts.data() : nullptr;
	// one slice: in steady state nothing is filled per call, so there is no host work to hide behind earlier kernels, and
	// one launch per kernel shape leaves the fewest launch tails
	int rc = batch_plan(ws, b->ctxs, b->plan.n ? b->plan.n : (int)b->last_instances.size(), opts, 1, b->plan);
	if (rc != VVM_OK) return rc;
	if (ws.stream) HIP_TRY(hipStreamSynchronize(ws.stream));
	rc = batch_ensure_buffers(ws, b->plan);
	if (rc != VVM_OK) return rc;
	std::vector<uint8_t> scratch;
	for (int i = 0; i < b->plan.n; ++i) {
		b->last_instances[(size_t)i] = b->ctxs[i].NumInstances;
		if (b->plan.single[(size_t)i]) continue;
		rc = batch_fill_exec(ws, b->plan, b->ctxs, opts, i, scratch);
		if (rc != VVM_OK) return rc;
	}
	b->uploaded = false;
	return VVM_OK;
}
}  // namespace

VVM_API vvm_batch* vvm_batch_create(vvm_exec_context* ctxs, int32_t n, const vvm_exec_options* opts)
{
	if (n <= 0 || !ctxs) { fail(VVM_E_INVALID_ARG, "bad batch"); return nullptr; }
	vvm_batch* b = new vvm_batch();
	b->ctxs = ctxs;
	if (opts) { b->opts.assign(opts, opts + n); b->have_opts = true; }
	b->last_instances.assign((size_t)n, 0);
	b->plan.n = n;
	int rc = batch_bind_device(b->ws);
	if (rc == VVM_OK) rc = batch_handle_rebuild(b);
	if (rc != VVM_OK) { b->ws.release(); delete b; return nullptr; }
	return b;
}

VVM_API void vvm_batch_destroy(vvm_batch* b)
{
	if (!b) return;
	b->ws.release();
	delete b;
}

VVM_API int vvm_batch_exec(vvm_batch* b, uint32_t changed)
{
	if (!b) return fail(VVM_E_INVALID_ARG, "null batch");
	BatchWorkspace& ws = b->ws;
	std::lock_guard<std::mutex> lock(ws.mutex);
	int rc = batch_bind_device(ws);
	if (rc != VVM_OK) return rc;
	BatchPlan& plan = b->plan;
	const vvm_exec_options* opts = b->have_opts ? b->opts.data() : nullptr;
	rc = batch_settle_states(plan, b->ctxs);
	if (rc != VVM_OK) return rc;
	size_t upload = b->uploaded ? 0 : plan.slices[0].used_bytes;
	if (changed & VVM_BATCH_CHANGED_INSTANCES) {
		// Sizes: the Execs whose NumInstances differs are prepared again.  A new size may move an Exec to another kernel (the
		// eight-wide one from VVM_EIGHT_WIDE_MIN_INSTANCES on) or change the size of nothing but its tile count; in the first case
		// the whole batch is planned again, in the second the descriptors get their new tile numbers.
		bool any = false, replan = false;
		for (int i = 0; i < plan.n && !replan; ++i) {
			if (b->ctxs[i].NumInstances == b->last_instances[(size_t)i]) continue;
			any = true;
			if (plan.single[(size_t)i]) { b->last_instances[(size_t)i] = b->ctxs[i].NumInstances; continue; }
			PreparedExec pe;
			pe.in_batch = true;
			rc = prepare(b->ctxs[i].VVMState, &b->ctxs[i], opts ? &opts[i] : nullptr, pe);
			if (rc != VVM_OK) return rc;
			if (kernel_key(pe) != kernel_key(plan.pes[(size_t)i]) || pe.bytes != plan.pes[(size_t)i].bytes) replan = true;
			else { plan.pes[(size_t)i] = pe; b->last_instances[(size_t)i] = b->ctxs[i].NumInstances; }
		}
		if (replan) {
			rc = batch_handle_rebuild(b);
			if (rc != VVM_OK) return rc;
			upload = plan.slices[0].used_bytes;
			changed &= ~(uint32_t)(VVM_BATCH_CHANGED_STREAMS | VVM_BATCH_CHANGED_CONSTANTS);      // (everything has just been filled)
		} else if (any) {
			batch_layout(plan, b->ctxs);
			if (plan.total_tiles > ws.status_capacity) {
				HIP_TRY(hipStreamSynchronize(ws.stream));
				rc = batch_ensure_buffers(ws, plan);
				if (rc != VVM_OK) return rc;
			}
			for (int i = 0; i < plan.n; ++i) {
				if (plan.single[(size_t)i]) continue;
				VvmExecDesc* d = (VvmExecDesc*)(ws.host + plan.desc_off[(size_t)i]);
				d->num_instances = (uint32_t)b->ctxs[i].NumInstances;
				d->num_tiles = plan.pes[(size_t)i].tiles;
				d->first_tile = plan.first_tile[(size_t)i];
				d->status = ws.d_status + plan.status_off[(size_t)i];
			}
			upload = std::max(upload, plan.slices[0].desc_bytes);
			if (plan.any_lean) {
				// the lean members' ride words and copy-kernel blocks move with the look-back words: those Execs are filled again
				if (ws.ride_capacity < plan.total_tiles) {
					HIP_TRY(hipStreamSynchronize(ws.stream));
					rc = batch_ensure_buffers(ws, plan);
					if (rc != VVM_OK) return rc;
				}
				std::vector<uint8_t> scratch;
				for (int i : plan.slices[0].through_members) {
					rc = batch_fill_exec(ws, plan, b->ctxs, opts, i, scratch);
					if (rc != VVM_OK) return rc;
				}
				upload = plan.slices[0].used_bytes;
			}
		}
	}
	if (changed & VVM_BATCH_CHANGED_STREAMS) {
		std::vector<uint8_t> scratch;
		for (int i = 0; i < plan.n; ++i) {
			if (plan.single[(size_t)i]) continue;
			rc = batch_fill_exec(ws, plan, b->ctxs, opts, i, scratch);
			if (rc != VVM_OK) return rc;
		}
		upload = plan.slices[0].used_bytes;
	} else if (changed & VVM_BATCH_CHANGED_CONSTANTS) {
		for (int i = 0; i < plan.n; ++i) {
			if (plan.single[(size_t)i]) continue;
			rc = fill_constants(b->ctxs[i].VVMState, &b->ctxs[i], (uint32_t*)(ws.host + plan.const_off[(size_t)i]), *plan.pes[(size_t)i].lp);
			if (rc != VVM_OK) return rc;
		}
		upload = std::max(upload, plan.slices[0].head_bytes);
	}
	if (plan.total_counts) HIP_TRY(hipMemsetAsync(ws.d_counts, 0, plan.total_counts * 4, ws.stream));
	rc = batch_launch_slice(ws, plan, 0, upload);
	if (rc != VVM_OK) return rc;
	b->uploaded = true;
	rc = batch_enqueue_singles(ws, plan, b->ctxs, opts);
	if (rc != VVM_OK) return rc;
	return batch_collect(ws, plan, b->ctxs, nullptr);
}

// --- introspection / helpers ------------------------------------------------------------------------
VVM_API int32_t vvm_state_last_exec_width(const vvm_state* st) { return st ? (int32_t)st->last_width : 0; }
VVM_API const int32_t* vvm_state_device_counts(const vvm_state* st) { return st ? (st->counts_src ? st->counts_src : st->d_counts) : nullptr; }
VVM_API int32_t vvm_state_max_output_dataset(const vvm_state* st) { return st ? (int32_t)st->script.max_output_dataset : 0; }

static void fill_info(const vvm::ThawedScript& s, const vvm::LinkedProgram& lp, vvm_program_info* out)
{
	out->NumTempRegisters = s.num_temps;
	out->NumConstBuffers = s.num_consts;
	out->NumInputBuffers = s.num_inputs;
	out->NumOutputBuffers = s.num_output_instructions;
	out->NumInstructions = (uint32_t)lp.code.size();
	out->NumVirtualRegisters = lp.num_vregs;
	out->InstancesPerLane = lp.instances_per_lane;
	out->LdsSpillRegisters = lp.lds_spill_slots;
	out->BytesInPerInstance = lp.bytes_in_per_instance;
	out->BytesOutPerInstance = lp.bytes_out_per_instance;
	out->Flags = s.flags | (lp.full_kernel ? 0x10000u : 0u) | (lp.num_chains > 1 ? 0x40000u : 0u) | std::min<uint32_t>(lp.num_passes, 15u) << 20;      // (vvm_state_info adds 0x20000: a program for the light eight-wide kernel exists; 0x40000: two acquired datasets in the first pass; [20:24): passes, i.e. launches, per Exec)
}

VVM_API int vvm_state_info(const vvm_state* st, vvm_program_info* out)
{
	if (!st || !out) return fail(VVM_E_INVALID_ARG, "null argument");
	fill_info(st->script, st->wide, out);
	if (st->light_ready) out->Flags |= 0x20000u;
	if (st->lean_ready) out->Flags |= 0x80000u | ((uint32_t)std::min<size_t>(st->through.size(), 255) << 24);
	return VVM_OK;
}

VVM_API int vvm_inspect_blob(const uint8_t* blob, size_t num_bytes, vvm_program_info* out)
{
	if (!out) return fail(VVM_E_INVALID_ARG, "null argument");
	vvm::ThawedScript s;
	vvm::LinkedProgram lp;
	std::string err;
	int rc = vvm::thaw_blob(blob, num_bytes, s, err);
	if (rc == VVM_OK) rc = vvm::link_program(s, false, lp, err);
	if (rc != VVM_OK) return fail(rc, err);
	fill_info(s, lp, out);
	// (the flags vvm_state_info adds for the state's other programs)
	vvm::LinkedProgram other;
	std::string why;
	const bool fits_light = lp.instances_per_lane == 4 && !lp.full_kernel && !lp.many_registers && lp.num_chains == 1 && !lp.uses_ids &&
	                        vvm::link_program(s, false, other, why, 0, true, nullptr, 0) == VVM_OK;      // (the first pass's, as vvm_alloc_state links it)
	if (fits_light) out->Flags |= 0x20000u;
	std::vector<uint32_t> through;
	if (plan_through_programs(s, lp, fits_light, other, through)) out->Flags |= 0x80000u | ((uint32_t)std::min<size_t>(through.size(), 255) << 24);
	return VVM_OK;
}

VVM_API int vvm_inspect_code(const uint8_t* blob, size_t num_bytes, int narrow, uint32_t* out_words, int32_t capacity_instructions)
{
	vvm::ThawedScript s;
	vvm::LinkedProgram lp;
	std::string err;
	int rc = vvm::thaw_blob(blob, num_bytes, s, err);
	if (rc == VVM_OK) rc = (narrow == 2) ? vvm::link_program(s, false, lp, err, 0, true) : vvm::link_program(s, narrow != 0, lp, err);
	if (rc != VVM_OK) { fail(rc, err); return -rc; }
	const int32_t n = (int32_t)lp.code.size();
	if (out_words && capacity_instructions > 0)
		memcpy(out_words, lp.code.data(), sizeof(VvmIns) * (size_t)std::min(n, capacity_instructions));
	return n;
}

VVM_API int vvm_inspect_device_code(const uint8_t* blob, size_t num_bytes, int narrow, uint32_t* out_words, int32_t capacity_instructions)
{
	vvm::ThawedScript s;
	vvm::LinkedProgram lp;
	std::string err;
	int rc = vvm::thaw_blob(blob, num_bytes, s, err);
	if (rc == VVM_OK) rc = (narrow == 2) ? vvm::link_program(s, false, lp, err, 0, true) : vvm::link_program(s, narrow != 0, lp, err);
	if (rc != VVM_OK) { fail(rc, err); return -rc; }
	std::vector<VvmDevIns> dev;
	encode_device_program(lp, dev);
	const int32_t n = (int32_t)dev.size();
	if (out_words && capacity_instructions > 0)
		memcpy(out_words, dev.data(), sizeof(VvmDevIns) * (size_t)std::min(n, capacity_instructions));
	return n;
}

VVM_API void* vvm_device_alloc(size_t num_bytes)
{
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, num_bytes ? num_bytes : 1);
	if (e != hipSuccess) { hip_fail(e, "hipMalloc"); return nullptr; }
	return p;
}
VVM_API void vvm_device_free(void* p) { if (p) (void)hipFree(p); }
VVM_API int vvm_device_upload(void* dst, const void* src, size_t n) { HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); return VVM_OK; }
VVM_API int vvm_device_download(void* dst, const void* src, size_t n) { HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); return VVM_OK; }
VVM_API int vvm_device_memset(void* dst, int v, size_t n) { HIP_TRY(memset_and_wait(dst, v, n)); return VVM_OK; }
VVM_API int vvm_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return VVM_OK; }
VVM_API int vvm_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
VVM_API int vvm_set_device(int ordinal) { HIP_TRY(hipSetDevice(ordinal)); return VVM_OK; }

VVM_API int vvm_state_enable_timing(vvm_state* st, int enable)
{
	if (!st) return fail(VVM_E_INVALID_ARG, "null state");
	st->timing = enable != 0;
	st->timing_span = enable == 2;
	st->ev_used = 0;
	st->span_open = false;
	st->span_closed = false;
	st->span_launches = 0;
	return VVM_OK;
}

VVM_API int vvm_state_collect_kernel_ms(vvm_state* st, float* sum_ms, int32_t* launches)
{
	if (!st || !sum_ms || !launches) return fail(VVM_E_INVALID_ARG, "null argument");
	float total = 0.0f;
	if (st->timing_span) {
		*sum_ms = 0.0f;
		*launches = 0;
		if (st->span_open && st->last_stream) {
			if (!st->span_closed) HIP_TRY(hipEventRecord(st->ev_stop[0], st->last_stream));
			HIP_TRY(hipEventSynchronize(st->ev_stop[0]));
			HIP_TRY(hipEventElapsedTime(sum_ms, st->ev_start[0], st->ev_stop[0]));
			*launches = (int32_t)st->span_launches;
		}
		st->span_open = false;
		st->span_closed = false;
		st->span_launches = 0;
		return VVM_OK;
	}
	for (size_t i = 0; i < st->ev_used; ++i) {
		HIP_TRY(hipEventSynchronize(st->ev_stop[i]));
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, st->ev_start[i], st->ev_stop[i]));
		total += ms;
	}
	*sum_ms = total;
	*launches = (int32_t)st->ev_used;
	st->ev_used = 0;
	return VVM_OK;
}

VVM_API int vvm_state_copy_counts(vvm_state* st, void* dst_device, void* stream)
{
	if (!st || !dst_device) return fail(VVM_E_INVALID_ARG, "null argument");
	const size_t ncounts = std::max<uint32_t>(st->script.max_output_dataset, 1);
	HIP_TRY(hipMemcpyAsync(dst_device, st->counts_src ? st->counts_src : st->d_counts, ncounts * 4, hipMemcpyDeviceToDevice, stream ? (hipStream_t)stream : st->stream));
	return VVM_OK;
}

}  // extern "C"

// --- instance ranges sharded over several GPUs -----------------------------------------------------------
// One process per GPU; every rank runs one Exec over its contiguous range of instances, and the ranks exchange their survivor
// counts (4 bytes per dataset and rank): the global output position of rank r's survivors is the sum of the counts of the ranks
// before it, because concatenating stable compactions of contiguous ranges in range order is the stable compaction of the
// whole (VectorVMRuntime.cpp:786-804 keeps instance order).  The payload never moves.  The reference has no counterpart: its
// Exec is one thread in one address space.
#include <dlfcn.h>

namespace {
struct vvm_nccl_id { char internal[128]; };      // ncclUniqueId
// RCCL is loaded on demand and by name: a process that has PyTorch loaded binds to the copy PyTorch brought (same soname),
// any other process to the system's; this library links against neither.
struct RcclApi {
	void* handle = nullptr;
	int (*get_unique_id)(vvm_nccl_id*) = nullptr;
	int (*comm_init_rank)(void**, int, vvm_nccl_id, int) = nullptr;
	int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
	int (*comm_destroy)(void*) = nullptr;
	int (*comm_count)(const void*, int*) = nullptr;
	const char* (*get_error_string)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (g_rccl.handle) return VVM_OK;
	void* h = nullptr;
	for (const char* name : {"librccl.so.1", "librccl.so"}) {
		h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (h) break;
	}
	if (!h) return fail(VVM_E_RCCL, std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?"));
	RcclApi api;
	api.handle = h;
	api.get_unique_id = (int (*)(vvm_nccl_id*))dlsym(h, "ncclGetUniqueId");
	api.comm_init_rank = (int (*)(void**, int, vvm_nccl_id, int))dlsym(h, "ncclCommInitRank");
	api.all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
	api.comm_destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
	api.comm_count = (int (*)(const void*, int*))dlsym(h, "ncclCommCount");
	api.get_error_string = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
	if (!api.get_unique_id || !api.comm_init_rank || !api.all_gather || !api.comm_destroy) return fail(VVM_E_RCCL, "librccl.so lacks an entry point this library needs");
	g_rccl = api;
	return VVM_OK;
}
int rccl_fail(int code, const char* where)
{
	return fail(VVM_E_RCCL, std::string(where) + ": " + (g_rccl.get_error_string ? g_rccl.get_error_string(code) : "error") );
}
constexpr int kCommSlots = 4;       // exchanges that may be pending at once
constexpr int kNcclInt32 = 2;       // ncclInt32
}  // namespace

struct vvm_comm {
	int32_t rank = 0, world = 1;
	int device = -1;
	void* nccl = nullptr;                    // ncclComm_t, or null for the callback forms
	vvm_all_gather_fn fn = nullptr; void* user = nullptr; int32_t where = VVM_COMM_BUFFERS_HOST;
	hipStream_t stream = nullptr;            // the exchanges run here, beside the next Exec
	int32_t* d_send = nullptr;               // [slot][VVM_MAX_SHARD_DATASETS]
	int32_t* d_recv = nullptr;               // [slot][world][VVM_MAX_SHARD_DATASETS]
	int32_t* h_send = nullptr; int32_t* h_recv = nullptr;      // pinned twins
	hipEvent_t counted[kCommSlots] = {}, exchanged[kCommSlots] = {};
	int32_t slot_datasets[kCommSlots] = {};
	uint32_t slot_acquired[kCommSlots] = {};  // bit d: the slot's script runs acquireindex for dataset d (only those words are written by its kernel)
	uint64_t issued = 0, waited = 0;          // exchanges enqueued / handed out so far
};

namespace {
vvm_comm* comm_create_common(int32_t rank, int32_t world)
{
	if (world < 1 || rank < 0 || rank >= world) { fail(VVM_E_INVALID_ARG, "bad rank / world"); return nullptr; }
	if (ensure_device_info() != VVM_OK) return nullptr;
	vvm_comm* c = new vvm_comm();
	c->rank = rank; c->world = world;
	auto bail = [&](hipError_t e, const char* where) -> vvm_comm* { hip_fail(e, where); vvm_comm_destroy(c); return nullptr; };
	hipError_t e;
	if ((e = hipGetDevice(&c->device)) != hipSuccess) return bail(e, "hipGetDevice");
	const size_t send_words = (size_t)kCommSlots * VVM_MAX_SHARD_DATASETS, recv_words = send_words * (size_t)world;
	if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
	if ((e = hipMalloc((void**)&c->d_send, send_words * 4)) != hipSuccess) return bail(e, "hipMalloc");
	if ((e = hipMalloc((void**)&c->d_recv, recv_words * 4)) != hipSuccess) return bail(e, "hipMalloc");
	if ((e = memset_and_wait(c->d_send, 0, send_words * 4)) != hipSuccess) return bail(e, "hipMemset");
	if ((e = memset_and_wait(c->d_recv, 0, recv_words * 4)) != hipSuccess) return bail(e, "hipMemset");
	if ((e = hipHostMalloc((void**)&c->h_send, send_words * 4, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
	if ((e = hipHostMalloc((void**)&c->h_recv, recv_words * 4, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
	for (int k = 0; k < kCommSlots; ++k) {
		if ((e = hipEventCreateWithFlags(&c->counted[k], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
		if ((e = hipEventCreateWithFlags(&c->exchanged[k], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
	}
	return c;
}
}  // namespace

extern "C" {

VVM_API int vvm_shard_offsets(int32_t rank, int32_t world, int32_t num_datasets, const int32_t* all_counts, vvm_shard_result* out)
{
	if (!all_counts || !out || world < 1 || rank < 0 || rank >= world || num_datasets < 0 || num_datasets > VVM_MAX_SHARD_DATASETS)
		return fail(VVM_E_INVALID_ARG, "bad arguments to vvm_shard_offsets");
	memset(out, 0, sizeof(*out));
	out->NumDataSets = num_datasets;
	for (int32_t d = 0; d < num_datasets; ++d) {
		int64_t before = 0, total = 0;
		for (int32_t r = 0; r < world; ++r) {
			const int32_t c = all_counts[(size_t)r * (size_t)num_datasets + (size_t)d];
			if (c < 0) return fail(VVM_E_INVALID_ARG, "negative survivor count");
			if (r < rank) before += c;
			total += c;
		}
		out->LocalSurvivors[d] = all_counts[(size_t)rank * (size_t)num_datasets + (size_t)d];
		out->GlobalOffset[d] = before;
		out->GlobalSurvivors[d] = total;
	}
	return VVM_OK;
}

VVM_API int vvm_comm_unique_id(uint8_t* out_128_bytes)
{
	if (!out_128_bytes) return fail(VVM_E_INVALID_ARG, "null argument");
	int rc = load_rccl();
	if (rc != VVM_OK) return rc;
	vvm_nccl_id id;
	memset(&id, 0, sizeof id);
	const int n = g_rccl.get_unique_id(&id);
	if (n != 0) return rccl_fail(n, "ncclGetUniqueId");
	memcpy(out_128_bytes, &id, sizeof id);
	return VVM_OK;
}

VVM_API vvm_comm* vvm_comm_create_rccl(int32_t rank, int32_t world, const uint8_t* unique_id_128_bytes)
{
	if (!unique_id_128_bytes) { fail(VVM_E_INVALID_ARG, "null unique id"); return nullptr; }
	if (load_rccl() != VVM_OK) return nullptr;
	vvm_comm* c = comm_create_common(rank, world);
	if (!c) return nullptr;
	vvm_nccl_id id;
	memcpy(&id, unique_id_128_bytes, sizeof id);
	const int n = g_rccl.comm_init_rank(&c->nccl, world, id, rank);
	if (n != 0) { rccl_fail(n, "ncclCommInitRank"); c->nccl = nullptr; vvm_comm_destroy(c); return nullptr; }
	return c;
}

VVM_API vvm_comm* vvm_comm_create_callback(int32_t rank, int32_t world, vvm_all_gather_fn fn, void* user, int32_t where)
{
	if (!fn || (where != VVM_COMM_BUFFERS_HOST && where != VVM_COMM_BUFFERS_DEVICE)) { fail(VVM_E_INVALID_ARG, "bad all-gather callback"); return nullptr; }
	vvm_comm* c = comm_create_common(rank, world);
	if (!c) return nullptr;
	c->fn = fn; c->user = user; c->where = where;
	return c;
}

VVM_API int32_t vvm_comm_world(const vvm_comm* c)
{
	if (!c) return 0;
	if (c->nccl && g_rccl.comm_count) {
		int n = 0;
		if (g_rccl.comm_count(c->nccl, &n) == 0) return (int32_t)n;
	}
	return c->world;
}

VVM_API void vvm_comm_destroy(vvm_comm* c)
{
	if (!c) return;
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->nccl && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->nccl);
	if (c->d_send) (void)hipFree(c->d_send);
	if (c->d_recv) (void)hipFree(c->d_recv);
	if (c->h_send) (void)hipHostFree(c->h_send);
	if (c->h_recv) (void)hipHostFree(c->h_recv);
	for (int k = 0; k < kCommSlots; ++k) {
		if (c->counted[k]) (void)hipEventDestroy(c->counted[k]);
		if (c->exchanged[k]) (void)hipEventDestroy(c->exchanged[k]);
	}
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

VVM_API int vvm_exec_sharded_async(vvm_exec_context* ctx, const vvm_exec_options* opt, vvm_comm* c)
{
	if (!ctx || !ctx->VVMState || !c) return fail(VVM_E_INVALID_ARG, "null context, state or communicator");
	vvm_state* st = ctx->VVMState;
	if (st->device != c->device) return fail(VVM_E_INVALID_ARG, "the state and the communicator belong to different devices");
	if (st->any_ids) return fail(VVM_E_UNSUPPORTED_OP, "scripts with update_id / acquire_id are not sharded: their ID tables are not coordinated between ranks");
	const int32_t nds = (int32_t)std::max<uint32_t>(st->script.max_output_dataset, 1);
	if (nds > VVM_MAX_SHARD_DATASETS) return fail(VVM_E_TOO_LARGE, "more output datasets than a sharded Exec exchanges counts for");
	if (c->issued - c->waited >= (uint64_t)kCommSlots) return fail(VVM_E_INVALID_ARG, "too many exchanges pending: call vvm_exec_sharded_wait");
	const int k = (int)(c->issued % kCommSlots);
	int32_t* const d_send = c->d_send + (size_t)k * VVM_MAX_SHARD_DATASETS;
	int32_t* const d_recv = c->d_recv + (size_t)k * VVM_MAX_SHARD_DATASETS * (size_t)c->world;
	int32_t* const h_send = c->h_send + (size_t)k * VVM_MAX_SHARD_DATASETS;
	int32_t* const h_recv = c->h_recv + (size_t)k * VVM_MAX_SHARD_DATASETS * (size_t)c->world;
	// the Exec leaves its counts where the exchange reads them
	vvm_exec_options o = opt ? *opt : vvm_exec_options{};
	o.DeviceCounts = d_send;
	int rc = vvm_exec_async(ctx, &o);
	if (rc != VVM_OK) return rc;
	hipStream_t exec_stream = st->last_stream;
	c->slot_datasets[k] = nds;
	// the kernel writes the count words of the datasets the script acquires and no others: the rest of the slot still holds what
	// an earlier Exec (of another script) left there, and is reported as zero
	c->slot_acquired[k] = 0;
	for (int32_t d = 0; d < nds; ++d)
		if ((size_t)d < st->wide.acquires_dataset.size() && st->wide.acquires_dataset[d]) c->slot_acquired[k] |= 1u << d;
	if (c->fn && c->where == VVM_COMM_BUFFERS_HOST) {
		// a host-side exchange (another transport, a test): the counts come to the host first; the call returns when the exchange is done
		HIP_TRY(hipMemcpyAsync(h_send, d_send, (size_t)nds * 4, hipMemcpyDeviceToHost, exec_stream));
		HIP_TRY(hipStreamSynchronize(exec_stream));
		rc = c->fn(c->user, h_send, h_recv, nds, nullptr);
		if (rc != 0) return fail(VVM_E_RCCL, "the all-gather callback reported a failure");
		HIP_TRY(hipEventRecord(c->exchanged[k], c->stream));
	} else {
		HIP_TRY(hipEventRecord(c->counted[k], exec_stream));
		HIP_TRY(hipStreamWaitEvent(c->stream, c->counted[k], 0));
		if (c->fn) {
			rc = c->fn(c->user, d_send, d_recv, nds, (void*)c->stream);
			if (rc != 0) return fail(VVM_E_RCCL, "the all-gather callback reported a failure");
		} else {
			const int n = g_rccl.all_gather(d_send, d_recv, (size_t)nds, kNcclInt32, c->nccl, c->stream);
			if (n != 0) return rccl_fail(n, "ncclAllGather");
		}
		HIP_TRY(hipMemcpyAsync(h_recv, d_recv, (size_t)nds * (size_t)c->world * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipEventRecord(c->exchanged[k], c->stream));
	}
	c->issued++;
	return VVM_OK;
}

VVM_API int vvm_exec_sharded_wait(vvm_comm* c, vvm_shard_result* out)
{
	if (!c || !out) return fail(VVM_E_INVALID_ARG, "null argument");
	if (c->waited == c->issued) return fail(VVM_E_INVALID_ARG, "no exchange is pending");
	const int k = (int)(c->waited % kCommSlots);
	HIP_TRY(hipEventSynchronize(c->exchanged[k]));
	c->waited++;
	int32_t* const all = c->h_recv + (size_t)k * VVM_MAX_SHARD_DATASETS * (size_t)c->world;
	const int32_t nds = c->slot_datasets[k];
	for (int32_t r = 0; r < c->world; ++r)
		for (int32_t d = 0; d < nds; ++d)
			if (!(c->slot_acquired[k] >> d & 1u)) all[(size_t)r * (size_t)nds + (size_t)d] = 0;
	return vvm_shard_offsets(c->rank, c->world, nds, all, out);
}

VVM_API int vvm_exec_sharded(vvm_exec_context* ctx, const vvm_exec_options* opt, vvm_comm* c, vvm_shard_result* out)
{
	if (c && c->issued != c->waited) return fail(VVM_E_INVALID_ARG, "earlier exchanges are still pending: call vvm_exec_sharded_wait first");
	int rc = vvm_exec_sharded_async(ctx, opt, c);
	if (rc != VVM_OK) return rc;
	rc = vvm_exec_finish(ctx);
	if (rc != VVM_OK) return rc;
	return vvm_exec_sharded_wait(c, out);
}

}  // extern "C"
