// vvm_device.h -- structures shared by the host launcher and the gfx950 interpreter kernel.
//
// The optimized bytecode the reference interprets (VectorVMRuntime.cpp:2135-2175) is byte-packed with
// unaligned u16 operands.  vvm_alloc_state re-encodes it once into fixed 16-byte device instructions that
// a wave can fetch with one scalar load; the re-encoding plays the role of the reference's per-Exec
// register pointer table (VectorVMRuntime.cpp:315-443): it resolves every register index to where the
// value lives on the GPU (per-lane register, constant, input stream).
#pragma once
#include <stdint.h>

#ifndef VVM_BLOCK_THREADS
#define VVM_BLOCK_THREADS 256
#endif

// ---- operand = class (3 bits) | index (13 bits) ----
enum : uint32_t {
	VVM_OPC_VREG  = 0,   // per-lane virtual register: index < resident count -> VGPR, else LDS slot
	VVM_OPC_UNIF  = 1,   // uniform register: one of the first VVM_MAX_UNIFORMS remapped constants (lane i of one VGPR holds constant i)
	VVM_OPC_CONST = 2,   // remapped constant read from the constant table (the ones past the uniform registers)
	VVM_OPC_NOADV = 4,   // no-advance input: element [InstanceOffset] of the stream, uniform (VectorVMRuntime.cpp:381-401)
	VVM_OPC_NOADV_HALF = 6,   // the same for a 16-bit stream, converted to float
	                          // (advancing input streams are loaded into per-lane registers by VVM_DOP_LOADN / VVM_DOP_LOADN16)
	VVM_OPC_NONE  = 7    // 0xFFFF in the bytecode: external-function dummy register
};
#define VVM_OPND(cls, idx) ((uint32_t)(((cls) << 13) | ((idx) & 0x1FFFu)))
#define VVM_OPND_CLS(o)    (((o) >> 13) & 7u)
#define VVM_OPND_IDX(o)    ((o) & 0x1FFFu)
#define VVM_MAX_OPND_INDEX 0x1FFFu
#define VVM_MAX_UNIFORMS   64u

// ---- device-only opcodes (the reference's opcodes keep their numbers 0..178) ----
enum : uint32_t {
	VVM_DOP_COPY = 189,    // an ordinary instruction (one operand, one destination): dst = o0, bit for bit
	VVM_DOP_LOADN  = 191,  // load up to 8 32-bit input streams into consecutive virtual registers (program prologue only):
	                       // aux = count, aux2 = first register, w1 = first slot of VvmExecDesc::load_ptrs
	VVM_DOP_LOADN16 = 190, // the same for 16-bit streams: the raw bits, zero-extended
	VVM_DOP_ACQUIRE = 192, // acquireindex: aux = chain (0; 1 = the second dataset of a program that acquires two in one pass), o0 = predicate (keep = its sign bit, aux2 = VVM_ACQ_SIGN), or, with the compare that
	                       // made the predicate folded in, keep = o0 <aux2> o1; every opcode from here on produces no register result
	VVM_DOP_OUTPUT32 = 193, // up to 3 (src,dst) pairs of outputdata_float / outputdata_int32: aux = count, aux2 = chain of the dataset,
	                        // o0..o2 = sources, w3 = the first pair's slot of VvmExecDesc::out_list (the pairs of one
	                        // instruction lie in one group of VVM_STAGE_SLOTS slots)
	VVM_DOP_OUTPUT16 = 194, // up to 3 pairs of outputdata_half (float register -> half stream)
	VVM_DOP_OUTPUT16_FROM16 = 195,
	VVM_DOP_OUTPUT32_FROM16 = 196,
	VVM_DOP_FLUSH = 197,    // between two groups of output slots: write out what the output instructions before it have staged
	VVM_DOP_UPDATE_ID = 198 // update_id (VectorVMRuntime.cpp:1129-1402): o0 = the ID register, w3 = its output-list entry; staged like an
	                        // output (one slot, all instances of the tile), applied to the ID tables when the tile's position is known
};

// acquireindex's aux2: how the keep bit is formed
enum : uint32_t {
	VVM_ACQ_SIGN = 0,
	VVM_ACQ_LT = 1, VVM_ACQ_LE, VVM_ACQ_GT, VVM_ACQ_GE, VVM_ACQ_EQ, VVM_ACQ_NE,          // float compares (VectorVMRuntime.cpp:1942-1947)
	VVM_ACQ_LTI, VVM_ACQ_LEI, VVM_ACQ_GTI, VVM_ACQ_GEI, VVM_ACQ_EQI, VVM_ACQ_NEI       // signed integer compares (:1960-1965)
};

// Output values wait in LDS staging slots until the tile's output position is known; one group of slots is written out at a time.
#define VVM_STAGE_SLOTS 8
#define VVM_MAX_ID_USES 8      // datasets with update_id / acquire_id per script
// The light eight-wide kernel has six staging slots: its programs stage into the first six entries of the output list.  Entries
// six and seven serve 32-bit outputs whose values stay where they are, in per-lane registers VVM_LIGHT_RIDE_REG and the one after
// it, until the next tile has resolved the position (no instruction: the kernel looks at the two entries).  A light program
// writes those two registers only after its acquireindex, and the kernel finishes the previous tile right after that instruction.
#define VVM_LIGHT_GROUP_SLOTS 6
#define VVM_LIGHT_RIDE_REG 8
// ... unless the tile keeps at most this many instances: then they lie compacted in the unused last quarter of the staging slots
// (three slots per output, 512 survivors each)
#define VVM_LIGHT_STASH_MAX 1536u
#define VVM_LIGHT_STASH_SLOTS 3u
// entry of VvmExecDesc::out_list: byte address in the low 48 bits (0 = nothing is written), flags above
#define VVM_OUT_IS16 (1ull << 62)         // the destination stream has 16-bit elements
#define VVM_OUT_GROUP16 (1ull << 61)      // on a group's first entry: some entry of the group has VVM_OUT_IS16 or VVM_OUT_UPDATE_ID
#define VVM_OUT_CHAIN1 (1ull << 59)       // the entry belongs to the program's second chain (the second of two datasets acquired in one pass)
#define VVM_OUT_UPDATE_ID (1ull << 60)    // the slot holds the ID register of an update_id; the address is the Exec's VvmIdDesc
#define VVM_OUT_ADDR_MASK 0x0000FFFFFFFFFFFFull

// Device instruction, 16 bytes:
//   w0: op[0:8) nin[8:11) nout[11:13) aux[16:24) aux2[24:32)
//   w1: o0 | o1 << 16      w2: o2 | o3 << 16      w3: o4 | o5 << 16
// Ordinary ops have exactly one destination: the inputs are o0.. in order and the destination is o5.
// nin = VVM_NIN_SPECIAL marks `done` and the device-only instructions (the interpreter's inner loop dispatches on nin first).
#define VVM_NIN_SPECIAL 7u
struct VvmIns {
	uint32_t w0, w1, w2, w3;
};
#define VVM_INS_W0(op, nin, nout, aux, aux2) \
	((uint32_t)(op) | ((uint32_t)(nin) << 8) | ((uint32_t)(nout) << 11) | ((uint32_t)(aux) << 16) | ((uint32_t)(aux2) << 24))
// The instruction as it lies in device memory, 32 bytes: the linker's four words, then what the hand-written core of the eight-wide
// kernels executes from (vvm_threaded.h; the host fills these when it uploads a program, the other kernels skip them).
struct VvmDevIns {
	VvmIns   ins;
	uint32_t routine;     // byte offset of the instruction's routine in the routine table; 0: none (not an ordinary instruction, or an op / operand kind without one)
	uint32_t opnds;       // operand bytes a | c << 8 | d << 16 | e << 24: 8 x register index, or the lane of a uniform register
	uint32_t opnd_b;      // operand byte b
	uint32_t dst;         // destination byte: 8 x register index
};
#define VVM_INS_STRIDE 32u

// Persistent-ID tables of the dataset an Exec's update_id / acquire_id works on.  Lives in device memory.
// The reference runs an Exec chunk by chunk (VectorVMRuntime.cpp:2218-2261) and two results depend on where the chunks end:
// update_id's MaxUsedID (the maximum over the first NumOutput entries of each chunk's ID register, :1141-1174) and the order of
// acquire_id's spawned-ID list (each chunk's IDs are copied in front of the previous chunk's, :1425-1433).  chunk_instances is
// that chunk size, so both come out the same here.
struct VvmIdDesc {
	int32_t*  id_table;          // IDTable->GetData()
	int32_t*  free_table;        // FreeIDTable->GetData()
	uint32_t* chunk_prefix;      // update_id: [chunk] survivors before the chunk's first instance (for the MaxUsedID pass)
	int32_t*  id_scratch;        // update_id: [instance] the ID register, as the MaxUsedID pass reads it
	int32_t*  max_id;            // update_id: device word that ends up holding MaxUsedID
	const int32_t* survivors;    // update_id: the dataset's survivor count (written by the Exec's last tile)
	uint32_t  id_table_num;      // IDTable->Num()
	uint32_t  free_table_max;    // FreeIDTable->Max()
	int32_t   instance_offset;
	int32_t   num_free_start;    // *NumFreeIDs when the Exec starts
	int32_t   num_spawned_start; // *NumSpawnedIDs when the Exec starts
	int32_t   acquire_tag;       // IDAcquireTag
	uint32_t  chunk_instances;   // MaxInstancesPerChunk of the reference's chunking for this Exec
	uint32_t  num_instances;
	uint32_t  num_chunks;
	int32_t   push_start;        // update_id: where in FreeIDTable the Exec's first freed ID goes (= num_free_start, unless the script also
	                             // takes IDs from this dataset: then the Exec is one chunk and they go where its own IDs came from)
	const int32_t* free_read;    // acquire_id reads FreeIDTable through this (= free_table, unless update_id writes the same entries in
	                             // the same launch: then a copy of them, addressed like the table)
};

// One Exec (one emitter) as the kernel sees it.  Lives in device memory.
struct VvmExecDesc {
	const VvmIns*   prog;
	const uint32_t* consts;        // NumConstBuffers dwords, remapped order, zero-padded to a multiple of 4 dwords
	const uint64_t* in_ptrs;       // byte address of element [InstanceOffset] of every remapped input stream (no-advance operands)
	const uint64_t* load_ptrs;     // the same addresses in load-list order: 8 per load instruction, 64-byte aligned
	const uint64_t* out_list;      // element [InstanceOffset] of every output in staging order (see VVM_OUT_*), in groups of
	                               // VVM_STAGE_SLOTS entries, 64-byte aligned
	unsigned long long* status;    // one look-back word per tile of this Exec
	int32_t*        counts;        // survivor count per output dataset (written by the Exec's last tile)
	const uint8_t*  perlin;        // 512-byte permutation table when the script calls a noise built-in
	uint32_t num_instances;
	uint32_t num_tiles;
	uint32_t first_tile;           // ticket number of this Exec's first tile within the launch
	int32_t  exec_index_base;
	uint32_t rand_seed_lo, rand_seed_hi;
	int32_t  rand_instance_base;
	uint32_t flags;
	uint32_t num_unif;             // constants copied into the uniform register, a multiple of 4, at most VVM_MAX_UNIFORMS
	uint32_t rand_stride;          // rand_words: elements from one draw's words to the next draw's
	const VvmIdDesc* ids;          // persistent-ID tables, one VvmIdDesc per dataset the script's update_id / acquire_id work on (acquire_id's aux2 = which)
	const uint32_t* rand_words;    // null: the random ops use the counter-based generator.  Otherwise the 32 random bits of every draw,
	                               // [ordinal of the draw][instance of the Exec], as the reference's xorwow yields them in its execution
	                               // order (vvm_xorwow_fill_kernel); read by the FULL kernels only
};

enum : uint32_t {
	VVM_DESC_USES_PERLIN = 1u << 0,
	VVM_DESC_RELATIVE    = 1u << 1,    // consts, in_ptrs, load_ptrs, out_list hold byte offsets from the descriptor, not addresses
	VVM_DESC_REGS_OF_4   = 1u << 2     // the program's routine operands count four VGPRs per register (a program of more than sixteen
	                                   // registers, for the four-wide kernel whose file is 44 registers of four instance slots); otherwise
	                                   // eight, which that kernel runs as well: sixteen registers, the upper four slots of each unused
};
// Largest parameter block (descriptor + tables) that is handed to the kernel as a kernel argument instead of through device memory.
#define VVM_INLINE_PARAM_BYTES 1024u

// The reference's random generator (VectorVMRuntime.cpp:445-483): four xorwow generators side by side (one per lane of a loop's
// four instances), stepped once per random draw and loop in execution order: chunk by chunk (:2276-2295), within a chunk
// instruction by instruction, within an instruction loop by loop (a random_2x steps twice per loop, :1093-1094).  So the n-th
// step's output belongs to one (draw, loop), and n follows from the chunking in closed form.  The generator's state moves
// linearly over GF(2) (plus a counter that advances by a constant), so the state after n steps is the start state times the n-th
// power of the step matrix: vvm_xorwow_fill_kernel jumps to the start of a segment of loops with the precomputed powers
// M^(2^b) and steps through the segment from there.
#define VVM_XORWOW_STATE_BITS 160u
#define VVM_XORWOW_JUMP_POWERS 40u        // n < 2^40: 255 draws x 2^29 loops
#define VVM_XORWOW_SEGMENT_LOOPS 64u      // loops one thread steps through
#define VVM_XORWOW_WEYL 362437u           // :480
// jump table: [power b][state bit j][5 words] = column j of M^(2^b) (the state after 2^b steps from the state with only bit j set;
// bit j of the state = bit j % 32 of word j / 32, word w = State[w] of the reference)
#define VVM_XORWOW_JUMP_WORDS (VVM_XORWOW_JUMP_POWERS * VVM_XORWOW_STATE_BITS * 5u)
struct VvmXorwowFill {
	uint32_t state[24];          // State[0..4][lane 0..3], Counters[lane 0..3] when the Exec starts
	uint32_t total_loops;        // loops of the whole Exec (four instances each)
	uint32_t chunk_loops;        // loops of every chunk but the last (MaxInstancesPerChunk / 4)
	uint32_t num_draws;          // draws per loop of the whole script
	uint32_t stride;             // elements between the words of two draws
	uint32_t num_instructions;   // random instructions
	uint32_t jump_powers;        // powers needed for the largest n of this Exec
	uint32_t pad[2];
	uint8_t  first[256];         // per random instruction: ordinal of its first draw
	uint8_t  width[256];         // ... and its draws per loop (1, or 2 for random_2x)
};

// Streams that a script hands through unchanged can stay out of its program (see vvm_host.cpp, plan_through): the program that
// runs on an eight-wide kernel then leaves, per tile and lane, the word the light kernel keeps of the lane's acquireindex result
// (keep bits [0:8) of the lane's eight instances, rank within the tile of its first survivor of the first half [8:20) and of the
// second half [20:32)) at ride[tile * VVM_BLOCK_THREADS + lane] (the entry three before the Exec's output list holds the
// address, 0 = none), and vvm_through_kernel moves the streams behind it: from the look-back words, which all hold inclusive
// counts once the Exec's kernel has ended, it takes where a tile's survivors start, from the ride words where each goes.
#define VVM_THROUGH_MAX_PAIRS 56u
#define VVM_THROUGH_IS16 (1ull << 63)
// (pairs one workgroup moves: measured on 24 / 40 streams each way at 32 Mi instances, all surviving / half dying, with 1, 2, 3, 4, 6
// pairs: 1.12 / 1.84 / 0.85 / 1.38 ms with one, 1.17 / 1.96 / 0.95 / 1.56 ms with four: the more workgroups in flight the better)
#ifndef VVM_THROUGH_PAIRS_PER_BLOCK
#define VVM_THROUGH_PAIRS_PER_BLOCK 1u
#endif
struct VvmThroughArgs {
	const unsigned long long* status;   // the Exec's look-back words
	const uint32_t* ride;
	uint32_t num_instances, num_tiles, num_pairs;
	uint32_t pad;                       // vvm_through_batch_kernel: the number of the Exec's first tile within the launch
	uint32_t instances_per_lane;        // of the kernel that left the ride words: 8 (as above), or 4: tiles of 1024 instances, keep bits [0:4),
	                                    // rank of the lane's first survivor [8:20) (vvm_through_kernel only)
	uint32_t reserved[3];
	uint64_t src[VVM_THROUGH_MAX_PAIRS];   // byte address of element [InstanceOffset] of the input stream
	uint64_t dst[VVM_THROUGH_MAX_PAIRS];   // ... of the output stream; VVM_THROUGH_IS16: both streams have 16-bit elements
};

// Launch-wide words in device memory; the last block to leave a launch re-arms them for the next one.
//
// Tiles are handed out in index order by ticket counters.  A device-scope atomic on ONE address completes about every 13 ns
// on MI355X (65536 tiles would take 0.85 ms for the tickets alone), so there are VVM_TICKET_CLASSES counters, each on a cache
// line of its own: workgroup b draws from counter (b + b / 16) % classes (launches of fewer than 32 workgroups: b % classes), and
// draw c of counter x is tile c * classes + x.  Tiles still start in (nearly) index order, which the look-back relies on, and every
// tile is claimed by a running workgroup.  (Why not b % classes throughout: see where vvm_exec_kernel works out its class.)
#define VVM_TICKET_CLASSES      16u
#define VVM_TICKET_STRIDE_BYTES 256u
struct VvmLaunchState {
	uint32_t unused;
	uint32_t done;       // blocks that have left
	uint32_t epoch;      // tag that makes a look-back word valid for this launch (never 0)
	uint32_t pad;        // error word: set when a look-back gave up
	// profiling aid, filled only when the launcher's debug bit 7 is set: look-backs, rounds, polls, clocks/16 spent in them,
	// clocks/16 waiting for tickets, in the load phase, in the interpreter, tiles
	unsigned long long stats[8];
	uint32_t reserved[(VVM_TICKET_STRIDE_BYTES - 80u) / 4u];
	uint32_t tickets[VVM_TICKET_CLASSES * VVM_TICKET_STRIDE_BYTES / 4u];   // counter x at tickets[x * VVM_TICKET_STRIDE_BYTES / 4]
};

// look-back word: tag[34:64) | flag[32:34) | value[0:32)
#define VVM_MAX_LAUNCH_TAG 0x3FFFFFFFu
#define VVM_LB_AGGREGATE 1ull
#define VVM_LB_INCLUSIVE 2ull
