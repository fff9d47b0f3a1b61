/*
 * vvm_abi.h -- C ABI of the MI355X VectorVM executor (libvvm_hip.so).
 *
 * Drop-in boundary for the reference entry points
 *   VectorVM::Runtime::AllocVectorVMState / FreeVectorVMState / ExecVectorVMState
 *   (Runtime/VectorVM/Public/VectorVMRuntime.h:30-33).
 * Struct fields keep the order and meaning of FVectorVMExecContext
 * (VectorVMRuntime.h:10-28) and FDataSetMeta (VectorVM.h:246-273); every
 * TArrayView<T> is flattened to {T* data; int32 num;} (ArrayView.h:826-827) and
 * every TArray<int32>* to a pointer to {int32* data; int32 num; int32 max;}.
 *
 * Memory spaces for libvvm_hip.so:
 *   - everything the structs themselves hold (pointer arrays, constant tables,
 *     counters) is HOST memory;
 *   - every register stream (InputRegisters.data[i], OutputRegisters.data[i])
 *     and every ID table payload (vvm_id_table.data) is DEVICE memory.
 * The CPU oracle (oracle/) uses the same structs with host memory throughout.
 */
#ifndef VVM_ABI_H
#define VVM_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define VVM_API
#else
#define VVM_API __attribute__((visibility("default")))
#endif

/* ---- status codes (the reference returns void and asserts; see RT.cpp:2184-2187, 2264-2267) ---- */
enum {
	VVM_OK                  = 0,
	VVM_E_INVALID_ARG       = 1,
	VVM_E_EXTFN_MISMATCH    = 2,  /* ExtFunctionTable.num != NumExtFns: the reference returns silently (RT.cpp:2184) */
	VVM_E_NO_BYTECODE       = 3,  /* RT.cpp:2264 */
	VVM_E_UNSUPPORTED_OP    = 4,
	VVM_E_BAD_BLOB          = 5,
	VVM_E_HIP               = 6,
	VVM_E_RCCL              = 7,  /* the count exchange of a sharded Exec failed (RCCL, or the caller's all-gather callback) */
	VVM_E_TOO_LARGE         = 8
};

/* ---- opcodes: numeric values of EVectorVMOp (VectorVM.h:50-240) ---- */
enum {
	VVM_OP_done = 0, VVM_OP_add = 1, VVM_OP_sub = 2, VVM_OP_mul = 3, VVM_OP_div = 4, VVM_OP_mad = 5, VVM_OP_lerp = 6,
	VVM_OP_rcp = 7, VVM_OP_rsq = 8, VVM_OP_sqrt = 9, VVM_OP_neg = 10, VVM_OP_abs = 11, VVM_OP_exp = 12, VVM_OP_exp2 = 13,
	VVM_OP_log = 14, VVM_OP_log2 = 15, VVM_OP_sin = 16, VVM_OP_cos = 17, VVM_OP_tan = 18, VVM_OP_asin = 19, VVM_OP_acos = 20,
	VVM_OP_atan = 21, VVM_OP_atan2 = 22, VVM_OP_ceil = 23, VVM_OP_floor = 24, VVM_OP_fmod = 25, VVM_OP_frac = 26,
	VVM_OP_trunc = 27, VVM_OP_clamp = 28, VVM_OP_min = 29, VVM_OP_max = 30, VVM_OP_pow = 31, VVM_OP_round = 32,
	VVM_OP_sign = 33, VVM_OP_step = 34, VVM_OP_random = 35, VVM_OP_noise = 36, VVM_OP_cmplt = 37, VVM_OP_cmple = 38,
	VVM_OP_cmpgt = 39, VVM_OP_cmpge = 40, VVM_OP_cmpeq = 41, VVM_OP_cmpneq = 42, VVM_OP_select = 43, VVM_OP_addi = 44,
	VVM_OP_subi = 45, VVM_OP_muli = 46, VVM_OP_divi = 47, VVM_OP_clampi = 48, VVM_OP_mini = 49, VVM_OP_maxi = 50,
	VVM_OP_absi = 51, VVM_OP_negi = 52, VVM_OP_signi = 53, VVM_OP_randomi = 54, VVM_OP_cmplti = 55, VVM_OP_cmplei = 56,
	VVM_OP_cmpgti = 57, VVM_OP_cmpgei = 58, VVM_OP_cmpeqi = 59, VVM_OP_cmpneqi = 60, VVM_OP_bit_and = 61, VVM_OP_bit_or = 62,
	VVM_OP_bit_xor = 63, VVM_OP_bit_not = 64, VVM_OP_bit_lshift = 65, VVM_OP_bit_rshift = 66, VVM_OP_logic_and = 67,
	VVM_OP_logic_or = 68, VVM_OP_logic_xor = 69, VVM_OP_logic_not = 70, VVM_OP_f2i = 71, VVM_OP_i2f = 72, VVM_OP_f2b = 73,
	VVM_OP_b2f = 74, VVM_OP_i2b = 75, VVM_OP_b2i = 76, VVM_OP_inputdata_float = 77, VVM_OP_inputdata_int32 = 78,
	VVM_OP_inputdata_half = 79, VVM_OP_inputdata_noadvance_float = 80, VVM_OP_inputdata_noadvance_int32 = 81,
	VVM_OP_inputdata_noadvance_half = 82, VVM_OP_outputdata_float = 83, VVM_OP_outputdata_int32 = 84,
	VVM_OP_outputdata_half = 85, VVM_OP_acquireindex = 86, VVM_OP_external_func_call = 87, VVM_OP_exec_index = 88,
	VVM_OP_noise2D = 89, VVM_OP_noise3D = 90, VVM_OP_enter_stat_scope = 91, VVM_OP_exit_stat_scope = 92,
	VVM_OP_update_id = 93, VVM_OP_acquire_id = 94, VVM_OP_half_to_float = 95, VVM_OP_fasi = 96, VVM_OP_iasf = 97,
	/* merged ops */
	VVM_OP_exec_indexf = 98, VVM_OP_exec_index_addi = 99, VVM_OP_cmplt_select = 100, VVM_OP_cmple_select = 101,
	VVM_OP_cmpeq_select = 102, VVM_OP_cmplti_select = 103, VVM_OP_cmplei_select = 104, VVM_OP_cmpeqi_select = 105,
	VVM_OP_cmplt_logic_and = 106, VVM_OP_cmple_logic_and = 107, VVM_OP_cmpgt_logic_and = 108, VVM_OP_cmpge_logic_and = 109,
	VVM_OP_cmpeq_logic_and = 110, VVM_OP_cmpne_logic_and = 111, VVM_OP_cmplti_logic_and = 112, VVM_OP_cmplei_logic_and = 113,
	VVM_OP_cmpgti_logic_and = 114, VVM_OP_cmpgei_logic_and = 115, VVM_OP_cmpeqi_logic_and = 116, VVM_OP_cmpnei_logic_and = 117,
	VVM_OP_cmplt_logic_or = 118, VVM_OP_cmple_logic_or = 119, VVM_OP_cmpgt_logic_or = 120, VVM_OP_cmpge_logic_or = 121,
	VVM_OP_cmpeq_logic_or = 122, VVM_OP_cmpne_logic_or = 123, VVM_OP_cmplti_logic_or = 124, VVM_OP_cmplei_logic_or = 125,
	VVM_OP_cmpgti_logic_or = 126, VVM_OP_cmpgei_logic_or = 127, VVM_OP_cmpeqi_logic_or = 128, VVM_OP_cmpnei_logic_or = 129,
	VVM_OP_mad_add = 130, VVM_OP_mad_sub0 = 131, VVM_OP_mad_sub1 = 132, VVM_OP_mad_mul = 133, VVM_OP_mad_sqrt = 134,
	VVM_OP_mad_mad0 = 135, VVM_OP_mad_mad1 = 136, VVM_OP_mul_mad0 = 137, VVM_OP_mul_mad1 = 138, VVM_OP_mul_add = 139,
	VVM_OP_mul_sub0 = 140, VVM_OP_mul_sub1 = 141, VVM_OP_mul_mul = 142, VVM_OP_mul_max = 143, VVM_OP_mul_2x = 144,
	VVM_OP_add_mad1 = 145, VVM_OP_add_add = 146, VVM_OP_sub_cmplt1 = 147, VVM_OP_sub_neg = 148, VVM_OP_sub_mul = 149,
	VVM_OP_div_mad0 = 150, VVM_OP_div_f2i = 151, VVM_OP_div_mul = 152, VVM_OP_muli_addi = 153, VVM_OP_addi_bit_rshift = 154,
	VVM_OP_addi_muli = 155, VVM_OP_b2i_2x = 156, VVM_OP_i2f_div0 = 157, VVM_OP_i2f_div1 = 158, VVM_OP_i2f_mul = 159,
	VVM_OP_i2f_mad0 = 160, VVM_OP_i2f_mad1 = 161, VVM_OP_f2i_select1 = 162, VVM_OP_f2i_maxi = 163, VVM_OP_f2i_addi = 164,
	VVM_OP_fmod_add = 165, VVM_OP_bit_and_i2f = 166, VVM_OP_bit_rshift_bit_and = 167, VVM_OP_neg_cmplt = 168,
	VVM_OP_bit_or_muli = 169, VVM_OP_bit_lshift_bit_or = 170, VVM_OP_random_add = 171, VVM_OP_random_2x = 172,
	VVM_OP_max_f2i = 173, VVM_OP_select_mul = 174, VVM_OP_select_add = 175, VVM_OP_sin_cos = 176,
	VVM_OP_outputdata_float_from_half = 177, VVM_OP_outputdata_half_from_half = 178,
	VVM_NUM_OPCODES = 179
};

/* EVectorVMFlags (VectorVM.h:529-536) */
enum {
	VVM_FLAG_OPT_SAVE_INTERMEDIATE = 1 << 0,
	VVM_FLAG_OPT_OMIT_STATS        = 1 << 1,
	VVM_FLAG_LARGE_SCRIPT          = 1 << 2,
	VVM_FLAG_HAS_RAND_INSTRUCTION  = 1 << 3,
	VVM_FLAG_DATA_MAP_CACHE_SETUP  = 1 << 4
};

/* ---- frozen blob (VectorVMBridge.cpp:14-148): packed little-endian header, then 16-byte aligned sections ---- */
#define VVM_BLOB_HEADER_PACKED_BYTES 54  /* what the archive writes, VectorVMBridge.cpp:81-101 */
#define VVM_BLOB_FIRST_SECTION       64  /* Align(sizeof(FRuntimeContextData)=56, 16), VectorVMBridge.cpp:116 */
#define VVM_BLOB_EXTFN_ENTRY_BYTES   16  /* sizeof(FVectorVMExtFunctionData) = {ptr, int32, int32}, VectorVM.h:360-365 */
#define VVM_INVALID_REG              0xFFFFu

/* ---- flattened containers ---- */
typedef struct vvm_id_table {       /* TArray<int32>: GetData(), Num(), Max() */
	int32_t* data;
	int32_t  num;
	int32_t  max;
} vvm_id_table;

typedef struct vvm_input_register_view {   /* TArrayView<uint8 const* RESTRICT const>, VectorVM.h:248 */
	const uint8_t* const* data;
	int32_t               num;
} vvm_input_register_view;

typedef struct vvm_output_register_view {  /* TArrayView<uint8* RESTRICT const>, VectorVM.h:249 */
	uint8_t* const* data;
	int32_t         num;
} vvm_output_register_view;

/* FDataSetMeta (VectorVM.h:246-273) */
typedef struct vvm_dataset_meta {
	vvm_input_register_view  InputRegisters;
	vvm_output_register_view OutputRegisters;
	uint32_t InputRegisterTypeOffsets[3];   /* first index of float / int32 / half streams */
	uint32_t OutputRegisterTypeOffsets[3];
	int32_t  DataSetAccessIndex;            /* out: survivors - 1 (RT.cpp:2305-2308) */
	int32_t  InstanceOffset;                /* first instance processed; also base of the outputs */
	vvm_id_table* IDTable;
	vvm_id_table* FreeIDTable;
	vvm_id_table* SpawnedIDsTable;
	int32_t* NumFreeIDs;
	int32_t* MaxUsedID;
	int32_t* NumSpawnedIDs;
	int32_t  IDAcquireTag;
} vvm_dataset_meta;

/* External functions.
 * The reference binds a TDelegate per function (VectorVM.h:358). The GPU executor resolves a
 * stated built-in set by id; the CPU oracle additionally accepts a host callback with the
 * reference's register-walk convention (VectorVM.h:315-356, RT.cpp:1442-1489). */
enum {
	VVM_EXTFN_HOST_CALLBACK = 0,  /* oracle only */
	VVM_EXTFN_PERLIN3       = 1,  /* 3 in (x,y,z) -> 1 out : FMath::PerlinNoise3D (UnrealMath.cpp:3658-3699) */
	VVM_EXTFN_CURL_NOISE3   = 2   /* 3 in (x,y,z) -> 3 out : curl of three offset Perlin potentials */
};

typedef struct vvm_extfn_context {          /* FVectorVMExternalFunctionContext, VectorVM.h:315-334 */
	uint32_t** RegisterData;
	uint8_t*   RegInc;                      /* 0 = constant, 1 = advance one 32-bit element per instance */
	int32_t    RegReadCount;
	int32_t    NumRegisters;
	int32_t    StartInstance;
	int32_t    NumInstances;
	int32_t    NumLoops;
	int32_t    PerInstanceFnInstanceIdx;
	void**     UserPtrTable;
	int32_t    NumUserPtrs;
	void*      RandStream;
	int32_t**  RandCounters;
	struct vvm_dataset_meta* DataSets;
	int32_t    NumDataSets;
} vvm_extfn_context;

typedef struct vvm_ext_function {           /* stands in for FVMExternalFunction */
	int32_t builtin_id;                     /* VVM_EXTFN_* */
	void  (*host_fn)(vvm_extfn_context*);   /* VVM_EXTFN_HOST_CALLBACK only */
	void*   user;
} vvm_ext_function;

typedef struct vvm_state vvm_state;         /* FVectorVMState (VectorVMTypes.h:23-75), opaque */

/* FVectorVMExecContext (VectorVMRuntime.h:10-28) */
typedef struct vvm_exec_context {
	struct {
		uint32_t NumBytesRequiredPerBatch;
		uint32_t PerBatchRegisterDataBytesRequired;
		uint32_t MaxChunksPerBatch;
		uint32_t MaxInstancesPerChunk;
	} Internal;
	vvm_state* VVMState;
	struct { vvm_dataset_meta* data; int32_t num; } DataSets;
	struct { const vvm_ext_function* const* data; int32_t num; } ExtFunctionTable;
	struct { void** data; int32_t num; } UserPtrTable;
	int32_t NumInstances;
	const uint8_t* const* ConstantTableData;
	const int*            ConstantTableNumBytes;
	int32_t               ConstantTableCount;
} vvm_exec_context;

/* Options with no counterpart in FVectorVMExecContext. All zero = reference behaviour. */
typedef struct vvm_exec_options {
	uint64_t RandSeed;          /* key of the counter-based generator used by random / randomi / random_add / random_2x */
	int32_t  ExecIndexBase;     /* added to exec_index*: global index of this shard's first instance */
	int32_t  RandInstanceBase;  /* added to the instance index that keys the generator (shard-invariant streams) */
	void*    Stream;            /* hipStream_t to enqueue on; NULL = the state's stream: one of four per device that the library's states share in turn */
	uint32_t Flags;             /* VVM_EXEC_* */
	int32_t* DeviceCounts;      /* device memory, MaxOutputDataSet ints, or NULL: where the Exec leaves the survivor count of every
	                             * dataset it runs acquireindex for, instead of the state's own words (vvm_state_device_counts). For
	                             * callers that hand the counts on device-side (the multi-GPU all-gather) without a copy in between.
	                             * Ignored for scripts with ID instructions and by vvm_exec_batch. */
	const uint32_t* RandXorwowState; /* host memory, 24 words, or NULL.  Not NULL: random / randomi / random_add / random_2x draw from the
	                             * reference's own generator (the four-lane xorwow of VectorVMRuntime.cpp:468-483, stepped once per draw
	                             * and loop in the reference's execution order: chunk by chunk, instruction by instruction), started
	                             * from this state instead of the cycle-counter seed of :445-466, which no caller can reproduce:
	                             * State[0][lane 0..3] .. State[4][lane 0..3], then Counters[lane 0..3] (FVectorVMBatchState::RandState;
	                             * vvm_xorwow_seed derives it the way the reference does).  The output then depends on the instance's
	                             * place in the Exec and on vm.PageSizeInKB (vvm_set_page_size_kb), as in the reference; RandSeed and
	                             * RandInstanceBase are not used.  The draws are laid out by a kernel of their own ahead of the Exec
	                             * (4 bytes of device memory per draw and instance, held by the state) and the Exec runs four
	                             * instances per lane or one.  Refused (VVM_E_UNSUPPORTED_OP): vvm_exec_noalloc, vvm_exec_batch /
	                             * vvm_batch_*.  A script whose acquired datasets go two to a pass runs one program per dataset for
	                             * such an Exec (the kernels that read the words have one chain). */
} vvm_exec_options;

enum {
	VVM_EXEC_NO_VGPR_WIDE = 1u << 0,  /* force the one-instance-per-lane kernel (debug / comparison) */
	/* A plain script (no transcendental / noise / ID instructions, all registers resident) over at least
	 * VVM_EIGHT_WIDE_MIN_INSTANCES instances runs with eight instances per lane instead of four: half the interpreter
	 * overhead per instance.  The results are the same either way. */
	VVM_EXEC_NO_EIGHT_WIDE = 1u << 1,    /* never */
	VVM_EXEC_FORCE_EIGHT_WIDE = 1u << 2, /* whenever the script allows it, whatever the size (tests) */
	/* The per-Exec parameter block (descriptor, constants, stream addresses) of a single Exec normally reaches the kernel as a
	 * kernel argument when it fits 1 KB, and otherwise through a host-to-device copy enqueued ahead of the launch.  This flag
	 * takes the copy whatever the size (tests / comparison; the environment variable VVM_NO_INLINE_PARAMS does the same). */
	VVM_EXEC_PARAMS_IN_MEMORY = 1u << 3,
	/* Scripts that need no more than eight per-lane registers run their eight-wide Execs on a kernel with a smaller register
	 * file, of which three workgroups fit a CU instead of two.  This flag keeps them on the general eight-wide kernel (tests /
	 * comparison).  The results are the same either way. */
	VVM_EXEC_NO_LIGHT_KERNEL = 1u << 4,
	/* A script that runs acquireindex for two datasets normally runs both in one pass of the kernel.  Where that form does not
	 * apply (more datasets; transcendental, noise or ID instructions beside them; the one-instance-per-lane kernel) the script is
	 * executed as one program per dataset (a pass), and passes that run four or one instance per lane go out as one launch that
	 * runs all of them on a tile of instances before it moves on.  This flag launches the passes one after the other instead
	 * (tests / comparison).  The results are the same. */
	VVM_EXEC_SEPARATE_PASSES = 1u << 5,
	/* A script whose particle record is wider than the register file of the light eight-wide kernel (every stream it reads is in a
	 * per-lane register from the top of a tile), and whose extra width is attributes it hands through unchanged, is kept in a
	 * lean form as well: a program with everything the script computes (for the light eight-wide kernel; where that needs nine to
	 * sixteen registers for the general eight-wide one, with transcendental or noise functions for the four-wide kernel that has
	 * them), and the list of the handed-through streams, which a copy kernel moves behind that program's launch (it takes every
	 * survivor's place from what the program's acquireindex left).  A single Exec of at least VVM_THROUGH_MIN_INSTANCES instances (VVM_THROUGH_MIN_INSTANCES_EIGHT_WIDE
	 * where the whole script still fits the general eight-wide kernel) runs that way, and so does a member of a batch of at least
	 * VVM_BATCH_THROUGH_MIN_INSTANCES instances whose whole script needs more than sixteen registers (one copy-kernel launch serves
	 * all such members of a slice) (bit
	 * 0x80000 of vvm_state_info's Flags: the state has the form; the top byte holds the number of handed-through streams).
	 * This flag runs the script whole instead (tests / comparison; so does VVM_NO_THROUGH=1 in the environment at
	 * vvm_alloc_state).  The results are the same either way. */
	VVM_EXEC_NO_THROUGH_KERNEL = 1u << 6
};
#define VVM_EIGHT_WIDE_MIN_INSTANCES (256 * 1024)
/* the lean form of a wide-record script (VVM_EXEC_NO_THROUGH_KERNEL) runs from this size on: it is two launches, which still measured
 * faster than the whole script on the kernels with 32 registers at every size tried (24 streams each way, 2 Ki / 32 Ki / 64 Ki
 * instances: 21 / 22 / 23 us per Exec against 27 / 29 / 29 us) */
#define VVM_THROUGH_MIN_INSTANCES 1024
/* ... and from this size on where the whole script fits the general eight-wide kernel (nine to sixteen per-lane registers): 12 to
 * 16 streams each way measured 1 to 3 % faster lean at 2 Mi instances, 8 to 12 % at 4 Mi, 7 to 17 % at 32 Mi, slower at 1 Mi */
#define VVM_THROUGH_MIN_INSTANCES_EIGHT_WIDE (4 * 1024 * 1024)
/* ... and for a member of a batch (whose launches are shared) whose whole script needs more than sixteen registers */
#define VVM_BATCH_THROUGH_MIN_INSTANCES 1024
/* ... and inside vvm_exec_batch / vvm_batch_exec, where the launch is shared, a member whose script fits the light eight-wide
 * kernel (eight per-lane registers; bit 0x20000 of vvm_state_info's Flags) runs eight-wide from this size on */
#define VVM_BATCH_LIGHT_MIN_INSTANCES (8 * 1024)

/* ---- entry points ---- */

/* AllocVectorVMState(TConstArrayView<uint8>) -- RT.h:30, RT.cpp:2311-2417.
 * Returns NULL on an empty or malformed blob (vvm_last_error() says why). Unlike the reference
 * (VectorVMBridge.cpp:191-198) the state copies what it needs; the blob may be freed afterwards. */
VVM_API vvm_state* vvm_alloc_state(const uint8_t* blob, size_t num_bytes);

/* FreeVectorVMState -- RT.h:31, RT.cpp:2419-2425. NULL is allowed. */
VVM_API void vvm_free_state(vvm_state* state);

/* ExecVectorVMState -- RT.h:33, RT.cpp:2182-2309. Synchronous: on return every output stream holds
 * its compacted survivors in [InstanceOffset, InstanceOffset+S) and DataSets[i].DataSetAccessIndex = S-1.
 * One Exec at a time per state (the reference mutates its state the same way). */
VVM_API int vvm_exec(vvm_exec_context* ctx);
VVM_API int vvm_exec_ex(vvm_exec_context* ctx, const vvm_exec_options* opt);

/* SetupRandStateForBatch (VectorVMRuntime.cpp:445-464) with its three sources of entropy as arguments: the two readings of the
 * cycle counter (:447, :449) and the address of the batch state (:448).  Fills the 24 words vvm_exec_options.RandXorwowState takes. */
VVM_API void vvm_xorwow_seed(uint64_t cycles_first, uint64_t cycles_second, uint64_t batch_state_address, uint32_t* state24);

/* Enqueue-only form of vvm_exec_ex and its completion. Between the two calls the context, its dataset
 * array and the state must stay alive and untouched. vvm_exec_finish blocks on the stream, then writes
 * DataSetAccessIndex. Several vvm_exec_async calls on one state may be queued before one finish; the
 * finish reports the counts of the LAST one.  Scripts with update_id / acquire_id are the exception: the
 * dataset's NumFreeIDs / MaxUsedID / NumSpawnedIDs are host words read at enqueue and written by the
 * finish, so a second vvm_exec_async before the finish returns VVM_E_INVALID_ARG.  A script with acquire_id AND
 * update_id on one dataset hands IDs from chunk to chunk of the reference's chunking (VectorVMRuntime.cpp:1175,
 * 1411-1423): its Exec runs one launch per chunk, each finished before the next, inside vvm_exec_async itself (the
 * call is synchronous for such a script and vvm_exec_finish only reports).
 *
 * Under HIP graph capture: a vvm_exec_async whose vvm_exec_options.Stream is being captured adds kernel nodes to the graph and
 * nothing else, provided that (a) the state's last Exec ran on that same stream with at least as many instances (its buffers
 * are sized and no other stream has to be waited for), (b) the parameter block travels as a kernel argument (no ID
 * instructions, at most 1 KB of constants and stream addresses, no VVM_EXEC_PARAMS_IN_MEMORY), (c) RandXorwowState is not set
 * and kernel timing is off.  Every replay of the graph is then that Exec again, on the stream addresses, constants and
 * NumInstances it was captured with; vvm_exec_finish, called once the replays have been waited for, reports the last one's
 * counts (or leave them on the device: DeviceCounts).  tests/test_gpu_graph_capture.py. */
VVM_API int vvm_exec_async(vvm_exec_context* ctx, const vvm_exec_options* opt);
VVM_API int vvm_exec_finish(vvm_exec_context* ctx);

/* vvm_exec_ex for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that must not
 * disturb the allocator while a HIP graph is being captured on ANOTHER stream): same arguments, same results, but the call makes
 * no hipMalloc / hipFree / hipHostMalloc and creates no stream or event.  It is synchronous like vvm_exec_ex (it copies the
 * survivor counts back and waits for its stream), so it cannot itself be issued on a stream that is being captured.  The launch words, survivor counts and look-back words of the Exec live in an array that is part of the library's
 * code object (5 MiB + 4.25 KiB per device, held from the library's first use on that device until the process ends) instead
 * of in buffers sized per Exec; 4 MiB of it are the words through which a script with a wide record runs in its lean form (see
 * VVM_EXEC_NO_THROUGH_KERNEL) up to 4096 tiles, i.e. 8 Mi instances: larger Execs of such a script run it whole, with the same
 * results.  Limits, each refused with a message (vvm_last_error):
 *   - at most 131072 tiles: 128 Mi instances at four instances per lane, 256 Mi at eight (VVM_E_TOO_LARGE);
 *   - the script's parameter block (constants, stream addresses) fits the 1 KB that travels as a kernel argument: about
 *     40 streams and constants together, for an Exec in the lean form those of the lean program only (VVM_E_TOO_LARGE);
 *   - no update_id / acquire_id (their scratch is sized per Exec), no VVM_EXEC_NO_VGPR_WIDE before a vvm_exec_ex has used it on
 *     the state (VVM_E_UNSUPPORTED_OP).
 * Every call on a device shares that device's array: concurrent calls of this entry point on one device take turns (a lock
 * per device is held from enqueue to completion).  The array keeps the last call's contents, so vvm_state_device_counts of a
 * state whose last Exec went through this entry point is valid only until the next such call on that device.
 * No counterpart in the reference, whose Exec mallocs its batch memory on every call (VectorVMRuntime.cpp:2268, 2302). */
VVM_API int vvm_exec_noalloc(vvm_exec_context* ctx, const vvm_exec_options* opt);

/* Test hook, no counterpart in the reference.  Every launch tags its look-back words with an epoch that counts 1 .. 2^30 - 1 and
 * round again; the launch that follows the wrap clears the words first.  This sets the epoch the next launch will use, through
 * the state's own launch words (scratch = 0) or through the scratch array of the state's device (scratch = 1, what
 * vvm_exec_noalloc runs on), as if that many launches had gone before, so that a test can cross the wrap. */
VVM_API int vvm_debug_set_launch_epoch(vvm_state* state, uint32_t epoch, int scratch);
/* Test hook: how many times vvm_exec_batch has planned a batch (sorted its Execs into launches) in this process.  It keeps the plan of
 * its last call and plans again only when the states, sizes, table counts or options differ from that call's, or a state has been
 * made or released since; stream addresses and constants are read on every call. */
VVM_API uint64_t vvm_debug_batch_plans_made(void);

/* Many independent Execs (one state each, e.g. one per emitter) in one call: one kernel launch per kernel shape (a batch of 1024
 * Execs or more is cut into up to four slices that are filled and launched one after the other, so that filling the parameter
 * blocks overlaps the kernels).  Scripts with update_id / acquire_id, or with acquireindex on several datasets, take launches
 * of their own behind the batched ones.  Same result as calling vvm_exec_ex on each context in order.  opts may be NULL or an
 * array of n.  No counterpart in the reference, which runs one Exec per call (VectorVMRuntime.cpp:2182). */
VVM_API int vvm_exec_batch(vvm_exec_context* ctxs, int32_t n, const vvm_exec_options* opts);

/* A batch that is kept from tick to tick.  vvm_exec_batch validates, sizes and fills the parameter blocks of all its Execs on
 * every call; an emitter set that persists pays that once here: vvm_batch_create does it and keeps the blocks in device memory,
 * vvm_batch_exec runs the batch again and touches only what the caller says has changed since the handle's last call:
 *   VVM_BATCH_CHANGED_INSTANCES  NumInstances of some contexts (tile counts; an Exec that moves to another kernel re-plans the batch)
 *   VVM_BATCH_CHANGED_CONSTANTS  contents of constant tables (the tables' addresses and sizes are read again as well)
 *   VVM_BATCH_CHANGED_STREAMS    stream addresses, InstanceOffset, DataSets arrays, ID tables: every block is filled again.
 *                                (A caller that ping-pongs two sets of streams keeps one handle per set and alternates.)
 *   0                            nothing: no host work on the blocks and no upload at all.
 * The `ctxs` array (and what it points to) must stay where it is until vvm_batch_destroy; the options are copied.  Results land in
 * the contexts exactly as with vvm_exec_batch, and a state may be used by other calls between two vvm_batch_exec calls (one Exec
 * at a time per state, as always).  One call at a time per handle; handles are independent of each other and of vvm_exec_batch.
 * No counterpart in the reference, which runs one Exec per call (VectorVMRuntime.cpp:2182). */
typedef struct vvm_batch vvm_batch;
enum {
	VVM_BATCH_CHANGED_INSTANCES = 1u << 0,
	VVM_BATCH_CHANGED_CONSTANTS = 1u << 1,
	VVM_BATCH_CHANGED_STREAMS   = 1u << 2
};
VVM_API vvm_batch* vvm_batch_create(vvm_exec_context* ctxs, int32_t n, const vvm_exec_options* opts);
VVM_API int        vvm_batch_exec(vvm_batch* batch, uint32_t changed);
VVM_API void       vvm_batch_destroy(vvm_batch* batch);

/* ---- instance ranges sharded over several GPUs (one process per GPU) ----
 * Rank r of `world` holds a contiguous range of an Exec's instances in its own device memory and runs the Exec over it, with
 * vvm_exec_options.ExecIndexBase = RandInstanceBase = the global index of the range's first instance, so that exec_index and the
 * random draws do not depend on the sharding.  The ranks then exchange their survivor counts, 4 bytes per dataset and rank; the
 * survivors stay where each rank wrote them, and the logical global array is the concatenation of the ranks' compacted prefixes
 * in rank order, which is the reference's survivor order (VectorVMRuntime.cpp:786-804).  The context's DataSetAccessIndex holds
 * the LOCAL count - 1 (what the rank's streams hold); vvm_shard_result holds the global figures.
 * The communicator is either RCCL (librccl.so is loaded when the first one is created: rank 0 obtains an id with
 * vvm_comm_unique_id, hands it to the other ranks by any means, and every rank calls vvm_comm_create_rccl with its device
 * current), or a callback that gathers `count` ints per rank in rank order: with VVM_COMM_BUFFERS_HOST it receives host
 * pointers and runs synchronously (stream = NULL), with VVM_COMM_BUFFERS_DEVICE device pointers and the stream to enqueue on.
 * No counterpart in the reference: its Exec is one thread in one address space (VectorVMRuntime.cpp:2276-2295). */
#define VVM_MAX_SHARD_DATASETS 8
typedef struct vvm_comm vvm_comm;
typedef int (*vvm_all_gather_fn)(void* user, const int32_t* send, int32_t* recv, int32_t count, void* stream);      /* 0 = success */
enum { VVM_COMM_BUFFERS_HOST = 0, VVM_COMM_BUFFERS_DEVICE = 1 };
typedef struct vvm_shard_result {
	int32_t NumDataSets;
	int32_t LocalSurvivors[VVM_MAX_SHARD_DATASETS];
	int64_t GlobalOffset[VVM_MAX_SHARD_DATASETS];      /* survivors of the ranks before this one: where this rank's first survivor stands globally */
	int64_t GlobalSurvivors[VVM_MAX_SHARD_DATASETS];   /* over all ranks; the global DataSetAccessIndex is this - 1 */
} vvm_shard_result;
VVM_API int       vvm_comm_unique_id(uint8_t* out_128_bytes);
VVM_API vvm_comm* vvm_comm_create_rccl(int32_t rank, int32_t world, const uint8_t* unique_id_128_bytes);
VVM_API vvm_comm* vvm_comm_create_callback(int32_t rank, int32_t world, vvm_all_gather_fn fn, void* user, int32_t where);
VVM_API void      vvm_comm_destroy(vvm_comm* comm);
/* The number of ranks of the communicator: what ncclCommCount reports for an RCCL communicator, `world` for a callback one. */
VVM_API int32_t   vvm_comm_world(const vvm_comm* comm);
/* One rank's part of a sharded Exec: vvm_exec_ex over the local range, then the exchange.  Synchronous. */
VVM_API int vvm_exec_sharded(vvm_exec_context* ctx, const vvm_exec_options* opt, vvm_comm* comm, vvm_shard_result* out);
/* The same in two halves, so that the exchange of one step runs (on the communicator's own stream) beside the next step's Exec:
 * vvm_exec_sharded_async enqueues Exec and exchange (up to four exchanges may be pending), vvm_exec_sharded_wait blocks for the
 * oldest pending exchange and returns its figures; the local results reach the context through vvm_exec_finish as usual. */
VVM_API int vvm_exec_sharded_async(vvm_exec_context* ctx, const vvm_exec_options* opt, vvm_comm* comm);
VVM_API int vvm_exec_sharded_wait(vvm_comm* comm, vvm_shard_result* out);
/* The arithmetic of the exchange alone, no device involved: all_counts[r * num_datasets + d] = survivors of rank r in dataset d. */
VVM_API int vvm_shard_offsets(int32_t rank, int32_t world, int32_t num_datasets, const int32_t* all_counts, vvm_shard_result* out);

/* Device pointer (int32[MaxOutputDataSet]) to the survivor counts of the last vvm_exec* call made on this state
 * (the caller's vvm_exec_options.DeviceCounts when that was given); valid after the Exec's kernel has run on its
 * stream. Lets a multi-GPU caller all-gather the counts without a host round trip.  After vvm_exec_batch the counts of
 * the batch's Execs are only in the contexts (DataSetAccessIndex): this pointer then still names an earlier Exec's. */
VVM_API const int32_t* vvm_state_device_counts(const vvm_state* state);
VVM_API int32_t        vvm_state_max_output_dataset(const vvm_state* state);

/* Introspection of the linked program (tests, reported register and byte counts). */
typedef struct vvm_program_info {
	uint32_t NumTempRegisters, NumConstBuffers, NumInputBuffers, NumOutputBuffers;
	uint32_t NumInstructions;       /* linked device instructions */
	uint32_t NumVirtualRegisters;   /* per-lane registers the kernel needs (temps + output-aliased) */
	uint32_t InstancesPerLane;      /* 4 = 128-bit stream access, 1 = narrow fallback */
	uint32_t LdsSpillRegisters;     /* registers living in LDS, on the kernel the script is linked for */
	uint32_t BytesInPerInstance, BytesOutPerInstance; /* algorithmic stream bytes */
	uint32_t Flags;                 /* [0:16) the script's EVectorVMFlags; 0x10000 the program needs the kernels with the transcendental /
	                                 * noise / ID instructions; 0x20000 the script (its first pass) has a program for the light eight-wide
	                                 * kernel; 0x40000 the first pass runs two acquired datasets (two chains); 0x80000 the script has a lean
	                                 * form (VVM_EXEC_NO_THROUGH_KERNEL), [24:32) its handed-through streams; [20:24) passes, i.e. launches,
	                                 * per Exec (one, unless the script acquires three or more datasets, or two that cannot share a pass) */
} vvm_program_info;
VVM_API int vvm_state_info(const vvm_state* state, vvm_program_info* out);

/* Thaw and link a blob without touching a device: the checks vvm_alloc_state makes, and the same vvm_program_info.
 * Lets host-side tooling validate a script on a machine with no GPU. */
VVM_API int vvm_inspect_blob(const uint8_t* blob, size_t num_bytes, vvm_program_info* out);

/* Instances per lane (8, 4 or 1) of the kernel the state's last vvm_exec* ran on; 0 before the first.  No counterpart in the
 * reference (its loop is always four wide, VectorVMRuntime.cpp:2135-2175). */
VVM_API int32_t vvm_state_last_exec_width(const vvm_state* state);
/* vm.PageSizeInKB (VectorVMRuntime.cpp:44-50, default 64): the page the reference sizes its chunks by (:2210-2261).  The
 * executor does not chunk, but Internal.* and the chunk-dependent results of update_id / acquire_id follow it.  Process-wide. */
VVM_API void vvm_set_page_size_kb(int kb);
/* Thaw and link a blob without touching a device and copy out the linked device program: 4 dwords per instruction
 * (renderer_amd/csrc/vvm_device.h gives the encoding), at most `capacity_instructions` of them.  Returns the number of
 * instructions of the program (negative status code on failure).  narrow = 1 links for the 1-instance-per-lane kernel,
 * narrow = 2 for the light eight-wide kernel (VVM_E_UNSUPPORTED_OP when the script does not fit it). */
VVM_API int vvm_inspect_code(const uint8_t* blob, size_t num_bytes, int narrow, uint32_t* out_words, int32_t capacity_instructions);
/* The same program as it is uploaded: 8 dwords per instruction, the four above followed by what the eight-wide kernels' hand-written
 * core executes from (renderer_amd/csrc/vvm_threaded.h: routine offset, operand bytes, destination byte; zeros for an instruction
 * that takes the compiled path), and one more `done` at the end.  Returns the number of entries.  No counterpart in the reference. */
VVM_API int vvm_inspect_device_code(const uint8_t* blob, size_t num_bytes, int narrow, uint32_t* out_words, int32_t capacity_instructions);

/* ---- script compile side (host only, no device) ----
 * OptimizeVectorVMScript (VectorVMOptimizer.h:17, VectorVMOptimizer.cpp:700-3177) followed by FreezeOptimizerContext
 * (VectorVMBridge.h:25, VectorVMBridge.cpp:151-178): takes bytecode in the format the Niagara compiler emits (decoder
 * VectorVMOptimizer.cpp:744-1025) plus the input / output counts of its external functions, and returns the frozen blob that
 * vvm_alloc_state consumes, in malloc'd memory released with vvm_free_blob.  A NULL or one-byte script yields no blob
 * (*blob = NULL) and VVM_OK, as the reference returns 0 and leaves the context empty (:703-712).  On failure returns
 * VVM_E_BAD_BLOB with the reference's EVectorVMOptimizeError bits (VectorVMEditor.h:104-120) in *error_flags. */
typedef struct vvm_ext_fn_io {      /* the counts of FVectorVMExtFunctionData, VectorVM.h:360-365 */
	int32_t NumInputs;
	int32_t NumOutputs;
} vvm_ext_fn_io;
enum {
	VVM_OPT_ERR_OUT_OF_MEMORY         = 1 << 0,
	VVM_OPT_ERR_OVERFLOW              = 1 << 1,
	VVM_OPT_ERR_BYTECODE              = 1 << 2,
	VVM_OPT_ERR_REGISTER_USAGE        = 1 << 3,
	VVM_OPT_ERR_CONST_REMAP           = 1 << 4,
	VVM_OPT_ERR_INSTRUCTIONS          = 1 << 5,
	VVM_OPT_ERR_INPUT_MERGE_BUFFER    = 1 << 6,
	VVM_OPT_ERR_INSTRUCTION_REORDER   = 1 << 7,
	VVM_OPT_ERR_SSA_REMAP             = 1 << 8,
	VVM_OPT_ERR_OPTIMIZED_BYTECODE    = 1 << 9,
	VVM_OPT_ERR_EXTERNAL_FUNCTION     = 1 << 10,
	VVM_OPT_ERR_REDUNDANT_INSTRUCTION = 1 << 11
};
VVM_API int vvm_optimize_script(const uint8_t* bytecode, int32_t num_bytes, const vvm_ext_fn_io* ext_fns, int32_t num_ext_fns,
                                uint64_t hash_id, uint32_t flags, uint8_t** blob, size_t* blob_bytes, uint32_t* error_flags);
VVM_API void vvm_free_blob(uint8_t* blob);
/* GenerateHumanReadableScript (VectorVMOptimizer.h:18, VectorVMOptimizer.cpp:220-253): the optimizer's instruction list for the
 * script, as text in the reference's layout (summary, constant-table section, one "<position>\t| <instruction>;" line per
 * instruction that survives, registers as R / C / I / O [index]); op names as in editor builds of the reference.  The reference
 * prints from the intermediate state that VVMFlag_OptSaveIntermediateState (flags bit 0, VectorVM.h:531) keeps in the optimizer
 * context; this call optimizes the script for the listing.  *text is NUL-terminated malloc'd memory, released with
 * vvm_free_text; error reporting as vvm_optimize_script. */
VVM_API int vvm_generate_human_readable_script(const uint8_t* bytecode, int32_t num_bytes, const vvm_ext_fn_io* ext_fns, int32_t num_ext_fns,
                                               uint32_t flags, char** text, size_t* text_bytes, uint32_t* error_flags);
VVM_API void vvm_free_text(char* text);

/* Thread-local text of the last failure in this library. */
VVM_API const char* vvm_last_error(void);

/* Device memory helpers for callers without a HIP runtime binding of their own. */
VVM_API void* vvm_device_alloc(size_t num_bytes);
VVM_API void  vvm_device_free(void* p);
VVM_API int   vvm_device_upload(void* dst_device, const void* src_host, size_t num_bytes);
VVM_API int   vvm_device_download(void* dst_host, const void* src_device, size_t num_bytes);
VVM_API int   vvm_device_memset(void* dst_device, int byte_value, size_t num_bytes);
VVM_API int   vvm_device_synchronize(void);
VVM_API int   vvm_device_count(void);
VVM_API int   vvm_set_device(int ordinal);

/* Kernel timing with HIP events recorded on the stream each kernel is enqueued on.  While enabled, every Exec of the state
 * records one event pair around its kernel (two event records per Exec, at most 4096 pairs between collects).
 * vvm_state_collect_kernel_ms blocks until the recorded kernels have finished, returns the sum of their durations and how
 * many there were, and starts a new collection.  enable = 2: one event pair around ALL Execs between two collects instead (the
 * first event ahead of the first Exec's kernel, the second behind the last, recorded by the vvm_exec_finish that follows it, else by the collect): the sum is then the span on
 * the stream, gaps between the kernels included, and the Execs themselves carry no events (an event pair per Exec costs about
 * 8 us of stream time).  Execs that run as several passes, or on a stream that also waits for other work, spread the span. */
VVM_API int vvm_state_enable_timing(vvm_state* state, int enable);
VVM_API int vvm_state_collect_kernel_ms(vvm_state* state, float* sum_ms, int32_t* launches);

/* Enqueue a device-to-device copy of the state's survivor counts (int32[max(1, MaxOutputDataSet)]) on `stream`
 * (NULL = the state's stream).  For callers that exchange the counts between GPUs without a host round trip. */
VVM_API int vvm_state_copy_counts(vvm_state* state, void* dst_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VVM_ABI_H */
