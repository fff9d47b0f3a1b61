// vvm_kernels.h -- host-callable entry points of vvm_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "vvm_device.h"

extern "C" {
// Enqueue the interpreter over `total_tiles` tiles described by `descs` (device memory).  instances_per_lane is 8, 4 or 1;
// many_registers != 0 selects the kernels with 32 resident registers plus lds_spill_slots registers in LDS; full != 0 the
// kernels that also implement the transcendental ops, the noise built-ins and the ID instructions (full == 2, four instances per
// lane without many_registers: the instantiation of that kernel that leaves the ride words for vvm_through_kernel).
// A launch of one Exec may hand its parameter block over as a kernel argument instead: descs = null, inline_block = the block
// as filled on the host with VVM_DESC_RELATIVE set, inline_bytes <= VVM_INLINE_PARAM_BYTES (copied before the call returns).
// tag: 0 = the look-back words of this launch are tagged with the epoch kept in `ls`; otherwise the tag itself, 1..VVM_MAX_LAUNCH_TAG
// (for launches that share one array of look-back words through several sets of launch words).
int vvm_launch_exec_kernel(int instances_per_lane, const VvmExecDesc* descs, uint32_t num_execs, uint32_t total_tiles,
                           VvmLaunchState* ls, int many_registers, uint32_t lds_spill_slots, int full, uint32_t grid, hipStream_t stream,
                           const void* inline_block, uint32_t inline_bytes, uint32_t tag, int fused);
// (fused != 0: the `num_execs` descriptors are the passes of one Exec, run tile by tile on the same instances by a FULL kernel;
// total_tiles = the tiles of one pass)
// update_id's MaxUsedID pass over an Exec that has run (same stream).
int vvm_launch_max_id_kernel(const VvmIdDesc* ids, uint32_t num_chunks, hipStream_t stream);
// The reference's random generator for one Exec: fills words[draw * stride + instance] (device memory, stride a multiple of 4 and
// at least 4 * total_loops) with the 32 bits of every draw; jump = the VVM_XORWOW_JUMP_WORDS words of the jump table in device memory.
int vvm_launch_xorwow_fill(const VvmXorwowFill* params, const uint32_t* jump, uint32_t* words, hipStream_t stream);
// The handed-through streams of an Exec whose program has run on the light eight-wide kernel and left its ride words (same stream).
int vvm_launch_through_kernel(const VvmThroughArgs* args, hipStream_t stream);
// The same for the lean members of a batch: `num_execs` blocks in device memory, each with the number of its first tile within
// this launch in `pad` (increasing), `total_tiles` tiles in all, no Exec with more than `max_pairs` pairs.
int vvm_launch_through_batch_kernel(const VvmThroughArgs* device_execs, uint32_t num_execs, uint32_t total_tiles, uint32_t max_pairs, hipStream_t stream);
int vvm_kernel_max_dynamic_lds(void);
uint32_t vvm_kernel_resident_registers(int instances_per_lane, int many_registers);
uint32_t vvm_kernel_lds_fixed_bytes(int instances_per_lane, int full);
// VVM_T_LAYOUT of the routine table (vvm_threaded_gen.h) that the eight-wide kernels were built with
uint32_t vvm_kernel_threaded_layout(void);
}
