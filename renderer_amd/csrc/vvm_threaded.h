// vvm_threaded.h -- the core of the eight-wide interpreter kernels: runs of ordinary instructions executed by hand-written gfx950
// routines ("threaded code").  Shared by the host (which picks every instruction's routine when it uploads a program) and the
// kernels (which hold the routines: vvm_threaded_gen.h, written by gen_threaded.py, where the register conventions are).
//
// Why: compiled from C++, an interpreted instruction cost about 36 scalar, 13 branch and 29 vector instructions per wave (a chain of
// compares to find the operand count and the opcode, then a GPR-index switch and eight moves per operand and for the result), and
// the kernel was bound by exactly that for scripts longer than a few instructions.  Here the instruction names its routine: the
// host stores the routine's byte offset in the instruction, the kernel adds it to the table's address and jumps (s_setpc_b64), and
// every routine ends by jumping to the next instruction's routine.  The per-lane register file lies in fixed VGPRs, REGISTER-MAJOR:
// the eight instance slots of virtual register i are VGPRs FILE + 8 i .. FILE + 8 i + 7, so one GPR-index setting reaches a whole
// register (as four 64-bit pairs for the packed fp32 instructions) and operands are read in place by the arithmetic instruction
// itself.  Measured with no memory traffic (tools/asm_probe.hip): 92 ns per interpreted instruction and workgroup pass against
// 230 ns.
//
// Device instruction (vvm_device.h: VvmDevIns, 32 bytes): words 0..3 as the linker made them (what the kernel's C++ reads for the
// instructions that have no routine), then
//   word 4  byte offset of the instruction's routine from the start of the routine table; 0 = none: the assembly hands it to the C++
//   word 5  operand bytes a | c << 8 | d << 16 | e << 24        word 6  operand byte b        word 7  destination byte
//           per-lane register: 8 x its index (its first VGPR relative to FILE); uniform register: its lane in the uniform VGPR.
// Which of the two an operand is does not show in the byte: it selects the routine.  An op has one routine per combination of
// operand kinds (the "uniform mask": bit k set = operand k is a uniform register), all-uniform excepted (the linker folds the common
// ones of those into derived constants; what is left takes the C++ path, like every op without a routine: the rarely used ones, the
// transcendental ones, those with operands that are neither of the two kinds).
//
// Two instruction buffers, X and Y (eight SGPRs each): an instruction executes from one while the next one arrives in the other, and
// every routine exists twice, once per buffer it reads its operands from, so nothing is ever copied.
#pragma once
#include <stdint.h>
#include "vvm_threaded_gen.h"

// The routine of an ordinary instruction: its byte offset in the table, or 0 when it has none.  uniform_mask: bit k = operand k is a
// uniform register (the caller has checked that every other operand, and the destination, is a per-lane register of the file).
static inline uint32_t vvm_threaded_routine(uint32_t op, uint32_t nin, uint32_t uniform_mask)
{
	if (nin < 1 || nin > 5 || uniform_mask + 1u >= (1u << nin)) return 0;      // (all operands uniform: no routine)
	static const uint32_t slot[6] = {0, VVM_T_SLOT1, VVM_T_SLOT2, VVM_T_SLOT3, VVM_T_SLOT4, VVM_T_SLOT5};
#define VVM_T_CASE(opcode, operands, first) if (op == (uint32_t)(opcode) && nin == (operands)) return (first) + uniform_mask * slot[operands];
	VVM_T_OPS(VVM_T_CASE)
#undef VVM_T_CASE
	return 0;
}
