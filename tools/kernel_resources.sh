#!/bin/bash
# usage: tools/kernel_resources.sh [pattern]   registers, scratch and spills of every kernel in vvm_kernels.hip as compiled for gfx950
# (device-only compile into /tmp, metadata notes of the code object)
set -e
cd "$(dirname "$0")/../renderer_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -structurizecfg-skip-uniform-regions=true --cuda-device-only -c vvm_kernels.hip -o /tmp/vvm_kernels.co
cd /tmp
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hip-amdgcn-amd-amdhsa-unknown-gfx950 --input=/tmp/vvm_kernels.co --output=/tmp/vvm_kernels.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/vvm_kernels.elf | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed|vgpr_spill|sgpr_spill|group_segment_fixed" | paste - - - - - - - | grep -E "${1:-.}" | sed -e 's/  */ /g'
