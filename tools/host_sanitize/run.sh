#!/bin/bash
# usage: tools/host_sanitize/run.sh   (no GPU needed)  builds the linker and the optimizer with AddressSanitizer and
# UndefinedBehaviorSanitizer (host code only: g++, no HIP) and runs them over the corpus of corpus.py: every blob is thawed and
# linked for the wide, the one-instance-per-lane and the light kernels, every compiler-format script is optimized and listed
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
work=${TMPDIR:-/tmp}/vvm_host_sanitize
rm -rf "$work" && mkdir -p "$work/corpus"
flags="-std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I$root/renderer_amd/csrc -I$root/include"
g++ $flags "$here/link_driver.cpp" "$root/renderer_amd/csrc/vvm_program.cpp" -o "$work/link_driver"
g++ $flags "$here/optimize_driver.cpp" "$root/renderer_amd/csrc/vvm_optimizer.cpp" -o "$work/optimize_driver"
python3 "$here/corpus.py" "$work/corpus"
"$work/link_driver" "$work"/corpus/*.bin
"$work/optimize_driver" "$work"/corpus/*.bc
