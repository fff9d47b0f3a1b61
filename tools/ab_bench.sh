#!/bin/bash
# usage (on the GPU box): tools/ab_bench.sh <rounds> <variant>...   A/B timing of builds of the library on ONE box: the named files
# renderer_amd/csrc/variants/libvvm_hip_<variant>.so are put in the library's place in turn, <rounds> times round robin, and the
# bench kernel and the arithmetic / half-dying configurations are timed with each.
rounds=$1; shift
cp renderer_amd/csrc/libvvm_hip.so /tmp/libvvm_hip_saved.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp renderer_amd/csrc/variants/libvvm_hip_$v.so renderer_amd/csrc/libvvm_hip.so
    b=$(timeout -k 10 200 python bench.py --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['kernel_ms'])") || exit 1
    c=$(timeout -k 10 200 python tools/bench_configs.py --which arith,kill_half --arith-ops ${ARITH_OPS:-1,16,64} --steps 20 2>/dev/null | python -c "
import sys, json
print(' '.join('%.4f' % json.loads(l)['kernel_ms'] for l in sys.stdin if 'kernel_ms' in l))") || exit 1
    echo "round $r variant $v: fountain $b | arith ${ARITH_OPS:-1,16,64} then half-dying: $c"
  done
done
cp /tmp/libvvm_hip_saved.so renderer_amd/csrc/libvvm_hip.so
