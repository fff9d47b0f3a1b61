#!/bin/bash
# usage (on the GPU box): tools/ab_death.sh <rounds> <variant>...   like tools/ab_bench.sh, for the sweep over death rates
# (tools/bench_configs.py --which death_sweep): kernel times of every rate with each build of the library in turn.
rounds=$1; shift
cp renderer_amd/csrc/libvvm_hip.so /tmp/libvvm_hip_saved.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp renderer_amd/csrc/variants/libvvm_hip_$v.so renderer_amd/csrc/libvvm_hip.so
    c=$(timeout -k 10 300 python tools/bench_configs.py --which death_sweep --steps 20 2>/dev/null | python -c "
import sys, json
print(' '.join('%.4f' % json.loads(l)['kernel_ms'] for l in sys.stdin if 'kernel_ms' in l))") || exit 1
    echo "round $r variant $v: 0 1 5 10 20 25 30 40 50 75 % dying: $c"
  done
done
cp /tmp/libvvm_hip_saved.so renderer_amd/csrc/libvvm_hip.so
