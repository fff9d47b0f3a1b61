#!/bin/bash
# usage (on the GPU box): tools/ab_configs.sh <rounds> <variant>...   like tools/ab_bench.sh, over the configurations of tools/bench_configs.py:
# the named builds renderer_amd/csrc/variants/libvvm_hip_<variant>.so are put in the library's place in turn and every configuration's
# kernel time is printed (tabulate with the variant names as columns)
rounds=$1; shift
cp renderer_amd/csrc/libvvm_hip.so /tmp/libvvm_hip_saved.so
for r in $(seq 1 $rounds); do for v in "$@"; do
  cp renderer_amd/csrc/variants/libvvm_hip_$v.so renderer_amd/csrc/libvvm_hip.so
  echo "== round $r $v"
  timeout -k 10 400 python tools/bench_configs.py --which ${CONFIGS:-transcendental,death_sweep,two_datasets,many_datasets,wide_record,one_million,kill_half_250m,batch} --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    k = d.get('kernel_ms', d.get('kernel_ms_per_exec', d.get('us_per_exec', d.get('kernel_us', d.get('ms_per_step')))))
    if k is not None: print('   %-70s %.4f' % (d['config'][:70], k))
"
done; done
cp /tmp/libvvm_hip_saved.so renderer_amd/csrc/libvvm_hip.so
