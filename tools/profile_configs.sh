#!/bin/bash
# usage: tools/profile_configs.sh <tag> <which>   (on the GPU box)  kernel trace + stats of tools/bench_configs.py --which <which>
tag=$1; which=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o cfg --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --which $which > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep "^{" $out.log | cut -c1-400
grep vvm_ $out/cfg_kernel_stats.csv | cut -c1-200
