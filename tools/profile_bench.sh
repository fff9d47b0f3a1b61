#!/bin/bash
# usage: tools/profile_bench.sh <tag>   (on the GPU box)  kernel trace + stats of the default bench -> gpurun_out/<tag>/
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
tail -2 $out.log | cut -c1-600
ls $out
