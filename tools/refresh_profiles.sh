#!/bin/bash
# on the GPU box: usage tools/refresh_profiles.sh <tag>  -- the numbers profiles/ holds (kernel-trace stats of the bench command, the bench line,
# the other configurations, the ULP report), written under gpurun_out/ with <tag> in their names
set -e
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o bench --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
timeout -k 10 200 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
timeout -k 10 400 python tools/bench_configs.py --which arith,transcendental,kill_half,kill_half_250m,death_sweep,one_million,wide_record,two_datasets,many_datasets,batch > gpurun_out/configs_$tag.jsonl 2> gpurun_out/configs_$tag.err
timeout -k 10 200 python tools/ulp_report.py > gpurun_out/ulp_$tag.json 2> gpurun_out/ulp_$tag.err
ls gpurun_out/prof_$tag
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_configs -o configs --output-format csv -- python3 $root/tools/bench_configs.py --which transcendental,kill_half,one_million,wide_record,batch --steps 10 > $root/gpurun_out/prof_${tag}_configs.log 2>&1
cd $root
ls gpurun_out/prof_${tag}_configs
