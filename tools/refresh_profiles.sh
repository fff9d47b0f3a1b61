#!/bin/bash
# on the GPU box: the numbers profiles/ holds (kernel-trace stats of the bench command, the bench line, the other configurations, the ULP report)
set -e
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r01f -o bench --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $root/gpurun_out/prof_r01f.log 2>&1
cd $root
timeout -k 10 200 python bench.py > gpurun_out/bench_r01f.json 2> gpurun_out/bench_r01f.err
timeout -k 10 400 python tools/bench_configs.py --which arith,transcendental,kill_half,kill_half_250m,one_million,batch > gpurun_out/configs_r01f.jsonl 2> gpurun_out/configs_r01f.err
timeout -k 10 200 python tools/ulp_report.py > gpurun_out/ulp_r01f.json 2> gpurun_out/ulp_r01f.err
ls gpurun_out/prof_r01f
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r01f_configs -o configs --output-format csv -- python3 $root/tools/bench_configs.py --which transcendental,kill_half,one_million,batch --steps 10 > $root/gpurun_out/prof_r01f_configs.log 2>&1
cd $root
ls gpurun_out/prof_r01f_configs
