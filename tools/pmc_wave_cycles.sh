#!/bin/bash
# usage: [SKIP=n] [WHAT='tools/bench_configs.py --which arith --arith-ops 1 --steps 5'] tools/pmc_wave_cycles.sh <tag>   (on the GPU box)  where the waves of the bench kernel spend their cycles: one rocprofv3 pass of SQ counters
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export VVM_DEBUG_SKIP=${SKIP:-0}
WHAT=${WHAT:-bench.py --steps 5 --warmup 2 --no-cpu-baseline}
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $root/gpurun_out/$tag -o pmc --output-format csv -- python3 $root/$WHAT > $root/gpurun_out/$tag.log 2>&1 || { tail -5 $root/gpurun_out/$tag.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $root/gpurun_out/${tag}_b -o pmc --output-format csv -- python3 $root/$WHAT > $root/gpurun_out/${tag}_b.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_b.log; exit 1; }
python3 - <<PY
import csv, collections
for sub in ("$tag", "${tag}_b"):
    rows = [r for r in csv.DictReader(open("$root/gpurun_out/%s/pmc_counter_collection.csv" % sub)) if "vvm_exec" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(by.items()):
        print("%-22s mean per launch %.4g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
