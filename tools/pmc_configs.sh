#!/bin/bash
# usage: tools/pmc_configs.sh <tag> <which>   (on the GPU box)  instruction-mix counters of tools/bench_configs.py --which <which>
tag=$1; which=$2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $root/gpurun_out/$tag -o pmc --output-format csv -- python3 $root/tools/bench_configs.py --which $which --steps 2 > $root/gpurun_out/$tag.log 2>&1 || { tail -5 $root/gpurun_out/$tag.log; exit 1; }
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('$root/gpurun_out/$tag/pmc_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'vvm_exec' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
