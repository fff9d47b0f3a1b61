#!/bin/bash
# usage: tools/pmc_sq.sh <tag> [bench args]; instruction-mix counters of the bench kernels -> gpurun_out/<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/$tag.log; exit 1; }
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/$tag/pmc_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'vvm_exec' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
