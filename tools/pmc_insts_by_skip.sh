#!/bin/bash
# usage (on the GPU box): tools/pmc_insts_by_skip.sh <tag> [skip values...]   instruction counts of the Exec kernel per launch with parts of
# a tile's work switched off (VVM_DEBUG_SKIP, see tools/ablate_arith.sh), on the one-instruction arithmetic script (WHAT overrides).
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
WHAT=${WHAT:-tools/bench_configs.py --which arith --arith-ops 1 --steps 5}
for skip in "$@"; do
  export VVM_DEBUG_SKIP=$skip
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY -d $root/gpurun_out/${tag}_$skip -o pmc --output-format csv -- python3 $root/$WHAT > $root/gpurun_out/${tag}_$skip.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_$skip.log; exit 1; }
  python3 - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$root/gpurun_out/${tag}_$skip/pmc_counter_collection.csv")) if "vvm_exec" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("skip %-4s" % "$skip", "  ".join("%s %.0f" % (k.replace("SQ_", ""), sum(v) / len(v) / 131072.0) for k, v in sorted(by.items())), " (per wave and tile)")
PY
done
