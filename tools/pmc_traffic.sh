#!/bin/bash
# usage: [PMC_BENCH_ARGS="--workload fountain_kill_half"] tools/pmc_traffic.sh <tag>   (on the GPU box)  HBM traffic of the bench kernel: one rocprofv3 pass per counter
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
	timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctr -d $root/gpurun_out/${tag}_$ctr -o pmc --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline $PMC_BENCH_ARGS > $root/gpurun_out/${tag}_$ctr.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_$ctr.log; exit 1; }
done
python3 - <<PY
import csv, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("$root/gpurun_out/${tag}_%s/pmc_counter_collection.csv" % ctr)))
    v = [float(r["Counter_Value"]) for r in rows if "vvm_exec" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    print(ctr, "launches", len(v), "mean per launch (counter units, KB)", sum(v) / len(v))
PY
