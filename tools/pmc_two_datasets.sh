#!/bin/bash
# usage: tools/pmc_two_datasets.sh <tag>   (on the GPU box)  HBM bytes fetched per Exec of the two-dataset script of tools/bench_configs.py, with both
# datasets in one pass (two chains, the default) and with one program and launch per dataset (VVM_NO_CHAINS=1, --separate-passes):
# one rocprofv3 pass per counter and mode
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in chains separate; do
	extra=""; unset VVM_NO_CHAINS; [ $mode = separate ] && { extra="--separate-passes"; export VVM_NO_CHAINS=1; }
	for ctr in FETCH_SIZE WRITE_SIZE; do
		timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $ctr -d $root/gpurun_out/${tag}_${mode}_$ctr -o pmc --output-format csv -- python3 $root/tools/bench_configs.py --which two_datasets --steps 3 $extra > $root/gpurun_out/${tag}_${mode}_$ctr.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_${mode}_$ctr.log; exit 1; }
	done
done
python3 - <<PY
import csv, json
out = {}
for mode in ("chains", "separate"):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = list(csv.DictReader(open("$root/gpurun_out/${tag}_%s_%s/pmc_counter_collection.csv" % (mode, ctr))))
        v = [float(r["Counter_Value"]) for r in rows if "vvm_exec" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        execs = 6      # three warm-up Execs and three timed ones
        out["%s_%s_kb_per_exec" % (mode, ctr)] = sum(v) / execs
        out["%s_launches_per_exec" % mode] = len(v) / execs
line = [l for l in open("$root/gpurun_out/${tag}_chains_FETCH_SIZE.log").read().splitlines() if l.startswith("{")][-1]
out["algorithmic_bytes"] = json.loads(line)["algorithmic_bytes"]
out["note"] = "counter unit KB; FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes on gfx950: double it for bytes"
print(json.dumps(out))
PY
