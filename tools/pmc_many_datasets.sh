#!/bin/bash
# usage: tools/pmc_many_datasets.sh <tag>   (on the GPU box)  HBM bytes fetched and written per Exec of the three- and four-dataset scripts of
# tools/bench_configs.py (--which many_datasets), with the datasets two to a pass (the default) and with one program and launch per
# dataset (VVM_NO_CHAINS=1): one rocprofv3 pass per counter and mode
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in paired separate; do
	unset VVM_NO_CHAINS; [ $mode = separate ] && export VVM_NO_CHAINS=1
	for ctr in FETCH_SIZE WRITE_SIZE; do
		timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $ctr -d $root/gpurun_out/${tag}_${mode}_$ctr -o pmc --output-format csv -- python3 $root/tools/bench_configs.py --which many_datasets --steps 3 > $root/gpurun_out/${tag}_${mode}_$ctr.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_${mode}_$ctr.log; exit 1; }
	done
done
python3 - <<PY
import csv, json
out = {"note": "counter unit KB; FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes on gfx950: doubled here; bytes per Exec, "
               "three warm-up Execs and three timed ones per script, the three-dataset script first",
       "algorithmic_input_bytes_per_exec": 32.0 * 64 * 1024 * 1024}
for mode in ("paired", "separate"):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = list(csv.DictReader(open("$root/gpurun_out/${tag}_%s_%s/pmc_counter_collection.csv" % (mode, ctr))))
        rows = [r for r in rows if "vvm_exec" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        v = [float(r["Counter_Value"]) for r in rows]
        # launches per Exec: paired 2 and 2, separate 3 and 4; six Execs per script
        per = (2, 2) if mode == "paired" else (3, 4)
        cut = 6 * per[0]
        assert len(v) == 6 * (per[0] + per[1]), (mode, ctr, len(v))
        scale = 2048.0 if ctr == "FETCH_SIZE" else 1024.0
        out["%s_%s_bytes_per_exec" % (mode, ctr)] = {"three_datasets": sum(v[:cut]) / 6 * scale, "four_datasets": sum(v[cut:]) / 6 * scale}
        out["%s_launches_per_exec" % mode] = {"three_datasets": per[0], "four_datasets": per[1]}
print(json.dumps(out))
PY
