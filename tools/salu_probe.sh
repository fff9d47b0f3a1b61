#!/bin/bash
# usage: tools/salu_probe.sh <tag> [4|8]   (on the GPU box)  per-wave instruction counts of the probe ladder -> gpurun_out/<tag>/
tag=$1
width=${2:-4}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH -d $root/gpurun_out/$tag -o pmc --output-format csv -- python3 $root/tools/salu_probe.py --width $width > $root/gpurun_out/$tag.log 2>&1 || { tail -5 $root/gpurun_out/$tag.log; exit 1; }
python3 - <<PY
import csv, collections, re
names = [l.split()[1] for l in open("$root/gpurun_out/$tag.log") if l.startswith("variant ")]
ms = [float(l.split()[-1]) for l in open("$root/gpurun_out/$tag.log") if l.startswith("variant ")]
rows = [r for r in csv.DictReader(open("$root/gpurun_out/$tag/pmc_counter_collection.csv")) if "vvm_exec" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
tiles = 16 * 1024 * 1024 // (256 * $width) * 4       # wave-tiles per launch (4 waves per tile of 256 x width instances)
tab = {}
for i, name in enumerate(names):
    d = by[ids[2 * i + 1]]
    tab[name] = {k: v / tiles for k, v in d.items()}
    print("%-8s %.3f ms  per wave-tile:" % (name, ms[i]), "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), v) for k, v in sorted(tab[name].items()) if k != "SQ_WAVES"))
base = tab.get("base")
if base:
    for name in names:
        if name in ("add_vv", "add_vu", "mad_vuv", "madmul"):
            print("%-8s per instruction:" % name, "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), (tab[name][k] - base[k]) / 8) for k in sorted(base) if k != "SQ_WAVES"))
PY
