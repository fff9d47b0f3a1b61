#!/usr/bin/env python3
"""usage: copy_profiles.py <run tag> <profile prefix>   (here, after tools/refresh_profiles.sh <run tag> ran on the GPU box)  copies what profiles/ holds out of gpurun_out/, e.g. r02a r02."""
import csv
import json
import os
import shutil
import sys

RUN, PRE = sys.argv[1], sys.argv[2]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
for src, dst in (("prof_%s/" % RUN + "bench_kernel_stats.csv", PRE + "_bench_kernel_stats.csv"), ("bench_%s.json" % RUN, PRE + "_bench_line.json"),
                 ("configs_%s.jsonl" % RUN, PRE + "_other_configs.jsonl"), ("ulp_%s.json" % RUN, PRE + "_ulp_report.json"),
                 ("prof_%s_configs/" % RUN + "configs_kernel_stats.csv", PRE + "_configs_kernel_stats.csv")):
    shutil.copy(os.path.join(G, src), os.path.join(P, dst))
rows = [r for r in csv.DictReader(open(os.path.join(G, "prof_%s_configs/" % RUN + "configs_kernel_trace.csv"))) if "vvm_exec" in r["Kernel_Name"] or "vvm_through" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = []
for r in rows:
    o = {"kernel": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": int(r["Grid_Size_X"]),
         "us": round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 2)}
    # (an Exec may be two kernels, the program and the copy kernel for handed-through streams: look a few groups back)
    g = next((c for c in groups[-3:][::-1] if c["kernel"] == o["kernel"] and c["grid"] == o["grid"] and 0.7 < o["us"] / c["us"][0] < 1.4), None)
    if g:
        g["us"].append(o["us"])
    else:
        o["us"] = [o["us"]]
        groups.append(o)
path = os.path.join(P, PRE + "_configs_kernel_trace_summary.jsonl")
head = open(path).readline() if os.path.exists(path) else open(os.path.join(P, "r01_configs_kernel_trace_summary.jsonl")).readline()
with open(path, "w") as f:
    f.write(head)
    for g in groups:
        u = g["us"]
        f.write(json.dumps({"kernel": g["kernel"], "grid_threads": g["grid"], "launches": len(u), "mean_us": round(sum(u) / len(u), 2),
                            "min_us": min(u), "max_us": max(u)}) + "\n")
        print(g["kernel"], g["grid"], len(u), round(sum(u) / len(u), 2))
b = json.load(open(os.path.join(P, PRE + "_bench_line.json")))
print(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"]["frac"])
for line in open(os.path.join(P, PRE + "_other_configs.jsonl")):
    d = json.loads(line)
    if "ms_per_step" in d:
        print(d["config"][:50], round(d["ms_per_step"], 4), round(d.get("kernel_ms", 0), 4))
