#!/usr/bin/env python3
"""Condense gpurun_out/final (tools/final_profile.sh) into profiles/: per-kernel rocprofv3 averages and PMC medians."""
import csv
import glob
import json
import os
import statistics
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final4"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
tag = sys.argv[3] if len(sys.argv) > 3 else "r04"


def newest(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


import hashlib
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = {"kernels": {}, "kernel_stats": [],
       # the sources these passes were collected on: bench.py reports `traffic` from this file only while they still have this hash
       "device_source_sha16": __import__("lasgun_amd").device_source_sha16()}
ks = newest("prof/**/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            if "lg::" in r["Name"]:
                w.writerow(r)
                out["kernel_stats"].append({"name": r["Name"].replace("void ", "").replace("(lg::DParams)", ""), "calls": int(r["Calls"]),
                                            "avg_ms": float(r["AverageNs"]) / 1e6})
ko = newest("prof_overlap/**/*kernel_stats.csv")
if ko:
    rows = [r for r in csv.DictReader(open(ko)) if "lg::" in r["Name"]]
    with open(os.path.join(dst, tag + "_kernel_stats_overlapped_frames.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = newest(os.path.relpath(d, src) + "/**/*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if "lg::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(lg::DParams)", "")
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, vals in cs.items():
            out["kernels"].setdefault(k, {})[c] = statistics.median(vals)
json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:3000])
