#!/usr/bin/env python3
"""Condense gpurun_out/final (tools/final_profile.sh) into profiles/: per-kernel rocprofv3 averages and PMC medians."""
import csv
import glob
import json
import os
import statistics
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final6"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
tag = sys.argv[3] if len(sys.argv) > 3 else "r06"


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
# the clock the chip held under each kernel: GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration, dispatch by dispatch (counter rows joined with the
# kernel trace of the same run by dispatch id, or by the timestamps the counter file carries itself), median per kernel
out["clock_ghz"] = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_grbm*"))):
    f = newest(os.path.relpath(d, src) + "/**/*counter_collection.csv")
    t = newest(os.path.relpath(d, src) + "/**/*kernel_trace.csv")
    if not f:
        continue
    dur = {}
    if t:
        for r in csv.DictReader(open(t)):
            try:
                dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    per = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE" or "lg::" not in r["Kernel_Name"]:
            continue
        ns = dur.get(r.get("Dispatch_Id"))
        if ns is None and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if not ns or ns < 3e5:  # (the quotient reads high on dispatches shorter than about 0.3 ms)
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(lg::DParams)", "")
        per.setdefault(k, []).append(float(r["Counter_Value"]) / 8.0 / ns)
    for k, v in per.items():
        out["clock_ghz"][k] = {"median": statistics.median(v), "min": min(v), "max": max(v), "dispatches": len(v)}
for name in ("prof_c4", "prof_c5"):
    ks2 = newest(name + "/**/*kernel_stats.csv")
    if ks2:
        rows = [r for r in csv.DictReader(open(ks2)) if "lg::" in r["Name"]]
        if rows:
            with open(os.path.join(dst, tag + "_" + name.replace("prof_c", "config").replace("config5", "config5").replace("config4", "config4") + "_kernel_stats.csv"), "w") as fo:
                w = csv.DictWriter(fo, fieldnames=rows[0].keys())
                w.writeheader()
                w.writerows(rows)
json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
# the other outputs of tools/final_profile.sh, under the round's tag
import shutil
for name in ("bench.json", "configs_parity.jsonl", "configs_fast.jsonl", "configs_megakernel.jsonl", "configs_wavefront.jsonl", "configs_queue.jsonl",
             "configs_unpruned.jsonl", "configs_pruned.jsonl", "share_time.jsonl", "multi_2x_same_device.json", "multi_2x_same_device_rccl.json",
             "config4_pmc.txt", "config5_pmc.txt", "progressive.jsonl", "org_choice.jsonl", "host_capture.json", "host_capture_one_band.json",
             "queue_levels.jsonl", "prune_audit.jsonl", "prune_audit_full.jsonl", "ss_par.jsonl", "tail_probe.jsonl", "host_capture_examples.jsonl", "build_times.log", "rule_first_launch.jsonl"):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copyfile(f, os.path.join(dst, tag + "_" + name))
print(json.dumps(out, indent=1, sort_keys=True)[:3000])
