#!/bin/bash
# round 6, GPU batch 1: tests on the new trig + graph replay, small-frame A/B, trig cost A/B, config 4 level by level launch by launch, bench
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6b1; mkdir -p "$O"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?"; tail -2 "$O/gputest.log"
timeout -k 10 500 bash tools/ab_small_frames.sh 3 > "$O/ab_small_frames.jsonl" 2> "$O/ab_small_frames.err"; echo "small frames rc=$?"
timeout -k 10 400 bash tools/ab_configs.sh "3 sph" 4 main oldtrig > "$O/ab_trig.jsonl" 2>/dev/null; echo "ab trig rc=$?"
timeout -k 10 300 bash tools/ab_configs.sh "1b 2G 4m" 2 main oldtrig >> "$O/ab_trig.jsonl" 2>/dev/null
export GPU_MAX_HW_QUEUES=16
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace_c4_wf" -- python3 tools/bench_configs.py --org=wavefront "4 mesh" > "$O/trace_c4_wf.log" 2>&1; echo "trace rc=$?"
python3 - "$O" <<'PY'
import csv, glob, sys, os
O = sys.argv[1]
fs = sorted(glob.glob(os.path.join(O, "trace_c4_wf", "**", "*kernel_trace.csv"), recursive=True))
if fs:
    rows = [r for r in csv.DictReader(open(fs[-1])) if "lg::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(os.path.join(O, "c4_wavefront_launches.csv"), "w") as f:
        f.write("start_us,dur_ms,grid,kernel\n")
        t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
        for r in rows:
            f.write("%.1f,%.4f,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"].replace("void ", "").replace("(lg::DParams)", "")))
PY
rm -rf "$O/trace_c4_wf"
timeout -k 10 500 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
tail -c 300 "$O/bench.json"
