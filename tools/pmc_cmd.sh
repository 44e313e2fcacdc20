#!/bin/bash
# Counter passes over any python command (each --pmc set is its own run; no trace flags with --pmc).
# usage (on the GPU box): bash tools/pmc_cmd.sh <tag> tools/bench_configs.py --fast "4 mesh"   -> gpurun_out/pmcc_<tag>/summary.txt
set -eu
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16  # (the profiler initialises HIP before the program can set it)
tag=$1; shift
O=gpurun_out/pmcc_$tag; rm -rf "$O"; mkdir -p "$O"
run() { name=$1; shift; rocprofv3 --pmc $1 --output-format csv -d "$O/$name" -- python3 "${CMD[@]}" > "$O/$name.log" 2>&1 || echo "pass $name failed" >> "$O/summary.txt"; }
CMD=("$@")
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"
run fetch "FETCH_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "${CMD[@]}" > "$O/stats.log" 2>&1 || true
python3 - "$O" >> "$O/summary.txt" <<'PY'
import csv, glob, collections, statistics, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lg::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("void ", "").replace("(lg::DParams)", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-26s %.4g  (n=%d, max %.4g)" % (c, statistics.median(acc[k][c]), len(acc[k][c]), max(acc[k][c])))
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lg::" in r["Name"]:
            print("%-70s calls %s avg_ns %s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
PY
cat "$O/summary.txt"
