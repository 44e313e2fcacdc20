#!/bin/bash
# SQ counter passes on bench.py's workload (each --pmc set is its own run; no trace flags with --pmc).
# usage (on the GPU box): bash tools/pmc_quick.sh <tag>   -> gpurun_out/pmcq_<tag>/summary.txt
set -eu
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16  # (the profiler initialises HIP before the program can set it)
tag=${1:-x}
O=gpurun_out/pmcq_$tag; rm -rf "$O"; mkdir -p "$O"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$O/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$O/$name.log" 2>&1 || echo "pass $name failed" >> "$O/summary.txt"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
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
        print("   %-26s %.4g  (n=%d)" % (c, statistics.median(acc[k][c]), len(acc[k][c])))
PY
cat "$O/summary.txt"
