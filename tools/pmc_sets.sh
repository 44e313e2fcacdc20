#!/bin/bash
# Counter passes of one's own choosing over a python command (each quoted set is its own run; no trace flags with --pmc):
#   bash tools/pmc_sets.sh <tag> "<set 1>" ["<set 2>" ...] -- tools/bench_configs.py "4 mesh"     -> gpurun_out/pmcs_<tag>/summary.txt
set -eu
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
tag=$1; shift
O=gpurun_out/pmcs_$tag; rm -rf "$O"; mkdir -p "$O"
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
i=0
for s in "${sets[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $s --output-format csv -d "$O/set$i" -- python3 "$@" > "$O/set$i.log" 2>&1 || echo "pass $i failed" >> "$O/summary.txt"
done
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
        print("   %-28s %.4g  (n=%d, max %.4g)" % (c, statistics.median(acc[k][c]), len(acc[k][c]), max(acc[k][c])))
PY
cat "$O/summary.txt"
