set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c3; rm -rf $O; mkdir -p $O
LASGUN_DEBUG=1 timeout -k 10 120 python tools/bench_configs.py --org=queue "3 sph" "2G" > $O/dbg.jsonl 2> $O/dbg.err; grep -a "lasgun\]" $O/dbg.err | sort | uniq -c | head
run() { name=$1; shift; pmc=$1; shift; rocprofv3 --pmc $pmc --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.log" 2>&1 || echo "pass $name failed"; }
for org in queue megakernel; do
  run ${org}_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" tools/bench_configs.py --org=$org "3 sph" "4m"
  run ${org}_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" tools/bench_configs.py --org=$org "3 sph" "4m"
done
python3 - "$O" <<'PY'
import csv, glob, collections, statistics, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lg::" in r["Kernel_Name"] and ("queue_kernel" in r["Kernel_Name"] or "trace_kernel<false" in r["Kernel_Name"]):
            acc[r["Kernel_Name"].replace("void ", "").replace("(lg::DParams)", "") + " grid=" + r.get("Grid_Size","?") ][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-26s %.4g  (n=%d, max %.4g)" % (c, statistics.median(acc[k][c]), len(acc[k][c]), max(acc[k][c])))
PY
