# LDS / SQ counter passes on the bench for the traversal kernels (each --pmc set is its own run; no trace flags with --pmc)
set -eu
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16  # (the profiler initialises HIP before the program can set it)
cd "${GRAFT_REPO_ROOT:?}"
run() { name=$1; shift; LASGUN_PACKET=${LASGUN_PACKET:-0} rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcl_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcl_$name.log 2>&1; }
run a SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob("gpurun_out/pmcl_%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "stream_trace" not in k and "stream_packet" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k)
            for c, v in sorted(acc[k].items()): print("   %-28s %.4g per launch" % (c, v / n[(k, c)]))
PY
