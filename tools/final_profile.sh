# Round-end evidence run on the 1-GPU box: tests, bench, rocprofv3 kernel stats, PMC passes, all configs.
set -eu
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tcc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_tcc.log 2>&1
timeout -k 10 300 python3 tools/bench_configs.py > $O/configs_parity.jsonl 2>/dev/null
timeout -k 10 300 python3 tools/bench_configs.py --fast > $O/configs_fast.jsonl 2>/dev/null
python3 - "$O" > $O/pmc_summary.txt <<'PY'
import csv, glob, collections, sys
K = "stream_trace_kernel<false, true, true, false>"
print("# per-launch counters of lg::%s (shadow traversal, scene tables in LDS; 4096^2, config 3): every dispatch's value" % K)
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_tcc"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (sys.argv[1], d), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if K in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(r["Counter_Value"])
        for c, v in sorted(acc.items()):
            print(c, " ".join(v))
PY
cat $O/pmc_summary.txt; tail -1 $O/pytest_gpu.log; tail -1 $O/bench.log | cut -c1-400
