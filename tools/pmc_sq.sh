# SQ / cache counter passes on the bench (each --pmc set is its own run; no trace flags with --pmc)
set -eu
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16  # (the profiler initialises HIP before the program can set it)
cd "${GRAFT_REPO_ROOT:?}"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
for d in sq1 sq2 sq3 tcc tcp; do f=$(find gpurun_out/pmc_$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep 'trace_kernel<false>' $f | awk -F, '{n=NF; print $(n-3), $(n-2)}' | sort | uniq -c | awk '{print $2, $3}' | sort -u; done
