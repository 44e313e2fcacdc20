#!/bin/bash
# Round-end evidence run on the 1-GPU box: bench, rocprofv3 kernel stats, PMC passes, all configs.
# usage (gpurun, two calls of <= 20 minutes): bash tools/final_profile.sh a ; bash tools/final_profile.sh b  -> gpurun_out/final6/ ;
# then python3 tools/summarize_profiles.py gpurun_out/final6 profiles r06
set -u
export TMPDIR=/tmp
# the profiler's preloaded library initialises HIP before bench.py can set this: the frames in flight need a hardware queue each
export GPU_MAX_HW_QUEUES=16
cd "${GRAFT_REPO_ROOT:?}"
part=${1:-a}
O=gpurun_out/final6; mkdir -p "$O"
if [ "$part" = a ]; then
timeout -k 10 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
# per-kernel durations: frames one after the other (the form bench.py's roofline.kernel_ms_avg is measured in) ...
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- python3 bench.py --steps 5 --warmup 1 --no-extras --sequential > "$O/prof.log" 2>&1; echo "stats rc=$?"
# ... and the default command (consecutive frames overlap, four in flight: kernels of several frames share the chip, durations stretch)
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_overlap" -- python3 bench.py --steps 5 --warmup 1 --no-extras > "$O/prof_overlap.log" 2>&1; echo "stats(overlap) rc=$?"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc_$name" -- python3 bench.py --steps 2 --warmup 1 --no-extras > "$O/pmc_$name.log" 2>&1; echo "pmc $name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# the clock the chip holds under these kernels: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration (MI355X_MICROARCH.md, DVFS give-back);
# frames one after the other, so a dispatch's duration is its own; --kernel-trace beside --pmc gives the timestamps (no other trace domain with counters)
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_grbm" -- python3 bench.py --steps 3 --warmup 1 --no-extras --sequential > "$O/pmc_grbm.log" 2>&1; echo "pmc grbm rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_grbm_c4" -- python3 tools/bench_configs.py "4 mesh" "5 mixed" > "$O/pmc_grbm_c4.log" 2>&1; echo "pmc grbm c4/c5 rc=$?"
# configs[4] (8192^2 mixed) and configs[3]: per-kernel durations of their own
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c5" -- python3 tools/bench_configs.py "5 mixed" > "$O/prof_c5.log" 2>&1; echo "stats c5 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4" -- python3 tools/bench_configs.py "4 mesh" > "$O/prof_c4.log" 2>&1; echo "stats c4 rc=$?"
timeout -k 10 300 python3 tools/bench_configs.py > "$O/configs_parity.jsonl" 2>/dev/null; echo "configs rc=$?"
timeout -k 10 300 python3 tools/bench_configs.py --fast > "$O/configs_fast.jsonl" 2>/dev/null
timeout -k 10 300 python3 tools/bench_configs.py --org=megakernel > "$O/configs_megakernel.jsonl" 2>/dev/null
timeout -k 10 300 python3 tools/bench_configs.py --org=wavefront > "$O/configs_wavefront.jsonl" 2>/dev/null
LASGUN_PRUNE=0 timeout -k 10 300 python3 tools/bench_configs.py > "$O/configs_unpruned.jsonl" 2>/dev/null
LASGUN_PRUNE=1 timeout -k 10 300 python3 tools/bench_configs.py > "$O/configs_pruned.jsonl" 2>/dev/null
timeout -k 10 300 python3 tools/bench_configs.py --org=queue > "$O/configs_queue.jsonl" 2>/dev/null
timeout -k 10 200 python3 tools/share_time.py > "$O/share_time.jsonl" 2>/dev/null
timeout -k 10 200 python3 tools/bench_multi.py --devices 0,0 --steps 5 > "$O/multi_2x_same_device.json" 2>/dev/null
LASGUN_MULTI_FORCE_RCCL=1 timeout -k 10 200 python3 tools/bench_multi.py --devices 0,0 --steps 5 > "$O/multi_2x_same_device_rccl.json" 2>/dev/null
# config 4's kernel (the queue organisation's persistent launch): counter passes of its own
bash tools/pmc_cmd.sh c4 tools/bench_configs.py "4 mesh" > "$O/pmc_c4.log" 2>&1; cp gpurun_out/pmcc_c4/summary.txt "$O/config4_pmc.txt" 2>/dev/null
bash tools/pmc_cmd.sh c5 tools/bench_configs.py "5 mixed" > "$O/pmc_c5.log" 2>&1; cp gpurun_out/pmcc_c5/summary.txt "$O/config5_pmc.txt" 2>/dev/null
tail -c 400 "$O/bench.json"
exit 0
fi
timeout -k 10 400 python3 tools/bench_configs.py --progressive=100 > "$O/progressive.jsonl" 2>/dev/null
timeout -k 10 300 python3 tools/bench_configs.py --progressive=7 "3 sph" "2G" >> "$O/progressive.jsonl" 2>/dev/null
timeout -k 10 600 python3 tools/bench_configs.py --org-choice > "$O/org_choice.jsonl" 2>/dev/null; echo "org choice rc=$?"
timeout -k 10 200 python3 tools/host_capture_probe.py > "$O/host_capture.json" 2>/dev/null
LASGUN_CAPTURE_BANDS=1 timeout -k 10 200 python3 tools/host_capture_probe.py > "$O/host_capture_one_band.json" 2>/dev/null
timeout -k 10 300 python3 tools/queue_levels.py glass > "$O/queue_levels.jsonl" 2>/dev/null
timeout -k 10 600 python3 tools/ss_probe.py 256 512 1024 > "$O/ss_par.jsonl" 2>/dev/null; echo "ss_par rc=$?"
timeout -k 10 300 python3 tools/tail_probe.py "3 sph" "4 mesh" "4m" "5 mixed" > "$O/tail_probe.jsonl" 2>/dev/null
(timeout -k 10 100 python3 tools/host_capture_probe.py 512 simple; timeout -k 10 100 python3 tools/host_capture_probe.py 512 cornell_glass; timeout -k 10 100 python3 tools/host_capture_probe.py 768 spooky) > "$O/host_capture_examples.jsonl" 2>/dev/null
LASGUN_DEBUG_TIMES=1 timeout -k 10 200 python3 tools/build_times.py > "$O/build_times.log" 2>&1
(LASGUN_AUTOTUNE=0 timeout -k 10 600 python3 tools/ss_probe.py 256 512 1024; LASGUN_AUTOTUNE=0 timeout -k 10 300 python3 tools/bench_configs.py) > "$O/rule_first_launch.jsonl" 2>/dev/null
LASGUN_AUDIT_SEEDS=100:300 LASGUN_AUDIT_LOG="$O/prune_audit.jsonl" timeout -k 10 600 python3 -m pytest tests/test_gpu_prune_audit.py -m gpu -x -q > "$O/audit.log" 2>&1; tail -2 "$O/audit.log"
# every row of configs 4 / 4m / 5: the pruned walk's audit (lg_audit_prune) -- and fast mode's (lg_audit_fast) on the same films
timeout -k 10 900 python3 tools/audit_full.py --out "$O/prune_audit_full.jsonl" > "$O/prune_audit_full.log" 2>&1; echo "full audit rc=$?"
ls "$O" | wc -l
