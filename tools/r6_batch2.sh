#!/bin/bash
# round 6, GPU batch 2: the refilling shadow pass -- its parity test first (a hang stops the batch), then the suite, then the A/B on the headline
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6b2; mkdir -p "$O"
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "refilling" > "$O/refill_test.log" 2>&1; rc=$?; echo "refill test rc=$rc"; tail -3 "$O/refill_test.log"
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 120 python tools/bench_configs.py "3 sph" > "$O/first_frame.jsonl" 2> "$O/first_frame.err"; rc=$?; echo "config 3 rc=$rc"; cut -c1-260 "$O/first_frame.jsonl"
if [ $rc -ne 0 ]; then exit 1; fi
# A/B, variants in turn: no refill, refill at 16 (shipped), 8, 32, 48 lanes done
for r in 1 2 3 4; do
  LASGUN_REFILL=0 python tools/bench_configs.py "3 sph" 2>/dev/null | sed "s/^{/{\"variant\": \"no refill\", \"round\": $r, /"
  for v in main r8 r32 r48; do
    if [ "$v" = main ]; then unset LASGUN_HIP_LIB; else export LASGUN_HIP_LIB=lasgun_amd/liblasgun_hip_$v.so; fi
    python tools/bench_configs.py "3 sph" 2>/dev/null | sed "s/^{/{\"variant\": \"refill $v\", \"round\": $r, /"
  done
  unset LASGUN_HIP_LIB
done > "$O/ab_refill.jsonl"
echo "ab refill done"
python - "$O/ab_refill.jsonl" <<'PY'
import json, sys, collections, statistics
a = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); a[d["variant"]].append((d["ms"], d["kernels_ms"].get("trace<shadow>")))
for k, v in a.items(): print(k, "frame ms", [x[0] for x in v], "shadow ms", [x[1] for x in v])
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?"; tail -2 "$O/gputest.log"
# lane use of the shadow pass, before / after (one counter set per run)
LASGUN_REFILL=0 bash tools/pmc_sets.sh norefill "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" -- tools/bench_configs.py "3 sph" > /dev/null 2>&1; cp gpurun_out/pmcs_norefill/summary.txt "$O/pmc_norefill.txt"
bash tools/pmc_sets.sh refill "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" -- tools/bench_configs.py "3 sph" > /dev/null 2>&1; cp gpurun_out/pmcs_refill/summary.txt "$O/pmc_refill.txt"
timeout -k 10 500 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
LASGUN_REFILL=0 timeout -k 10 500 python3 bench.py --no-cpu-baseline > "$O/bench_norefill.json" 2> "$O/bench_norefill.err"; echo "bench (no refill) rc=$?"
python3 - "$O" <<'PY'
import json, sys
for n in ("bench.json", "bench_norefill.json"):
    d = json.loads(open(sys.argv[1] + "/" + n).read().strip().splitlines()[-1])
    print(n, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "kernels", d["roofline"]["kernels_ms_avg"], "bit_exact", d["bit_exact"], "kernel", d["roofline"]["kernel"])
PY
