#!/bin/bash
# round 6: ONE capped fuzz campaign on the kernels as shipped (five generators x 1,200 seeds = 6,000 scenes; the libm budget at 1 pixel per scene)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6fuzz; mkdir -p "$O"
LASGUN_FUZZ_SEEDS=${1:-7000:8200} LASGUN_FUZZ_LOG="$O/fuzz.jsonl" timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > "$O/fuzz.log" 2>&1; echo "fuzz rc=$?"
tail -3 "$O/fuzz.log"; cut -c1-300 "$O/fuzz.jsonl"
