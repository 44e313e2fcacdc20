set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/fuzz4; mkdir -p $O
t0=$(date +%s)
LASGUN_FUZZ_SEEDS=60000:60600 LASGUN_FUZZ_LOG=$O/fuzz_a.jsonl timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_a.log 2>&1; echo "fuzz rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 $O/fuzz_a.log
cat $O/fuzz_a.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['generator'], d['scenes'], d['renders'], len(d['mismatches']), d['libm_sensitive_pixels'], d['fast_refused'])"
