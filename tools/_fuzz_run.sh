set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/fuzz4; mkdir -p $O
t0=$(date +%s)
LASGUN_FUZZ_SEEDS=61000:63600 LASGUN_FUZZ_LOG=$O/fuzz_c.jsonl timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "progression_soup or random" > $O/fuzz_c.log 2>&1; echo "fuzz rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 $O/fuzz_c.log
LASGUN_FUZZ_FILM=97x61 LASGUN_FUZZ_SEEDS=70000:71500 LASGUN_FUZZ_LOG=$O/fuzz_d.jsonl timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz_d.log 2>&1; echo "fuzz(97x61) rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 $O/fuzz_d.log
cat $O/fuzz_c.jsonl $O/fuzz_d.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['generator'], d['film'], d['scenes'], d['renders'], len(d['mismatches']), d['libm_sensitive_pixels'], d['fast_refused'])"
