set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/fuzz4; mkdir -p $O; rm -f $O/audit.jsonl
LASGUN_AUDIT_SEEDS=100:400 LASGUN_AUDIT_LOG=$O/audit.jsonl timeout -k 10 300 python3 -m pytest tests/test_gpu_prune_audit.py -m gpu -x -q > $O/audit.log 2>&1; tail -3 $O/audit.log
python3 - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/fuzz4/audit.jsonl")]
print(len(rows), "audits; violations", sum(r["violations"] for r in rows), "max margin used", max(r["max_margin_used_nodes"] for r in rows), "min slack nodes", min(r["min_slack_nodes"] for r in rows))
for r in rows[:4]: print(r["scene"], r["max_margin_used_nodes"], r["min_slack_nodes"])
PY
t0=$(date +%s)
LASGUN_FUZZ_SEEDS=61000:63600 LASGUN_FUZZ_LOG=$O/fuzz_b.jsonl timeout -k 10 850 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_b.log 2>&1; echo "fuzz rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 $O/fuzz_b.log
cat $O/fuzz_b.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['generator'], d['scenes'], d['renders'], len(d['mismatches']), d['libm_sensitive_pixels'], d['fast_refused'])"
