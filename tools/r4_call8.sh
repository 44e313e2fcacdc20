set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c8; rm -rf $O; mkdir -p $O
LASGUN_AUDIT_LOG=$O/audit.jsonl timeout -k 10 900 python -m pytest tests/test_gpu_prune_audit.py -m gpu -x -q --durations=5 > $O/pytest_audit.log 2>&1; echo "pytest(audit) rc=$?"; tail -25 $O/pytest_audit.log; cat $O/audit.jsonl | head -5; python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r4c8/audit.jsonl")]
print(len(rows), "audits; violations", sum(r["violations"] for r in rows), "min slack nodes", min(r["min_slack_nodes"] for r in rows), "runs", min(r["min_slack_runs"] for r in rows))
PY
