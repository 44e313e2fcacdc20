set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c2; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "queue" > $O/pytest_queue.log 2>&1; rc=$?; echo "pytest(queue) rc=$rc"; tail -15 $O/pytest_queue.log
if [ $rc -eq 0 ]; then
timeout -k 10 300 python tools/bench_configs.py --org=queue "2G" "4 mesh" "4m" "5 mixed" "3 sph" > $O/configs_queue.jsonl 2>$O/configs_queue.err; echo "configs(queue) rc=$?"
timeout -k 10 300 python tools/bench_configs.py "2G" "4 mesh" "4m" "5 mixed" > $O/configs_default.jsonl 2>/dev/null; echo "configs rc=$?"
python - <<'PY'
import json
for f in ("configs_queue","configs_default"):
    for l in open("gpurun_out/r4c2/%s.jsonl"%f):
        d=json.loads(l); print(f, d["config"], d["ms"], d["kernels_ms"])
PY
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
