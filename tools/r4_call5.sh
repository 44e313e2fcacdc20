set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c5; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "queue" > $O/pytest_queue.log 2>&1; rc=$?; echo "pytest(queue) rc=$rc"; tail -15 $O/pytest_queue.log
if [ $rc -eq 0 ]; then
for ord in 0 1; do for u in 1 4; do
LASGUN_QUEUE_ORDER=$ord LASGUN_QUEUE_UNIT=$u timeout -k 10 300 python tools/bench_configs.py --org=queue "4 mesh" "4m" "5 mixed" > $O/configs_queue_o${ord}_u$u.jsonl 2>$O/configs_queue.err; echo "configs(queue order=$ord u=$u) rc=$?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c5/configs_*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("/")[-1], d["config"], d["ms"], d["kernels_ms"])
PY
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
