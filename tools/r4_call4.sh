set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c4; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "queue" > $O/pytest_queue.log 2>&1; rc=$?; echo "pytest(queue) rc=$rc"; tail -15 $O/pytest_queue.log
if [ $rc -eq 0 ]; then
for u in 1 2 4 8 16; do
LASGUN_QUEUE_UNIT=$u timeout -k 10 300 python tools/bench_configs.py --org=queue "2G" "4 mesh" "4m" > $O/configs_queue_u$u.jsonl 2>$O/configs_queue.err; echo "configs(queue u=$u) rc=$?"
done
timeout -k 10 300 python tools/bench_configs.py --org=queue "3 sph" "5 mixed" "1b" "2P" > $O/configs_queue_rest.jsonl 2>/dev/null
timeout -k 10 300 python tools/bench_configs.py "2G" "4 mesh" "4m" > $O/configs_default.jsonl 2>/dev/null; echo "configs rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c4/configs_*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("/")[-1], d["config"], d["ms"], d["kernels_ms"])
PY
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
