set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c12; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for org in default megakernel wavefront queue; do
timeout -k 10 300 python tools/bench_configs.py --org=$org "2G" "4 mesh" "4m" "5 mixed" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin: d=json.loads(l); print('$org', d['config'], d['ms'], d['kernels_ms'])"
done
