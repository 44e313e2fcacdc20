set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c14; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for w in 0 4; do
LASGUN_MEGA_WAVES=$w timeout -k 10 300 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin: d=json.loads(l); print('mega_waves=$w default', d['config'], d['ms'], d['kernels_ms'])"
done
LASGUN_MEGA_WAVES=3 timeout -k 10 300 python tools/bench_configs.py --org=megakernel "4 mesh" "4m" "5 mixed" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin: d=json.loads(l); print('w3 megakernel', d['config'], d['ms'])"
