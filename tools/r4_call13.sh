set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
for rep in 1 2; do for v in 1 0; do for org in megakernel queue wavefront; do
LASGUN_ACCEL_LDS=$v timeout -k 10 300 python tools/bench_configs.py --org=$org "4 mesh" "4m" "5 mixed" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin: d=json.loads(l); print('accel_lds=$v', '$org', d['config'], d['ms'])"
done; done; done
LASGUN_PRUNE=1 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline with prune forced', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_avg'])"
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline default', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_avg'])"
