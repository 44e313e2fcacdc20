set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c7; rm -rf $O; mkdir -p $O
for w in 3 4 5; do
  lib=$PWD/lasgun_amd/liblasgun_hip_w$w.so; [ $w = 4 ] && lib=$PWD/lasgun_amd/liblasgun_hip.so
  for org in queue megakernel; do
    LASGUN_QUEUE_UNIT=1 LASGUN_QUEUE_ORDER=0 LASGUN_HIP_LIB=$lib timeout -k 10 200 python tools/bench_configs.py --org=$org "4 mesh" "4m" "2G" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin: d=json.loads(l); print('waves=$w', '$org', d['config'], d['ms'])"
  done
done
