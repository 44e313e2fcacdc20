set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c6; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/queue_levels.py glass > $O/levels.jsonl 2>$O/levels.err; cat $O/levels.jsonl
LASGUN_SLAB_SIGNS=0 timeout -k 10 300 python tools/bench_configs.py --org=queue "4 mesh" "4m" > $O/nosigns.jsonl 2>/dev/null; python -c "
import json
for l in open('$O/nosigns.jsonl'): d=json.loads(l); print('nosigns', d['config'], d['ms'])"
bash tools/pmc_cmd.sh q4 tools/bench_configs.py --org=queue "4 mesh" > $O/pmc.log 2>&1; tail -45 gpurun_out/pmcc_q4/summary.txt
