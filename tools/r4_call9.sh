set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c9; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python tools/bench_configs.py --progressive=100 "3 sph" "2G" "2P" "1a" > $O/progressive.jsonl 2>$O/progressive.err; cat $O/progressive.jsonl
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c9/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["bit_exact"])
m=d["roofline_mesh"]; print({k:m[k] for k in ("ms_per_frame","kernel_ms_avg","frac","kernel","megakernel")}, m["plain_walk"]["ms_per_frame"])
PY
