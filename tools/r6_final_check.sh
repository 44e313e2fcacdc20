#!/bin/bash
# round 6, last GPU call: what the driver runs at round end -- the GPU suite, smoke(), bench.py -- on the tree as committed; plus bench.py's other forms for the record
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6final; mkdir -p "$O"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?"; tail -2 "$O/gputest.log"
timeout -k 10 300 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$O/smoke.log"
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > "$O/bench_gloo2.json" 2> "$O/bench_gloo2.err"; echo "bench --gpus 2 (gloo, no launcher) rc=$?"
LASGUN_MULTI_FORCE_RCCL=1 timeout -k 10 300 python3 bench.py --gpus 2 --collective library --steps 5 --warmup 2 > "$O/bench_library2.json" 2> "$O/bench_library2.err"; echo "bench --collective library rc=$?"
timeout -k 10 400 python3 bench.py --workload configs4 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > "$O/bench_configs4.json" 2> "$O/bench_configs4.err"; echo "bench configs4 rc=$?"
python3 - "$O" <<'PY'
import json, sys
for n in ("bench.json", "bench_gloo2.json", "bench_library2.json", "bench_configs4.json"):
    try:
        d = json.loads(open(sys.argv[1] + "/" + n).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "unreadable", e); continue
    r = d.get("roofline") or {}
    print(n, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "n_gpus", d["n_gpus"], "frac", r.get("frac"), "clock", r.get("clock_ghz"), "frac_at_clock", r.get("frac_at_clock"),
          "traffic", r.get("traffic"), "bit_exact", d.get("bit_exact"), "gathered", d.get("gathered_equals_single_gpu"), "gather_ms", d.get("gather_ms"))
PY
