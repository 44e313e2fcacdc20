#!/bin/bash
# round 6, GPU batch 3: what the correctly rounded trig costs where it is inlined (main), as a call (trigcall), against the old algorithm (oldtrig)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6b3; mkdir -p "$O"
timeout -k 10 1000 bash tools/ab_configs.sh "1b 3 4 5" 4 main oldtrig trigcall > "$O/ab_trig_all.jsonl" 2>/dev/null; echo "ab rc=$?"
python3 - "$O/ab_trig_all.jsonl" <<'PY'
import json, sys, collections, statistics
a = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); a[(d["config"][:12], d["lib"])].append(d["ms"])
for k in sorted(a): print(k, [round(x, 3) for x in a[k]], "median", round(statistics.median(a[k]), 3))
PY
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "refilling or trig or kat" > "$O/tests.log" 2>&1; tail -2 "$O/tests.log"
