#!/bin/bash
# round 6, GPU batch 4: the suite on the split host units; the sincos of phi and theta as ONE inlined instance (trigshared) against two (main) and the old algorithm
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6b4; mkdir -p "$O"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?"; tail -2 "$O/gputest.log"
timeout -k 10 1000 bash tools/ab_configs.sh "1b 2G 3 4 5" 4 main oldtrig trigshared > "$O/ab_trig_shared.jsonl" 2>/dev/null; echo "ab rc=$?"
python3 - "$O/ab_trig_shared.jsonl" <<'PY'
import json, sys, collections, statistics
a = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); a[(d["config"][:12], d["lib"])].append(d["ms"])
for k in sorted(a): print(k, [round(x, 3) for x in a[k]], "median", round(statistics.median(a[k]), 3))
PY
