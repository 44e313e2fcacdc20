#!/bin/bash
# round 6, GPU batch 5: lg_audit_fast (tests + the meshes the gate refuses), the sincos instance A/B
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6b5; mkdir -p "$O"
LASGUN_AUDIT_LOG="$O/fast_audit.jsonl" timeout -k 10 600 python -m pytest tests/test_gpu_prune_audit.py -m gpu -x -q -k "fast" > "$O/audit_test.log" 2>&1; echo "fast audit tests rc=$?"; tail -4 "$O/audit_test.log"
# the meshes fast mode refuses (coordinates > 2^20 x the smallest triangle), with the gate taken out: what the audit finds there
LASGUN_FAST_NO_GATE=1 timeout -k 10 400 python3 - > "$O/fast_audit_ungated.jsonl" 2> "$O/fast_audit_ungated.err" <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import lasgun_amd as la
G, S = la.api, la.scenes
for gen in (S.progression_soup_scene, S.adversarial_mesh_scene):
    tot = {"generator": gen.__name__, "scenes": 0, "rays": 0, "fallbacks": 0, "violations": 0, "scenes_with_violations": []}
    for seed in range(0, 600):
        try:
            acc = G.Accel(gen(G, seed)); G.set_mode(acc, True)
        except la.LasgunError:
            continue
        r = G.audit_fast(acc, 64, 48)
        tot["scenes"] += 1
        for k in ("rays", "fallbacks", "violations"): tot[k] += r[k]
        if r["violations"]: tot["scenes_with_violations"].append([seed, r["violations"]])
    print(json.dumps(tot), flush=True)
PY
echo "ungated audit rc=$?"; cat "$O/fast_audit_ungated.jsonl" | cut -c1-400
timeout -k 10 1000 bash tools/ab_configs.sh "1b 3 4" 4 main oldtrig trigshared > "$O/ab_trig_shared.jsonl" 2>/dev/null; echo "ab rc=$?"
python3 - "$O/ab_trig_shared.jsonl" <<'PY'
import json, sys, collections, statistics
a = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); a[(d["config"][:12], d["lib"])].append(d["ms"])
for k in sorted(a): print(k, [round(x, 3) for x in a[k]], "median", round(statistics.median(a[k]), 3))
PY
