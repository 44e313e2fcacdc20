set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c1; rm -rf $O; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python bench.py > $O/bench_signs.json 2> $O/bench_signs.err; echo "bench rc=$?"
LASGUN_SLAB_SIGNS=0 timeout -k 10 200 python bench.py --no-extras > $O/bench_nosigns.json 2> $O/bench_nosigns.err; echo "bench(nosigns) rc=$?"
timeout -k 10 300 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null; echo "configs rc=$?"
LASGUN_SLAB_SIGNS=0 timeout -k 10 300 python tools/bench_configs.py > $O/configs_nosigns.jsonl 2>/dev/null; echo "configs(nosigns) rc=$?"
python - <<'PY'
import json
for f in ("bench_signs","bench_nosigns"):
    try:
        d=json.loads(open("gpurun_out/r4c1/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("kernel_ms_avg"), d.get("bit_exact"))
    except Exception as e: print(f, "ERR", e)
for f in ("configs","configs_nosigns"):
    for l in open("gpurun_out/r4c1/%s.jsonl"%f):
        d=json.loads(l); print(f, d["config"], d["ms"], d["kernels_ms"])
PY
