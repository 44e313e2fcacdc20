"""The committed bench line (profiles/r03_bench.json, produced by `python bench.py` on the GPU box) carries
every field of the driver's contract; BASELINE.json's metric string is the one bench.py prints."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        rows = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(rows[-1])


def test_bench_line_has_the_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f64" and d["unit"] == "Mrays/s" and d["value"] > 100.0  # north-star floor: 100 Mrays/s
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the resource that binds the dominant kernel is named: f64 VALU issue (the SURVEY 8(d) byte figure sits beside it)
    assert r["bound"] == "valu_f64" and r["unit"] == "TFLOP/s" and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic_source"]
    assert "hbm_algorithmic" in r and "lds" in r and r["hbm_algorithmic"]["measured_copy_peak"] > 1000.0
    assert d["latency_ms"] > 0.0 and d["config"]["rays_per_frame"] == d["config"]["primary"] + d["config"]["shadow"]
    # the line proves its own "bit-exact RGBA8 vs CPU": the timed frame against the oracle film, in the same run
    assert d["bit_exact"] is True and d["mismatched_bytes"] == 0 and d["bit_exact_check"]["checked_pixels"] >= 65536
    assert d["value_single_frame"] > 100.0 and abs(d["value_single_frame"] - d["config"]["rays_per_frame"] / d["latency_ms"] / 1e3) < 1e-6 * d["value_single_frame"]
    # the spread of the run: the same K steps timed three times back to back, the first of them being `value`
    assert len(d["value_repeats"]) == 3 and d["value_repeats"][0] == d["value"] and min(d["value_repeats"]) > 0.9 * d["value"]
    m = d["roofline_mesh"]
    assert m["bound"] == "valu_f64" and 0.0 < m["frac"] < 1.0 and m["work_per_frame"]["triangles_tested"] > 0 and m["ms_per_frame"] > 0.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["cpu"]


def test_bench_metric_is_baselines_metric():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    want = base["metric"].replace("×", "x")
    assert _line()["metric"] == want
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert want in src


def test_bench_gpus_n_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with no launcher in front spawns torch.distributed.run as a child and hands its exit code through (no GPU
    here: the ranks fail at their first device call, and that failure -- not a usage message -- is what comes back)."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr[-2000:]
        return
    assert p.returncode != 0 and "launch with" not in p.stderr
    assert "torch.distributed" in p.stderr or "ChildFailedError" in p.stderr or "rank" in p.stderr.lower(), p.stderr[-2000:]


def test_round6_bench_line_carries_the_mixed_config_and_the_clock():
    """profiles/r06_bench.json (the same command on the GPU box in round 6): `roofline_mixed` -- BASELINE configs[4], 8192^2, whole on one GPU -- beside
    `roofline` and `roofline_mesh`, and in each block `clock_ghz` / `frac_at_clock` (null until the PMC passes of these sources are committed)."""
    with open(os.path.join(ROOT, "profiles", "r06_bench.json")) as f:
        d = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
    assert d["bit_exact"] is True and d["value"] > 4000.0 and d["n_gpus"] == 1
    for key in ("roofline", "roofline_mesh", "roofline_mixed"):
        b = d[key]
        assert "clock_ghz" in b and "frac_at_clock" in b and 0.0 < b["frac"] < 1.0, key
    m = d["roofline_mixed"]
    assert "configs[4]" in m["workload"] and m["rays_per_frame"] == 2 * 8192 * 8192 - 2 and m["work_per_frame"]["triangles_tested"] > 0
    assert m["kernel_launches_per_frame"] >= 1 and abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-9 and m["frame_frac"] < m["frac"] + 1e-9
    with open(os.path.join(ROOT, "profiles", "r06_pmc.json")) as f:
        pmc = json.load(f)
    clocks = [v["median"] for v in pmc["clock_ghz"].values()]
    assert clocks and all(1.5 < c < 2.6 for c in clocks)  # GRBM_GUI_ACTIVE / 8 / duration
