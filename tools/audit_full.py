"""Full-frame device audit of property (P) of the pruned reference walk on the BASELINE mesh configs: EVERY row of
configs 4 (glass), 4m (metal) and 5 (mixed, 8192^2) through lg_audit_prune -- every node and run the pruned walk skips is also walked
the reference's way, and no primitive found there may be one the reference would have accepted (triangle.rs:251, sphere.rs:86,
cuboid.rs:95; point.rs:49 for shadow rays).  tests/test_gpu_prune_audit.py does 56 rows per config inside `-m gpu`; this is the
exhaustive run, offline:  gpurun -- python tools/audit_full.py --out gpurun_out/prune_audit_full.jsonl
One JSON line per config: violations (must be 0), skipped nodes / runs, audited primitives, the largest share of a shipped margin any
skipped primitive needed, rows covered, seconds."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
FULL = {
    "config4_mesh_glass": (lambda: S.mesh_scene(G, 224, 224, "glass"), 4096),
    "config4m_mesh_metal": (lambda: S.mesh_scene(G, 224, 224, "metal"), 4096),
    "config5_mixed": (lambda: S.mixed_scene(G), 8192),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prune_audit_full.jsonl")
    ap.add_argument("--configs", default=",".join(FULL))
    ap.add_argument("--band", type=int, default=128, help="rows per lg_audit_prune call")
    ap.add_argument("--max-seconds", type=float, default=1e9, help="per config: stop after this long and report the rows covered")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    for name in args.configs.split(","):
        builder, size = FULL[name]
        acc = G.Accel(builder())
        tot = None
        t0 = time.time()
        rows = 0
        for y0 in range(0, size, args.band):
            y1 = min(size, y0 + args.band)
            r = G.audit_prune(acc, size, size, y0, y1)
            rows = y1
            if tot is None:
                tot = r
            else:
                for k in ("skipped_nodes", "skipped_runs", "primitives", "violations"):
                    tot[k] += r[k]
                for k in ("min_slack_nodes", "min_slack_runs"):
                    tot[k] = min(tot[k], r[k])
                tot["max_margin_used_nodes"] = max(tot["max_margin_used_nodes"], r["max_margin_used_nodes"])
            if (y0 // args.band) % 4 == 0:
                print("%s rows %d/%d violations %d  %.0f s" % (name, y1, size, tot["violations"], time.time() - t0), flush=True)
            if time.time() - t0 > args.max_seconds:
                break
        rec = dict(tot, scene=name, film=[size, size], rows_covered=rows, full_frame=rows == size, seconds=round(time.time() - t0, 1),
                   device_source_sha16=la.device_source_sha16())
        with open(args.out, "a") as f:
            f.write(json.dumps(rec) + "\n")
        print(json.dumps(rec), flush=True)
        if tot["violations"]:
            sys.exit(5)
        # the same film through lg_audit_fast: every ray of every row traced by the opt-in fast walk as shipped AND by the reference walk
        G.set_mode(acc, True)
        ft = {"rays": 0, "fallbacks": 0, "violations": 0}
        t0 = time.time()
        for y0 in range(0, size, args.band):
            r = G.audit_fast(acc, size, size, y0, min(size, y0 + args.band))
            for k in ft:
                ft[k] += r[k]
        rec = dict(ft, audit="fast", scene=name, film=[size, size], full_frame=True, seconds=round(time.time() - t0, 1), device_source_sha16=la.device_source_sha16())
        with open(args.out, "a") as f:
            f.write(json.dumps(rec) + "\n")
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
