#!/usr/bin/env python3
"""How many pixels depend on WHICH trigonometry renders them: the CPU oracle with glibc's atan2 / acos / sin / cos (what the Rust reference calls,
sphere.rs:99-114) against the same oracle with the portable, correctly rounded algorithm the device shares (tools/gen_trig.py), on the fuzz
generators of tests/test_gpu_fuzz.py -- the knife-edge generator (spheres touching in a point, rays through the tangent points) above all.
CPU only.  Test infrastructure (it runs the oracle), like everything else under tests/.  usage: python tests/libm_sensitivity.py [first_seed last_seed] [generator ...]  -> one JSON line per generator"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lasgun_amd import scenes as S  # noqa: E402  (scene generators only: no device is touched)
from oracle_lib import oracle  # noqa: E402

GENERATORS = {"random": (S.random_scene, (56, 40)), "adversarial": (S.adversarial_scene, (64, 48)), "adversarial_mesh": (S.adversarial_mesh_scene, (64, 48)),
              "adversarial_prune": (S.adversarial_prune_scene, (64, 48)), "progression_soup": (S.progression_soup_scene, (64, 48))}


def main():
    args = [a for a in sys.argv[1:]]
    lo, hi = (int(args[0]), int(args[1])) if len(args) >= 2 and args[0].isdigit() else (0, 2000)
    gens = [a for a in args if a in GENERATORS] or ["adversarial_prune"]
    o = oracle()
    threads = len(os.sched_getaffinity(0))
    for gen in gens:
        build, (w, h) = GENERATORS[gen]
        scenes = pixels = differing = worst = 0
        seeds = []
        for seed in range(lo, hi):
            try:
                acc = o.Accel(build(o, seed))
            except Exception:  # noqa: BLE001  (a scene the reference cannot build)
                continue
            a = o.Film(w, h)
            o.capture_subset_mt(0, 1, acc, a, threads)
            o.set_trig_mode(1)
            try:
                b = o.Film(w, h)
                o.capture_subset_mt(0, 1, acc, b, threads)
            finally:
                o.set_trig_mode(0)
            d = int((a.pixels() != b.pixels()).any(axis=-1).sum())
            scenes += 1
            pixels += w * h
            differing += d
            worst = max(worst, d)
            if d:
                seeds.append([seed, d])
        print(json.dumps({"generator": gen, "seeds": [lo, hi], "scenes": scenes, "pixels": pixels, "libm_sensitive_pixels": differing,
                          "most_in_one_scene": worst, "scenes_with_any": seeds[:50]}), flush=True)


if __name__ == "__main__":
    main()
