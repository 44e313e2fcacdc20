#!/usr/bin/env python3
"""Counter-example search for the opt-in fast mode: scenes built to stress its pruning margin (giant spheres as walls,
rays grazing them, lights and camera close to surfaces, boxes with extreme aspect, coincident and nearly coincident
primitives), fast traversal vs the reference traversal on the GPU (films and radiance bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
M = G.Material

def scene(seed):
    return S.adversarial_scene(G, seed)


def scene2(seed):
    return S.adversarial_mesh_scene(G, seed)


def bits(x):
    x = np.ascontiguousarray(x, dtype=np.float64); u = x.view(np.uint64).copy(); u[np.isnan(x)] = np.uint64(0x7FF8000000000000); return u

a, b = int(sys.argv[1]), int(sys.argv[2]); w, h = 128, 96
gen = scene2 if len(sys.argv) > 3 and sys.argv[3] == "mesh" else scene
bad = 0
for seed in range(a, b):
    try:
        acc = G.Accel(gen(seed))
    except la.LasgunError as e:
        continue
    outs = []
    try:
        G.set_mode(acc, True)
    except la.LasgunError:
        refused = globals().get("refused", 0) + 1; globals()["refused"] = refused
        continue
    for fast in (False, True):
        G.set_mode(acc, fast)
        f = G.Film(w, h); G.capture_subset(0, 1, acc, f)
        outs.append((f.pixels(), bits(G.capture_radiance(acc, w, h))))
    if not (np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])):
        bad += 1
        print("MISMATCH seed", seed, "bytes", int((outs[0][0] != outs[1][0]).sum()), "radiance words", int((outs[0][1] != outs[1][1]).sum()), flush=True)
print("adversarial seeds", a, b, "mismatches", bad, "fast mode refused for", globals().get("refused", 0))
