#!/usr/bin/env python3
"""Counter-example search for the opt-in fast mode: scenes built to stress its pruning margin (giant spheres as walls,
rays grazing them, lights and camera close to surfaces, boxes with extreme aspect, coincident and nearly coincident
primitives), fast traversal vs the reference traversal on the GPU (films and radiance bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lasgun_amd as la
G = la.api; S = la.scenes
M = G.Material

def scene(seed):
    rng = np.random.default_rng(seed)
    sc = G.Scene.new()
    cam = sc.set_perspective_camera(float(rng.uniform(20, 100)))
    eye = rng.uniform(-1, 1, 3) * [1.5, 1.0, 1.0] + [0, 0, 4.5]
    cam.look_at(eye.tolist(), (rng.uniform(-0.5, 0.5, 3)).tolist(), [0, 1, 0])
    sc.set_ambient_light([0.1, 0.1, 0.1])
    R = float(10.0 ** rng.uniform(2, 6))           # giant wall spheres
    mats = [M.matte(rng.uniform(0.2, 1, 3).tolist(), 0.0), M.plastic(rng.uniform(0.2, 1, 3).tolist(), [0.5, 0.5, 0.5], 0.3)]
    root = sc.root
    for axis, sign in ((1, -1), (1, 1), (0, -1), (0, 1), (2, -1)):
        c = [0.0, 0.0, 0.0]; c[axis] = sign * (R + 2.0)
        root.add_sphere(c, R, mats[int(rng.integers(2))])
    n = int(rng.integers(20, 200))
    for i in range(n):
        c = rng.uniform(-1.8, 1.8, 3)
        r = float(10.0 ** rng.uniform(-3, -0.5))
        root.add_sphere(c.tolist(), r, mats[i % 2])
        if rng.random() < 0.1:   # nearly coincident twin
            root.add_sphere((c + rng.uniform(-1, 1, 3) * 1e-9).tolist(), r * (1 + float(rng.uniform(-1, 1)) * 1e-9), mats[(i + 1) % 2])
    for i in range(int(rng.integers(0, 12))):
        lo = rng.uniform(-1.8, 1.2, 3); d = 10.0 ** rng.uniform(-4, 0, 3)
        root.add_box(lo.tolist(), (lo + d).tolist(), mats[i % 2])
    for i in range(int(rng.integers(1, 4))):      # lights, some very close to a wall
        p = rng.uniform(-1.9, 1.9, 3)
        if rng.random() < 0.5: p[1] = 2.0 - 10.0 ** rng.uniform(-6, -1)
        sc.add_point_light(p.tolist(), rng.uniform(0.2, 0.9, 3).tolist(), [1.0, 0.0, 0.0])
    return sc

def sliver_obj(rng, n):
    """OBJ text: needle and sliver triangles, nearly edge-on fans, duplicated vertices."""
    lines = []
    for i in range(n):
        c = rng.uniform(-1.2, 1.2, 3)
        kind = rng.integers(4)
        if kind == 0:   # needle
            a = c; b = c + rng.uniform(-1, 1, 3); d = a + (b - a) * 0.5 + rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(-9, -3)
        elif kind == 1: # tiny
            a = c; b = c + rng.uniform(-1, 1, 3) * 1e-6; d = c + rng.uniform(-1, 1, 3) * 1e-6
        elif kind == 2: # big, axis aligned (edge-on for axis-parallel rays)
            a = c; b = c + [float(rng.uniform(0.2, 1.5)), 0.0, 0.0]; d = c + [0.0, float(rng.uniform(0.2, 1.5)), 0.0]
        else:           # generic
            a = c; b = c + rng.uniform(-0.6, 0.6, 3); d = c + rng.uniform(-0.6, 0.6, 3)
        for v in (a, b, d):
            lines.append("v %.9g %.9g %.9g" % tuple(v))
        lines.append("f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    return "\n".join(lines) + "\n"


def scene2(seed):
    """Meshes of degenerate triangles under extreme nested transforms, orthographic or distant cameras."""
    rng = np.random.default_rng(seed + 100000)
    sc = G.Scene.new()
    if rng.random() < 0.3:
        cam = sc.set_orthographic_camera(float(rng.uniform(2, 6)))
    else:
        cam = sc.set_perspective_camera(float(rng.uniform(5, 90)))
    dist = float(10.0 ** rng.uniform(0.5, 3))
    eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * dist
    cam.look_at(eye.tolist(), (rng.uniform(-0.3, 0.3, 3)).tolist(), [0, 1, 0])
    sc.set_ambient_light([0.2, 0.2, 0.2])
    mats = [M.matte(rng.uniform(0.2, 1, 3).tolist(), 0.0), M.plastic(rng.uniform(0.2, 1, 3).tolist(), [0.5, 0.5, 0.5], 0.3)]
    mesh = sc.parse_obj(sliver_obj(rng, int(rng.integers(20, 400))))
    root = sc.root
    for k in range(int(rng.integers(1, 4))):
        g = G.Aggregate.new()
        g.scale(float(10.0 ** rng.uniform(-2, 1)), float(10.0 ** rng.uniform(-2, 1)), float(10.0 ** rng.uniform(-2, 1)))
        ax = rng.normal(size=3)
        if rng.random() < 0.9: ax = ax / np.linalg.norm(ax)  # a non-unit axis makes transform and inverse disagree: fast mode is refused
        g.rotate(float(rng.uniform(0, 360)), ax.tolist())
        g.translate(rng.uniform(-1, 1, 3).tolist())
        g.add_obj_of(mesh, mats[k % 2])
        if rng.random() < 0.5:
            inner = G.Aggregate.new()
            inner.rotate_x(float(rng.uniform(0, 360)))
            inner.scale(float(10.0 ** rng.uniform(-1, 1)), 1.0, 1.0)
            for i in range(int(rng.integers(1, 30))):
                inner.add_sphere(rng.uniform(-1, 1, 3).tolist(), float(10.0 ** rng.uniform(-3, -0.3)), mats[i % 2])
            g.add_group(inner)
        root.add_group(g)
    for i in range(int(rng.integers(0, 40))):
        root.add_sphere(rng.uniform(-1.5, 1.5, 3).tolist(), float(10.0 ** rng.uniform(-3, -0.5)), mats[i % 2])
    for i in range(int(rng.integers(1, 3))):
        sc.add_point_light((rng.normal(size=3) * dist * 0.7).tolist(), rng.uniform(0.3, 0.9, 3).tolist(), [1.0, 0.0, 0.0])
    return sc


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
