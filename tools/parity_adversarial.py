#!/usr/bin/env python3
"""The adversarial scene generators of fast_adversarial.py (giant spheres, needle boxes, degenerate meshes, nested
anisotropic transforms, non-unit rotation axes, distant and orthographic cameras) against the CPU ORACLE: films and
radiance bits of the reference traversal on the GPU, megakernel and forced streaming pipeline (and packet walk)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import lasgun_amd as la
from oracle_lib import oracle
src = open(os.path.join(root, "tools", "fast_adversarial.py")).read().split("a, b = int(sys.argv[1])")[0]
G = la.api; o = oracle()

def make(api):
    ns = {"__file__": os.path.join(root, "tools", "fast_adversarial.py")}
    exec(compile(src, "fast_adversarial", "exec"), ns)
    ns["G"] = api; ns["M"] = api.Material
    return ns
nsG, nsO = make(G), make(o)

def bits(x):
    x = np.ascontiguousarray(x, dtype=np.float64); u = x.view(np.uint64).copy(); u[np.isnan(x)] = np.uint64(0x7FF8000000000000); return u

a, b = int(sys.argv[1]), int(sys.argv[2]); gen = "scene2" if len(sys.argv) > 3 and sys.argv[3] == "mesh" else "scene"
w, h = 96, 72
bad = 0; skipped = 0
for seed in range(a, b):
    try:
        oacc = o.Accel(nsO[gen](seed))
    except la.LasgunError:
        skipped += 1
        try:
            G.Accel(nsG[gen](seed)); print("BUILD MISMATCH seed", seed, "(oracle refuses, product builds)"); bad += 1
        except la.LasgunError:
            pass
        continue
    of = o.Film(w, h); o.capture_subset_mt(0, 1, oacc, of, 16)
    o.set_trig_mode(1); orad = o.capture_radiance(oacc, w, h, nthreads=16); o.set_trig_mode(0)
    acc = G.Accel(nsG[gen](seed))
    for streaming, packet in ((0, False), (2, False), (2, True)):
        G.set_streaming(acc, streaming); G.set_packet(acc, packet); G.set_mode(acc, False)
        f = G.Film(w, h); G.capture_subset(0, 1, acc, f)
        r = G.capture_radiance(acc, w, h)
        if not (np.array_equal(f.pixels(), of.pixels()) and np.array_equal(bits(r), bits(orad))):
            bad += 1
            print("MISMATCH seed", seed, "streaming", streaming, "packet", packet, "bytes", int((f.pixels() != of.pixels()).sum()), "radiance words", int((bits(r) != bits(orad)).sum()), flush=True)
print("parity adversarial", gen, a, b, "mismatches", bad, "unbuildable", skipped)
