"""One-off wider fuzz run: seeds [a, b) of lasgun_amd.scenes.random_scene, every mode, GPU vs oracle."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import lasgun_amd as la
from oracle_lib import oracle
S = la.scenes; G = la.api; o = oracle()
a, b = int(sys.argv[1]), int(sys.argv[2]); w, h = 72, 56
def bits(x):
    x = np.ascontiguousarray(x, dtype=np.float64); u = x.view(np.uint64).copy(); u[np.isnan(x)] = np.uint64(0x7FF8000000000000); return u
bad = 0; skipped = 0; ties = 0
for seed in range(a, b):
    try:
        oacc = o.Accel(S.random_scene(o, seed))
    except la.LasgunError:
        skipped += 1; continue
    of = o.Film(w, h); o.capture_subset_mt(0, 1, oacc, of, 16)
    o.set_trig_mode(1); orad = o.capture_radiance(oacc, w, h, nthreads=16); o.set_trig_mode(0)
    acc = G.Accel(S.random_scene(G, seed))
    for streaming, fast, packet in ((0, False, False), (0, True, False), (2, False, False), (2, True, False), (2, False, True)):
        if True:
            G.set_streaming(acc, streaming); G.set_mode(acc, fast); G.set_packet(acc, packet)
            f = G.Film(w, h); G.capture_subset(0, 1, acc, f)
            r = G.capture_radiance(acc, w, h)
            if not (np.array_equal(f.pixels(), of.pixels()) and np.array_equal(bits(r), bits(orad))):
                bad += 1; print("MISMATCH seed", seed, "streaming", streaming, "fast", fast, "packet", packet, int((f.pixels() != of.pixels()).sum()), "bytes", flush=True)
print("seeds", a, b, "mismatches", bad, "skipped (unbuildable in the reference)", skipped)
