#!/usr/bin/env python3
"""Lane use of the pruned walk on the mesh configs (diagnostic build: tools/build_variant.sh stamps "-DLG_STAMPS", then
LASGUN_HIP_LIB=lasgun_amd/liblasgun_hip_stamps.so python tools/stamp_mesh.py [glass|metal]): per phase of the walk, wave-level
trips and the lanes that take part in them."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api
lib = G.lib
which = sys.argv[1] if len(sys.argv) > 1 else "metal"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
acc = G.Accel(la.scenes.cornell_scene(G, which.split("_")[1]) if which.startswith("cornell_") else la.scenes.mesh_scene(G, 224, 224, which))
G.set_streaming(acc, 2)  # the level-by-level pipeline's traversal kernels carry the stamps
film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0); G.synchronize(acc)
out = (C.c_ulonglong * 32)()  # two DStats records of 16 words
lib.lg_debug_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.lg_debug_stats(acc.h, 1, out)
G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0); G.synchronize(acc)
lib.lg_debug_stats(acc.h, 0, out)
names = ["setup", "A nodes", "B mesh leaves", "B leaf slots", "enter", "C returns"]
tot = sum(out[i] for i in range(6))
for i, n in enumerate(names):
    print("%-14s %16d cycles  %5.1f %%" % (n, out[i], 100.0 * out[i] / tot))
walks = out[7]
r = out[8:15]
print("phase C inside (cycles per walk): loop entry %.0f, frame + parent fetch %.0f, level record %.0f, ray %.0f, prune constants %.0f, next state %.0f; iterations per walk %.1f" % tuple([v / walks for v in r[:6]] + [r[6] / walks]))
c = out[16:32]
print("walks (wave level) %d, outer trips per walk %.1f" % (walks, out[6] / walks))
print("node trips per walk %.1f: %.1f lanes stepping, %.1f done" % (c[0] / walks, c[5] / max(c[0], 1), c[6] / max(c[0], 1)))
print("outer trips with a mesh leaf per walk %.2f (mesh-leaf calls are per dominant axis)" % (c[1] / walks))
print("mesh-leaf calls per walk %.2f, lanes in a leaf at the call %.1f" % (c[13] / walks, c[14] / max(c[13], 1)))
print("  record (seek) trips per call %.1f: %.1f lanes seeking" % (c[9] / max(c[13], 1), c[10] / max(c[9], 1)))
print("  triangle-pair (test) trips per call %.1f: %.1f lanes testing" % (c[11] / max(c[13], 1), c[12] / max(c[11], 1)))
print("leaf-slot trips per walk %.1f: %.1f lanes stepping" % (c[2] / walks, c[7] / max(c[2], 1)))
