import os
#!/usr/bin/env python3
"""Where the reference walk spends its cycles (diagnostic build: make -C lasgun_amd/csrc EXTRA_DEVFLAGS=-DLG_STAMPS
EXTRA_HOSTFLAGS=-DLG_STAMPS OUT=../liblasgun_hip_stamps.so, then LASGUN_HIP_LIB=lasgun_amd/liblasgun_hip_stamps.so python tools/stamp_phases.py).
Read the shares with care: the accumulators of the diagnostic build are registers the walk does not have to spare (128 per lane): its kernels spill
more than the shipped ones, and a phase that reloads what was spilled (phase C: the root ray) looks worse here than it is."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api
lib = G.lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
acc = G.Accel(la.scenes.spheres_scene(G))
film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0); G.synchronize(acc)
out = (C.c_ulonglong * 32)()  # two DStats records of 16 words: cycles per phase + walks, then the trip counts
lib.lg_debug_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.lg_debug_stats(acc.h, 1, out)
G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0); G.synchronize(acc)
lib.lg_debug_stats(acc.h, 0, out)
names = ["setup", "A nodes", "B mesh leaves", "B leaf slots", "enter", "C returns", "trips", "walks"]
tot = sum(out[i] for i in range(6))
for i, n in enumerate(names):
    print("%-14s %14d  %5.1f %%" % (n, out[i], 100.0 * out[i] / tot if i < 6 else 0.0))
print("cycles per walk %.0f, trips per walk %.1f" % (tot / out[7], out[6] / out[7]))
r = out[8:15]
print("phase C inside, busiest lane, cycles per walk: loop entry %.0f, frame + parent fetch %.0f, level record %.0f, ray %.0f, prune constants %.0f, next state %.0f; iterations per walk %.1f" % tuple([v / out[7] for v in r[:6]] + [r[6] / out[7]]))
cn = ["node-loop trips", "mesh-leaf blocks", "leaf-slot trips", "enter blocks", "return blocks", "lanes stepping, summed over node trips", "lanes already done, summed over node trips",
      "lanes stepping, summed over leaf-slot trips", "lanes already done, summed over leaf-slot trips"]
for i, n in enumerate(cn):
    print("%-48s %12d  %7.1f per walk" % (n, out[16 + i], out[16 + i] / out[7]))
print("node trips: %.1f lanes stepping, %.1f done;  leaf-slot trips: %.1f lanes stepping, %.1f done" % (out[21] / out[16], out[22] / out[16], out[23] / out[18], out[24] / out[18]))
