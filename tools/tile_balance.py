#!/usr/bin/env python3
"""How even are contiguous row tiles? Times each of N row tiles of config 3 separately on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
from lasgun_amd.distributed import row_tile
G = la.api
G.set_device(0)
w = h = 4096
acc = G.Accel(la.scenes.spheres_scene(G))
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    times = []
    for r in range(n):
        y0, y1 = row_tile(r, n, h)
        t = torch.zeros((y1 - y0, w, 4), dtype=torch.uint8, device="cuda")
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            G.capture_rows_device(acc, w, h, y0, y1, t.data_ptr(), stream=stream)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        times.append(dt)
    print("N=%d tiles ms: %s  max %.2f mean %.2f  ideal speedup %.2f" % (n, " ".join("%.2f" % x for x in times), max(times), sum(times) / n, times and (sum(times) / max(times))))
