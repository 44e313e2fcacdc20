#!/usr/bin/env python3
"""Host time to ENQUEUE a frame (the library's launches, memsets and bookkeeping, no synchronisation) beside the frame's device time:
python tools/enqueue_cost.py [glass|plastic|readme|spheres] [size] -- is a small frame bound by the host's launch calls?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
G.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "glass"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
scene = S.readme_scene(G) if which == "readme" else S.spheres_scene(G) if which == "spheres" else S.cornell_scene(G, which)
for org, st in (("default", 1), ("megakernel", 0), ("wavefront", 2), ("queue", 3)):
    acc = G.Accel(scene)
    G.set_streaming(acc, st)
    film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s %d^2 %-10s enqueue %.3f ms per frame, frame %.3f ms" % (which, size, org, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
