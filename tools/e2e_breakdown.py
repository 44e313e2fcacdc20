#!/usr/bin/env python3
"""Where the time of the drop-in entry point lg_capture(scene, host film) goes (config 3, 4096^2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
G.set_device(0)
w = h = 4096
scene = S.spheres_scene(G)
film = G.Film(w, h)
G.capture(scene, film)  # warm: module load, first-touch of the film
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ms_capture, _ = t(lambda: G.capture(scene, film))
ms_accel, acc = t(lambda: G.Accel(scene))
ms_subset, _ = t(lambda: G.capture_subset(0, 1, acc, film))
import torch  # after the library on purpose: both must share one HIP runtime
dev = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
def dev_render():
    G.capture_rows_device(acc, w, h, 0, h, dev.data_ptr(), row0=0); G.synchronize(acc)
ms_dev, _ = t(dev_render)
print("lg_capture %.2f ms = lg_accel_from %.2f + lg_capture_subset(host film) %.2f (device render %.2f, rest = D2H of %d MiB)" % (
    ms_capture, ms_accel, ms_subset, ms_dev, w * h * 4 >> 20))
