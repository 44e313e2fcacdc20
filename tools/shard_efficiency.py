#!/usr/bin/env python3
"""How much of the single-GPU frame time does one rank's share cost?  (interleaved 64-row blocks, config 3, 4096^2)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
acc = G.Accel(S.spheres_scene(G))
w = h = 4096
stream = torch.cuda.current_stream().cuda_stream
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
full = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
t1 = t(lambda: G.capture_rows_device(acc, w, h, 0, h, full.data_ptr(), row0=0, stream=stream))
print("1 rank : %.3f ms" % t1)
for label, streaming in (("streaming pipeline", 1), ("megakernel", 0)):
  G.set_streaming(acc, streaming)
  print(label)
  for n in (2, 4, 8):
    tile = torch.zeros((h // n, w, 4), dtype=torch.uint8, device="cuda")
    worst = 0.0
    for r in range(n):
        worst = max(worst, t(lambda: G.capture_interleaved_device(acc, w, h, 64, n, r, tile.data_ptr(), stream=stream), reps=10))
    print("  %d ranks: slowest share %.3f ms = %.1f %% of ideal (%.3f ms)" % (n, worst, 100.0 * (t1 / n) / worst, t1 / n))
G.set_streaming(acc, 1)
for n in (1, 8):
    tile = torch.zeros((h // n, w, 4), dtype=torch.uint8, device="cuda")
    G.profile_enable(acc, True)
    for _ in range(10):
        if n == 1: G.capture_rows_device(acc, w, h, 0, h, tile.data_ptr(), row0=0, stream=stream)
        else: G.capture_interleaved_device(acc, w, h, 64, n, 3, tile.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    ms, launches = G.profile_read(acc)
    kinds = G.profile_read_kinds(acc)
    G.profile_enable(acc, False)
    print("n=%d: whole share %.3f ms (events); per kernel:" % (n, ms / launches), {k: round(v[0] / v[1], 3) for k, v in kinds.items() if v[1]})
# consecutive frames on two streams (what bench.py does for N > 1): throughput per share
G.set_streaming(acc, 1)
for n in (1, 2, 4, 8):
    tiles = [torch.zeros((h // n, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    def frames(count, two):
        for k in range(count):
            s = ss[k % 2] if two else ss[0]
            G.capture_interleaved_device(acc, w, h, 64, n, n // 2, tiles[k % 2].data_ptr(), stream=s.cuda_stream)
    res = []
    for two in (False, True):
        frames(4, two); torch.cuda.synchronize()
        t0 = time.perf_counter(); frames(40, two); torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 40 * 1e3)
    print("share 1/%d: one stream %.3f ms/frame, two streams %.3f ms/frame (ideal %.3f)" % (n, res[0], res[1], t1 / n))
