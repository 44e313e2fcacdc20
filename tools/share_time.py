#!/usr/bin/env python3
"""What one rank's share of the headline frame costs at N = 1, 2, 4, 8 (one GPU, no gather): the fixed per-frame costs
that bound strong scaling.  usage: python tools/share_time.py [size] [megakernel]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G = la.api
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G.set_device(0)
acc = G.Accel(la.scenes.spheres_scene(G))
if len(sys.argv) > 2 and sys.argv[2] == "megakernel":  # A/B: one launch per frame instead of three
    G.set_streaming(acc, 0)
NS = int(os.environ.get("LASGUN_SHARE_STREAMS", "2"))  # frames in flight in the overlapped form (bench.py keeps four)
streams = [torch.cuda.Stream() for _ in range(NS)]
for world in (1, 2, 4, 8):
    tiles = [torch.zeros((size // world, size, 4), dtype=torch.uint8, device="cuda") for _ in range(NS)]
    out = {"world": world, "frames_in_flight": NS}
    for overlap in (False, True):
        def frame(k):
            s = streams[k % NS] if overlap else streams[0]
            G.capture_interleaved_device(acc, size, size, 64, world, 0, tiles[k % NS].data_ptr(), stream=s.cuda_stream)
        for k in range(2 * NS):
            frame(k)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for k in range(n):
            frame(k)
        torch.cuda.synchronize()
        out["overlapped_ms" if overlap else "sequential_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
    out["ideal_ms"] = None
    print(json.dumps(out), flush=True)
