#!/usr/bin/env python3
"""What a literal caller of capture(scene, film) pays on config 3: the film in pageable memory (Film.new) against the same film in
pinned memory (new_with_output over a pinned buffer), and the pieces (accel build, render into HBM, D2H alone).
python tools/host_capture_probe.py [size]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
G.set_device(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else "spheres"  # spheres (config 3) | simple (src/examples/simple.rs, 9 spp) | cornell_glass | spooky
scene = {"spheres": lambda: S.spheres_scene(G), "simple": lambda: S.simple_scene(G, 2), "cornell_glass": lambda: S.cornell_scene(G, "glass", 2),
         "spooky": lambda: S.spooky_scene(G)}[which]()


def timed(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return round(min(ts), 3), round(sorted(ts)[len(ts) // 2], 3)


out = {"size": size, "scene": which}
film = G.Film.new(size, size)
out["capture_pageable_film_ms(min,median)"] = timed(lambda: G.capture(scene, film))
pinned = torch.empty((size, size, 4), dtype=torch.uint8, pin_memory=True)
film_p = G.Film.new_with_output(size, size, pinned.numpy())
out["capture_pinned_film_ms(min,median)"] = timed(lambda: G.capture(scene, film_p))
out["accel_from_ms(min,median)"] = timed(lambda: G.Accel(scene))
acc = G.Accel(scene)
dev = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
out["render_into_hbm_ms(min,median)"] = timed(lambda: G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st))
host = torch.empty((size, size, 4), dtype=torch.uint8)
out["d2h_pageable_ms(min,median)"] = timed(lambda: host.copy_(dev))
out["d2h_pinned_ms(min,median)"] = timed(lambda: pinned.copy_(dev))
out["capture_subset_whole_ms(min,median)"] = timed(lambda: G.capture_subset(0, 1, acc, film))
print(json.dumps(out))
