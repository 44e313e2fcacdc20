#!/usr/bin/env python3
"""Where does the pruned walk (and the culling records + strips its accel build makes) start to pay?  Meshes of growing size: the accel build's
stages (LASGUN_DEBUG_TIMES on stderr) and one frame with the pruned walk off and on, default organisation.
python tools/prune_threshold_probe.py  2> build_stages.txt"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")
os.environ["LASGUN_DEBUG_TIMES"] = "1"
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
G.set_device(0)
SCENES = [("playground 1840 tris", lambda: S.playground_scene(G), 512), ("spooky 1514 tris", lambda: S.spooky_scene(G), 768)]
for n in (16, 24, 32, 48, 64, 96, 128):
    SCENES.append(("mesh %dx%d metal (%d tris)" % (n, n, 2 * n * n), (lambda n=n: S.mesh_scene(G, n, n, "metal")), 1024))
    SCENES.append(("mesh %dx%d glass (%d tris)" % (n, n, 2 * n * n), (lambda n=n: S.mesh_scene(G, n, n, "glass")), 1024))
for name, make, size in SCENES:
    scene = make()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); acc = G.Accel(scene); ts.append((time.perf_counter() - t) * 1e3)
    out = {"scene": name, "size": size, "accel_from_ms": round(min(ts), 3), "prune_default": bool(G.get_prune(acc))}
    dev = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for label, prune in (("unpruned_ms", False), ("pruned_ms", True)):
        G.set_prune(acc, prune)
        best = 1e9
        for i in range(6):
            e0.record(); G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st); e1.record(); torch.cuda.synchronize()
            if i > 1: best = min(best, e0.elapsed_time(e1))
        out[label] = round(best, 3)
        out[label.replace("_ms", "_is")] = G.last_organisation(acc)
    print(json.dumps(out), flush=True)
