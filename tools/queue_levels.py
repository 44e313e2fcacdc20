#!/usr/bin/env python3
"""Config 4 (glass torus + mirror sphere) at recursion 0..3 in the megakernel and in the queue organisation: what each recursion
level adds (ms, rays) -- where the queue organisation's time goes."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
size = int(os.environ.get("SIZE", "4096"))
G.set_device(0)
film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for material in sys.argv[1:] or ["glass"]:
    for rec in (0, 1, 2, 3):
        scene = S.mesh_scene(G, 224, 224, material)
        scene.set_max_recursion_depth(rec)
        acc = G.Accel(scene)
        row = {"material": material, "recursion": rec}
        for org, code in (("megakernel", 0), ("queue", 3)):
            G.set_streaming(acc, code)
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            row[org + "_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        pk = (C.c_ulonglong * 8)()
        G.lib.lg_debug_queue_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        G.lib.lg_debug_queue_packets(acc.h, C.c_void_p(stream), pk)  # the queue launch just timed (the last on this stream)
        row["queue_packets_per_level"] = [int(v) for v in pk][1:rec + 1]
        st = G.capture_stats(acc, size, size)
        row.update({k: st[k] for k in ("primary_rays", "shadow_rays", "secondary_rays", "hits", "nodes_tested", "triangles_tested")})
        if row["queue_packets_per_level"]:
            row["rays_per_deep_packet"] = round(st["secondary_rays"] / max(1, sum(row["queue_packets_per_level"])), 1)
        print(json.dumps(row), flush=True)
