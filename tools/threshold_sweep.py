#!/usr/bin/env python3
"""Where does the streaming pipeline overtake the megakernel?  Scenes of growing traversal work at 2048^2 (and 1024^2),
both organisations, with the per-ray work counters beside the times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
G.set_device(0)
cases = [("cornell plastic", lambda: S.cornell_scene(G, "plastic")), ("simple ss0", lambda: S.simple_scene(G, 0))]
cases += [("spheres%d" % n, (lambda n=n: S.spheres_scene(G, n))) for n in (16, 64, 256, 1024, 4096)]
cases += [("mesh 32x32 metal", lambda: S.mesh_scene(G, 32, 32, "metal")), ("mesh 224x224 metal", lambda: S.mesh_scene(G, 224, 224, "metal")), ("mixed", lambda: S.mixed_scene(G))]
for name, build in cases:
    acc = G.Accel(build())
    for size in (1024, 2048):
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        row = []
        fits = False
        for streaming, lds in ((0, True), (2, True), (2, False), (1, True)):
            G.set_streaming(acc, streaming); fits = G.set_lds_scene(acc, lds)
            for _ in range(2):
                G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / reps * 1e3)
        st = G.capture_stats(acc, size, size)
        rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
        print("%-20s %5d^2 mega %8.3f  stream+LDS %8.3f  stream %8.3f  default %8.3f ms | per ray: nodes %.1f sph+cub %.1f tri %.1f entries %.1f | lds-fit %d" % (
            name, size, *row, st["nodes_tested"] / rays, (st["spheres_tested"] + st["cuboids_tested"]) / rays, st["triangles_tested"] / rays,
            st["accel_entries"] / rays, fits), flush=True)
