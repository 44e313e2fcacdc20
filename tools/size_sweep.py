import os
#!/usr/bin/env python3
"""Frame time against film size, per kernel organisation, on the small scenes (python tools/size_sweep.py readme|plastic|glass|spheres|simple1|simple2
[sizes]): where a launch's fixed cost -- claims, cold code, one wave's latency -- and where its throughput sets the time.  Round 4's default rules
(launch.cpp, enqueue) come from these tables."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
G.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "glass"
for org, st in (("megakernel", 0), ("wavefront", 2)):
    for size in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "8,64,256,512,1024".split(","))]:
        acc = G.Accel(S.readme_scene(G) if which == "readme" else S.spheres_scene(G) if which == "spheres" else S.simple_scene(G, 1) if which == "simple1" else S.simple_scene(G, 2) if which == "simple2" else S.cornell_scene(G, which))
        G.set_streaming(acc, st)
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        # one frame at a time (latency)
        t1 = []
        for _ in range(10):
            t0 = time.perf_counter()
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            t1.append((time.perf_counter() - t0) * 1e3)
        G.profile_enable(acc, True)
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        kinds = {k: round(v[0], 3) for k, v in G.profile_read_kinds(acc).items() if v[1]}
        G.profile_read(acc); G.profile_enable(acc, False)
        print(which, org, size, "back-to-back %.3f ms" % ms, "single %.3f" % min(t1), kinds, flush=True)
