import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la
G = la.api; S = la.scenes
G.set_device(0)
for name, build, size in (("4m mesh metal", lambda: S.mesh_scene(G, 224, 224, "metal"), 4096), ("5 mixed", lambda: S.mixed_scene(G), 8192), ("5 mixed", lambda: S.mixed_scene(G), 4096)):
    acc = G.Accel(build())
    film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for fast in (False, True):
        G.set_mode(acc, fast)
        row = []
        for streaming in (0, 2):
            G.set_streaming(acc, streaming)
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / reps * 1e3)
        print("%-16s %5d^2 fast=%d mega %8.3f  stream %8.3f ms" % (name, size, fast, *row), flush=True)
