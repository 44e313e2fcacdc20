#!/usr/bin/env python3
"""Reference walk and fast walk, each with its trees in LDS and in L2 (headline scene, or `nspheres` of it):
python tools/ab_lds_scene.py [nspheres] [size]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
G.set_device(0)
nsph = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ref = None
for fast in (False, True):
    for lds in (True, False):
        acc = G.Accel(S.spheres_scene(G, nsph))
        G.set_streaming(acc, 2)
        G.set_mode(acc, fast)
        fits = G.set_lds_scene(acc, lds)
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=st)
        torch.cuda.synchronize()
        if ref is None:
            ref = film.clone()
        G.profile_enable(acc, True)
        for _ in range(3):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=st)
        torch.cuda.synchronize()
        kinds = {k: round(v[0] / max(v[1], 1), 3) for k, v in G.profile_read_kinds(acc).items() if v[1]}
        print(json.dumps({"spheres": nsph, "fast": fast, "lds": lds, "ref_image_fits": bool(fits), "kernels_ms": kinds, "identical": bool(torch.equal(film, ref))}), flush=True)
