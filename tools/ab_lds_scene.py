import sys, json, time
sys.path.insert(0, '/root/repo')
import torch, lasgun_amd as la
G, S = la.api, la.scenes
G.set_device(0)
size = 4096
for fast in (False, True):
    for lds in (True, False):
        acc = G.Accel(S.spheres_scene(G))
        G.set_mode(acc, fast)
        G.set_lds_scene(acc, lds)
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=st)
        torch.cuda.synchronize()
        G.profile_enable(acc, True)
        for _ in range(3):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=st)
        torch.cuda.synchronize()
        kinds = {k: round(v[0] / max(v[1], 1), 3) for k, v in G.profile_read_kinds(acc).items() if v[1]}
        print(json.dumps({"fast": fast, "lds": lds, "kernels_ms": kinds}), flush=True)
