import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import lasgun_amd as la
from oracle_lib import oracle
G = la.api; S = la.scenes
w = h = 4096
build = lambda api: S.mesh_scene(api, 224, 224, "glass")
acc = G.Accel(build(G))
G.set_streaming(acc, 0)
films = {}
for label, fast, prune in (("parity", False, False), ("pruned", False, True), ("fast", True, False)):
    G.set_mode(acc, fast); G.set_prune(acc, prune)
    f = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    G.capture_rows_device(acc, w, h, 0, h, f.data_ptr(), row0=0); G.synchronize(acc)
    films[label] = f.cpu().numpy().reshape(-1, 4)
d = np.nonzero((films["parity"] != films["fast"]).any(axis=1))[0]
print("parity vs fast:", len(d), "pixels differ;", "parity vs pruned:", int((films["parity"] != films["pruned"]).any(axis=1).sum()))
idx = d[:24].astype(np.uint64)
if len(idx):
    o = oracle()
    want, _ = o.capture_pixels(o.Accel(build(o)), w, h, idx, radiance=False, nthreads=32)
    for k, i in enumerate(idx):
        i = int(i)
        print(i % w, i // w, "oracle", want.reshape(-1, 4)[k][:3], "parity", films["parity"][i][:3], "fast", films["fast"][i][:3])
