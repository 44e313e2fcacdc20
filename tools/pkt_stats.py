#!/usr/bin/env python3
"""Analysis (needs the -DLG_PKT_STATS build, LASGUN_HIP_LIB=.../liblasgun_hip_pkstats.so): wave-level work of the packet walk."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lasgun_amd as la
G = la.api; S = la.scenes
lib = G.lib if hasattr(G, "lib") else ctypes.CDLL(la.LIB_PATH)
lib.lg_debug_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
acc = G.Accel(S.spheres_scene(G))
w = h = 4096
st = G.capture_stats(acc, w, h)
rays = st["primary_rays"] + st["shadow_rays"]
print("private walks: per ray nodes %.1f spheres %.1f" % (st["nodes_tested"] / rays, st["spheres_tested"] / rays))
G.set_packet(acc, True)
film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
G.capture_rows_device(acc, w, h, 0, h, film.data_ptr(), row0=0); G.synchronize(acc)
lib.lg_debug_stats(acc.h, 1, None)
G.capture_rows_device(acc, w, h, 0, h, film.data_ptr(), row0=0); G.synchronize(acc)
out = (ctypes.c_ulonglong * 9)()
lib.lg_debug_stats(acc.h, 0, out)
# DStats order: primary, shadow, secondary, nodes, spheres, cuboids, triangles, entries, hits
nodes_w, sph_w, sph_lanes, node_lanes = out[3], out[4], out[5], out[8]
waves = rays / 64
print("packet walk: per wave-traversal node visits %.1f (lanes in mask %.1f avg), sphere tests %.1f (lanes in leaf %.1f avg)" % (
    nodes_w / waves, node_lanes / max(nodes_w, 1), sph_w / waves, sph_lanes / max(sph_w, 1)))
