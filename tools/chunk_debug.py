import sys, json
sys.path.insert(0, '/root/repo')
import lasgun_amd as la
G, S = la.api, la.scenes
G.set_device(0)
for name, build, size in (("4m", lambda: S.mesh_scene(G, 224, 224, "metal"), 2048), ("4", lambda: S.mesh_scene(G, 224, 224, "glass"), 2048)):
    acc = G.Accel(build())
    G.set_prune(acc, True)
    base = G.capture_stats(acc, size, size, 0, size)
    for kind in (1, 2):
        st = G.capture_stats_kind(acc, size, size, kind, 0, size)
        rays = {1: base["primary_rays"] + base["secondary_rays"], 2: base["shadow_rays"]}[kind]
        print(json.dumps({"scene": name, "kind": "closest" if kind == 1 else "shadow", "rays": rays, "records": st["cuboids_tested"], "counted": st["spheres_tested"], "nodes+records": st["nodes_tested"],
                          "triangles": st["triangles_tested"], "entries": st["accel_entries"]}), flush=True)
