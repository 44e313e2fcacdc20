#!/usr/bin/env python3
"""Time every BASELINE.json config on one GPU (device-resident film) and print one JSON line each.
Not the driver's bench (that is bench.py, config 3); this fills BASELINE.md's results table."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G = la.api
S = la.scenes
CONFIGS = [
    ("1a readme 512^2", lambda: S.readme_scene(G), 512),
    ("1b simple ss2 512^2 (9 spp)", lambda: S.simple_scene(G, 2), 512),
    ("2P cornell plastic 512^2", lambda: S.cornell_scene(G, "plastic"), 512),
    ("2G cornell glass 512^2", lambda: S.cornell_scene(G, "glass"), 512),
    ("3 spheres1024 4096^2", lambda: S.spheres_scene(G), 4096),
    ("4 mesh100k glass+mirror 4096^2", lambda: S.mesh_scene(G, 224, 224, "glass"), 4096),
    ("4m mesh100k metal 4096^2", lambda: S.mesh_scene(G, 224, 224, "metal"), 4096),
    ("5 mixed 8192^2", lambda: S.mixed_scene(G), 8192),
]


def progressive(name, scene, size, n):
    """The web worker's progressive form (www/renderer.ts:103-120): n shuffled capture_subset(k, n) calls into ONE film (each call
    the pixels {k, k + n, ...} of the row-major image) against one whole frame."""
    import random
    acc = G.Accel(scene)
    film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    ref = torch.zeros_like(film)
    stream = torch.cuda.current_stream().cuda_stream
    G.capture_rows_device(acc, size, size, 0, size, ref.data_ptr(), row0=0, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        G.capture_rows_device(acc, size, size, 0, size, ref.data_ptr(), row0=0, stream=stream)
    torch.cuda.synchronize()
    frame_ms = (time.perf_counter() - t0) / 3 * 1e3
    order = list(range(n))
    random.Random(7).shuffle(order)
    for k in order[:2]:  # warm-up (launch contexts, kernels)
        G.capture_subset_device(k, n, acc, size, size, film.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    film.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in order:
        G.capture_subset_device(k, n, acc, size, size, film.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    prog_ms = (time.perf_counter() - t0) * 1e3
    single_ok = bool(torch.equal(film, ref))
    # the batched entry (lg_capture_subsets_device): the same n shuffled subsets in batches of b -- one render per batch
    batched = {}
    for b in (n, max(n // 10, 1)):
        G.capture_subsets_device(order[:b], n, acc, size, size, film.data_ptr(), stream=stream)  # warm-up
        torch.cuda.synchronize()
        film.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for at in range(0, n, b):
            G.capture_subsets_device(order[at:at + b], n, acc, size, size, film.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        batched["batches_of_%d" % b] = {"ms": round(ms, 3), "ratio": round(ms / frame_ms, 3), "identical_to_frame": bool(torch.equal(film, ref))}
    print(json.dumps({"config": name, "progressive_subsets": n, "frame_ms": round(frame_ms, 3), "progressive_ms": round(prog_ms, 3),
                      "ratio": round(prog_ms / frame_ms, 3), "ms_per_subset": round(prog_ms / n, 4),
                      "identical_to_frame": single_ok, "batched": batched}), flush=True)


def glass_spheres_scene(api):
    """config 3's shell and spheres, every sphere glass: recursion 3 over an LDS-resident scene"""
    rng = S.SplitMix64(0x1A560001)
    scene = api.Scene.new()
    S._cornell_shell(api, scene, 0)
    for _ in range(1024):
        c = [rng.uniform(-1.8, 1.8), rng.uniform(-1.8, 1.8), rng.uniform(-1.8, 1.8)]
        scene.root.add_sphere(c, rng.uniform(0.02, 0.06), api.Material.glass([1.0, 0.7, 1.0], [0.7, 1.0, 0.7], 1.25))
        rng.next_f64()
    return scene


# scenes NO rule was ever fitted to (the rule of rounds 2-4 was fitted to CONFIGS): the reference's other example programs at their own sizes,
# the suite's feature scenes, meshes of other tessellations, other film sizes
UNFITTED = [
    ("u01 simple.rs 1 spp 1024^2", lambda: S.simple_scene(G, 0), 1024),
    ("u02 simplereflect.rs 512^2", lambda: S.simple_scene(G, 2, True), 512),
    ("u03 playground.rs 512^2 (9 spp)", lambda: S.playground_scene(G), 512),
    ("u04 spooky.rs 768^2 (9 spp)", lambda: S.spooky_scene(G), 768),
    ("u05 simplecows.rs 512^2 (9 spp)", lambda: S.simplecows_scene(G), 512),
    ("u06 kitchen_sink 1024^2", lambda: S.kitchen_sink_scene(G, "perspective"), 1024),
    ("u07 instanced meshes 2048^2", lambda: S.instanced_scene(G), 2048),
    ("u08 mesh 32x32 glass 2048^2", lambda: S.mesh_scene(G, 32, 32, "glass"), 2048),
    ("u09 mesh 64x64 metal 2048^2", lambda: S.mesh_scene(G, 64, 64, "metal"), 2048),
    ("u10 mesh 128x128 plastic 4096^2", lambda: S.mesh_scene(G, 128, 128, "plastic"), 4096),
    ("u11 glass spheres1024 2048^2", lambda: glass_spheres_scene(G), 2048),
    ("u12 cornell glass 1536^2", lambda: S.cornell_scene(G, "glass"), 1536),
    ("u13 readme sphere 2048^2", lambda: S.readme_scene(G), 2048),
    ("u14 spheres 300 1024^2", lambda: S.spheres_scene(G, 300, seed=7), 1024),
]


def org_choice(only):
    """What the default picks (measured on the launch's first use, launch.cpp: tuned_choice) against every organisation forced, one frame at a time."""
    def timed(acc, film, size, stream, reps=3):
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best
    for name, build, size in CONFIGS + UNFITTED:
        if only and not any(o in name for o in only):
            continue
        acc = G.Accel(build())
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        forced = {}
        for label, code in (("megakernel", 0), ("wavefront", 2), ("queue", 3)):
            G.set_streaming(acc, code)
            forced[label] = round(timed(acc, film, size, stream), 4)
        G.set_streaming(acc, 1)
        t0 = time.perf_counter()
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)  # (the first default launch of this kind measures)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        default_ms = timed(acc, film, size, stream)
        used = G.last_organisation(acc)
        best = min(forced, key=forced.get)
        print(json.dumps({"config": name, "default_ms": round(default_ms, 4), "default_is": used, "forced_ms": forced, "best": best,
                          "default_over_best": round(default_ms / forced[best], 4), "first_default_launch_ms": round(first_ms, 3),
                          "fitted": not name.startswith("u")}), flush=True)


def main():
    if "--org-choice" in sys.argv:
        G.set_device(0)
        org_choice([a for a in sys.argv[1:] if not a.startswith("--")])
        return
    prog = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--progressive=")]
    if prog:
        only = [a for a in sys.argv[1:] if not a.startswith("--")]
        G.set_device(0)
        for name, build, size in CONFIGS:
            if only and not any(o in name for o in only):
                continue
            progressive(name, build(), size, int(prog[0]))
        return
    fast = "--fast" in sys.argv
    org = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--org=")]  # default | megakernel | wavefront | queue
    org = org[0] if org else "default"
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    G.set_device(0)
    for name, build, size in CONFIGS:
        if only and not any(o in name for o in only):
            continue
        scene = build()
        t0 = time.perf_counter()
        acc = G.Accel(scene)
        build_s = time.perf_counter() - t0
        G.set_mode(acc, fast)
        if org == "megakernel":
            G.set_streaming(acc, 0)
        elif org == "wavefront":
            G.set_streaming(acc, 2)
        elif org == "queue":
            G.set_streaming(acc, 3)
        order = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--tile-order=")]  # 0 top-down | 1 bottom-up | 2 middle-out (default: the library's)
        if order:
            G.set_tile_order(acc, int(order[0]))
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        G.profile_enable(acc, True)
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        kinds = {k: round(v[0], 3) for k, v in G.profile_read_kinds(acc).items() if v[1]}
        G.profile_read(acc)
        G.profile_enable(acc, False)
        st = G.capture_stats(acc, size, size)
        rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
        print(json.dumps({"config": name, "mode": "fast" if fast else "parity", "org": org, "ms": round(ms, 3), "kernels_ms": kinds, "Mrays_s": round(rays / ms / 1e3, 1), "rays": rays,
                          "accel_build_s": round(build_s, 3), "info": G.accel_info(acc), "stats": st}), flush=True)


if __name__ == "__main__":
    main()
