#!/usr/bin/env python3
"""Supersampled frames, a pixel's samples one after the other (rounds 1-4) against side by side (lg_accel_set_sample_order), in every
organisation and as the default picks: frame time of each, and that every film is the same bytes.
python tools/ss_probe.py [size ...]     (SS_PROBE_SCENES=a,b: only those)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
G.set_device(0)
sizes = [int(a) for a in sys.argv[1:]] or [512]
SCENES = {
    "simple_ss3": lambda: S.simple_scene(G, 3),
    "simple_reflect_ss3": lambda: S.simple_scene(G, 3, True),
    "cornell_glass_ss2": lambda: S.cornell_scene(G, "glass", 2),
    "cornell_glass_ss3": lambda: S.cornell_scene(G, "glass", 3),
    "spooky_ss2": lambda: S.spooky_scene(G),
    "playground_ss2": lambda: S.playground_scene(G),
    "simplecows_ss2": lambda: S.simplecows_scene(G),
    "kitchen_sink_ss2": lambda: S.kitchen_sink_scene(G, "perspective", 2, 2),
    "spheres1024_ss2": lambda: S.spheres_scene(G, supersampling=2),
    "mesh_glass_ss2": lambda: S.mesh_scene(G, supersampling=2),
    "mesh_plastic_ss2": lambda: S.mesh_scene(G, material="plastic", supersampling=2),
    "mixed_ss2": lambda: S.mixed_scene(G, supersampling=2),
}
only = os.environ.get("SS_PROBE_SCENES")
if only:
    SCENES = {k: v for k, v in SCENES.items() if k in only.split(",")}
ORGS = {"megakernel": 0, "wavefront": 2, "queue": 3, "default": 1}


def frame_ms(acc, size, dev, n=7):
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0.record()
        G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return round(min(ts), 3)


for size in sizes:
    for name, make in SCENES.items():
        scene = make()
        out = {"scene": name, "size": size}
        ref = None
        differs = 0
        for org, code in ORGS.items():
            for order, oname in ((1, "in_a_row"), (0, "side_by_side")) if org != "default" else ((None, "ms"),):
                acc = G.Accel(scene)
                G.set_streaming(acc, code)
                G.set_sample_order(acc, order)
                dev = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
                out[org + "_" + oname] = frame_ms(acc, size, dev)
                if org == "default":
                    out["default_is"] = G.last_organisation(acc)
                film = dev.cpu()
                if ref is None:
                    ref = film
                else:
                    differs += int((film != ref).any(dim=-1).sum())
        out["pixels_that_differ_between_any_two"] = differs
        print(json.dumps(out), flush=True)
