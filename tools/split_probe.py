#!/usr/bin/env python3
"""Small frames, one sample per pixel: the megakernel handing its tiles out whole and in quarters (lg_accel_set_tile_parts), the other
organisations, and what the default picks."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LASGUN_AUTOTUNE", "2")
import lasgun_amd as la
G, S = la.api, la.scenes
G.set_device(0)
SCENES = {"cornell_glass": lambda: S.cornell_scene(G, "glass"), "kitchen_sink": lambda: S.kitchen_sink_scene(G, "perspective", 2, 0),
          "mesh_glass": lambda: S.mesh_scene(G), "mesh_metal": lambda: S.mesh_scene(G, material="metal"), "spooky_ss0": lambda: S.spooky_scene(G, supersampling=0),
          "simple_reflect": lambda: S.simple_scene(G, 0, True), "spheres1024": lambda: S.spheres_scene(G), "spooky_ss2": lambda: S.spooky_scene(G)}
for size in [int(a) for a in sys.argv[1:]] or [256, 512]:
    for name, make in SCENES.items():
        acc = G.Accel(make())
        dev = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = {"scene": name, "size": size}
        for org, code in (("megakernel", 0), ("megakernel_quarters", 0), ("wavefront", 2), ("queue", 3), ("queue_quarters", 3), ("default", 1)):
            G.set_streaming(acc, code)
            G.set_tile_parts(acc, 4 if org.endswith("_quarters") else None)
            ts = []
            for i in range(8):
                e0.record(); G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st); e1.record(); torch.cuda.synchronize()
                if i: ts.append(e0.elapsed_time(e1))
            out[org + "_ms"] = round(min(ts), 3)
            if org == "default":
                out["default_is"] = G.last_organisation(acc)
        print(json.dumps(out), flush=True)
