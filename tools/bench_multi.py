#!/usr/bin/env python3
"""Single-process multi-GPU capture (lg_multi_*: one accel per device, ONE grouped RCCL gather over xGMI into the root's HBM).
usage: python tools/bench_multi.py [--devices 0,1,2,3] [--size 4096] [--steps 10] [--block-rows 64]
On a 1-GPU box a device may be named several times (device-local copies; LASGUN_MULTI_FORCE_RCCL=1 sends them through RCCL)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--block-rows", type=int, default=64)
    args = ap.parse_args()
    G = la.api
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(G.device_count()))
    w = h = args.size
    scene = la.scenes.spheres_scene(G)
    t0 = time.perf_counter()
    m = G.Multi(scene, devices, args.block_rows)
    create_s = time.perf_counter() - t0
    torch.cuda.set_device(devices[0])
    film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    m.capture_device(w, h, film.data_ptr())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.capture_device(w, h, film.data_ptr())  # synchronous: render on every device + gather
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    st = G.capture_stats(m.accel(0), w, h)
    rays = st["primary_rays"] + st["shadow_rays"]
    ref = torch.zeros_like(film)
    torch.cuda.synchronize()
    acc = G.Accel(scene)
    G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0)
    G.synchronize(acc)
    print(json.dumps({"devices": devices, "uses_rccl": m.uses_rccl, "ms_per_frame": ms, "Mrays_s": rays / ms / 1e3, "create_s": create_s,
                      "identical_to_single_device_film": bool(torch.equal(film, ref))}))


if __name__ == "__main__":
    main()
