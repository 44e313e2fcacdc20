#!/usr/bin/env python3
"""lg_accel_from of every config and of the reference's example scenes, four times each (LASGUN_DEBUG_TIMES=1: the build's stages on stderr)."""
import os, sys, time, json
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/lasgun_amd') else '.')
sys.path.insert(0, 'tools')
import torch, lasgun_amd as la
from bench_configs import CONFIGS
G, S = la.api, la.scenes
G.set_device(0)
scenes = {"spooky": lambda: S.spooky_scene(G), "playground": lambda: S.playground_scene(G), "simplecows": lambda: S.simplecows_scene(G)}
for name, build, size in CONFIGS:
    scenes[name] = build
for name, build in scenes.items():
    scene = build()
    ts = []
    for i in range(4):
        t = time.perf_counter(); acc = G.Accel(scene); ts.append((time.perf_counter() - t) * 1e3); del acc
    print(json.dumps({"scene": name, "accel_from_ms": [round(x, 3) for x in ts]}), flush=True)
