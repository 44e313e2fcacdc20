#!/usr/bin/env python3
"""How much of a queue-organisation frame its waves spend with nothing to claim (the launch's tail and the waits between levels).
Needs the diagnostic build: tools/build_variant.sh qidle -DLG_QIDLE ; LASGUN_HIP_LIB=lasgun_amd/liblasgun_hip_qidle.so python tools/queue_idle.py ["4 mesh" ...]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402
from bench_configs import CONFIGS  # noqa: E402

G = la.api
G.set_device(0)
filters = sys.argv[1:] or ["4 mesh"]
G.lib.lg_debug_queue_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
for name, build, size in CONFIGS:
    if not any(name.startswith(f) for f in filters):
        continue
    acc = G.Accel(build())
    G.set_streaming(acc, 3)
    dev = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for order in (0, 1, 2):
        G.set_tile_order(acc, order)
        for i in range(3):
            e0.record()
            G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        pk = (C.c_ulonglong * 8)()
        G.lib.lg_debug_queue_packets(acc.h, C.c_void_p(st), pk)
        info = G.accel_info(acc)
        waves = 256 * 16  # the persistent grid: 4 waves per SIMD
        idle_ms_per_wave = pk[7] / 1e5 / waves
        print(json.dumps({"config": name, "tile_order": order, "frame_ms": round(ms, 3), "idle_ticks": int(pk[7]), "waves_assumed": waves,
                          "idle_ms_per_wave": round(idle_ms_per_wave, 3), "idle_share": round(idle_ms_per_wave / ms, 3)}), flush=True)
    # the megakernel: every wave's start and exit (one launch = one grid of persistent waves)
    G.lib.lg_debug_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    G.set_streaming(acc, 0)
    for order in (0, 1, 2):
        G.set_tile_order(acc, order)
        for i in range(3):
            e0.record()
            G.capture_rows_device(acc, size, size, 0, size, dev.data_ptr(), row0=0, stream=st)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        out = (C.c_ulonglong * 32)()
        G.lib.lg_debug_stats(acc.h, 0, out)
        start, end, exits, nw = (1 << 62) - out[0], out[1], out[2], out[3]
        span = (end - start) / 1e5
        idle = (nw * end - exits) / 1e5 / max(nw, 1)
        print(json.dumps({"config": name, "organisation": "megakernel", "tile_order": order, "frame_ms": round(ms, 3), "waves": int(nw), "first_start_to_last_exit_ms": round(span, 3),
                          "idle_ms_per_wave": round(idle, 3), "idle_share": round(idle / span, 3) if span > 0 else None}), flush=True)
