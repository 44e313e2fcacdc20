#!/usr/bin/env python3
"""How much of a frame is the LATENCY of its slowest tiles rather than throughput: row bands of growing height around the middle of a
config's film (8 rows = 512 tiles at 4096^2: far fewer than the machine has waves), in every organisation.
python tools/tail_probe.py ["4 mesh" ...]"""
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
for name, build, size in CONFIGS:
    if not any(name.startswith(f) for f in filters):
        continue
    scene = build()
    dev = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for org, code in (("megakernel", 0), ("wavefront", 2), ("queue", 3)):
        acc = G.Accel(scene)
        G.set_streaming(acc, code)
        out = {"config": name, "organisation": org}
        for rows in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, size):
            y0 = (size - rows) // 2 // 8 * 8
            ts = []
            for i in range(4):
                e0.record()
                G.capture_rows_device(acc, size, size, y0, y0 + rows, dev.data_ptr(), row0=y0, stream=st)
                e1.record()
                torch.cuda.synchronize()
                if i:
                    ts.append(e0.elapsed_time(e1))
            out["rows_%d" % rows] = round(min(ts), 3)
        print(json.dumps(out), flush=True)
