#!/usr/bin/env python3
"""Small frames -- the sizes the reference's own examples use -- one after the other: latency (a frame alone, synchronised: median of 40) and
back-to-back rate (40 frames on the stream, one synchronise), level by level (the organisation whose launch chain a HIP graph replays) and
as the default picks.  Run once per variant (LASGUN_GRAPH=0 / 1) in turn: tools/ab_small_frames.sh.  Every film is compared with the
first variant's through a checksum."""
import hashlib
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
os.environ.setdefault("LASGUN_AUTOTUNE", "2")
import lasgun_amd as la  # noqa: E402

G, S = la.api, la.scenes
FRAMES = [
    ("1a readme 512^2", lambda: S.readme_scene(G), 512),
    ("1b simple ss2 512^2 (9 spp)", lambda: S.simple_scene(G, 2), 512),
    ("2P cornell plastic 512^2", lambda: S.cornell_scene(G, "plastic"), 512),
    ("2G cornell glass 512^2", lambda: S.cornell_scene(G, "glass"), 512),
    ("spooky.rs 768^2 (9 spp)", lambda: S.spooky_scene(G), 768),
    ("playground.rs 512^2 (9 spp)", lambda: S.playground_scene(G), 512),
    ("simplecows.rs 512^2 (9 spp)", lambda: S.simplecows_scene(G), 512),
    ("2G cornell glass 256^2", lambda: S.cornell_scene(G, "glass"), 256),
    ("3 spheres1024 1024^2", lambda: S.spheres_scene(G), 1024),
]


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    G.set_device(0)
    stream = torch.cuda.Stream()
    for name, build, size in FRAMES:
        if only and not any(o in name for o in only):
            continue
        acc = G.Accel(build())
        film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        for label, code in (("level by level", 2), ("default", 1)):
            G.set_streaming(acc, code)
            with torch.cuda.stream(stream):
                for _ in range(6):
                    G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                lat = []
                for _ in range(40):
                    t0 = time.perf_counter()
                    G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
                    torch.cuda.synchronize()
                    lat.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter()
                for _ in range(40):
                    G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                rate = (time.perf_counter() - t0) / 40 * 1e3
            print(json.dumps({"frame": name, "organisation": label, "ran_as": G.last_organisation(acc), "graph": os.environ.get("LASGUN_GRAPH", "1"),
                              "latency_ms_median": round(statistics.median(lat), 4), "latency_ms_min": round(min(lat), 4), "back_to_back_ms": round(rate, 4),
                              "film_sha": hashlib.sha1(film.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)


if __name__ == "__main__":
    main()
