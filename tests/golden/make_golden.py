#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the CPU oracle (oracle/liblasgun_oracle.so).

The reference is Rust and cannot be executed in this pipeline (no toolchain), so
these are NOT outputs of the reference binary: they pin the oracle (and through it the GPU
path) against regressions.  The vectors that DO come from the reference are its 17 inline
known-answer tests, restated as data in tests/kats.py.

Each fixture holds, for one scene of lasgun_amd/scenes.py rendered at a small size:
  rgba      (h, w, 4) uint8   -- oracle, libm trig (what the Rust binary would call)
  radiance  (h, w, 3) float64 -- oracle, PORTABLE trig (bit-comparable with the GPU)
  stats     deterministic work counters of that render (closest-hit shadow rays, as in the reference)
CROPS (golden_cases.py) are 64x64 crops of the full-size films of BASELINE.json's configs[3] and configs[4]
(4096x4096 100k-triangle mesh, 8192x8192 mixed), evaluated pixel by pixel with orc_capture_pixels.
Usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from golden_cases import CASES, CROPS  # noqa: E402
from oracle_lib import oracle  # noqa: E402


def main():
    o = oracle()
    for name, (builder, w, h) in CASES.items():
        scene = builder(o)
        acc = o.Accel(scene)
        o.set_trig_mode(0)
        film = o.Film(w, h)
        o.stats_reset()
        o.capture_subset(0, 1, acc, film)
        stats = o.stats_read()
        rgba = film.pixels()
        o.set_trig_mode(1)
        rad = o.capture_radiance(acc, w, h, nthreads=4)
        film2 = o.Film(w, h)
        o.capture_subset(0, 1, acc, film2)
        o.set_trig_mode(0)
        assert np.array_equal(rgba, film2.pixels()), "RGBA8 depends on the trig implementation for " + name
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=rgba, radiance=rad,
                            stats=np.frombuffer(json.dumps(stats).encode(), dtype=np.uint8))
        print(name, rgba.shape, stats)
    # crops of the full-size config films: rgba (libm trig), radiance (portable trig), no stats (a crop's counters
    # are not a quantity of the reference)
    for name, (builder, w, h, x0, y0, cw, ch) in CROPS.items():
        acc = o.Accel(builder(o))
        o.set_trig_mode(0)
        rgba, _ = o.capture_rect(acc, w, h, x0, y0, x0 + cw, y0 + ch, radiance=False)
        o.set_trig_mode(1)
        rgba2, rad = o.capture_rect(acc, w, h, x0, y0, x0 + cw, y0 + ch)
        o.set_trig_mode(0)
        assert np.array_equal(rgba, rgba2), "RGBA8 depends on the trig implementation for " + name
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=rgba, radiance=rad, rect=np.array([w, h, x0, y0, cw, ch]))
        print(name, rgba.shape, "std %.1f" % rgba[..., :3].std())


if __name__ == "__main__":
    main()
