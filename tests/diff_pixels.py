#!/usr/bin/env python3
"""Where and how a GPU render differs from the oracle: tests/diff_pixels.py <generator> <seed> [w h]
(generator: a function of lasgun_amd.scenes taking (api, seed)).  Prints the differing pixels with both radiances and
the GPU's trace of each (primary hit, shadow rays); the Python witness (tests/pyref.py) can then be put on the same pixel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
import lasgun_amd as la  # noqa: E402
from oracle_lib import oracle  # noqa: E402


def main():
    gen, seed = sys.argv[1], int(sys.argv[2])
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 48)
    build = getattr(la.scenes, gen)
    G, o = la.api, oracle()
    oacc = o.Accel(build(o, seed))
    o.set_trig_mode(1)
    orad = np.asarray(o.capture_radiance(oacc, w, h, nthreads=8))
    o.set_trig_mode(0)
    acc = G.Accel(build(G, seed))
    G.set_streaming(acc, 0)
    G.set_prune(acc, False)
    rad = np.asarray(G.capture_radiance(acc, w, h))
    same = ((rad.view(np.uint64) == orad.view(np.uint64)) | (np.isnan(rad) & np.isnan(orad))).all(axis=-1)
    ys, xs = np.nonzero(~same)
    print("%s seed %d: %d of %d pixels differ" % (gen, seed, len(ys), w * h))
    for y, x in list(zip(ys, xs))[:12]:
        print("pixel (%d, %d): gpu %s oracle %s" % (x, y, rad[y, x].tolist(), orad[y, x].tolist()))
        print("   gpu trace:", G.trace_pixel(acc, w, h, int(x), int(y)))
    print("info", G.accel_info(acc))


if __name__ == "__main__":
    main()
