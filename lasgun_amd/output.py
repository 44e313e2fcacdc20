"""`output::render(&scene, [w, h], filename)` of the reference (src/output.rs:5-18): capture + save as PNG.

The reference delegates PNG encoding to the third-party `image` crate; here a minimal RGBA8 PNG
encoder (zlib + CRC from the standard library) does the file part on the host.  The pixels come
from the HIP path exactly as `render()` returns them.
"""
import struct
import zlib

import numpy as np


def write_png(path, rgba):
    """Write an (h, w, 4) uint8 array as an 8-bit RGBA PNG."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, c = a.shape
    if c != 4:
        raise ValueError("expected RGBA")
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def render(api, scene, resolution, filename):
    """src/output.rs:5-18 -- render `scene` at `resolution` through `api` and save it to `filename`."""
    film = api.render(scene, resolution)
    write_png(filename, film.pixels())
    return film
