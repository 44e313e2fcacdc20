"""`output::render` (src/output.rs:5-18): capture + save as PNG.  The file part is lasgun_amd/output.py's own RGBA8 PNG
writer (the reference delegates it to the `image` crate); these tests decode what it writes with an independent
minimal PNG reader (signature, chunk CRCs, IHDR, zlib stream, per-row filter byte) and compare pixel for pixel."""
import os
import struct
import zlib

import numpy as np
import pytest

from lasgun_amd import output


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(data):
        (n,) = struct.unpack(">I", data[at:at + 4])
        tag, body = data[at + 4:at + 8], data[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", data[at + 8 + n:at + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF, tag
        chunks.append((tag, body))
        at += 12 + n
    assert [t for t, _ in chunks][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, depth, ctype, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, lace) == (8, 6, 0, 0, 0)  # 8-bit RGBA, deflate, adaptive filtering, no interlace
    raw = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    assert len(raw) == h * (1 + 4 * w)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 4 * w)
    assert not rows[:, 0].any()  # filter type 0 (None) on every scanline
    return rows[:, 1:].reshape(h, w, 4)


@pytest.mark.parametrize("w, h", [(1, 1), (7, 3), (64, 48), (300, 2)])
def test_png_writer_round_trip(tmp_path, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    path = str(tmp_path / "a.png")
    output.write_png(path, rgba)
    assert np.array_equal(read_png(path), rgba)
    with pytest.raises(ValueError):
        output.write_png(path, rgba[..., :3])


@pytest.mark.gpu
def test_output_render_writes_the_film_as_png(tmp_path):
    import lasgun_amd as la
    from lasgun_amd import scenes as S
    from oracle_lib import oracle
    G = la.api
    w, h = 96, 64
    path = str(tmp_path / "cornell.png")
    film = output.render(G, S.cornell_scene(G, "glass"), (w, h), path)  # src/output.rs:5-18
    got = read_png(path)
    assert np.array_equal(got, film.pixels())
    o = oracle()
    assert np.array_equal(got, o.render(S.cornell_scene(o, "glass"), (w, h)).pixels())
    assert np.all(got[..., 3] == 255)
