"""The lattice addressing of strided subsets (lasgun_amd/csrc/shade.h, pixel_of modes 4 and 5; launch.cpp, set_subset / set_subsets), restated
in Python and checked as arithmetic: every (tile, lane) pair that the formula marks active is a pixel of the subset, every pixel of the
subset is reached exactly once, and the compact output index is the pixel's place in the subset.  The device code itself is compared with
the frame's bytes in tests/test_gpu_parity.py (test_strided_subsets_tile_by_lattice_column); this file is about the formula.

Reference semantics: capture_subset(k, n) writes the pixels {k + i*n < w*h} of the row-major film (lib.rs:110-162)."""
import itertools

import pytest


def rows(w, h, n):
    """the host's table (launch.cpp, lattice_rows): floor(y*w / n) and (y*w) mod n per film row"""
    return [divmod(y * w, n) for y in range(h)]


def place(rt, kk, kdiv, c, n):
    """shade.h: the pixel's x within its row and its place q in its subset, without a division: (x, q)"""
    base, rem = rt
    ph = kk - rem
    wrapped = 1 if ph < 0 else 0
    if wrapped:
        ph += n
    return ph + n * c, base + c + wrapped - kdiv


def single(w, h, n, k):
    """mode 4: a tile = 64 consecutive rows of one lattice column c; -> [(offset, compact index)]"""
    cols, tab = -(-w // n), rows(w, h, n)
    out = []
    for tile in range(-(-h // 64) * cols):
        ty, c = divmod(tile, cols)
        for lane in range(64):
            y = ty * 64 + lane
            if y >= h:
                continue
            x, q = place(tab[y], k % n, k // n, c, n)
            if x < w and q >= 0:
                out.append((y * w + x, q))
    return out


def batch(w, h, n, ks):
    """mode 5: a tile = 64 // m rows of one lattice column, lane = row * m + j; -> [(offset, j, compact index q * m + j)]"""
    m = len(ks)
    nrows, cols, tab = 64 // m, -(-w // n), rows(w, h, n)
    out = []
    for tile in range(-(-h // nrows) * cols):
        ty, c = divmod(tile, cols)
        for lane in range(64):
            r, j = divmod(lane, m)
            y = ty * nrows + r
            if r >= nrows or y >= h:
                continue
            x, q = place(tab[y], ks[j] % n, ks[j] // n, c, n)
            if x < w and q >= 0:
                out.append((y * w + x, j, q * m + j))
    return out


SHAPES = [(96, 80, 10), (200, 131, 100), (131, 67, 8), (64, 200, 64), (257, 19, 33), (120, 72, 16), (97, 61, 97), (97, 61, 50), (4096, 130, 100), (33, 300, 9)]


@pytest.mark.parametrize("w, h, n", SHAPES)
def test_a_subset_is_covered_once_and_nothing_else(w, h, n):
    area = w * h
    for k in sorted({0, 1, n // 2, n - 1, n, n + 3, area - 1, area - n, area // 2}):
        if k < 0:
            continue
        got = single(w, h, n, k)
        want = list(range(k, area, n))
        assert sorted(o for o, _ in got) == want, (w, h, n, k)
        assert all(o == k + i * n for o, i in got), (w, h, n, k)  # the compact index is the pixel's place in the subset


@pytest.mark.parametrize("w, h, n", SHAPES)
def test_a_batch_of_subsets_is_covered_once_and_nothing_else(w, h, n):
    area = w * h
    for m in (2, 3, 7, 10, 32):
        if m > n:
            continue
        ks = sorted(set(itertools.islice(itertools.cycle(range(0, n, max(1, n // m))), m)))  # m distinct residues of n
        ks[-1] = min(ks[-1] + 0, n - 1)
        ks = sorted(set(ks))
        got = batch(w, h, n, ks)
        for j, kj in enumerate(ks):
            mine = sorted(o for o, jj, _ in got if jj == j)
            assert mine == list(range(kj, area, n)), (w, h, n, ks, j)
        assert all(o == ks[j] + (i // len(ks)) * n and i % len(ks) == j for o, j, i in got), (w, h, n, ks)
        assert len({i for _, _, i in got}) == len(got)  # compact places are distinct
