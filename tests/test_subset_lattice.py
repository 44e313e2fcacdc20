"""The lattice addressing of strided subsets (lasgun_amd/csrc/shade.h, pixel_of modes 4 and 5; capi.cpp, set_subset / set_subsets), restated
in Python and checked as arithmetic: every (tile, lane) pair that the formula marks active is a pixel of the subset, every pixel of the
subset is reached exactly once, and the compact output index is the pixel's place in the subset.  The device code itself is compared with
the frame's bytes in tests/test_gpu_parity.py (test_strided_subsets_tile_by_lattice_column); this file is about the formula.

Reference semantics: capture_subset(k, n) writes the pixels {k + i*n < w*h} of the row-major film (lib.rs:110-162)."""
import itertools

import pytest


def single(w, h, n, k):
    """mode 4: a tile = 64 consecutive rows of one lattice column m; -> [(offset, compact index)]"""
    cols = -(-w // n)
    out = []
    for tile in range(-(-h // 64) * cols):
        ty, m = divmod(tile, cols)
        for lane in range(64):
            y = ty * 64 + lane
            r, kk = (y * w) % n, k % n
            x = (kk - r if kk >= r else kk + n - r) + n * m
            off = y * w + x
            if y < h and x < w and off >= k:
                out.append((off, (off - k) // n))
    return out


def batch(w, h, n, ks):
    """mode 5: a tile = 64 // m rows of one lattice column, lane = row * m + j; -> [(offset, j, compact index q * m + j)]"""
    m = len(ks)
    rows, cols = 64 // m, -(-w // n)
    out = []
    for tile in range(-(-h // rows) * cols):
        ty, c = divmod(tile, cols)
        for lane in range(64):
            r, j = divmod(lane, m)
            y = ty * rows + r
            kj = ks[j]
            rr, kk = (y * w) % n, kj % n
            x = (kk - rr if kk >= rr else kk + n - rr) + n * c
            off = y * w + x
            if r < rows and y < h and x < w and off >= kj:
                out.append((off, j, ((off - kj) // n) * m + j))
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
