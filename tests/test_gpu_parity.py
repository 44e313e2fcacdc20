"""Parity of the HIP path (through the C ABI) against the CPU oracle on a real MI355X.

Bars: RGBA8 byte-identical; f64 radiance BIT-identical to the oracle in portable-trig mode
(same published algorithm on both sides) and within 1 ulp of fp32 -- the north-star tolerance --
of the oracle in libm mode (what the Rust reference would call).
"""
import json
import os

import numpy as np
import pytest

import lasgun_amd as la
from golden_cases import CASES
from kats import KATS, run_kat, run_surface_kat
from lasgun_amd import scenes as S
from oracle_lib import oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = la.api


def bits(a):
    """Bit patterns with every NaN canonicalised: an invalid operation yields the NaN 0xFFF8... on
    x86 and 0x7FF8... on gfx950 -- same value (NaN, quantised to byte 0 by img.rs:66), different sign
    bit.  NaN POSITIONS must still coincide; everything else is compared bit for bit."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = a.view(np.uint64).copy()
    b[np.isnan(a)] = np.uint64(0x7FF8000000000000)
    return b


def test_device_present_and_library_loaded():
    assert G.device_count() >= 1
    G.set_device(0)


# ---- arithmetic primitives the bit-exactness argument rests on -----------------------------
def _edge_values():
    rng = np.random.default_rng(1234)
    a = np.concatenate([
        rng.uniform(0, 10, 20000), 10.0 ** rng.uniform(-300, 300, 20000), rng.uniform(0, 1e-300, 2000),
        np.array([0.0, -0.0, 1.0, 2.0, 4.0, 0.5, np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308])])
    b = np.concatenate([rng.uniform(-10, 10, a.size - 11), np.array([1.0, 3.0, -0.0, 0.0, np.inf, 7.0, 2.0, 1.0, 3.0, 3.0, 0.1])])
    return a, b


@pytest.mark.parametrize("op,name", [(0, "sqrt"), (1, "div"), (6, "fmin"), (7, "fmax")])
def test_ieee_ops_are_correctly_rounded_on_device(op, name):
    a, b = _edge_values()
    if op == 0:
        a = np.abs(a)
    got, want = G.math_eval(op, a, b), oracle().math_eval(op, a, b)
    if op in (6, 7):  # fmin/fmax of (+0, -0) may legitimately return either zero
        assert np.array_equal(got, want, equal_nan=True)
    else:
        assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("op,name", [(2, "sin"), (3, "cos"), (4, "atan2"), (5, "acos")])
def test_portable_trig_is_bit_identical_cpu_gpu(op, name):
    rng = np.random.default_rng(99 + op)
    n = 60000
    if op in (2, 3):
        a = np.concatenate([rng.uniform(0, 2 * np.pi, n), rng.uniform(-1e4, 1e4, n), [0.0, np.pi, np.pi / 2, 2 * np.pi, 1e-300]])
        b = np.zeros_like(a)
    elif op == 4:
        a = np.concatenate([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-12, 0, n), [0.0, -0.0, 1.0, -1.0, 0.0]])
        b = np.concatenate([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), [1.0, -1.0, 0.0, -0.0, 0.0]])
    else:
        a = np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-16, 0, n), [1.0, -1.0, 0.0, 0.5, -0.5]])
        b = np.zeros_like(a)
    assert np.array_equal(bits(G.math_eval(op, a, b)), bits(oracle().math_eval(op, a, b)))


def test_to_byte_quantisation():
    rng = np.random.default_rng(5)
    k = np.arange(0, 256)
    a = np.concatenate([rng.uniform(-0.5, 1.5, 50000), (k + 0.5) / 255.0, np.nextafter((k + 0.5) / 255.0, 0), np.nextafter((k + 0.5) / 255.0, 2),
                        [np.nan, np.inf, -np.inf, -0.0, 1.0, 0.0]])
    assert np.array_equal(G.math_eval(8, a), oracle().math_eval(8, a))


# ---- the reference's 17 inline KATs on the device intersectors -----------------------------
@pytest.mark.parametrize("kat", KATS, ids=[k[0] for k in KATS])
def test_reference_kat_on_device(kat):
    got = run_kat(G, kat)
    o = oracle()
    o.set_trig_mode(1)
    try:
        want = run_kat(o, kat)
    finally:
        o.set_trig_mode(0)
    assert got["t"] == want["t"]
    assert np.array_equal(bits(got["ng"]), bits(want["ng"])) and np.array_equal(bits(got["ns"]), bits(want["ns"]))


def test_surface_interaction_kat_on_device():
    run_surface_kat(G)


# ---- golden fixtures ------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_matches_golden(name):
    builder, w, h = CASES[name]
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    scene = builder(G)
    acc = G.Accel(scene)
    film = G.Film(w, h)
    G.capture_subset(0, 1, acc, film)
    assert np.array_equal(film.pixels(), z["rgba"])
    rad = G.capture_radiance(acc, w, h)
    assert np.array_equal(bits(rad), bits(z["radiance"]))
    stats = json.loads(bytes(z["stats"]).decode())
    got = G.capture_stats(acc, w, h)
    for key in ("primary_rays", "shadow_rays", "secondary_rays", "hits"):
        assert got[key] == stats[key], key
    # any-hit shadow rays may only REMOVE work relative to the reference's closest-hit shadow rays
    for key in ("nodes_tested", "spheres_tested", "cuboids_tested", "triangles_tested", "accel_entries"):
        assert got[key] <= stats[key], key


# ---- seeded scenes at sizes the oracle finishes in seconds ------------------------------------
MID = {
    "cornell_plastic_256": (lambda api: S.cornell_scene(api, "plastic"), 256, 256),
    "cornell_glass_256": (lambda api: S.cornell_scene(api, "glass"), 256, 256),
    "cornell_plastic_ss1": (lambda api: S.cornell_scene(api, "plastic", supersampling=1), 96, 96),
    "simple_ss2_160": (lambda api: S.simple_scene(api, 2), 160, 160),
    "simplereflect_160": (lambda api: S.simple_scene(api, 0, True), 160, 128),
    "spheres_512": (S.spheres_scene, 512, 512),
    "spheres_seed7_300": (lambda api: S.spheres_scene(api, 300, seed=7), 200, 136),
    "mesh_glass_128": (lambda api: S.mesh_scene(api, 64, 64, "glass"), 128, 128),
    "mesh_plastic_flat_128": (lambda api: S.mesh_scene(api, 48, 48, "plastic", smoothing=False), 128, 96),
    "mixed_128": (lambda api: S.mixed_scene(api, 256, 48, 48), 128, 128),
    "instanced_mesh_176": (S.instanced_scene, 176, 144),
    "kitchen_sink_persp": (lambda api: S.kitchen_sink_scene(api, "perspective"), 192, 160),
    "kitchen_sink_ortho": (lambda api: S.kitchen_sink_scene(api, "orthographic", recursion=4, supersampling=0), 160, 120),
    "kitchen_sink_rec0": (lambda api: S.kitchen_sink_scene(api, "perspective", recursion=0, supersampling=2), 96, 72),
    # OBJ forms beyond `f a b c`: negative indices, v//vn, a 4- and a 5-vertex polygon (first three vertices), o / g groups
    "exotic_obj_smooth": (lambda api: S.exotic_obj_scene(api, True), 160, 120),
    "exotic_obj_flat": (lambda api: S.exotic_obj_scene(api, False), 96, 72),
    # the reference's other example programs: a mesh added straight to the root; white ambient light, glass beside a mesh three groups deep,
    # the root rotated in place before its groups are added (spooky.rs renders 768 x 768: a third of it here); a polygon OBJ, arches of
    # scaled cubes and spheres three groups deep, the root rotated in place AFTER its children were added (simplecows.rs:90)
    "playground_ss2": (S.playground_scene, 128, 128),
    "spooky_ss2_256": (S.spooky_scene, 256, 256),
    "simplecows_ss2": (S.simplecows_scene, 160, 160),
    # ragged / tiny films: tiles cut by the right and bottom edges, and a single pixel
    "ragged_67x13": (lambda api: S.cornell_scene(api, "glass"), 67, 13),
    "ragged_5x131": (lambda api: S.spheres_scene(api, 64, seed=11), 5, 131),
    "one_pixel": (lambda api: S.simple_scene(api, 1), 1, 1),
    # sparse hits: one small sphere in front of the background (most 8x8 tiles have no hit or a few: the compacted part of the hit queue)
    "readme_sparse_200x120": (S.readme_scene, 200, 120),
}


@pytest.mark.parametrize("name", list(MID))
def test_gpu_matches_oracle(name):
    builder, w, h = MID[name]
    o = oracle()
    oacc = o.Accel(builder(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    gscene = builder(G)
    gfilm = G.render(gscene, (w, h))  # lib.rs:46 path: Accel::from + capture
    assert np.array_equal(gfilm.pixels(), ofilm.pixels())
    gacc = G.Accel(gscene)
    grad = G.capture_radiance(gacc, w, h)
    o.set_trig_mode(1)
    try:
        orad_p = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    assert np.array_equal(bits(grad), bits(orad_p))  # same algorithm on both sides: bit-exact
    orad = o.capture_radiance(oacc, w, h, nthreads=8)  # libm trig, as the Rust reference
    tol = np.maximum(np.abs(orad), 1e-30) * 2.0 ** -23  # 1 ulp of fp32
    assert np.all(np.abs(grad - orad) <= tol)


# ---- opt-in fast mode: must reproduce the parity path's bytes and radiance bits ------------------
@pytest.mark.parametrize("name", list(MID))
def test_fast_mode_matches_parity_mode(name):
    builder, w, h = MID[name]
    acc = G.Accel(builder(G))
    film = G.Film(w, h)
    G.capture_subset(0, 1, acc, film)
    rad = G.capture_radiance(acc, w, h)
    st = G.capture_stats(acc, w, h)
    G.set_mode(acc, True)
    ffilm = G.Film(w, h)
    G.capture_subset(0, 1, acc, ffilm)
    frad = G.capture_radiance(acc, w, h)
    fst = G.capture_stats(acc, w, h)
    assert np.array_equal(ffilm.pixels(), film.pixels())
    assert np.array_equal(bits(frad), bits(rad))
    for key in ("primary_rays", "shadow_rays", "secondary_rays", "hits"):
        assert fst[key] == st[key]


def test_fast_mode_full_size_films_are_identical():
    import torch
    cases = [(S.spheres_scene, 4096), (lambda api: S.mesh_scene(api, 224, 224, "glass"), 2048),
             (lambda api: S.mixed_scene(api), 2048), (lambda api: S.kitchen_sink_scene(api, "perspective", 3, 0), 1024)]
    for builder, size in cases:
        acc = G.Accel(builder(G))
        a = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        b = torch.zeros_like(a)
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        G.capture_rows_device(acc, size, size, 0, size, a.data_ptr(), row0=0)
        G.synchronize(acc)
        G.set_mode(acc, True)
        G.capture_rows_device(acc, size, size, 0, size, b.data_ptr(), row0=0)
        G.synchronize(acc)
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", list(MID))
def test_wavefront_pipeline_matches_oracle(name):
    """li() level by level (hits compacted into queues, specular children queued per recursion level, levels combined
    bottom-up in the (output + reflected) + refracted order of integrate.rs:79) against the oracle: bytes and radiance
    bits, whole film and a strided subset, reference and fast traversal -- on every scene of the suite, recursion 0 .. 4,
    supersampling, ragged and one-pixel films included."""
    builder, w, h = MID[name]
    o = oracle()
    oacc = o.Accel(builder(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    o.set_trig_mode(1)
    try:
        orad = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    acc = G.Accel(builder(G))
    G.set_streaming(acc, 2)  # streamed whatever the size of the launch
    for fast in (False, True):
        G.set_mode(acc, fast)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), ofilm.pixels()), fast
        assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), fast
        sub = G.Film.new_with_output(w, h, np.full((h, w, 4), 7, np.uint8))
        G.capture_subset(1, 3, acc, sub)
        idx = np.arange(1, w * h, 3)
        got, want = sub.pixels().reshape(-1, 4), ofilm.pixels().reshape(-1, 4)
        assert np.array_equal(got[idx], want[idx])
        mask = np.ones(w * h, bool); mask[idx] = False
        assert np.all(got[mask] == 7)


def test_wavefront_pipeline_in_small_chunks():
    """A memory budget that cuts the film into many chunks (every chunk sized so that its worst-case queues fit): same film."""
    w, h = 200, 152
    o = oracle()
    for builder in (lambda api: S.cornell_scene(api, "glass"), lambda api: S.kitchen_sink_scene(api, "perspective", recursion=4, supersampling=1)):
        ofilm = o.Film(w, h)
        o.capture_subset_mt(0, 1, o.Accel(builder(o)), ofilm, 8)
        os.environ["LASGUN_WF_BUDGET_MB"] = "1"  # read when an accel first uses the pipeline: 64 MiB is the floor
        try:
            acc = G.Accel(builder(G))
            G.set_streaming(acc, 2)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
        finally:
            del os.environ["LASGUN_WF_BUDGET_MB"]
        assert np.array_equal(film.pixels(), ofilm.pixels())


@pytest.mark.parametrize("name", list(MID))
def test_queue_organisation_matches_oracle(name):
    """The queue organisation (k_queue.hip: every recursion level of the film in ONE persistent launch, 64-ray packets pulled
    from per-level queues, levels combined bottom-up in the (output + reflected) + refracted order of integrate.rs:79) against
    the oracle: bytes and radiance bits, whole film and a strided subset, plain and pruned reference walk, scene tables in LDS
    and in L2 -- on every scene of the suite, recursion 0 .. 4, supersampling, ragged and one-pixel films included."""
    builder, w, h = MID[name]
    o = oracle()
    oacc = o.Accel(builder(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    o.set_trig_mode(1)
    try:
        orad = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    acc = G.Accel(builder(G))
    G.set_streaming(acc, 3)
    for prune, lds in ((False, True), (True, True), (False, False), (True, False)):
        G.set_prune(acc, prune); G.set_lds_scene(acc, lds)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), ofilm.pixels()), (prune, lds)
        assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), (prune, lds)
        sub = G.Film.new_with_output(w, h, np.full((h, w, 4), 7, np.uint8))
        G.capture_subset(1, 3, acc, sub)
        idx = np.arange(1, w * h, 3)
        got, want = sub.pixels().reshape(-1, 4), ofilm.pixels().reshape(-1, 4)
        assert np.array_equal(got[idx], want[idx])
        mask = np.ones(w * h, bool); mask[idx] = False
        assert np.all(got[mask] == 7)


def test_queue_organisation_in_small_chunks():
    """A memory budget that cuts the film into many chunks (each sized so that its worst-case queues fit), several frames one
    after the other on one launch context (the control words are cleared per launch): same film every time."""
    w, h = 200, 152
    o = oracle()
    for builder in (lambda api: S.cornell_scene(api, "glass"), lambda api: S.kitchen_sink_scene(api, "perspective", recursion=4, supersampling=1)):
        ofilm = o.Film(w, h)
        o.capture_subset_mt(0, 1, o.Accel(builder(o)), ofilm, 8)
        os.environ["LASGUN_QUEUE_BUDGET_MB"] = "4"  # read when an accel first uses the organisation
        try:
            acc = G.Accel(builder(G))
            G.set_streaming(acc, 3)
            for _ in range(3):
                film = G.Film(w, h)
                G.capture_subset(0, 1, acc, film)
                assert np.array_equal(film.pixels(), ofilm.pixels())
        finally:
            del os.environ["LASGUN_QUEUE_BUDGET_MB"]


SUPERSAMPLED = {
    "simple_ss3": (lambda api: S.simple_scene(api, 3), 104, 72),
    "simple_reflect_ss2": (lambda api: S.simple_scene(api, 2, True), 96, 96),
    "cornell_glass_ss2": (lambda api: S.cornell_scene(api, "glass", supersampling=2), 96, 80),
    "kitchen_sink_ss2_rec3": (lambda api: S.kitchen_sink_scene(api, "perspective", recursion=3, supersampling=2), 88, 64),
    "mesh_glass_ss2": (lambda api: S.mesh_scene(api, 32, 32, "glass", supersampling=2), 72, 72),
    "ragged_ss3_13x9": (lambda api: S.cornell_scene(api, "glass", supersampling=3), 13, 9),
}


@pytest.mark.parametrize("name", list(SUPERSAMPLED))
def test_a_pixels_samples_side_by_side_or_in_a_row_are_the_same_film(name):
    """lg_accel_set_sample_order: a supersampled launch's level-0 work items as (tile, sample) pairs, every sample's li() parked and summed in
    the reference's order by the resolve pass (integrate.rs:17-20) -- against one sample after the other (rounds 1-4) and against the oracle,
    in every organisation: bytes, radiance bits, a strided subset, a batch of subsets."""
    builder, w, h = SUPERSAMPLED[name]
    o = oracle()
    oacc = o.Accel(builder(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    o.set_trig_mode(1)
    try:
        orad = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    want = ofilm.pixels().reshape(-1, 4)
    acc = G.Accel(builder(G))
    for org in (0, 2, 3, 1):
        G.set_streaming(acc, org)
        for order in (0, 1, None):
            G.set_sample_order(acc, order)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            assert np.array_equal(film.pixels(), ofilm.pixels()), (org, order)
            if org == 0 and order is not None:
                assert G.last_organisation(acc) == "megakernel, middle-out" + (", samples in a row" if order == 1 else "")
            assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), (org, order)
            buf = np.full((h, w, 4), 7, np.uint8)
            G.capture_subset(2, 5, acc, G.Film.new_with_output(w, h, buf))
            G.capture_subsets((0, 3), 5, acc, G.Film.new_with_output(w, h, buf))
            got = buf.reshape(-1, 4)
            for k in (0, 2, 3):
                assert np.array_equal(got[k::5], want[k::5]), (org, order, k)
            for k in (1, 4):
                assert np.all(got[k::5] == 7), (org, order, k)
    with pytest.raises(la.LasgunError):
        G.set_sample_order(acc, 2)


@pytest.mark.parametrize("name", ["cornell_glass_256", "kitchen_sink_persp", "mesh_glass_128", "ragged_67x13", "one_pixel", "spooky_ss2_256", "simple_ss2_160"])
@pytest.mark.parametrize("org", [0, 3])
def test_tiles_in_parts_are_the_same_film(name, org):
    """lg_accel_set_tile_parts: the megakernel and the queue organisation hand a tile out whole or in 2 / 4 / 8 parts of 32 / 16 / 8 lanes (more
    waves at work on a small launch; the queue's deeper packets are then as narrow as their parents) -- bytes, radiance bits, a strided subset
    and a batch of subsets against the oracle, with a pixel's samples side by side and in a row."""
    builder, w, h = MID[name]
    o = oracle()
    oacc = o.Accel(builder(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    o.set_trig_mode(1)
    try:
        orad = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    want = ofilm.pixels().reshape(-1, 4)
    acc = G.Accel(builder(G))
    G.set_streaming(acc, org)
    for parts in (2, 4, 8, 1, None):
        G.set_tile_parts(acc, parts)
        for order in (0, 1):
            G.set_sample_order(acc, order)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            assert np.array_equal(film.pixels(), ofilm.pixels()), (parts, order)
            if w * h > 64:
                assert ("tiles in parts" in G.last_organisation(acc)) == (parts in (2, 4, 8)), (parts, G.last_organisation(acc))
            assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), (parts, order)
            buf = np.full((h, w, 4), 7, np.uint8)
            G.capture_subset(2, 5, acc, G.Film.new_with_output(w, h, buf))
            G.capture_subsets((0, 3), 5, acc, G.Film.new_with_output(w, h, buf))
            got = buf.reshape(-1, 4)
            for k in (0, 2, 3):
                assert np.array_equal(got[k::5], want[k::5]), (parts, order, k)
            for k in (1, 4):
                assert np.all(got[k::5] == 7), (parts, order, k)
    with pytest.raises(la.LasgunError):
        G.set_tile_parts(acc, 3)


def test_samples_side_by_side_in_small_chunks():
    """The level-by-level pipeline and the queue organisation under a memory budget that cuts a supersampled film into many chunks (a chunk
    holds pixel tiles x samples work items): same film."""
    w, h = 168, 120
    o = oracle()
    builder = lambda api: S.kitchen_sink_scene(api, "perspective", recursion=3, supersampling=2)
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, o.Accel(builder(o)), ofilm, 8)
    for env, org in (("LASGUN_WF_BUDGET_MB", 2), ("LASGUN_QUEUE_BUDGET_MB", 3)):
        os.environ[env] = "4"
        try:
            acc = G.Accel(builder(G))
            G.set_streaming(acc, org)
            for _ in range(2):
                film = G.Film(w, h)
                G.capture_subset(0, 1, acc, film)
                assert np.array_equal(film.pixels(), ofilm.pixels()), org
        finally:
            del os.environ[env]


@pytest.mark.parametrize("name", ["spheres_512", "cornell_plastic_ss1", "simple_ss2_160", "mixed_128", "instanced_mesh_176", "ragged_5x131", "one_pixel"])
def test_streaming_pipeline_and_megakernel_agree(name):
    """The kernel organisations (and both traversal modes under each) give the same bytes and bits."""
    builder, w, h = MID[name]
    acc = G.Accel(builder(G))
    outs = []
    for streaming, prune in ((2, False), (2, True), (0, False), (0, True), (3, False), (3, True)):  # 2 = the wavefront pipeline even for small films, 0 = megakernel, 3 = queue organisation; plain / pruned reference walk
        for fast in (False, True):
            G.set_streaming(acc, streaming)
            G.set_prune(acc, prune)
            G.set_mode(acc, fast)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            sub = G.Film.new_with_output(w, h, np.full((h, w, 4), 7, np.uint8))
            G.capture_subset(1, 3, acc, sub)
            outs.append((film.pixels(), bits(G.capture_radiance(acc, w, h)), sub.pixels()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])


@pytest.mark.parametrize("name", ["spheres_512", "cornell_plastic_ss1", "simple_ss2_160", "mixed_128", "instanced_mesh_176", "ragged_5x131", "one_pixel"])
def test_lds_resident_scene_matches_global_tables(name):
    """Streaming traversal kernels with the scene tables in LDS (1024-lane workgroups) vs in HBM/L2."""
    builder, w, h = MID[name]
    acc = G.Accel(builder(G))
    G.set_streaming(acc, 2)
    outs = []
    for lds in (True, False):
        fits = G.set_lds_scene(acc, lds)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        sub = G.Film.new_with_output(w, h, np.full((h, w, 4), 7, np.uint8))
        G.capture_subset(2, 5, acc, sub)
        outs.append((film.pixels(), bits(G.capture_radiance(acc, w, h)), sub.pixels()))
    assert fits or name == "mixed_128", "these scenes are small enough for the LDS variant"
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    ofilm = oracle().Film(w, h)
    o = oracle()
    o.capture_subset_mt(0, 1, o.Accel(builder(o)), ofilm, 16)
    assert np.array_equal(outs[0][0], ofilm.pixels())


@pytest.mark.parametrize("nlights, w, h", [(3, 160, 96), (32, 96, 64), (33, 96, 64), (40, 64, 64)])
def test_many_lights(nlights, w, h):
    """One any-hit pass per light; 32 lights is the last count the streaming pipeline's visibility word holds,
    from 33 on the megakernel takes over whatever the setting."""
    def build(api):
        sc = S.spheres_scene(api, 600)
        for i in range(nlights - 1):
            sc.add_point_light([-1.5 + 3.0 * i / max(nlights - 1, 1), 1.5, 1.0 - 0.05 * i], [0.05, 0.04, 0.06], [1.0, 0.0, 0.05])
        return sc
    o = oracle()
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, o.Accel(build(o)), ofilm, 16)
    acc = G.Accel(build(G))
    for streaming in (0, 2, 3):
        G.set_streaming(acc, streaming)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), ofilm.pixels()), streaming


REFILL_CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import lasgun_amd as la
from oracle_lib import oracle
G, S = la.api, la.scenes
def build(api, scene):
    if scene == "spheres_3_lights":
        sc = S.spheres_scene(api, 600)
        for i in range(2):
            sc.add_point_light([-1.5 + 3.0 * i, 1.5, 1.0 - 0.05 * i], [0.05, 0.04, 0.06], [1.0, 0.0, 0.05])
        return sc
    return S.readme_scene(api) if scene == "readme_sparse" else S.cornell_scene(api, "glass")
w = h = 1024
for scene in ("spheres_3_lights", "readme_sparse", "cornell_glass"):
    acc = G.Accel(build(G, scene))
    films = {}
    for streaming in (0, 2):
        G.set_streaming(acc, streaming)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        films[streaming] = film.pixels().copy()
    assert np.array_equal(films[0], films[2]), scene
    o = oracle()
    n = 97
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, n, o.Accel(build(o, scene)), ofilm, 16)
    assert np.array_equal(films[2].reshape(-1, 4)[::n], ofilm.pixels().reshape(-1, 4)[::n]), scene
print("refill ok")
"""


def test_refilling_shadow_pass_is_the_same_film():
    """The opt-in experiment of round 6 (LASGUN_REFILL=1; measured slower, k_wavefront.hip: launch_wf_trace): level by level, a level-0 shadow
    pass of >= 4 tiles per wave over an LDS-resident scene as ONE persistent walk per wave whose lanes take the next hit as soon as their own
    shadow ray is done (ShadowRefill; walk.h: REFILL).  1024^2 is the smallest film that takes that path: several lights per hit (the next
    light's ray without a new claim), a film whose hits are sparse (appended part of the hit queue, holes), a recursive scene (only level 0
    refills).  Same bytes as the megakernel -- which has no such pass -- and as the oracle on a strided sample; in a child process, because the
    switch is read once.  Reference: src/light/point.rs:42-54 (one shadow ray per hit and light; `isect.t < 1.0`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LASGUN_REFILL="1")
    p = subprocess.run([sys.executable, "-c", REFILL_CHILD % (root, os.path.join(root, "tests"))], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "refill ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


@pytest.mark.parametrize("w, h", [(4096, 1), (1, 777), (3, 3), (8192, 2)])
def test_extreme_film_shapes(w, h):
    o = oracle()
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, o.Accel(S.cornell_scene(o, "plastic")), ofilm, 16)
    acc = G.Accel(S.cornell_scene(G, "plastic"))
    for streaming in (0, 2, 3):
        G.set_streaming(acc, streaming)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), ofilm.pixels()), streaming


# films whose 8x8 tiles number just below, at and just above the sizes at which a persistent launch changes how it ends (kcommon.h: fewer
# tiles than the grid has waves -- 4,096 on an MI355X with the scene in LDS --, at most twice as many: "final" tiles; more: plain bands)
@pytest.mark.parametrize("w, h", [(8, 8), (360, 728), (512, 512), (136, 1928), (8, 65528), (1024, 512), (24, 21848), (1000, 600)])
@pytest.mark.parametrize("kind", ["glass", "plastic"])
def test_launches_end_without_losing_tiles(w, h, kind):
    """1, 4095, 4096, 4097, 8191, 8192, 8193 and 9375 tiles: every pixel of the film against the oracle, in each organisation and by default
    (a tile nobody claimed would keep the bytes the film was created with)."""
    o = oracle()
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, o.Accel(S.cornell_scene(o, kind)), ofilm, 16)
    acc = G.Accel(S.cornell_scene(G, kind))
    for streaming in (1, 0, 2, 3):
        G.set_streaming(acc, streaming)
        film = G.Film.new_with_output(w, h, np.full((h, w, 4), 99, np.uint8))
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), ofilm.pixels()), streaming


def test_fast_mode_adversarial_scenes():
    """Scenes built against fast mode's margins (wall-sized spheres, lights a hair from a surface, needle boxes, twin
    spheres); seeds 75 and 375 differed by one pixel each before the fast walk's winner was put to the reference tree's
    own box tests (a shadow ray leaving a giant sphere 3e-11 above its box)."""
    ns = {"scene": lambda seed: S.adversarial_scene(G, seed), "scene2": lambda seed: S.adversarial_mesh_scene(G, seed)}
    w, h = 128, 96
    for seed in (75, 375, 1, 2, 3, 4, 5, 6):
        acc = G.Accel(ns["scene"](seed))
        outs = []
        for fast in (False, True):
            G.set_mode(acc, fast)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            outs.append((film.pixels(), bits(G.capture_radiance(acc, w, h))))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), seed
    # degenerate meshes under nested anisotropic transforms (second generator); a non-unit rotation axis makes an
    # aggregate's transform and inverse disagree (transform.rs:144-148) and fast mode is refused for such a scene
    refused = 0
    for seed in (381, 393, 414, 589, 591, 7, 8, 9):
        acc2 = G.Accel(ns["scene2"](seed))
        try:
            G.set_mode(acc2, True)
        except la.LasgunError as e:
            # ... and for a mesh whose smallest triangle is far below its coordinates' precision (round 4)
            assert "inverse" in str(e) or "smallest triangle" in str(e)
            refused += 1
            continue
        outs = []
        for fast in (False, True):
            G.set_mode(acc2, fast)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc2, film)
            outs.append(film.pixels())
        assert np.array_equal(outs[0], outs[1]), seed
    # the trace hook that found the cause: the primary hit is a wall sphere in both modes
    r0, r1 = G.trace_pixel(acc, w, h, 40, 50, False), G.trace_pixel(acc, w, h, 40, 50, True)
    assert r0["ref"] == r1["ref"] and r0["t"] == r1["t"] and [s[1] for s in r0["shadow"]] != []


# ---- fuzz parity: seeded random scenes with duplicated / touching primitives (exact ties in t) ----
@pytest.mark.parametrize("seed", list(range(32)))
def test_random_scene_parity(seed):
    w, h = 56, 40
    o = oracle()
    try:
        oacc = o.Accel(S.random_scene(o, seed))
    except la.LasgunError:
        with pytest.raises(la.LasgunError):
            G.Accel(S.random_scene(G, seed))  # what the reference cannot build, neither side builds
        return
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    o.set_trig_mode(1)
    try:
        orad = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    acc = G.Accel(S.random_scene(G, seed))
    # megakernel and wavefront pipeline in either traversal mode, the queue organisation
    for streaming, fast, packet in ((0, False, False), (0, True, False), (2, False, False), (2, True, False), (3, False, False)):
        G.set_streaming(acc, streaming)
        G.set_mode(acc, fast)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), ofilm.pixels()), (seed, streaming, fast, packet)
        assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), (seed, streaming, fast, packet)


# ---- driver semantics (lib.rs:55-162) -------------------------------------------------------
def test_capture_subset_partitions_and_preserves_other_pixels():
    w, h = 96, 80
    scene = S.cornell_scene(G, "glass")
    acc = G.Accel(scene)
    full = G.Film(w, h)
    G.capture_subset(0, 1, acc, full)
    want = full.pixels()
    # progressive use: n shuffled subsets into ONE film (www/renderer.ts:103-120)
    n = 7
    buf = np.full((h, w, 4), 9, np.uint8)
    film = G.Film.new_with_output(w, h, buf)
    for k in (3, 0, 6, 1, 5):
        G.capture_subset(k, n, acc, film)
    flat, wf = buf.reshape(-1, 4), want.reshape(-1, 4)
    done = np.zeros(w * h, bool)
    for k in (3, 0, 6, 1, 5):
        done[k::n] = True
    assert np.array_equal(flat[done], wf[done])
    assert np.all(flat[~done] == 9)  # untouched pixels keep their previous bytes
    for k in (2, 4):
        G.capture_subset(k, n, acc, film)
    assert np.array_equal(buf, want)
    # k beyond the area writes nothing; ragged last tile
    G.capture_subset(w * h + 5, 3, acc, film)
    assert np.array_equal(buf, want)


def test_alternating_periods_and_films_on_one_stream_keep_their_row_tables_apart():
    """A caller that alternates periods n -- and film sizes -- on ONE stream (progressive refinement with a varying n, two films): the lattice
    addressing keeps a row table per (w, h, n) in the stream's launch context, the last four of them, each uploaded from pinned staging on the
    stream itself (launch.cpp: lattice_rows; round 5 kept one and synchronised the whole device on every change).  Seven (film, period) pairs
    taken in turn three times over, no synchronisation in between: tables are evicted while launches that read older ones are still queued.
    Every film ends as the frame."""
    import torch
    acc = G.Accel(S.cornell_scene(G, "glass"))
    G.set_streaming(acc, 2)
    cases = [(160, 96, 9), (160, 96, 17), (160, 96, 23), (131, 67, 8), (131, 67, 50), (160, 96, 64), (200, 80, 11)]
    stream = torch.cuda.Stream()
    refs, films = {}, {}
    for w, h, n in cases:
        if (w, h) not in refs:
            full = G.Film(w, h)
            G.capture_subset(0, 1, acc, full)
            refs[(w, h)] = full.pixels().copy()
        films[(w, h, n)] = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for third in range(3):  # each film gets a third of its subsets per round: every (w, h, n) comes back after six other tables were used
            for w, h, n in cases:
                for k in range(third, n, 3):
                    G.capture_subset_device(k, n, acc, w, h, films[(w, h, n)].data_ptr(), stream=stream.cuda_stream)
    torch.cuda.synchronize()
    for (w, h, n), film in films.items():
        assert np.array_equal(film.cpu().numpy(), refs[(w, h)]), (w, h, n)


@pytest.mark.parametrize("w, h, n, ks", [(96, 80, 10, (0, 3, 9)), (200, 131, 100, (0, 57, 99)), (131, 67, 8, (7, 2)), (64, 200, 64, (0, 63, 31)),
                                         (257, 19, 33, (32, 5)), (120, 72, 16, (120 * 72 - 3, 130, 15 + 16 * 7)), (97, 61, 97, (0, 96)), (97, 61, 50, (49,))])
def test_strided_subsets_tile_by_lattice_column(w, h, n, ks):
    """capture_subset(k, n) with 8 <= n <= w is addressed as the lattice it is (shade.h, mode 4: a tile = 64 rows of one lattice column): exactly
    the pixels {k + i*n} get the frame's bytes -- into a host film (compact device output scattered by the host) and into a device film, in every
    organisation, k >= n included -- and nothing else is touched."""
    import torch
    scene = S.kitchen_sink_scene(G, "perspective", recursion=2, supersampling=1)
    acc = G.Accel(scene)
    full = G.Film(w, h)
    G.capture_subset(0, 1, acc, full)
    want = full.pixels().reshape(-1, 4)
    for org in (0, 2, 3):
        G.set_streaming(acc, org)
        buf = np.full((h, w, 4), 9, np.uint8)
        film = G.Film.new_with_output(w, h, buf)
        dev = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda")
        done = np.zeros(w * h, bool)
        for k in ks:
            G.capture_subset(k, n, acc, film)
            G.capture_subset_device(k, n, acc, w, h, dev.data_ptr())
            done[k::n] = True
        G.synchronize(acc)
        torch.cuda.synchronize()
        # ... and the same subsets as ONE batch (lg_capture_subsets: mode 5, 64 / m rows of a lattice column per tile)
        bbuf = np.full((h, w, 4), 9, np.uint8)
        G.capture_subsets(list(ks), n, acc, G.Film.new_with_output(w, h, bbuf))
        bdev = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda")
        G.capture_subsets_device(list(ks), n, acc, w, h, bdev.data_ptr())
        G.synchronize(acc)
        torch.cuda.synchronize()
        for got in (buf.reshape(-1, 4), dev.cpu().numpy().reshape(-1, 4), bbuf.reshape(-1, 4), bdev.cpu().numpy().reshape(-1, 4)):
            assert np.array_equal(got[done], want[done]), (org, w, h, n)
            assert np.all(got[~done] == 9), (org, w, h, n)


def test_capture_subsets_batch_equals_the_single_calls_and_the_oracle():
    """lg_capture_subsets: several subsets of one n as ONE render (the progressive caller's batch, www/renderer.ts:103-120) writes
    exactly what the single capture_subset calls write -- which the oracle's capture_subset (lib.rs:110-162) pins -- and nothing else."""
    import torch
    w, h = 97, 61  # (not a multiple of 8 or of n: ragged tiles, a ragged last period)
    for scene_of, n, ks in ((lambda api: S.cornell_scene(api, "glass"), 7, (3, 0, 5)), (lambda api: S.simple_scene(api, 1), 100, (17, 99, 0, 42, 41, 63))):
        acc = G.Accel(scene_of(G))
        o = oracle()
        oacc = o.Accel(scene_of(o))
        want = np.full((h, w, 4), 9, np.uint8)
        ofilm = o.Film.new_with_output(w, h, want)
        for k in ks:
            o.capture_subset(k, n, oacc, ofilm)
        buf = np.full((h, w, 4), 9, np.uint8)
        film = G.Film.new_with_output(w, h, buf)
        G.capture_subsets(list(ks) + [ks[0], w * h + 3], n, acc, film)  # a repeated k counts once, a k behind the film is an empty subset
        assert np.array_equal(buf, want)
        done = np.zeros(w * h, bool)
        for k in ks:
            done[k::n] = True
        assert np.all(buf.reshape(-1, 4)[~done] == 9)
        # the device-film form; then the rest of the subsets: the whole frame
        dev = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda")
        G.capture_subsets_device(ks, n, acc, w, h, dev.data_ptr())
        G.synchronize(acc)
        assert np.array_equal(dev.cpu().numpy(), want)
        G.capture_subsets_device([k for k in range(n) if k not in ks], n, acc, w, h, dev.data_ptr())
        G.synchronize(acc)
        full = G.Film(w, h)
        G.capture_subset(0, 1, acc, full)
        assert np.array_equal(dev.cpu().numpy(), full.pixels())
        # every k of 0 .. n-1 in one batch IS the frame (rendered as one); an empty batch and n larger than the film are fine
        buf2 = np.full((h, w, 4), 9, np.uint8)
        film2 = G.Film.new_with_output(w, h, buf2)
        G.capture_subsets(list(range(n))[::-1], n, acc, film2)
        assert np.array_equal(buf2, full.pixels())
        G.capture_subsets([], n, acc, film2)
        big = np.full((h, w, 4), 9, np.uint8)
        G.capture_subsets([5, w * h - 1], w * h + 10, acc, G.Film.new_with_output(w, h, big))
        flat = big.reshape(-1, 4)
        assert np.array_equal(flat[5], full.pixels().reshape(-1, 4)[5]) and np.array_equal(flat[-1], full.pixels().reshape(-1, 4)[-1])
        assert np.all(flat[6:-1] == 9) and np.all(flat[:5] == 9)
    with pytest.raises(la.LasgunError):
        G.capture_subsets([0], 0, acc, film)


def test_capture_subsets_in_every_organisation():
    """the batched addressing mode through the megakernel, the level-by-level pipeline and the queue organisation: same film"""
    w, h, n, ks = 120, 72, 10, (9, 2, 4, 7)
    scene = S.cornell_scene(G, "glass")
    acc = G.Accel(scene)
    full = G.Film(w, h)
    G.capture_subset(0, 1, acc, full)
    want = np.full((h * w, 4), 9, np.uint8)
    for k in ks:
        want[k::n] = full.pixels().reshape(-1, 4)[k::n]
    for org in (0, 2, 3):
        G.set_streaming(acc, org)
        buf = np.full((h, w, 4), 9, np.uint8)
        G.capture_subsets(ks, n, acc, G.Film.new_with_output(w, h, buf))
        assert np.array_equal(buf.reshape(-1, 4), want), org


def test_the_default_organisation_is_measured_once_per_kind_and_changes_no_byte():
    """lg_accel_set_streaming(1), the default: the first launch of a kind takes the fitted rule's choice (a program that renders one frame
    pays nothing), the SECOND renders with every organisation that can take it and keeps the fastest for the process (launch.cpp + tune.cpp,
    tuned_choice; LASGUN_AUTOTUNE=2: already the first); later launches -- of another accel of the same scene too -- only enqueue.  The film
    is the oracle's whichever organisation runs, and a forced organisation is reported as such."""
    w, h = 200, 136
    o = oracle()
    build = lambda api: S.kitchen_sink_scene(api, "perspective")
    want = o.render(build(o), (w, h)).pixels()
    names = {0: "megakernel", 2: "wavefront", 3: "queue"}
    picked = []
    for _ in range(3):  # three accels of one scene: one kind
        acc = G.Accel(build(G))
        assert G.last_organisation(acc) is None
        for _ in range(2):  # (the kind's first launch in the process: by the rule; from its second: as measured)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            assert np.array_equal(film.pixels(), want)
            assert G.last_organisation(acc).split(",")[0] in names.values()  # ("megakernel, bottom-up": the direction its tiles are claimed in is measured with it)
        picked.append(G.last_organisation(acc))
    assert len(set(picked)) == 1, picked  # remembered, not measured again with another outcome
    for code, name in names.items():
        G.set_streaming(acc, code)
        for order in (0, 1, 2, None):  # lg_accel_set_tile_order: the tiles claimed from the film's top, from its bottom, from its middle outwards (the default of a forced organisation) -- the same film
            G.set_tile_order(acc, order)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            assert np.array_equal(film.pixels(), want), (name, order)
            assert G.last_organisation(acc) == name + ("" if name == "wavefront" else {0: "", 1: ", bottom-up", 2: ", middle-out", None: ", middle-out"}[order])
    with pytest.raises(la.LasgunError):
        G.set_tile_order(acc, 3)


def test_the_table_of_measured_choices_can_be_exported_pinned_and_cleared():
    """lg_tune_export / lg_tune_import / lg_tune_clear (round 6): a measured kind shows up in the export; after a clear nothing is known;
    an imported entry PINS the kind -- the next launch runs what the entry says, without a measurement, and renders the same bytes; an entry
    that holds no organisation is refused."""
    w, h = 192, 144
    o = oracle()
    build = lambda api: S.spheres_scene(api, 300, seed=11)
    want = o.render(build(o), (w, h)).pixels()
    G.tune_clear()
    assert G.tune_export() == []
    acc = G.Accel(build(G))
    for _ in range(3):  # (measured at the kind's first or second call, LASGUN_AUTOTUNE=2 or 1)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        assert np.array_equal(film.pixels(), want)
    table = G.tune_export()
    assert len(table) >= 1 and all(len(k) == 12 and 0 <= c < 256 for k, c in table)
    measured = G.last_organisation(acc)
    # pin every known kind to another organisation (the megakernel top-down = 0, or level by level = 1 if the megakernel was what ran)
    other = 1 if measured.startswith("megakernel") else 0
    G.tune_clear()
    assert G.tune_export() == []
    G.tune_import([(k, other) for k, _ in table])
    assert sorted(G.tune_export()) == sorted((k, other) for k, _ in table)
    film = G.Film(w, h)
    G.capture_subset(0, 1, acc, film)
    assert np.array_equal(film.pixels(), want)
    assert G.last_organisation(acc).split(",")[0] == ("wavefront" if other == 1 else "megakernel"), (measured, G.last_organisation(acc))
    with pytest.raises(la.LasgunError):
        G.tune_import([(table[0][0], 7)])  # (organisations are 0 .. 2)
    G.tune_clear()


TUNE_FAIL_CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import lasgun_amd as la
from oracle_lib import oracle
G, S = la.api, la.scenes
w, h = 160, 120
o = oracle()
want = o.render(S.cornell_scene(o, "glass"), (w, h)).pixels()
acc = G.Accel(S.cornell_scene(G, "glass"))
for _ in range(3):
    film = G.Film(w, h)
    G.capture_subset(0, 1, acc, film)   # the race runs inside one of these calls; the injected failures must not reach the caller
    assert np.array_equal(film.pixels(), want)
ran = G.last_organisation(acc)
assert ran is not None and not ran.startswith(%r), ran
assert len(G.tune_export()) >= 1
print("tune ok", ran)
"""


@pytest.mark.parametrize("fail_org, name", [(1, "wavefront"), (0, "megakernel")])
def test_a_candidate_that_cannot_run_drops_out_of_the_measurement(fail_org, name):
    """an organisation whose buffers do not fit (hipMalloc of the megakernel's accumulator, of the level-by-level state) used to fail the
    caller's render from inside the measurement, and every later launch of the kind again.  LASGUN_TUNE_FAIL makes every candidate of one
    organisation throw in the race (a test hook, launch.cpp): the render succeeds with the bytes of the oracle, a choice is remembered, and it is
    not the organisation that failed.  In a child process (the switch is read once; the table is the process's)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LASGUN_TUNE_FAIL=str(fail_org), LASGUN_AUTOTUNE="2")
    p = subprocess.run([sys.executable, "-c", TUNE_FAIL_CHILD % (root, os.path.join(root, "tests"), name)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "tune ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


TUNE_FILE_CHILD = r"""
import sys
sys.path.insert(0, %r)
import lasgun_amd as la
G, S = la.api, la.scenes
acc = G.Accel(S.cornell_scene(G, "glass"))
for _ in range(int(sys.argv[1])):
    G.capture_subset(0, 1, acc, G.Film(200, 150))
print("RAN", G.last_organisation(acc), len(G.tune_export()))
"""


def test_a_one_frame_program_runs_on_what_an_earlier_run_measured(tmp_path):
    """LASGUN_TUNE_FILE: a first process measures (LASGUN_AUTOTUNE=2) and leaves its table in the file; a second process in the default mode renders
    ONE frame -- it would take the fitted rule's choice and never measure -- and finds the kind in the file: same organisation, no race."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "tune.txt")

    def child(frames, autotune):
        env = dict(os.environ, LASGUN_TUNE_FILE=path, LASGUN_AUTOTUNE=autotune)
        p = subprocess.run([sys.executable, "-c", TUNE_FILE_CHILD % root, str(frames)], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        _, org, n = p.stdout.strip().splitlines()[-1].split(" ", 2)[0], p.stdout.strip().splitlines()[-1].split(" ", 1)[1].rsplit(" ", 1)[0], int(p.stdout.strip().rsplit(" ", 1)[1])
        return org, n
    measured, n1 = child(2, "2")
    assert n1 >= 1 and os.path.getsize(path) > 0
    again, n2 = child(1, "1")
    assert again == measured and n2 == n1, (measured, again)


def test_capture_rebuilds_and_render_matches():
    w, h = 64, 48
    scene = S.simple_scene(G, 1)
    a = G.render(scene, (w, h)).pixels()
    film = G.Film(w, h)
    G.capture(scene, film)
    assert np.array_equal(a, film.pixels())
    assert np.all(a[..., 3] == 255)


def test_row_tiles_on_device_assemble_to_the_full_film():
    import torch
    w, h = 200, 123
    acc = G.Accel(S.spheres_scene(G, 200, seed=3))
    full = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
    G.capture_rows_device(acc, w, h, 0, h, full.data_ptr(), row0=0)
    G.synchronize(acc)
    from lasgun_amd.distributed import row_tile
    parts = []
    for r in range(3):
        y0, y1 = row_tile(r, 3, h)
        t = torch.zeros((y1 - y0, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        G.capture_rows_device(acc, w, h, y0, y1, t.data_ptr())
        G.synchronize(acc)
        parts.append(t)
    assert torch.equal(torch.cat(parts, 0), full)
    film = G.Film(w, h)
    G.capture_subset(0, 1, acc, film)
    assert np.array_equal(full.cpu().numpy(), film.pixels())


def test_cpp_host_example_matches_oracle(tmp_path):
    """examples/cornell.cpp (src/examples/cornell.rs through include/lasgun.hpp) == oracle, byte for byte."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    out = str(tmp_path / "cornell.rgba")
    subprocess.check_call([os.path.join(root, "examples", "cornell"), out, "96"])
    got = np.fromfile(out, dtype=np.uint8).reshape(96, 96, 4)
    o = oracle()
    want = o.render(S.cornell_scene(o, "glass", supersampling=2), (96, 96)).pixels()
    assert np.array_equal(got, want)


def test_interleaved_blocks_on_device_match_the_full_film():
    import torch
    from lasgun_amd.distributed import interleaved_rows
    w, h, b, n = 120, 96, 8, 3
    acc = G.Accel(S.cornell_scene(G, "glass"))
    full = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
    G.capture_rows_device(acc, w, h, 0, h, full.data_ptr(), row0=0)
    G.synchronize(acc)
    for r in range(n):
        t = torch.zeros((h // n, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        G.capture_interleaved_device(acc, w, h, b, n, r, t.data_ptr())
        G.synchronize(acc)
        assert torch.equal(t, full[interleaved_rows(r, n, h, b)])
    with pytest.raises(la.LasgunError):
        G.capture_interleaved_device(acc, w, h, 7, n, 0, full.data_ptr())


def test_errors_instead_of_panics():
    with pytest.raises(la.LasgunError):
        G.Accel(G.Scene.new())  # empty root aggregate
    acc = G.Accel(S.readme_scene(G))
    with pytest.raises(la.LasgunError):
        G.capture_subset(0, 0, acc, G.Film(8, 8))


# ---- BASELINE.json's full size: size-independent properties + sampled oracle pixels ---------
def test_headline_4096_properties():
    import torch
    w = h = 4096
    scene = S.spheres_scene(G)
    acc = G.Accel(scene)
    full = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
    G.capture_rows_device(acc, w, h, 0, h, full.data_ptr(), row0=0)
    G.synchronize(acc)
    # (1) idempotence / determinism: a second render is byte-identical
    again = torch.zeros_like(full)
    torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
    G.capture_rows_device(acc, w, h, 0, h, again.data_ptr(), row0=0)
    G.synchronize(acc)
    assert torch.equal(full, again)
    # (2) partition invariance: 8 row tiles (the multi-GPU sharding) assemble to the same bytes
    from lasgun_amd.distributed import row_tile
    for r in range(8):
        y0, y1 = row_tile(r, 8, h)
        t = torch.zeros((y1 - y0, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        G.capture_rows_device(acc, w, h, y0, y1, t.data_ptr())
        G.synchronize(acc)
        assert torch.equal(t, full[y0:y1])
    host = full.cpu().numpy()
    assert np.all(host[..., 3] == 255)
    # (3) a strided sample of 16384 pixels against the oracle's capture_subset(k, n) on the same film size
    o = oracle()
    oacc = o.Accel(S.spheres_scene(o))
    n, k = 1024, 77
    ofilm = o.Film(w, h)
    o.capture_subset_mt(k, n, oacc, ofilm, 8)
    idx = np.arange(k, w * h, n)
    assert np.array_equal(host.reshape(-1, 4)[idx], ofilm.pixels().reshape(-1, 4)[idx])
    # (4) ray accounting: every primary ray is counted, one shadow ray per hit and light
    st = G.capture_stats(acc, w, h)
    assert st["primary_rays"] == w * h and st["shadow_rays"] == st["hits"] and st["secondary_rays"] == 0


def test_headline_4096_full_frame_vs_oracle():
    """The headline claim literally: the whole 4096x4096 film of config 3, every byte, against the CPU
    oracle (all host cores, capped at 64 threads), in every kernel organisation."""
    import torch
    w = h = 4096
    o = oracle()
    oacc = o.Accel(S.spheres_scene(o))
    ofilm = o.Film(w, h)
    nthreads = max(1, min(64, len(os.sched_getaffinity(0))))
    o.capture_subset_mt(0, 1, oacc, ofilm, nthreads)
    ref = ofilm.pixels()
    acc = G.Accel(S.spheres_scene(G))
    film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
    for streaming, lds, fast, packet in ((1, True, False, False), (1, False, False, False), (0, True, False, False), (1, True, True, False),
                                         (3, True, False, False), (3, False, False, False)):
        G.set_streaming(acc, streaming); G.set_lds_scene(acc, lds); G.set_mode(acc, fast)
        film.zero_()
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        G.capture_rows_device(acc, w, h, 0, h, film.data_ptr(), row0=0)
        G.synchronize(acc)
        diff = int((film.cpu().numpy() != ref).sum())
        assert diff == 0, (streaming, lds, fast, packet, diff)


@pytest.mark.parametrize("devices, w, h, threads", [([0, 0], 160, 256, 0), ([0, 0, 0], 131, 200, 0), ([0, 0, 0, 0], 96, 512, 2), ([], 64, 128, 0)])
def test_host_film_capture_split_over_devices(devices, w, h, threads):
    """lg_capture split over the devices of lg_set_devices (one host thread each; an index may repeat, which is
    how a 1-GPU box exercises the split): same film as the single-device capture, for interleaved 64-row blocks
    (height % (64 n) == 0), contiguous row tiles otherwise, and with `scene.threads` capping the device count."""
    def build():
        sc = S.kitchen_sink_scene(G)
        sc.set_threads(threads)
        return sc
    try:
        G.set_devices([0])
        one = G.Film(w, h)
        G.capture(build(), one)
        G.set_devices(devices)
        many = G.Film.new_with_output(w, h, np.full((h, w, 4), 9, np.uint8))
        G.capture(build(), many)
        assert np.array_equal(one.pixels(), many.pixels())
        assert np.array_equal(G.render(build(), (w, h)).pixels(), one.pixels())
    finally:
        G.set_devices([0])


@pytest.mark.parametrize("gen", ["scene", "scene2"])
def test_adversarial_scenes_match_the_oracle(gen):
    """tools/fast_adversarial.py's generators (giant spheres, needle boxes, degenerate meshes, nested anisotropic groups,
    non-unit rotation axes, distant / orthographic cameras) through the reference traversal, every kernel organisation,
    against the CPU oracle: the reference's behaviour on ill-conditioned input is reproduced too."""
    o = oracle()
    build = S.adversarial_scene if gen == "scene" else S.adversarial_mesh_scene
    ns_g, ns_o = {gen: lambda seed: build(G, seed)}, {gen: lambda seed: build(o, seed)}
    w, h = 96, 72
    for seed in (75, 375, 381, 589, 591, 11, 12, 13, 14, 15):
        oacc = o.Accel(ns_o[gen](seed))
        ofilm = o.Film(w, h)
        o.capture_subset_mt(0, 1, oacc, ofilm, 16)
        o.set_trig_mode(1)
        try:
            orad = o.capture_radiance(oacc, w, h, nthreads=16)
        finally:
            o.set_trig_mode(0)
        acc = G.Accel(ns_g[gen](seed))
        for streaming, packet in ((0, False), (2, False), (3, False)):
            G.set_streaming(acc, streaming)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            assert np.array_equal(film.pixels(), ofilm.pixels()), (seed, streaming, packet)
            assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), (seed, streaming, packet)


def test_hooks_reject_bad_arguments():
    acc = G.Accel(S.readme_scene(G))
    with pytest.raises(la.LasgunError):
        G.trace_pixel(acc, 64, 64, 64, 0)          # pixel outside the film
    with pytest.raises(la.LasgunError):
        G.set_devices([0, 99])                     # no such device
    G.set_devices([0])
    r = G.trace_pixel(acc, 64, 64, 32, 32)         # the sphere in the middle of the README scene
    assert r["ref"] == 0 and r["accel"] == 0 and 0.0 < r["t"] < 1e3 and len(r["shadow"]) == 1
    log = G.trace_pixel_log(acc, 64, 64, 32, 32, False, 1)
    assert log and log[-1][0] == 9.0 and any(3.0 <= e[0] < 4.0 for e in log)


def test_library_first_then_torch_share_one_hip_runtime():
    """Importing lasgun_amd (and rendering) BEFORE torch must leave torch able to use the GPU: the package
    preloads the HIP runtime bundled with the torch wheel so the process never holds two runtimes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import lasgun_amd as la\n"
            "G = la.api; f = G.Film(64, 64); G.capture(la.scenes.readme_scene(G), f)\n"
            "import torch\n"
            "t = torch.zeros((64, 64, 4), dtype=torch.uint8, device='cuda'); torch.cuda.synchronize()\n"
            "acc = G.Accel(la.scenes.readme_scene(G)); G.capture_rows_device(acc, 64, 64, 0, 64, t.data_ptr(), row0=0); G.synchronize(acc)\n"
            "assert (t.cpu().numpy() == f.pixels()).all(); print('shared runtime ok')\n") % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "shared runtime ok" in p.stdout, p.stderr[-2000:]


def test_bench_multi_gpu_path_over_rccl_world1():
    """bench.py's N>1 code path (RCCL process group, interleaved tile, async gather overlapped with the
    next frame, all_reduce of counters) at world size 1: the gathered film must equal the plain film."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--size", "1024", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "verify: gathered 1-rank film == single-GPU film" in p.stderr
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["primary"] == 1024 * 1024
    assert line["gathered_equals_single_gpu"] is True and line["rccl_ranks"] == 1 and line["collective_backend"] == "nccl"


@pytest.mark.parametrize("collective", ["gather", "all-gather"])
def test_bench_two_ranks_end_to_end_under_torchrun(collective):
    """The driver's own launch line for N > 1, with two ranks sharing this box's one GPU and gloo carrying the tiles:
    `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --backend gloo`.  Rank 0's gathered film must equal
    the single-rank film (checked by bench.py itself at every N > 1) and the record must describe a two-rank run.  The ranks are child processes: this
    process starts them and reads their output, nothing is exec'ed from a process that has touched the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = str(29600 + (os.getpid() % 300) + (0 if collective == "gather" else 301))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "512", "--steps", "3", "--warmup", "1"]
    if collective == "all-gather":
        cmd.append("--all-gather")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "verify: gathered 2-rank film == single-GPU film" in p.stderr, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["primary"] == 512 * 512 and "2 rank(s)" in line["config"]["parallelism"]
    # N > 1: no cpu_baseline figure, but the gathered frame is verified: against rank 0's own full render, every byte, and against
    # the oracle on every 64th pixel
    assert "cpu_baseline" not in line and line["bit_exact"] is True and line["bit_exact_check"]["checked_pixels"] == 512 * 512 // 64
    assert line["gathered_equals_single_gpu"] is True and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo"


def test_bench_configs4_workload_two_ranks_under_torchrun():
    """`bench.py --workload configs4` (BASELINE configs[4]: mixed mesh + spheres, 8192^2 by default; 1024^2 here) under the driver's
    N > 1 launch line, two ranks sharing this box's GPU over gloo: the same partition, one gather per frame, the gathered film verified
    against rank 0's own render (every byte) and the oracle sample; `config.workload` names the config."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = str(30300 + (os.getpid() % 300))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "configs4", "--size", "1024", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "verify: gathered 2-rank film == single-GPU film" in p.stderr, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["workload_key"] == "configs4" and line["config"]["workload"].startswith("configs[4]: 1024x1024")
    assert line["config"]["work_per_frame"]["triangles_tested"] > 0 and line["config"]["work_per_frame"]["spheres_tested"] > 0
    assert line["bit_exact"] is True and line["gathered_equals_single_gpu"] is True and line["rccl_ranks"] == 2


def test_exact_ties_inside_fat_leaves_go_to_the_reference_winner():
    """Rays through edges and corners shared by two to six triangles of a mesh with per-corner shading normals: the winner of an
    exact tie in t shows in the picture.  The pruned walk scans a fat leaf run by run instead of in the reference's order and gives
    a tie to the lower original slot; plain walk, pruned walk (megakernel and wavefront pipeline) and fast mode must all paint
    the oracle's picture -- bytes and radiance bits."""
    w = h = 128
    o = oracle()
    oacc = o.Accel(S.tie_mesh_scene(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    o.set_trig_mode(1)
    try:
        orad = o.capture_radiance(oacc, w, h, nthreads=8)
    finally:
        o.set_trig_mode(0)
    assert len(np.unique(ofilm.pixels().reshape(-1, 4), axis=0)) > 1000
    acc = G.Accel(S.tie_mesh_scene(G))
    assert G.accel_info(acc)["triangles"] == 2 * 24 * 24
    for streaming in (0, 2):
        for prune, fast in ((False, False), (True, False), (False, True)):
            G.set_streaming(acc, streaming); G.set_prune(acc, prune); G.set_mode(acc, fast)
            film = G.Film(w, h)
            G.capture_subset(0, 1, acc, film)
            assert np.array_equal(film.pixels(), ofilm.pixels()), (streaming, prune, fast)
            assert np.array_equal(bits(G.capture_radiance(acc, w, h)), bits(orad)), (streaming, prune, fast)
