"""CPU-only checks of the product's host side: the C ABI loads and exports every declared symbol,
and the host HLBVH builder / transforms / OBJ reader agree bit-for-bit with the oracle's."""
import ctypes
import os
import re

import numpy as np
import pytest

import lasgun_amd as la
from golden_cases import CASES
from lasgun_amd import scenes as S
from oracle_lib import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lasgun_hip.h")).read()
    return sorted(set(re.findall(r"\b(lg_[a-z0-9_]+)\s*\(", text)))


def test_abi_exports_every_declared_symbol():
    lib = ctypes.CDLL(la.LIB_PATH)
    names = declared_symbols()
    assert len(names) > 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_a_device():
    if la.api.device_count() > 0:
        pytest.skip("a HIP device is present")
    scene = S.readme_scene(la.api)
    with pytest.raises(la.LasgunError, match="no HIP device"):
        la.Accel(scene)
    with pytest.raises(la.LasgunError, match="no HIP device"):
        la.render(scene, (8, 8))


BUILD_CASES = dict(CASES)
BUILD_CASES["mixed_small"] = (lambda api: S.mixed_scene(api, 300, 40, 40), 0, 0)
BUILD_CASES["mesh_100k"] = (lambda api: S.mesh_scene(api), 0, 0)


@pytest.mark.parametrize("name", list(BUILD_CASES))
def test_host_bvh_build_matches_oracle_bit_for_bit(name):
    builder = BUILD_CASES[name][0]
    f, i, info = la.api.host_build_dump(builder(la.api))
    o = oracle()
    acc = o.Accel(builder(o))
    of, oi = acc.dump()
    assert np.array_equal(i, oi)
    assert np.array_equal(f.view(np.uint64), of.view(np.uint64))
    assert info["max_stack"] <= 64 * 8


@pytest.mark.parametrize("name", list(BUILD_CASES))
def test_wide_records_of_the_fast_trees_cover_their_binary_trees(name):
    """Fast mode's wide node records (DNode4): every leaf of the binary fast tree is reached exactly once, every child box
    (f32, rounded outward) contains its node's f64 box, and the stack the flattening reserves covers the deepest walk."""
    r = la.api.host_check_wide_records(BUILD_CASES[name][0](la.api))
    assert r["violations"] == 0, r
    assert r["children"] >= 2 * r["records"] and r["children"] <= 4 * r["records"]
    assert r["deepest_stack"] <= r["reserved_stack"], r
    if name == "mesh_100k":
        assert r["leaves"] > 20000 and r["children"] > 2.5 * r["records"], r


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_incoherent_fat_leaves_build(seed):
    """A soup whose centroids form a geometric progression with random orientations (round 3's builder threw "too many culling
    records in one leaf" on it): the culling records are an optimisation and must never fail the build; the reference tree
    still equals the oracle's."""
    f, i, info = la.api.host_build_dump(S.progression_soup_scene(la.api, seed))
    o = oracle()
    of, oi = o.Accel(S.progression_soup_scene(o, seed)).dump()
    assert np.array_equal(i, oi) and np.array_equal(f.view(np.uint64), of.view(np.uint64))


def test_leaf_records_do_not_depend_on_the_thread_count(monkeypatch):
    """The fat leaves' runs, culling records and strips are made by several host threads, one leaf per task (host.cpp, build_chunks), and
    laid out in leaf order: the tables -- hashed as they would be uploaded -- are what one thread makes."""
    for build in (lambda: S.mesh_scene(la.api), lambda: S.mixed_scene(la.api, 300, 40, 40), lambda: S.adversarial_prune_scene(la.api, 5)):
        seen = set()
        for threads in ("1", "3", "8"):
            monkeypatch.setenv("LASGUN_HOST_THREADS", threads)
            r = la.api.host_check_strips(build())
            assert r["violations"] == 0
            seen.add((r["records_hash"], r["strips_hash"], r["runs"], r["entries"]))
        assert len(seen) == 1, seen


@pytest.mark.parametrize("name", ["mesh_100k", "mixed_small", "tie_mesh", "exotic", "soup0", "soup3", "slabs"])
def test_triangle_strips_cover_their_leaves(name):
    """The strips the pruned walk's leaf loop streams (DStrip): every triangle of every mesh leaf with culling records is exactly one
    strip triangle, made of its own three vertices; on a tessellated surface a strip entry serves most of a triangle."""
    builders = {"mesh_100k": lambda: S.mesh_scene(la.api), "mixed_small": lambda: S.mixed_scene(la.api, 300, 40, 40), "tie_mesh": lambda: S.tie_mesh_scene(la.api),
                "exotic": lambda: S.exotic_obj_scene(la.api), "soup0": lambda: S.progression_soup_scene(la.api, 0), "soup3": lambda: S.progression_soup_scene(la.api, 3),
                "slabs": lambda: S.adversarial_prune_scene(la.api, 5)}
    r = la.api.host_check_strips(builders[name]())
    assert r["violations"] == 0, r
    assert r["entries"] >= r["triangles"] and r["entries"] <= 3 * r["triangles"], r
    if name == "mesh_100k":
        assert r["triangles"] >= 100352 and r["entries"] < 1.5 * r["triangles"], r  # (three entries per triangle without strips)


def test_transform_concat_matches_oracle():
    o = oracle()
    outs = []
    for api in (la.api, o):
        a = api.Aggregate.new()
        a.scale(2.0, 0.5, 3.0).rotate_z(33.0).translate([1.0, -2.0, 0.25]).rotate_x(-71.5).rotate_y(12.0)
        a.rotate(40.0, [0.6, 0.0, 0.8]).scale(1.5, 1.5, 1.5)
        outs.append(a.transform())
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
    m, minv = outs[0]
    assert np.allclose(m.T @ minv.T, np.eye(4), atol=1e-12)  # column-major storage


def test_obj_reader_forms_and_errors():
    text = "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\n" \
           "f 1/1/1 2/2/1 3/3/1\nf -4//1 -3//1 -1//1 2//1\n# comment\ng grp\ns off\n"
    for api in (la.api, oracle()):
        sc = api.Scene.new()
        assert sc.parse_obj(text) == 0
        assert sc.parse_obj(S.PLANE_OBJ) == 1
        with pytest.raises(la.ObjError):
            sc.parse_obj("v 0 0\n")
        with pytest.raises(la.ObjError):
            sc.parse_obj("v 0 0 0\nf 1 2 3\n")
        with pytest.raises(la.ObjError):
            sc.parse_obj("bogus 1 2 3\n")
    # a mesh with normals whose face lacks vn indices panics in the reference (triangle.rs:60)
    sc = la.api.Scene.new()
    m = sc.parse_obj("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1 2 3\n")
    sc.root.add_obj(m)
    with pytest.raises(la.LasgunError):
        la.api.host_build_dump(sc)


def test_empty_aggregate_is_an_error():
    with pytest.raises(la.LasgunError):
        la.api.host_build_dump(la.api.Scene.new())


def test_material_pod_layout():
    m = la.Material.plastic([0.1, 0.2, 0.3], [0.4, 0.5, 0.6], 0.25)
    assert m.c.kind == 1 and list(m.c.p)[:7] == [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.25]
    d = la.Material.default()
    assert d.c.kind == 0 and list(d.c.p)[:4] == [0.5, 0.5, 0.5, 0.0]
    assert la.Material.matte([1, 1, 1], 120.0).c.p[3] == 90.0  # matte.rs:15 clamps sigma
    o = oracle()
    for name, args in (("glass", ([1, .7, 1], [.7, 1, .7], 1.25)), ("metal", ([.2, .9, 1.1], [3.9, 2.4, 2.2], .1, .2)),
                       ("mirror", ([.5, .5, .5],))):
        a = getattr(la.Material, name)(*args).c
        b = getattr(o.Material, name)(*args).c
        assert a.kind == b.kind and list(a.p) == list(b.p)
