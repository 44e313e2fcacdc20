"""The reference's 17 inline known-answer tests, as data.

Each entry restates the inputs and the exact-equality expectations of one `#[test]` of the
reference (file:line under /root/reference).  `run_kats(api)` executes them through an
`orc_`/`lg_` C ABI's kat hooks; tests/test_oracle_kats.py runs them on the CPU oracle and
tests/test_gpu_parity.py on the device intersectors.
"""
from lasgun_amd.scenes import PLANE_OBJ

UNIT_SPHERE = [0.0, 0.0, 0.0, 1.0]
CUBE1 = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
BOX11 = [-1.1, -1.1, -1.0, 1.1, 1.1, 1.0]


def _r(v):
    return tuple(float(round(c)) for c in v)


# (name, ref, kind, params, obj, origin, d, checks) ; checks: dict of t / ng / ns / ng_round / hit
KATS = [
    ("sphere.straight_on_intersection", "src/shape/sphere.rs:137-146", 0, UNIT_SPHERE, None, (0, 0, 2), (0, 0, -1),
     {"t": 1.0, "ng": (0.0, 0.0, 1.0)}),
    ("sphere.inside_intersection", "src/shape/sphere.rs:149-157", 0, UNIT_SPHERE, None, (0, 0, 0), (0, 0, 1),
     {"t": 1.0, "ng": (0.0, 0.0, -1.0)}),
    ("sphere.behind_intersection", "src/shape/sphere.rs:160-173", 0, UNIT_SPHERE, None, (0, 0, -2), (0, 0, 1),
     {"t": 1.0, "ng_round": (0.0, 0.0, -1.0)}),
    ("cuboid.straight_on_intersection", "src/shape/cuboid.rs:137-146", 1, CUBE1, None, (0, 0, -2), (0, 0, 1),
     {"t": 1.0, "ng": (0.0, 0.0, -1.0)}),
    ("cuboid.edge_intersection", "src/shape/cuboid.rs:148-157", 1, BOX11, None, (0, 0, -2), (1, 0, 1),
     {"t": 1.0, "ng": (0.0, 0.0, -1.0)}),
    ("cuboid.corner_intersection", "src/shape/cuboid.rs:159-168", 1, BOX11, None, (0, 0, -2), (1, 1, 1),
     {"t": 1.0, "ng": (0.0, 0.0, -1.0)}),
    ("cuboid.inside_intersection", "src/shape/cuboid.rs:170-179", 1, CUBE1, None, (0, 0, 0), (0, 0, 1),
     {"t": 1.0}),
    ("cuboid.inside_behind_intersection", "src/shape/cuboid.rs:182-191", 1, CUBE1, None, (0, 0, 0), (0, -1, 0),
     {"t": 1.0}),
    ("cuboid.inside_intersection_offset", "src/shape/cuboid.rs:194-202", 1, CUBE1, None, (0.5, 0.5, 0.5), (1, 0, 1),
     {"hit": True}),
    ("cuboid.behind_intersection", "src/shape/cuboid.rs:205-213", 1, CUBE1, None, (0, 0, 2), (0, 0, -1),
     {"t": 1.0, "ng": (0.0, 0.0, 1.0)}),
    ("cuboid.top_intersection", "src/shape/cuboid.rs:216-224", 1, CUBE1, None, (0, 2, 0), (0, -1, 0),
     {"t": 1.0, "ns": (0.0, 1.0, 0.0)}),
    ("cuboid.bottom_intersection", "src/shape/cuboid.rs:227-235", 1, CUBE1, None, (0, -2, 0), (0, 1, 0),
     {"t": 1.0, "ns": (0.0, -1.0, 0.0)}),
    ("cuboid.top_angled_intersection", "src/shape/cuboid.rs:238-246", 1, CUBE1, None, (0, 2, 2), (0, -0.5, -1),
     {"t": 2.0, "ng": (0.0, 1.0, 0.0)}),
    ("triangle.plane_intersection", "src/shape/triangle.rs:411-431", 2, None, PLANE_OBJ, (0, 1, 0), (0, -1, 0),
     {"t": 1.0, "ng": (0.0, 1.0, 0.0)}),
    ("triangle.plane_intersection_with_normals_and_texture", "src/shape/triangle.rs:434-454", 2, None, PLANE_OBJ,
     (0, 1, 0), (0, -1, 0), {"t": 1.0, "ng": (0.0, 1.0, 0.0)}),
]

# src/interaction/surface.rs:194-200
SURFACE_KAT = {"origin": (0, 0, 1), "d": (0, 0, -1), "t": 1.0, "dpdu": (1, 0, 0), "dpdv": (0, 1, 0), "ng": (0.0, 0.0, 1.0)}


def run_kat(api, kat):
    name, ref, kind, params, obj, origin, d, checks = kat
    got = api.kat_intersect(kind, params=params, obj_text=obj, origin=origin, d=d)
    assert got["hit"], name  # every reference test first asserts `is_some()`
    for key, want in checks.items():
        if key == "t":
            assert got["t"] == want, (name, ref, got)
        elif key == "ng":
            assert got["ng"] == want, (name, ref, got)  # == on floats: -0.0 == 0.0, as in Rust's assert_eq!
        elif key == "ns":
            assert got["ns"] == want, (name, ref, got)
        elif key == "ng_round":
            assert _r(got["ng"]) == want, (name, ref, got)
    return got


def run_surface_kat(api):
    k = SURFACE_KAT
    ng = api.kat_surface_interaction(k["origin"], k["d"], k["t"], k["dpdu"], k["dpdv"])
    assert ng == k["ng"], ng
    return ng
