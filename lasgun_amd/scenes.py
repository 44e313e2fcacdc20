"""Synthetic scenes of BASELINE.json's configs, built through the reference-shaped API.

Every builder takes an `api` (a bound `lasgun_amd._capi.Api`) and issues exactly the calls a
user of the reference would make (`Scene::new`, `set_perspective_camera`, `Aggregate::add_sphere`
...), so the same function drives the product and -- in tests only -- the CPU oracle.
Templates: /root/reference/README.md:57-94, src/examples/simple.rs:7-41,
src/examples/cornell.rs:7-69.  The reference's .obj meshes are Git-LFS stubs, so the plane is
the 4-vertex/2-face OBJ of src/shape/triangle.rs:412-421 and bigger meshes are generated here.
"""
import math

# OBJ text of the reference's own inline plane fixture (src/shape/triangle.rs:412-421)
PLANE_OBJ = """o plane
v -1 0 -1
v 1 0 -1
v 1 0 1
v -1 0 1

f 1 2 3
f 1 3 4
"""

MASK64 = (1 << 64) - 1


class SplitMix64:
    """Deterministic PRNG for scene generation (never used at render time)."""

    def __init__(self, seed):
        self.s = seed & MASK64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def next_f64(self):
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def uniform(self, lo, hi):
        return lo + (hi - lo) * self.next_f64()


def readme_scene(api):
    """Config 1a: README.md:57-94 -- one sphere, one point light."""
    scene = api.Scene.new()
    camera = scene.set_perspective_camera(45.0)
    camera.look_at([0, 0, 0], [0, 0, 1], [0, 1, 0])
    scene.add_point_light([100, 200, 400], [0.8, 0.8, 0.8], [1, 0, 0])
    mat = api.Material.plastic([0.7, 1.0, 0.7], [0.5, 0.7, 0.5], 0.25)
    node = api.Aggregate.new()
    node.add_sphere([0, 0, 100], 50, mat)
    scene.set_root(node)
    return scene


def simple_scene(api, supersampling=2, reflect=False):
    """Config 1b: src/examples/simple.rs:7-41 (simplereflect.rs when reflect=True) minus the LFS mesh."""
    scene = api.Scene.new()
    scene.set_ambient_light([0.2, 0.2, 0.2])
    if reflect:
        scene.set_radial_background([0.93, 0.87, 0.36], [0.94, 0.6, 0.1], 0.5)
        scene.set_max_recursion_depth(4)
    else:
        scene.set_radial_background([0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5)
    camera = scene.set_perspective_camera(45.0)
    camera.look_at([25.0, 0.0, 800.0], [25.0, 0.0, 0.0], [0.0, 1.0, 0.0])
    camera.set_supersampling(supersampling)
    M = api.Material
    if reflect:
        mat0 = M.glass([0.7, 1.0, 0.7], [0.5, 0.7, 0.5], 1.333)
        mat1 = M.mirror([0.5, 0.5, 0.5])
        mat2 = M.glass([1.0, 0.6, 0.1], [0.7, 0.7, 1.0], 1.75)
        mat3 = M.glass([0.7, 0.6, 1.0], [0.5, 0.4, 0.8], 1.5)
    else:
        mat0 = M.plastic([0.7, 1.0, 0.7], [0.5, 0.7, 0.5], 0.25)
        mat1 = M.plastic([0.5, 0.5, 0.5], [0.5, 0.7, 0.5], 0.25)
        mat2 = M.plastic([1.0, 0.6, 0.1], [0.5, 0.7, 0.5], 0.25)
        mat3 = M.plastic([0.7, 0.6, 1.0], [0.5, 0.4, 0.8], 0.25)
    scene.add_point_light([-100.0, 150.0, 400.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0])
    scene.add_point_light([400.0, 100.0, 150.0], [0.7, 0.0, 0.7], [1.0, 0.0, 0.0])
    root = scene.root
    root.add_sphere([0.0, 0.0, -400.0], 100.0, mat0)
    root.add_sphere([200.0, 50.0, -100.0], 150.0, mat0)
    root.add_sphere([0.0, -1200.0, -500.0], 1000.0, mat1)
    root.add_sphere([-100.0, 25.0, -300.0], 50.0, mat2)
    root.add_sphere([0.0, 100.0, -250.0], 25.0, mat0)
    root.add_cube([-200.0, -125.0, 0.0], 100.0, mat3)
    return scene


def _cornell_shell(api, scene, supersampling):
    """Camera, light and the five plane walls of src/examples/cornell.rs:7-62."""
    scene.set_ambient_light([0.2, 0.2, 0.2])
    camera = scene.set_perspective_camera(60.0)
    camera.look_at([0.0, 0.0, 5.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])
    camera.set_supersampling(supersampling)
    M = api.Material
    white = M.plastic([0.9, 0.9, 0.9], [0.5, 0.7, 0.5], 0.25)
    r = M.plastic([1.0, 0.0, 0.0], [0.5, 0.7, 0.5], 0.25)
    g = M.plastic([0.0, 1.0, 0.0], [0.5, 0.7, 0.5], 0.25)
    plane = scene.parse_obj(PLANE_OBJ)
    scene.add_point_light([0.0, 1.75, 0.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0])
    A = api.Aggregate

    floor = A.new(); floor.scale(2.0, 1.0, 2.0); floor.translate([0.0, -2.0, 0.0]); floor.add_obj_of(plane, white)
    scene.root.add_group(floor)
    ceiling = A.new(); ceiling.scale(2.0, 1.0, 2.0); ceiling.translate([0.0, 2.0, 0.0]); ceiling.add_obj_of(plane, white)
    scene.root.add_group(ceiling)
    left = A.new(); left.scale(2.0, 1.0, 2.0); left.rotate_z(90.0); left.translate([-2.0, 0.0, 0.0]); left.add_obj_of(plane, r)
    scene.root.add_group(left)
    right = A.new(); right.scale(2.0, 1.0, 2.0); right.rotate_z(90.0); right.translate([2.0, 0.0, 0.0]); right.add_obj_of(plane, g)
    scene.root.add_group(right)
    back = A.new(); back.scale(2.0, 1.0, 2.0); back.rotate_x(90.0); back.translate([0.0, 0.0, -2.0]); back.add_obj_of(plane, white)
    scene.root.add_group(back)
    return white


def cornell_scene(api, variant="plastic", supersampling=0):
    """Config 2: cornell.rs with 1 spp.  variant "plastic" (2P, the primary+shadow metric scene) or
    "glass" (2G, the reference's own materials: recursion 3)."""
    scene = api.Scene.new()
    white = _cornell_shell(api, scene, supersampling)
    if variant == "glass":
        mat = api.Material.glass([1.0, 0.7, 1.0], [0.7, 1.0, 0.7], 1.25)
    elif variant == "plastic":
        mat = white
    else:
        raise ValueError(variant)
    scene.root.add_sphere([1.0, -1.25, 0.0], 1.0, mat)
    scene.root.add_cube([-1.999, -1.999, 0.0], 1.0, mat)
    return scene


PALETTE = [
    [0.9, 0.9, 0.9], [1.0, 0.2, 0.2], [0.2, 1.0, 0.2], [0.2, 0.3, 1.0],
    [1.0, 0.8, 0.1], [0.8, 0.2, 0.9], [0.1, 0.8, 0.8], [0.95, 0.5, 0.1],
]


def spheres_scene(api, nspheres=1024, seed=0x1A560001, supersampling=0):
    """Config 3 (headline): Cornell shell + `nspheres` random plastic spheres, direct children of root.
    Draw order per sphere: cx, cy, cz, r, material index (SplitMix64, doubles = (x>>11)*2^-53)."""
    scene = api.Scene.new()
    _cornell_shell(api, scene, supersampling)
    mats = [api.Material.plastic(kd, [0.5, 0.7, 0.5], 0.25) for kd in PALETTE]
    rng = SplitMix64(seed)
    root = scene.root
    for _ in range(nspheres):
        cx = rng.uniform(-1.8, 1.8)
        cy = rng.uniform(-1.8, 1.8)
        cz = rng.uniform(-1.8, 1.8)
        r = rng.uniform(0.02, 0.06)
        mi = min(int(rng.next_f64() * len(mats)), len(mats) - 1)
        root.add_sphere([cx, cy, cz], r, mats[mi])
    return scene


def torus_obj(nu=224, nv=224, R=0.9, r=0.35, normals=True):
    """OBJ text of a torus: nu*nv quads -> 2*nu*nv triangles (224x224 -> 100,352).
    Coordinates are printed with %.6f and parsed back as f32 by the OBJ reader."""
    lines = ["o torus"]
    for i in range(nu):
        u = 2.0 * math.pi * i / nu
        cu, su = math.cos(u), math.sin(u)
        for j in range(nv):
            v = 2.0 * math.pi * j / nv
            cv, sv = math.cos(v), math.sin(v)
            x = (R + r * cv) * cu
            y = r * sv
            z = (R + r * cv) * su
            lines.append("v %.6f %.6f %.6f" % (x, y, z))
    if normals:
        for i in range(nu):
            u = 2.0 * math.pi * i / nu
            cu, su = math.cos(u), math.sin(u)
            for j in range(nv):
                v = 2.0 * math.pi * j / nv
                cv, sv = math.cos(v), math.sin(v)
                lines.append("vn %.6f %.6f %.6f" % (cv * cu, sv, cv * su))
    def vid(i, j):
        return (i % nu) * nv + (j % nv) + 1
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            if normals:
                lines.append("f %d//%d %d//%d %d//%d" % (a, a, d, d, c, c))
                lines.append("f %d//%d %d//%d %d//%d" % (a, a, c, c, b, b))
            else:
                lines.append("f %d %d %d" % (a, d, c))
                lines.append("f %d %d %d" % (a, c, b))
    return "\n".join(lines) + "\n"


def mesh_scene(api, nu=224, nv=224, material="glass", smoothing=True, supersampling=0):
    """Config 4: Cornell shell + a generated torus mesh in a transformed group + a mirror sphere."""
    scene = api.Scene.new()
    scene.set_mesh_smoothing(smoothing)
    _cornell_shell(api, scene, supersampling)
    M = api.Material
    if material == "glass":
        mat = M.glass([1.0, 0.7, 1.0], [0.7, 1.0, 0.7], 1.25)
    elif material == "metal":
        mat = M.metal([0.2, 0.9, 1.1], [3.9, 2.4, 2.2], 0.1, 0.1)
    elif material == "plastic":
        mat = M.plastic([0.2, 0.3, 1.0], [0.5, 0.7, 0.5], 0.25)
    elif material == "default":
        mat = None
    else:
        raise ValueError(material)
    mesh = scene.parse_obj(torus_obj(nu, nv, normals=True))
    grp = api.Aggregate.new()
    grp.scale(1.2, 1.2, 1.2)
    grp.rotate_x(35.0)
    grp.rotate_y(30.0)
    if mat is None:
        grp.add_obj(mesh)
    else:
        grp.add_obj_of(mesh, mat)
    scene.root.add_group(grp)
    scene.root.add_sphere([1.1, -1.4, 0.6], 0.6, M.mirror([0.5, 0.5, 0.5]))
    return scene


def mixed_scene(api, nspheres=1024, nu=224, nv=224, supersampling=0):
    """Config 5: config 3's spheres + config 4's mesh (plastic) in the shell."""
    scene = spheres_scene(api, nspheres=nspheres, supersampling=supersampling)
    mesh = scene.parse_obj(torus_obj(nu, nv, normals=True))
    grp = api.Aggregate.new()
    grp.scale(1.2, 1.2, 1.2)
    grp.rotate_x(35.0)
    grp.rotate_y(30.0)
    grp.add_obj_of(mesh, api.Material.plastic([0.2, 0.3, 1.0], [0.5, 0.7, 0.5], 0.25))
    scene.root.add_group(grp)
    return scene


def blob_obj(nu, nv, centre, radii, lumps, seed, normals=False, name="blob"):
    """OBJ text of a closed, lumpy UV surface (stand-in for the reference's bunny / skull meshes, which are Git-LFS stubs): a sphere whose
    radius is modulated by `lumps` low-frequency cosine lobes, poles closed by triangle fans; 2 * nu * (nv - 1) triangles.  Coordinates
    are printed with %.6f and parsed back as f32 by the OBJ reader."""
    rng = SplitMix64(seed)
    lobes = [(rng.uniform(0.05, 0.22), 1 + int(rng.uniform(0, 4)), 1 + int(rng.uniform(0, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28)) for _ in range(lumps)]

    def point(u, v):
        rho = 1.0 + sum(a * math.cos(ku * u + pu) * math.cos(kv * v + pv) * math.sin(v) for a, ku, kv, pu, pv in lobes)
        d = (math.sin(v) * math.cos(u), math.cos(v), math.sin(v) * math.sin(u))
        return (centre[0] + radii[0] * rho * d[0], centre[1] + radii[1] * rho * d[1], centre[2] + radii[2] * rho * d[2]), d

    lines = ["o " + name]
    verts, norms = [], []
    p, d = point(0.0, 0.0)
    verts.append(p); norms.append(d)
    for j in range(1, nv):
        for i in range(nu):
            p, d = point(2.0 * math.pi * i / nu, math.pi * j / nv)
            verts.append(p); norms.append(d)
    p, d = point(0.0, math.pi)
    verts.append(p); norms.append(d)
    for x, y, z in verts:
        lines.append("v %.6f %.6f %.6f" % (x, y, z))
    if normals:
        for x, y, z in norms:
            lines.append("vn %.6f %.6f %.6f" % (x, y, z))
    ring = lambda j, i: 2 + (j - 1) * nu + (i % nu)
    south = len(verts)
    f = (lambda a, b, c: "f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c)) if normals else (lambda a, b, c: "f %d %d %d" % (a, b, c))
    for i in range(nu):
        lines.append(f(1, ring(1, i + 1), ring(1, i)))
    for j in range(1, nv - 1):
        for i in range(nu):
            a, b, c, e = ring(j, i), ring(j, i + 1), ring(j + 1, i + 1), ring(j + 1, i)
            lines.append(f(a, b, c))
            lines.append(f(a, c, e))
    for i in range(nu):
        lines.append(f(south, ring(nv - 1, i), ring(nv - 1, i + 1)))
    return "\n".join(lines) + "\n"


def buckyball_obj():
    """OBJ text of a truncated icosahedron -- 60 vertices, 12 pentagons and 20 hexagons AS POLYGONS, the way a modelling tool writes a
    buckyball: the reference's TriangleIterator takes the FIRST THREE vertices of every polygon (src/shape/triangle.rs:52-53,315-371),
    so the rendered solid is the 32 corner triangles -- a stand-in for the reference's buckyball.obj (a Git-LFS stub), with unit circumradius."""
    phi = (1.0 + math.sqrt(5.0)) / 2.0
    ico = []
    for a in (-1.0, 1.0):
        for b in (-phi, phi):
            ico += [(0.0, a, b), (a, b, 0.0), (b, 0.0, a)]
    def near(p):  # an icosahedron vertex's five neighbours
        ds = sorted((sum((x - y) ** 2 for x, y in zip(p, q)), k) for k, q in enumerate(ico) if q != p)
        return [k for _, k in ds[:5]]
    verts, index = [], {}
    for k, p in enumerate(ico):  # every edge cut at a third from each end
        for m in near(p):
            q = ico[m]
            index[(k, m)] = len(verts)
            verts.append(tuple(p[i] + (q[i] - p[i]) / 3.0 for i in range(3)))
    scale = 1.0 / math.sqrt(sum(c * c for c in verts[0]))
    lines = ["o buckyball"] + ["v %.6f %.6f %.6f" % tuple(c * scale for c in v) for v in verts]
    def ordered(face_pts, centre):  # vertices of a planar convex face in order around its centre, counter-clockwise seen from outside
        n = centre
        ref = tuple(verts[face_pts[0]][i] - n[i] * sum(verts[face_pts[0]][j] * n[j] for j in range(3)) / sum(c * c for c in n) for i in range(3))
        def angle(idx):
            v = verts[idx]
            w = tuple(v[i] - n[i] * sum(v[j] * n[j] for j in range(3)) / sum(c * c for c in n) for i in range(3))
            cr = (ref[1] * w[2] - ref[2] * w[1], ref[2] * w[0] - ref[0] * w[2], ref[0] * w[1] - ref[1] * w[0])
            return math.atan2(sum(cr[i] * n[i] for i in range(3)) / math.sqrt(sum(c * c for c in n)), sum(ref[i] * w[i] for i in range(3)))
        return sorted(face_pts, key=angle)
    for k, p in enumerate(ico):  # pentagons: around every icosahedron vertex
        lines.append("f " + " ".join(str(i + 1) for i in ordered([index[(k, m)] for m in near(p)], p)))
    seen = set()
    for a in range(12):  # hexagons: one per icosahedron face
        for b in near(ico[a]):
            for c in near(ico[b]):
                if c in near(ico[a]) and len({a, b, c}) == 3 and frozenset((a, b, c)) not in seen:
                    seen.add(frozenset((a, b, c)))
                    pts = [index[(a, b)], index[(b, a)], index[(b, c)], index[(c, b)], index[(c, a)], index[(a, c)]]
                    centre = tuple(ico[a][i] + ico[b][i] + ico[c][i] for i in range(3))
                    lines.append("f " + " ".join(str(i + 1) for i in ordered(pts, centre)))
    return "\n".join(lines) + "\n"


def playground_scene(api, nu=40, nv=24, supersampling=2):
    """src/examples/playground.rs:5-27: one metal mesh added straight to `scene.root` (`add_obj_of`), radial background, one light,
    3x3 supersampling.  The bunny is a Git-LFS stub in the reference: a lumpy closed stand-in mesh (no normals, like the Stanford
    bunny's OBJ) sits where the camera looks."""
    scene = api.Scene.new()
    scene.set_ambient_light([0.1, 0.1, 0.1])
    scene.set_radial_background([0.93, 0.87, 0.36], [0.94, 0.6, 0.1], 0.8)
    camera = scene.set_perspective_camera(60.0)
    camera.look_at([0.0, 1.0, 4.0], [-0.1, 1.0, 3.0], [0.0, 1.0, 0.0])
    camera.set_supersampling(supersampling)
    mat0 = api.Material.metal([0.9, 0.1, 0.9], [0.7, 1.0, 0.7], 0.25, 0.25)
    bunny = scene.parse_obj(blob_obj(nu, nv, (-0.35, 0.95, 0.2), (0.9, 0.75, 0.7), 5, 0xB0771, normals=False, name="bunny"))
    scene.add_point_light([0.0, 2.0, 3.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0])
    scene.root.add_obj_of(bunny, mat0)
    return scene


def spooky_scene(api, nu=36, nv=22, supersampling=2):
    """src/examples/spooky.rs:6-52 at its own 768 x 768: white ambient light, two lights, a mesh under scale / rotate_y / translate inside
    a group inside the rotated root, glass cube and spheres beside a plastic one, a plane scaled by 100, `scene.root.rotate_y` applied to
    the root in place BEFORE the groups are added.  The skull is a Git-LFS stub: a lumpy stand-in with vertex normals."""
    scene = api.Scene.new()
    scene.set_ambient_light([1.0, 1.0, 1.0])
    scene.set_radial_background([0.39, 0.29, 0.29], [0.1, 0.0, 0.0], 1.0)
    camera = scene.set_perspective_camera(50.0)
    camera.look_at([-5.0, 2.0, 6.0], [-3.0, 2.2, 1.0], [0.0, 1.0, 0.0])
    camera.set_supersampling(supersampling)
    skull = scene.parse_obj(blob_obj(nu, nv, (0.0, 2.2, 0.0), (2.2, 2.6, 2.9), 6, 0x5C011, normals=True, name="skull"))
    plane = scene.parse_obj(PLANE_OBJ)
    M, A = api.Material, api.Aggregate
    floor = M.plastic([0.8, 0.7, 0.7], [0.0, 0.0, 0.0], 0.0)
    bone = M.plastic([0.7, 0.7, 0.5], [0.3, 0.3, 0.3], 0.20)
    purple = M.plastic([0.7, 0.6, 1.0], [0.8, 0.8, 0.8], 0.25)
    glass = M.glass([0.7, 0.6, 1.0], [0.8, 0.8, 0.8], 1.333)
    scene.add_point_light([-20.0, 15.0, 0.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0])
    scene.add_point_light([40.0, 10.0, 15.0], [1.0, 0.5, 0.0], [1.0, 0.0, 0.0])
    skull_group = A.new()
    skull_group.scale(0.5, 0.5, 0.5)
    skull_group.rotate_y(-60.0)
    skull_group.translate([4.0, 0.5, -4.0])
    skull_group.add_obj_of(skull, bone)
    item_group = A.new()
    item_group.add_group(skull_group)
    item_group.add_sphere([4.0, 4.0, -11.0], 4.0, purple)
    item_group.add_cube([-2.5, 0.001, -3.0], 1.75, glass)
    item_group.add_sphere([0.0, 2.0, -15.0], 2.0, glass)
    item_group.add_sphere([2.5, 1.0, -2.0], 1.0, glass)
    floor_group = A.new()
    floor_group.scale(100.0, 1.0, 100.0)
    floor_group.add_obj_of(plane, floor)
    scene.root.rotate_y(10.0)
    scene.root.add_group(item_group)
    scene.root.add_group(floor_group)
    return scene


def simplecows_scene(api, supersampling=2):
    """src/examples/simplecows.rs:5-104: the plane scaled by 30, a buckyball (here: the polygon OBJ of `buckyball_obj`, of which the
    reference's reader keeps each face's first three vertices), six arches of two scaled cubes and a scaled sphere each -- groups three
    deep, built with CHAINED transforms (`p1.scale(..).translate(..)`) --, three cows of seven spheres, and `scene.root.rotate_x(23)`
    applied to the root in place AFTER its children were added (:90)."""
    scene = api.Scene.new()
    scene.set_ambient_light([0.2, 0.2, 0.2])
    scene.set_radial_background([0.85, 0.82, 0.6], [0.69, 0.85, 0.73], 0.5)
    camera = scene.set_perspective_camera(50.0)
    camera.look_at([0.0, 2.0, 30.0], [0.0, 2.0, 29.0], [0.0, 1.0, 0.0])
    camera.set_supersampling(supersampling)
    scene.add_point_light([200.0, 202.0, 430.0], [0.8, 0.8, 0.8], [1.0, 0.0, 0.0])
    M, A = api.Material, api.Aggregate
    stone = M.metal([0.0, 0.0, 0.0], [0.7, 0.7, 0.7], 0.5, 0.5)
    grass = M.plastic([0.1, 0.7, 0.1], [0.0, 0.0, 0.0], 0.0)
    hide = M.plastic([0.84, 0.6, 0.53], [0.3, 0.3, 0.3], 0.2)
    planemesh = scene.parse_obj(PLANE_OBJ)
    buckyballmesh = scene.parse_obj(buckyball_obj())
    plane = A.new()
    plane.scale(30.0, 30.0, 30.0)
    plane.add_obj_of(planemesh, grass)
    scene.root.add_group(plane)
    buckyball = A.new()
    buckyball.scale(1.5, 1.5, 1.5)
    buckyball.add_obj_of(buckyballmesh, stone)
    scene.root.add_group(buckyball)
    for i in range(1, 7):
        p1 = A.new()
        p1.add_cube([0.0, 0.0, 0.0], 1.0, stone)
        p1.scale(0.8, 4.0, 0.8).translate([-2.4, 0.0, -0.4])
        p2 = A.new()
        p2.add_cube([0.0, 0.0, 0.0], 1.0, stone)
        p2.scale(0.8, 4.0, 0.8).translate([1.6, 0.0, -0.4])
        s = A.new()
        s.add_sphere([0.0, 0.0, 0.0], 1.0, stone)
        s.scale(4.0, 0.6, 0.6).translate([0.0, 4.0, 0.0])
        arc = A.new()
        arc.add_group(p1)
        arc.add_group(p2)
        arc.add_group(s)
        arc.translate([0.0, 0.0, -10.0])
        arc.rotate_y(float((i - 1) * 60))
        scene.root.add_group(arc)
    for translation, rotation in (([1.0, 1.3, 14.0], 20.0), ([5.0, 1.3, -11.0], 180.0), ([-5.5, 1.3, -3.0], -60.0)):
        cow = A.new()
        cow.scale(1.4, 1.4, 1.4).rotate_y(rotation).translate(translation)
        for center, radius in (([0.0, 0.0, 0.0], 1.0), ([0.9, 0.3, 0.0], 0.6), ([-0.94, 0.34, 0.0], 0.2), ([0.7, -0.7, -0.7], 0.3),
                               ([-0.7, -0.7, -0.7], 0.3), ([0.7, -0.7, 0.7], 0.3), ([-0.7, -0.7, 0.7], 0.3)):
            cow.add_sphere(center, radius, hide)
        scene.root.add_group(cow)
    scene.root.rotate_x(23.0)
    return scene


def instanced_scene(api, nu=24, nv=16, supersampling=0):
    """No glass / mirror (so every kernel organisation applies): one torus mesh instanced three times under
    different transforms and materials (the instances share BVH nodes and primrefs on the device), nested two
    groups deep, next to boxes, spheres and two lights."""
    scene = api.Scene.new()
    _cornell_shell(api, scene, supersampling)
    scene.add_point_light([-1.2, 1.2, 1.5], [0.4, 0.4, 0.5], [1.0, 0.0, 0.1])
    M = api.Material
    mesh = scene.parse_obj(torus_obj(nu, nv, normals=True))
    outer = api.Aggregate.new()
    outer.translate([0.1, -0.2, 0.0])
    for k, (mat, tr, rot) in enumerate([(M.plastic([0.9, 0.3, 0.2], [0.5, 0.7, 0.5], 0.2), [-0.9, -0.6, 0.2], 20.0),
                                        (M.matte([0.3, 0.8, 0.4], 25.0), [0.8, 0.1, -0.3], 75.0),
                                        (M.metal([0.2, 0.9, 1.1], [3.9, 2.4, 2.2], 0.15, 0.1), [0.0, 0.9, 0.4], 140.0)]):
        g = api.Aggregate.new()
        g.scale(0.55 + 0.1 * k, 0.55, 0.55)
        g.rotate_x(rot)
        g.rotate_z(15.0 * k)
        g.translate(tr)
        g.add_obj_of(mesh, mat)
        if k == 1:
            g.add_sphere([0.0, 0.0, 0.0], 0.25, M.plastic([0.2, 0.2, 0.9], [0.5, 0.5, 0.5], 0.3))
        outer.add_group(g)
    outer.add_cube([-0.3, -1.6, -0.9], 0.5, M.matte([0.8, 0.8, 0.2], 0.0))
    scene.root.add_group(outer)
    scene.root.add_box([1.0, -1.9, -1.0], [1.6, -1.2, -0.2], M.plastic([0.7, 0.4, 0.8], [0.4, 0.4, 0.4], 0.4))
    for i in range(40):
        scene.root.add_sphere([-1.6 + 0.08 * i, -1.7 + 0.02 * (i % 7), 1.2 - 0.05 * i], 0.06 + 0.002 * (i % 5), M.matte([0.6, 0.6, 0.6], 0.0))
    return scene


def quad_obj_with_uv():
    """A unit quad given as ONE 4-vertex polygon (only its first three vertices are used,
    src/shape/triangle.rs:41-53) plus two triangles, with vt and vn on every vertex."""
    return """o quad
v -1 -1 0
v 1 -1 0
v 1 1 0
v -1 1 0
v 0 2 0.5
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vt 0.5 0.25
vn 0 0 1
vn 0.2 0 1
vn 0 0.2 1
g first
f 1/1/1 2/2/2 3/3/3 4/4/1
g second
f 1/1/1 3/3/3 4/4/2
f -2/4/1 -3/3/2 -1/5/3
"""


def kitchen_sink_scene(api, camera="perspective", recursion=2, supersampling=1):
    """Every feature of the path in one scene: orthographic / perspective camera, radial background,
    three lights (one with distance falloff, one with f_att == 0), all five materials incl.
    Oren-Nayar and metal, cube + box, nested groups three levels deep with rotate(axis),
    swap_backface, the same mesh instanced three times (with / without material, with vt + vn),
    a flat-shaded mesh, supersampling."""
    scene = api.Scene.new()
    scene.set_ambient_light([0.15, 0.12, 0.1])
    scene.set_radial_background([0.2, 0.3, 0.6], [0.9, 0.8, 0.7], 0.7)
    scene.set_max_recursion_depth(recursion)
    if camera == "orthographic":
        cam = scene.set_orthographic_camera(7.0)
    else:
        cam = scene.set_perspective_camera(50.0)
    cam.look_at([0.5, 1.0, 9.0], [0.0, 0.0, 0.0], [0.1, 1.0, 0.0])
    cam.set_supersampling(supersampling)
    cam.set_aperture_radius(0.1)  # accepted and unused, as in the reference (camera.rs:142)
    M = api.Material
    matte = M.matte([0.8, 0.7, 0.6], 0.0)
    oren = M.matte([0.3, 0.8, 0.4], 35.0)
    plastic = M.plastic([0.2, 0.3, 1.0], [0.5, 0.7, 0.5], 0.25)
    shiny = M.plastic([0.0, 0.0, 0.0], [0.9, 0.9, 0.9], 0.05)  # kd == 0: microfacet only
    metal = M.metal([0.2, 0.9, 1.1], [3.9, 2.4, 2.2], 0.15, 0.3)
    glass = M.glass([1.0, 0.8, 1.0], [0.8, 1.0, 0.8], 1.4)
    mirror = M.mirror([0.7, 0.7, 0.7])
    scene.add_point_light([3.0, 5.0, 6.0], [0.9, 0.85, 0.8], [1.0, 0.0, 0.0])
    scene.add_point_light([-4.0, 2.0, 3.0], [0.5, 0.6, 0.9], [0.2, 0.05, 0.01])
    scene.add_point_light([0.0, 8.0, 0.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0])  # f_att == 0: contributes nothing
    quad = scene.parse_obj(quad_obj_with_uv())
    torus = scene.parse_obj(torus_obj(20, 12, normals=True))
    scene.set_mesh_smoothing(False)
    flat_torus = scene.parse_obj(torus_obj(10, 8, normals=True))  # normals dropped (scene.rs:111)
    scene.set_mesh_smoothing(True)
    root = scene.root
    root.rotate_y(8.0)
    root.add_sphere([0.0, -101.5, 0.0], 100.0, matte)           # ground
    root.add_sphere([-2.2, -0.5, 0.5], 1.0, glass)
    root.add_sphere([2.4, -0.7, 1.0], 0.8, mirror)
    root.add_sphere([0.2, -1.0, 2.6], 0.5, oren)
    root.add_cube([-0.6, -1.5, -0.6], 1.2, metal)
    root.add_box([1.0, -1.5, -2.5], [3.0, 0.5, -1.5], shiny)
    A = api.Aggregate
    g1 = A.new(); g1.translate([-3.0, 1.0, -2.0]).rotate(25.0, [0.0, 0.6, 0.8]).scale(1.2, 0.8, 1.0)
    g1.add_obj_of(torus, plastic)
    g2 = A.new(); g2.rotate_x(-60.0).translate([0.0, 1.6, 0.0])
    g2.add_obj(quad)                      # no material: Material::default()
    g3 = A.new(); g3.scale(0.5, 0.5, 0.5).rotate_z(45.0).translate([0.0, 0.0, 1.5])
    g3.add_obj_of(quad, oren)
    g3.add_sphere([0.0, 0.0, 1.0], 0.6, plastic)
    g3.swap_backface()
    g2.add_group(g3)
    g1.add_group(g2)
    root.add_group(g1)
    g4 = A.new(); g4.translate([2.5, 1.4, 0.0]).rotate_y(30.0)
    g4.add_obj_of(flat_torus, metal)
    g4.add_obj_of(torus, glass)           # same mesh, second instance
    g4.swap_backface(); g4.swap_backface()  # toggles twice: off
    root.add_group(g4)
    return scene


def random_scene(api, seed):
    """Seeded random scene for fuzz parity: every primitive / material / transform / camera feature,
    plus deliberately DUPLICATED primitives and touching boxes so that exact ties in t occur (the
    reference resolves them by visit order, bvh.rs:481-488)."""
    rng = SplitMix64(0xF00D0000 + seed)
    u = rng.uniform

    def pick(n):
        return min(int(rng.next_f64() * n), n - 1)

    def col(lo=0.0, hi=1.0):
        return [u(lo, hi), u(lo, hi), u(lo, hi)]

    M = api.Material

    def material():
        k = pick(7)
        if k == 0:
            return M.matte(col(), 0.0)
        if k == 1:
            return M.matte(col(), u(5.0, 60.0))
        if k == 2:
            return M.plastic(col(), col(0.2, 0.9), u(0.05, 0.6))
        if k == 3:
            return M.plastic([0.0, 0.0, 0.0], col(0.2, 0.9), u(0.05, 0.6))
        if k == 4:
            return M.metal(col(0.1, 1.5), col(1.5, 4.0), u(0.05, 0.5), u(0.05, 0.5))
        if k == 5:
            return M.glass(col(0.3, 1.0), col(0.3, 1.0), u(1.1, 1.8))
        return M.mirror(col(0.2, 0.9))

    scene = api.Scene.new()
    scene.set_ambient_light(col(0.0, 0.3))
    if pick(2):
        scene.set_radial_background(col(), col(), u(0.2, 1.0))
    else:
        scene.set_solid_background(col(0.0, 0.5))
    scene.set_max_recursion_depth(pick(4))
    cam = scene.set_orthographic_camera(u(6.0, 10.0)) if pick(4) == 0 else scene.set_perspective_camera(u(35.0, 70.0))
    cam.look_at([u(-2, 2), u(-1, 3), u(7, 10)], [u(-0.5, 0.5), u(-0.5, 0.5), 0.0], [u(-0.2, 0.2), 1.0, u(-0.2, 0.2)])
    cam.set_supersampling(pick(2))
    for _ in range(1 + pick(3)):
        fall = [[1.0, 0.0, 0.0], [0.5, 0.05, 0.0], [0.3, 0.02, 0.01]][pick(3)]
        scene.add_point_light([u(-6, 6), u(2, 8), u(-2, 8)], col(0.3, 1.0), fall)
    meshes = [scene.parse_obj(PLANE_OBJ), scene.parse_obj(quad_obj_with_uv()), scene.parse_obj(torus_obj(8 + pick(8), 6 + pick(6), normals=True))]
    scene.set_mesh_smoothing(False)
    meshes.append(scene.parse_obj(torus_obj(7, 5, normals=True)))
    scene.set_mesh_smoothing(True)

    def fill(agg, n, allow_group, depth):
        for _ in range(n):
            k = pick(10)
            c = [u(-3, 3), u(-2, 2), u(-3, 3)]
            if k <= 3:
                agg.add_sphere(c, u(0.2, 1.0), material())
            elif k == 4:  # the same sphere twice: an exact tie in t between two primitives
                m1, m2, r = material(), material(), u(0.3, 0.8)
                agg.add_sphere(c, r, m1)
                agg.add_sphere(c, r, m2)
            elif k == 5:
                agg.add_cube(c, u(0.3, 1.2), material())
            elif k == 6:  # two boxes sharing a face
                d = u(0.4, 1.0)
                agg.add_box(c, [c[0] + d, c[1] + d, c[2] + d], material())
                agg.add_box([c[0] + d, c[1], c[2]], [c[0] + 2 * d, c[1] + d, c[2] + d], material())
            elif k == 7:
                mi = pick(len(meshes))
                g = api.Aggregate.new()
                g.translate(c).scale(u(0.5, 1.5), u(0.5, 1.5), u(0.5, 1.5))
                if pick(2):
                    g.add_obj_of(meshes[mi], material())
                else:
                    g.add_obj(meshes[mi])
                agg.add_group(g)
            elif allow_group and depth < 3:
                g = api.Aggregate.new()
                t = pick(5)
                if t == 0:
                    g.rotate_x(u(-90, 90))
                elif t == 1:
                    g.rotate_y(u(-90, 90))
                elif t == 2:
                    g.rotate_z(u(-90, 90))
                elif t == 3:
                    g.rotate(u(-90, 90), [0.48, 0.6, 0.64])
                g.translate([u(-2, 2), u(-1, 1), u(-2, 2)])
                if pick(3) == 0:
                    g.swap_backface()
                fill(g, 2 + pick(4), True, depth + 1)
                agg.add_group(g)
            else:
                agg.add_sphere(c, u(0.2, 0.7), material())

    root = scene.root
    if pick(3) == 0:
        root.rotate_y(u(-20, 20))
    root.add_sphere([0.0, -103.0, 0.0], 100.0, material())
    fill(root, 6 + pick(30), True, 0)
    return scene


# ---- adversarial generators (built against the fast mode's margins; also run through the reference traversal and the
# oracle: the reference's behaviour on ill-conditioned input is reproduced too) ----------------------------------------
def adversarial_scene(api, seed):
    """Giant spheres as walls, rays grazing them, lights and camera close to surfaces, needle boxes, (nearly) coincident
    primitives (tools/fast_adversarial.py searches for fast-mode counter-examples with it)."""
    import numpy as np
    G = api; M = api.Material
    rng = np.random.default_rng(seed)
    sc = G.Scene.new()
    cam = sc.set_perspective_camera(float(rng.uniform(20, 100)))
    eye = rng.uniform(-1, 1, 3) * [1.5, 1.0, 1.0] + [0, 0, 4.5]
    cam.look_at(eye.tolist(), (rng.uniform(-0.5, 0.5, 3)).tolist(), [0, 1, 0])
    sc.set_ambient_light([0.1, 0.1, 0.1])
    R = float(10.0 ** rng.uniform(2, 6))           # giant wall spheres
    mats = [M.matte(rng.uniform(0.2, 1, 3).tolist(), 0.0), M.plastic(rng.uniform(0.2, 1, 3).tolist(), [0.5, 0.5, 0.5], 0.3)]
    root = sc.root
    for axis, sign in ((1, -1), (1, 1), (0, -1), (0, 1), (2, -1)):
        c = [0.0, 0.0, 0.0]; c[axis] = sign * (R + 2.0)
        root.add_sphere(c, R, mats[int(rng.integers(2))])
    n = int(rng.integers(20, 200))
    for i in range(n):
        c = rng.uniform(-1.8, 1.8, 3)
        r = float(10.0 ** rng.uniform(-3, -0.5))
        root.add_sphere(c.tolist(), r, mats[i % 2])
        if rng.random() < 0.1:   # nearly coincident twin
            root.add_sphere((c + rng.uniform(-1, 1, 3) * 1e-9).tolist(), r * (1 + float(rng.uniform(-1, 1)) * 1e-9), mats[(i + 1) % 2])
    for i in range(int(rng.integers(0, 12))):
        lo = rng.uniform(-1.8, 1.2, 3); d = 10.0 ** rng.uniform(-4, 0, 3)
        root.add_box(lo.tolist(), (lo + d).tolist(), mats[i % 2])
    for i in range(int(rng.integers(1, 4))):      # lights, some very close to a wall
        p = rng.uniform(-1.9, 1.9, 3)
        if rng.random() < 0.5: p[1] = 2.0 - 10.0 ** rng.uniform(-6, -1)
        sc.add_point_light(p.tolist(), rng.uniform(0.2, 0.9, 3).tolist(), [1.0, 0.0, 0.0])
    return sc

def _sliver_obj(rng, n):
    """OBJ text: needle and sliver triangles, nearly edge-on fans, duplicated vertices."""
    lines = []
    for i in range(n):
        c = rng.uniform(-1.2, 1.2, 3)
        kind = rng.integers(4)
        if kind == 0:   # needle
            a = c; b = c + rng.uniform(-1, 1, 3); d = a + (b - a) * 0.5 + rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(-9, -3)
        elif kind == 1: # tiny
            a = c; b = c + rng.uniform(-1, 1, 3) * 1e-6; d = c + rng.uniform(-1, 1, 3) * 1e-6
        elif kind == 2: # big, axis aligned (edge-on for axis-parallel rays)
            a = c; b = c + [float(rng.uniform(0.2, 1.5)), 0.0, 0.0]; d = c + [0.0, float(rng.uniform(0.2, 1.5)), 0.0]
        else:           # generic
            a = c; b = c + rng.uniform(-0.6, 0.6, 3); d = c + rng.uniform(-0.6, 0.6, 3)
        for v in (a, b, d):
            lines.append("v %.9g %.9g %.9g" % tuple(v))
        lines.append("f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    return "\n".join(lines) + "\n"


def adversarial_mesh_scene(api, seed):
    """Meshes of degenerate triangles under extreme nested transforms, orthographic or distant cameras."""
    import numpy as np
    G = api; M = api.Material
    rng = np.random.default_rng(seed + 100000)
    sc = G.Scene.new()
    if rng.random() < 0.3:
        cam = sc.set_orthographic_camera(float(rng.uniform(2, 6)))
    else:
        cam = sc.set_perspective_camera(float(rng.uniform(5, 90)))
    dist = float(10.0 ** rng.uniform(0.5, 3))
    eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * dist
    cam.look_at(eye.tolist(), (rng.uniform(-0.3, 0.3, 3)).tolist(), [0, 1, 0])
    sc.set_ambient_light([0.2, 0.2, 0.2])
    mats = [M.matte(rng.uniform(0.2, 1, 3).tolist(), 0.0), M.plastic(rng.uniform(0.2, 1, 3).tolist(), [0.5, 0.5, 0.5], 0.3)]
    mesh = sc.parse_obj(_sliver_obj(rng, int(rng.integers(20, 400))))
    root = sc.root
    for k in range(int(rng.integers(1, 4))):
        g = G.Aggregate.new()
        g.scale(float(10.0 ** rng.uniform(-2, 1)), float(10.0 ** rng.uniform(-2, 1)), float(10.0 ** rng.uniform(-2, 1)))
        ax = rng.normal(size=3)
        if rng.random() < 0.9: ax = ax / np.linalg.norm(ax)  # a non-unit axis makes transform and inverse disagree: fast mode is refused
        g.rotate(float(rng.uniform(0, 360)), ax.tolist())
        g.translate(rng.uniform(-1, 1, 3).tolist())
        g.add_obj_of(mesh, mats[k % 2])
        if rng.random() < 0.5:
            inner = G.Aggregate.new()
            inner.rotate_x(float(rng.uniform(0, 360)))
            inner.scale(float(10.0 ** rng.uniform(-1, 1)), 1.0, 1.0)
            for i in range(int(rng.integers(1, 30))):
                inner.add_sphere(rng.uniform(-1, 1, 3).tolist(), float(10.0 ** rng.uniform(-3, -0.3)), mats[i % 2])
            g.add_group(inner)
        root.add_group(g)
    for i in range(int(rng.integers(0, 40))):
        root.add_sphere(rng.uniform(-1.5, 1.5, 3).tolist(), float(10.0 ** rng.uniform(-3, -0.5)), mats[i % 2])
    for i in range(int(rng.integers(1, 3))):
        sc.add_point_light((rng.normal(size=3) * dist * 0.7).tolist(), rng.uniform(0.3, 0.9, 3).tolist(), [1.0, 0.0, 0.0])
    return sc


def _slab_mesh_obj(rng, n_quads, snap, normals=False):
    """OBJ text: axis-aligned quads in planes whose offsets are multiples of `snap` (rays of an axis-aligned orthographic
    camera on the same lattice lie IN those planes: triangles seen exactly edge-on), generic triangles, and long thin ones.
    With `normals` every triangle corner carries a shading normal of its own: two triangles that meet in an edge then shade
    differently, so WHICH of them wins an exact tie in t (a ray through the shared edge) shows in the picture."""
    lines, faces, nv, nn = [], [], 0, 0
    def quad(p, e1, e2):
        nonlocal nv, nn
        for v in (p, p + e1, p + e1 + e2, p + e2):
            lines.append("v %.9g %.9g %.9g" % tuple(v))
        for tri in ((1, 2, 3), (1, 3, 4)):
            if normals:
                face_n = np.cross(e1, e2)
                face_n = face_n / (np.linalg.norm(face_n) or 1.0)
                ids = []
                for _ in range(3):
                    n = face_n + rng.uniform(-0.4, 0.4, 3)
                    lines.append("vn %.6f %.6f %.6f" % tuple(n))
                    nn += 1
                    ids.append(nn)
                faces.append("f %d//%d %d//%d %d//%d" % (nv + tri[0], ids[0], nv + tri[1], ids[1], nv + tri[2], ids[2]))
            else:
                faces.append("f %d %d %d" % (nv + tri[0], nv + tri[1], nv + tri[2]))
        nv += 4
    import numpy as np
    for i in range(n_quads):
        kind = int(rng.integers(4))
        p = np.round(rng.uniform(-1.5, 1.5, 3) / snap) * snap
        if kind < 3:  # a quad in the plane {axis = const}
            e1 = np.zeros(3); e2 = np.zeros(3)
            e1[(kind + 1) % 3] = float(np.round(rng.uniform(0.05, 0.8) / snap) * snap) or snap
            e2[(kind + 2) % 3] = float(np.round(rng.uniform(0.05, 0.8) / snap) * snap) or snap
            quad(p, e1, e2)
        else:         # a generic or a needle quad
            e1 = rng.uniform(-0.5, 0.5, 3)
            e2 = rng.uniform(-0.5, 0.5, 3) * (10.0 ** rng.uniform(-6, 0))
            quad(p, e1, e2)
    return "\n".join(lines + faces) + "\n"


def adversarial_prune_scene(api, seed):
    """Built against the pruned reference walk (lg_accel_set_prune): meshes of several fat leaves (>= 600 triangles) full of
    coplanar, axis-aligned faces; cameras whose rays lie exactly in those planes (orthographic, axis-aligned, on the same
    lattice) or graze them; spheres resting on the planes and touching each other (rays through the poles and the tangent
    points); boxes sharing faces; tiny spheres seen from far away; lights inside, outside and on surfaces."""
    import numpy as np
    G = api; M = api.Material
    rng = np.random.default_rng(seed + 200000)
    sc = G.Scene.new()
    snap = 1.0 / 32.0
    mode = int(rng.integers(4))
    if mode == 0:    # orthographic, axis-aligned, film lattice = the mesh's lattice (64 x 48 film, scale 3: pixel pitch 1/16)
        cam = sc.set_orthographic_camera(3.0)
        eye = np.round(rng.uniform(-0.5, 0.5, 3) / snap) * snap + [0, 0, 6.0]
        cam.look_at(eye.tolist(), (eye - [0, 0, 6.0]).tolist(), [0, 1, 0])
    elif mode == 1:  # perspective from a lattice point: the central row and column of rays have an exactly zero component
        cam = sc.set_perspective_camera(float(rng.uniform(30, 70)))
        eye = np.round(rng.uniform(-1, 1, 3) / snap) * snap + [0, 0, 5.0]
        cam.look_at(eye.tolist(), (eye - [0, 0, 5.0]).tolist(), [0, 1, 0])
    elif mode == 2:  # far away: large |o - c| / r
        dist = float(10.0 ** rng.uniform(2, 4))
        cam = sc.set_perspective_camera(float(np.degrees(2.0 * np.arctan(2.0 / dist)) * rng.uniform(0.6, 1.4)))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        cam.look_at((d * dist).tolist(), rng.uniform(-0.3, 0.3, 3).tolist(), [0, 1, 0])
    else:            # generic, inside the cloud
        cam = sc.set_perspective_camera(float(rng.uniform(40, 100)))
        cam.look_at(rng.uniform(-1.2, 1.2, 3).tolist(), rng.uniform(-1, 1, 3).tolist(), [0, 1, 0])
    sc.set_ambient_light([0.15, 0.15, 0.15])
    sc.set_max_recursion_depth(int(rng.integers(0, 3)))
    mats = [M.matte(rng.uniform(0.2, 1, 3).tolist(), 0.0), M.plastic(rng.uniform(0.2, 1, 3).tolist(), [0.5, 0.5, 0.5], 0.3),
            M.mirror([0.6, 0.6, 0.6]), M.glass([0.9, 0.9, 0.9], [0.9, 0.9, 0.9], 1.3)]
    nmat = 4 if rng.random() < 0.4 else 2
    mesh = sc.parse_obj(_slab_mesh_obj(rng, int(rng.integers(300, 700)), snap, normals=bool(rng.random() < 0.5)))
    root = sc.root
    if rng.random() < 0.5:
        root.add_obj_of(mesh, mats[0])                       # the mesh accel directly in the root (identity all the way)
    else:
        g = G.Aggregate.new()
        if rng.random() < 0.5:
            g.translate((np.round(rng.uniform(-0.5, 0.5, 3) / snap) * snap).tolist())  # lattice-preserving
        else:
            g.rotate_y(float(rng.choice([90.0, 180.0, 37.0]))); g.scale(1.0, float(rng.choice([1.0, 0.5, 2.0])), 1.0)
        g.add_obj_of(mesh, mats[1])
        root.add_group(g)
    # spheres resting on lattice planes, chains of touching spheres, twins
    for i in range(int(rng.integers(10, 120))):
        r = float(rng.choice([snap, 2 * snap, 0.1, 10.0 ** rng.uniform(-3, -0.7)]))
        c = np.round(rng.uniform(-1.5, 1.5, 3) / snap) * snap
        c[1] += r if rng.random() < 0.5 else 0.0             # its lowest point on a lattice plane
        root.add_sphere(c.tolist(), r, mats[i % nmat])
        if rng.random() < 0.3:                               # a neighbour touching it exactly (along an axis)
            ax = int(rng.integers(3)); c2 = c.copy(); c2[ax] += 2 * r
            root.add_sphere(c2.tolist(), r, mats[(i + 1) % nmat])
    for i in range(int(rng.integers(0, 16))):                # boxes on the lattice, sharing faces
        lo = np.round(rng.uniform(-1.5, 1.0, 3) / snap) * snap
        d = np.round(rng.uniform(snap, 0.6, 3) / snap) * snap
        root.add_box(lo.tolist(), (lo + d).tolist(), mats[i % nmat])
        if rng.random() < 0.5:
            lo2 = lo.copy(); lo2[0] += d[0]
            root.add_box(lo2.tolist(), (lo2 + d).tolist(), mats[(i + 1) % nmat])
    for i in range(int(rng.integers(1, 4))):
        p = np.round(rng.uniform(-1.8, 1.8, 3) / snap) * snap if rng.random() < 0.5 else rng.uniform(-3, 3, 3)
        sc.add_point_light(p.tolist(), rng.uniform(0.3, 0.9, 3).tolist(), [1.0, 0.0, 0.0])
    return sc


def progression_soup_scene(api, seed):
    """A triangle soup whose centroids form a geometric progression along x (ratio 1.2 or 1.3) with random orientations:
    inside one fat leaf of the reference tree no run of three or more triangles is coherent and every largest-gap cut peels
    one triangle off -- the host's culling records (host.cpp, build_chunks) must fall back to plain runs instead of failing
    (round 3 refused such a scene: "too many culling records in one leaf").  Seen from the side, with spheres among it."""
    import numpy as np
    G = api; M = api.Material
    rng = np.random.default_rng(seed + 300000)
    ratio = float(rng.choice([1.2, 1.3]))
    n = int(rng.integers(250, 520))
    lines = []
    for i in range(n):
        cx = ratio ** (i % 254) * 1e-20 * (1.0 if i < 254 else -1.0)   # f32 range: 1.3^253 * 1e-20 ~ 7e8
        c = np.array([cx, float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))])
        size = max(abs(cx) * 0.4, 0.05)
        for _ in range(3):
            lines.append("v %.9g %.9g %.9g" % tuple(c + rng.uniform(-1, 1, 3) * size))
        lines.append("f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    sc = G.Scene.new()
    cam = sc.set_perspective_camera(float(rng.uniform(40, 80)))
    cam.look_at([float(rng.uniform(-2, 2)), float(rng.uniform(-1, 1)), 6.0], [0.0, 0.0, 0.0], [0, 1, 0])
    sc.set_ambient_light([0.2, 0.2, 0.2])
    sc.set_max_recursion_depth(int(rng.integers(0, 3)))
    mats = [M.matte(rng.uniform(0.2, 1, 3).tolist(), 0.0), M.plastic(rng.uniform(0.2, 1, 3).tolist(), [0.5, 0.5, 0.5], 0.3),
            M.mirror([0.6, 0.6, 0.6])]
    mesh = sc.parse_obj("\n".join(lines) + "\n")
    sc.root.add_obj_of(mesh, mats[int(rng.integers(3))])
    for i in range(int(rng.integers(2, 12))):
        sc.root.add_sphere(rng.uniform(-1.5, 1.5, 3).tolist(), float(10.0 ** rng.uniform(-1.5, -0.3)), mats[i % 3])
    for i in range(int(rng.integers(1, 3))):
        sc.add_point_light(rng.uniform(-4, 4, 3).tolist(), rng.uniform(0.3, 0.9, 3).tolist(), [1.0, 0.0, 0.0])
    return sc


def exotic_obj():
    """OBJ text in the forms the `obj` crate accepts beyond plain `f a b c`: comments, `o` / `g` statements, `v//vn` faces,
    NEGATIVE (relative) indices for positions and normals, a 4- and a 5-vertex polygon (the reference keeps the first three
    vertices of each, src/shape/triangle.rs:39-53), exponent notation and surplus whitespace."""
    return """# a small tent: two quads, a pentagon and two triangles
o tent
v -1.0 0.0 -1.0
v  1.0 0.0 -1.0
v  1.0 0.0  1.0
v -1.0 0.0  1.0
v  0.0 1.2e0 0.0
vn 0 1 0
vn -0.7 0.7 0
vn 0.7 0.7 0
vn 0 0.7 0.7
vn 0 0.7 -0.7
g floor
f 1//1 2//1 3//1 4//1
g sides
f -5//-4 -1//-4   -2//-4
f 2//3 3//3 5//3
f 3//4 4//4 5//4
o fin
g fin
v 1.5 0.0 -0.5
v 2.2 0.0 -0.2
v 2.4 0.9 0.0
v 1.9 1.4 0.2
v 1.4 0.8 0.1
vn 0.1 0.2 1.0
f -5//-1 -4//-1 -3//-1 -2//-1 -1//-1
f 1//5 5//5 2//5
"""


def exotic_obj_scene(api, smoothing=True):
    """The `exotic_obj` mesh (negative indices, v//vn, 4- and 5-vertex polygons, o / g groups) under two lights, once as
    parsed and once more in a rotated group, in front of a sphere: a render depends on every index being resolved as the
    `obj` crate resolves it."""
    scene = api.Scene.new()
    scene.set_ambient_light([0.15, 0.15, 0.2])
    scene.set_radial_background([0.3, 0.4, 0.6], [0.05, 0.05, 0.1], 0.6)
    cam = scene.set_perspective_camera(50.0)
    cam.look_at([1.0, 2.5, 6.0], [0.6, 0.4, 0.0], [0.0, 1.0, 0.0])
    scene.set_mesh_smoothing(smoothing)
    M = api.Material
    mesh = scene.parse_obj(exotic_obj())
    scene.add_point_light([3.0, 5.0, 4.0], [0.9, 0.85, 0.8], [1.0, 0.0, 0.0])
    scene.add_point_light([-4.0, 3.0, 2.0], [0.3, 0.4, 0.6], [1.0, 0.0, 0.0])
    scene.root.add_obj_of(mesh, M.plastic([0.8, 0.5, 0.3], [0.5, 0.5, 0.5], 0.3))
    g = api.Aggregate.new()
    g.rotate_y(140.0)
    g.translate([-1.8, 0.2, -1.0])
    g.add_obj_of(mesh, M.matte([0.4, 0.7, 0.5], 10.0))
    scene.root.add_group(g)
    scene.root.add_sphere([0.5, -50.0, 0.0], 49.9, M.matte([0.5, 0.5, 0.5], 0.0))
    return scene


def tie_mesh_scene(api, n=24, seed=5):
    """A flat n x n grid of quads (2 n^2 triangles, several fat leaves of the reference tree) whose corners carry random shading
    normals of their own, under an orthographic camera on the grid's lattice: at a 128 x 128 film every fourth row and column of
    rays passes exactly through edges and corners shared by two to six triangles -- exact ties in t, which the reference gives to the triangle
    that comes first in its leaf's order[] (and, across leaves, in its visit order).  The picture shows who won."""
    import numpy as np
    rng = np.random.default_rng(seed)
    step = 1.0 / 8.0
    lines, faces, nn = ["o grid"], [], 0
    for j in range(n + 1):
        for i in range(n + 1):
            lines.append("v %.9g %.9g 0" % ((i - n / 2) * step, (j - n / 2) * step))
    def vid(i, j):
        return j * (n + 1) + i + 1
    for j in range(n):
        for i in range(n):
            for tri in ((vid(i, j), vid(i + 1, j), vid(i + 1, j + 1)), (vid(i, j), vid(i + 1, j + 1), vid(i, j + 1))):
                ids = []
                for _ in range(3):
                    v = np.array([0.0, 0.0, 1.0]) + rng.uniform(-0.5, 0.5, 3)
                    lines.append("vn %.6f %.6f %.6f" % tuple(v))
                    nn += 1
                    ids.append(nn)
                faces.append("f %d//%d %d//%d %d//%d" % (tri[0], ids[0], tri[1], ids[1], tri[2], ids[2]))
    scene = api.Scene.new()
    scene.set_ambient_light([0.2, 0.2, 0.2])
    cam = scene.set_orthographic_camera(4.0)   # image plane 4 high: 128 rows of 1/32 -- every fourth row and column of rays runs along grid lines
    cam.look_at([0.0, 0.0, 4.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])
    mesh = scene.parse_obj("\n".join(lines + faces) + "\n")
    scene.add_point_light([1.0, 2.0, 3.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0])
    scene.add_point_light([-2.0, -1.0, 2.5], [0.4, 0.5, 0.7], [1.0, 0.0, 0.0])
    scene.root.add_obj_of(mesh, api.Material.plastic([0.8, 0.6, 0.3], [0.5, 0.5, 0.5], 0.3))
    g = api.Aggregate.new()
    g.translate([0.25, -0.125, -0.5])   # a second copy behind the first, also on the lattice
    g.add_obj_of(mesh, api.Material.matte([0.3, 0.6, 0.8], 0.0))
    scene.root.add_group(g)
    return scene
