"""A third, independent witness for the shading half of the path -- TEST INFRASTRUCTURE, like oracle/.

Plain-Python (IEEE f64 `float`, `math`) restatement of the reference's Whitted integrator for scenes made of spheres
boxes and OBJ triangle meshes in nested, transformed groups: camera, sphere / cuboid / triangle intersection with their
differentials and shading normals, Transform3 and
the aggregates' transform concatenation, backface swapping, SurfaceInteraction, the five materials and every BxDF they use,
point lights with shadow rays, specular recursion, background.  It was written from the
Rust sources cited below, not from oracle/lasgun_oracle.cpp, shares no code with it, and finds the closest hit by brute
force over every primitive of every group (no BVH: scenes must not contain exact ties in t, where the reference's visit
order decides, nor rays grazing a primitive's bounding box to the last bit).

It pins nothing (only the reference's own 17 known-answer tests do), but a transcription slip in the oracle's camera /
shading / recursion code would have to be made twice, in two languages, to go unseen.  tests/test_pyref_witness.py compares
it with the oracle (libm trigonometry on both sides: Rust and CPython both call glibc).

cgmath 0.17 operation order as in SURVEY.md Appendix A1 (dot = (x*x + y*y) + z*z, normalize = v * (1 / |v|), ...).
"""
import ctypes
import ctypes.util
import math

PI = math.pi
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
_libm.sincos.restype = None


def sincos(x):
    """(sin x, cos x) from glibc's sincos(): what a native build of the reference calls where it takes both of one argument
    (cgmath's Rad::sin_cos in the rotation matrices, `phi.cos()` / `phi.sin()` in sphere.rs:107-108) -- LLVM merges an
    fsin / fcos pair on one operand into the sincos libcall on *-linux-gnu.  It is not always bit-identical to
    (math.sin(x), math.cos(x)): this witness found that on its 36th random scene (rotate_x(71.69565646291534))."""
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(x, ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value
FRAC_1_PI = 0.318309886183790671537767526745028724  # std::f64::consts::FRAC_1_PI
INF = float("inf")


# ---- cgmath-shaped vector helpers -------------------------------------------------------------
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def mul(a, s): return (a[0] * s, a[1] * s, a[2] * s)          # v * s
def smul(s, a): return (s * a[0], s * a[1], s * a[2])          # s * v
def div(a, s): return (_div(a[0], s), _div(a[1], s), _div(a[2], s))
def neg(a): return (-a[0], -a[1], -a[2])
def mulv(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])  # mul_element_wise
def divv(a, b): return (a[0] / b[0], a[1] / b[1], a[2] / b[2])
def dot(a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def magnitude(a): return math.sqrt(dot(a, a))
def normalize(a): return mul(a, _div(1.0, magnitude(a)))  # a zero vector becomes NaNs, as in cgmath
def fmin(a, b): return b if (a != a) else (a if (b != b) else min(a, b))  # f64::min (NaN-ignoring)
def fmax(a, b): return b if (a != a) else (a if (b != b) else max(a, b))
ZERO = (0.0, 0.0, 0.0)


def signum(x):  # f64::signum: +0 -> 1, -0 -> -1, NaN -> NaN
    if x != x:
        return x
    return math.copysign(1.0, x)


# ---- scene description (duck-typed like lasgun_amd.api / the oracle's api, spheres only) -------
class Material:
    def __init__(self, kind, *p):
        self.kind, self.p = kind, p

    @staticmethod
    def matte(kd, sigma): return Material("matte", tuple(map(float, kd)), fmin(fmax(float(sigma), 0.0), 90.0))  # matte.rs:15
    @staticmethod
    def plastic(kd, ks, rough): return Material("plastic", tuple(map(float, kd)), tuple(map(float, ks)), float(rough))
    @staticmethod
    def metal(eta, k, ur, vr): return Material("metal", tuple(map(float, eta)), tuple(map(float, k)), float(ur), float(vr))
    @staticmethod
    def glass(kr, kt, eta): return Material("glass", tuple(map(float, kr)), tuple(map(float, kt)), float(eta))
    @staticmethod
    def mirror(kr): return Material("mirror", tuple(map(float, kr)))


class Camera:  # camera.rs:59-102
    def __init__(self, perspective, param):
        self.perspective, self.param = perspective, float(param)
        self.origin, self.view, self.up, self.aux = ZERO, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)
        self.root, self.distance = 1, 1.0
        self.image_plane_height = self._iph(1.0)
        self.pixel_separation = 0.0 if perspective else 1.0

    def _iph(self, focal):  # camera.rs:158-164
        return focal * math.tan(self.param * PI / 360.0) * 2.0 if self.perspective else self.param

    def look_at(self, origin, look, up):  # camera.rs:85-94
        origin, look, up = tuple(map(float, origin)), tuple(map(float, look)), tuple(map(float, up))
        view = sub(look, origin)
        aux = cross(view, up)
        self.origin = origin
        self.up = normalize(cross(aux, view))
        self.aux = normalize(aux)
        self.view = view
        self.image_plane_height = self._iph(magnitude(view))
        return self

    def set_supersampling(self, base):  # camera.rs:189-193
        self.root = int(base) + 1
        self.distance = 1.0 / float(self.root)
        return self

    def sample(self, x, y, w, h):  # camera.rs:113-146, Film's cached 1/w, 1/h, w/h (film.rs:40-42)
        winv, hinv, aspect = 1.0 / float(w), 1.0 / float(h), float(w) / float(h)
        iph = self.image_plane_height
        ipw = iph * aspect
        pixel_size = iph * hinv
        sep = self.distance * pixel_size
        sx = (float(x) * winv - 0.5) * ipw
        sy = (0.5 - float(y + 1) * hinv) * iph
        origin = add(add(self.origin, mul(self.up, sy * self.pixel_separation)), mul(self.aux, sx * self.pixel_separation))
        d = add(add(self.view, smul(sy, self.up)), smul(sx, self.aux))
        updiff, auxdiff = mul(self.up, sep), mul(self.aux, sep)
        halfdiff = add(mul(updiff, 0.5), mul(auxdiff, 0.5))
        rays = []
        for i in range(self.root):
            for j in range(self.root):
                rays.append((origin, add(add(add(d, smul(float(j), updiff)), smul(float(i), auxdiff)), halfdiff)))
        return rays


# ---- Transform3 (space/transform.rs; cgmath 0.17 Matrix4, column-major m[col][row]) ----------------
def mat_identity():
    return [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]


def mat_mul_vec4(m, v):  # ((c0*x + c1*y) + c2*z) + c3*w
    return tuple(((m[0][r] * v[0] + m[1][r] * v[1]) + m[2][r] * v[2]) + m[3][r] * v[3] for r in range(4))


def mat_mul(a, b):  # a * b: column j of the product = a * (column j of b)
    return [list(mat_mul_vec4(a, b[j])) for j in range(4)]


def mat_transpose(m):
    return [[m[r][c] for r in range(4)] for c in range(4)]


def transform_vector(m, v):
    return mat_mul_vec4(m, (v[0], v[1], v[2], 0.0))[:3]


def transform_point(m, p):
    h = mat_mul_vec4(m, (p[0], p[1], p[2], 1.0))
    return mul(h[:3], 1.0 / h[3])


def transform_normal(minv, n):  # transform.rs:202-209
    return tuple(minv[i][0] * n[0] + minv[i][1] * n[1] + minv[i][2] * n[2] for i in range(3))


class Transform:
    def __init__(self):
        self.m, self.minv = mat_identity(), mat_identity()

    def concat_self(self, m, minv):  # transform.rs:191-197: m <- other.m * m, minv <- minv * other.minv
        self.m, self.minv = mat_mul(m, self.m), mat_mul(self.minv, minv)


def _rot(theta_deg, axis):  # Matrix4::from_angle_{x,y,z}; Rad = Deg * (pi / 180)
    r = theta_deg * (PI / 180.0)
    s, c = sincos(r)
    m = mat_identity()
    if axis == 0:
        m[1], m[2] = [0.0, c, s, 0.0], [0.0, -s, c, 0.0]
    elif axis == 1:
        m[0], m[2] = [c, 0.0, -s, 0.0], [s, 0.0, c, 0.0]
    else:
        m[0], m[1] = [c, s, 0.0, 0.0], [-s, c, 0.0, 0.0]
    return m


class Aggregate:  # scene/node.rs:25-115
    def __init__(self):
        self.contents = []  # ("sphere", c, r, mat) / ("cuboid", min, max, mat) / ("group", Aggregate)
        self.transform = Transform()
        self.swap = False

    @staticmethod
    def new(): return Aggregate()

    def add_sphere(self, c, r, mat):
        self.contents.append(("sphere", tuple(map(float, c)), float(r), mat))
        return self

    def add_cube(self, origin, dim, mat):  # cuboid.rs:23-29: max = origin + Vector::from_value(dim)
        o = tuple(map(float, origin))
        self.contents.append(("cuboid", o, (o[0] + float(dim), o[1] + float(dim), o[2] + float(dim)), mat))
        return self

    def add_box(self, mn, mx, mat):
        self.contents.append(("cuboid", tuple(map(float, mn)), tuple(map(float, mx)), mat))
        return self

    def add_group(self, agg):
        self.contents.append(("group", agg))
        return self

    def add_obj(self, mesh):  # node.rs:69-76: a mesh with the default material
        self.contents.append(("mesh", mesh, None))
        return self

    def add_obj_of(self, mesh, mat):
        self.contents.append(("mesh", mesh, mat))
        return self

    def swap_backface(self):
        self.swap = not self.swap
        return self

    def translate(self, d):
        d = tuple(map(float, d))
        m, minv = mat_identity(), mat_identity()
        m[3] = [d[0], d[1], d[2], 1.0]
        minv[3] = [-d[0], -d[1], -d[2], 1.0]
        self.transform.concat_self(m, minv)
        return self

    def scale(self, x, y, z):
        x, y, z = float(x), float(y), float(z)
        m, minv = mat_identity(), mat_identity()
        m[0][0], m[1][1], m[2][2] = x, y, z
        minv[0][0], minv[1][1], minv[2][2] = 1.0 / x, 1.0 / y, 1.0 / z
        self.transform.concat_self(m, minv)
        return self

    def rotate(self, t, axis):  # Matrix4::from_axis_angle (cgmath 0.17), inverse = transpose (transform.rs:144-148)
        x, y, z = map(float, axis)
        s, c = sincos(float(t) * (PI / 180.0))
        k = 1.0 - c
        m = mat_identity()
        m[0] = [k * x * x + c, k * x * y + s * z, k * x * z - s * y, 0.0]
        m[1] = [k * x * y - s * z, k * y * y + c, k * y * z + s * x, 0.0]
        m[2] = [k * x * z + s * y, k * y * z - s * x, k * z * z + c, 0.0]
        self.transform.concat_self(m, mat_transpose(m))
        return self

    def rotate_x(self, t): m = _rot(float(t), 0); self.transform.concat_self(m, mat_transpose(m)); return self
    def rotate_y(self, t): m = _rot(float(t), 1); self.transform.concat_self(m, mat_transpose(m)); return self
    def rotate_z(self, t): m = _rot(float(t), 2); self.transform.concat_self(m, mat_transpose(m)); return self


class Scene:  # scene.rs:49-62 defaults
    def __init__(self):
        self.root = Aggregate()
        self.camera = Camera(True, 45.0)
        self.bg_inner = self.bg_outer = ZERO
        self.bg_scale = 1.0
        self.ambient = ZERO
        self.recursion = 3
        self.lights = []
        self.smoothing = True

    @staticmethod
    def new(): return Scene()

    def set_perspective_camera(self, fov):
        self.camera = Camera(True, fov)
        return self.camera

    def set_orthographic_camera(self, height):
        self.camera = Camera(False, height)
        return self.camera

    def set_solid_background(self, c): self.bg_inner = self.bg_outer = tuple(map(float, c)); self.bg_scale = 1.0
    def set_radial_background(self, a, b, s): self.bg_inner, self.bg_outer, self.bg_scale = tuple(map(float, a)), tuple(map(float, b)), float(s)
    def set_ambient_light(self, c): self.ambient = tuple(map(float, c))
    def set_max_recursion_depth(self, d): self.recursion = int(d)
    def add_point_light(self, pos, intensity, falloff): self.lights.append((tuple(map(float, pos)), tuple(map(float, intensity)), tuple(map(float, falloff))))
    def set_root(self, agg): self.root = agg
    def set_mesh_smoothing(self, on): self.smoothing = bool(on)

    def parse_obj(self, text):  # scene.rs:109-123: normals are dropped at add time when smoothing is off
        obj = ObjData(text)
        if not self.smoothing:
            obj.normal = []
        return obj


class Api:
    Scene, Aggregate, Material = Scene, Aggregate, Material


# ---- OBJ meshes and triangles (shape/triangle.rs:39-307, scene.rs:109-123; obj 0.10's data model) -----------
def parse_f32(text):
    """The f32 nearest to the decimal `text` (ties to even), as an f64 -- str::parse::<f32>() followed by `.into()`.
    (float(text) rounded to f32 would round twice.)"""
    import fractions
    import struct
    exact = fractions.Fraction(text)
    guess = struct.unpack("f", struct.pack("f", float(text)))[0]
    bits = struct.unpack("I", struct.pack("f", guess))[0]
    best = None
    for b in (bits - 1, bits, bits + 1):
        if b < 0:
            continue
        c = struct.unpack("f", struct.pack("I", b & 0xFFFFFFFF))[0]
        if c != c or c in (INF, -INF):
            continue
        err = abs(fractions.Fraction(c) - exact)
        key = (err, b & 1)
        if best is None or key < best[0]:
            best = (key, c)
    return best[1]


class ObjData:
    """position / texture / normal arrays (f32 widened) and the polygons in file order, each a list of (v, vt, vn) index
    tuples (0-based, None where absent).  Only what triangle.rs reads: the first three tuples of a polygon."""

    def __init__(self, text):
        self.position, self.texture, self.normal, self.polys = [], [], [], []
        for line in text.splitlines():
            f = line.split("#", 1)[0].split()
            if not f:
                continue
            if f[0] == "v":
                self.position.append(tuple(parse_f32(x) for x in f[1:4]))
            elif f[0] == "vt":
                self.texture.append(tuple(parse_f32(x) for x in f[1:3]))
            elif f[0] == "vn":
                self.normal.append(tuple(parse_f32(x) for x in f[1:4]))
            elif f[0] == "f":
                poly = []
                for tok in f[1:]:
                    parts = tok.split("/")
                    idx = []
                    for k, arr in enumerate((self.position, self.texture, self.normal)):
                        if k < len(parts) and parts[k] != "":
                            i = int(parts[k])
                            idx.append(i - 1 if i > 0 else len(arr) + i)
                        else:
                            idx.append(None)
                    poly.append(tuple(idx))
                self.polys.append(poly)


def max_dimension(v):  # space/mod.rs:33-36
    if v[0] > v[1]:
        return 0 if v[0] > v[2] else 2
    return 1 if v[1] > v[2] else 2


def coordinate_system(v1):  # space/mod.rs:39-47
    if abs(v1[0]) > abs(v1[1]):
        v2 = div((-v1[2], 0.0, v1[0]), math.sqrt(v1[0] * v1[0] + v1[2] * v1[2]))
    else:
        v2 = div((0.0, v1[2], -v1[1]), math.sqrt(v1[1] * v1[1] + v1[2] * v1[2]))
    return v2, cross(v1, v2)


def triangle_isect(obj, poly, o, d, best_t):  # Triangle::intersect, triangle.rs:161-304 -> isect fields or None
    p0, p1, p2 = obj.position[poly[0][0]], obj.position[poly[1][0]], obj.position[poly[2][0]]
    p0t, p1t, p2t = sub(p0, o), sub(p1, o), sub(p2, o)
    kz = max_dimension((abs(d[0]), abs(d[1]), abs(d[2])))
    kx = (kz + 1) % 3
    ky = (kx + 1) % 3
    dd = (d[kx], d[ky], d[kz])
    p0t, p1t, p2t = [p0t[kx], p0t[ky], p0t[kz]], [p1t[kx], p1t[ky], p1t[kz]], [p2t[kx], p2t[ky], p2t[kz]]
    sx, sy, sz = _div(-dd[0], dd[2]), _div(-dd[1], dd[2]), _div(1.0, dd[2])
    for q in (p0t, p1t, p2t):
        q[0] += sx * q[2]
        q[1] += sy * q[2]
    e0 = p1t[0] * p2t[1] - p1t[1] * p2t[0]
    e1 = p2t[0] * p0t[1] - p2t[1] * p0t[0]
    e2 = p0t[0] * p1t[1] - p0t[1] * p1t[0]
    if (e0 < 0.0 or e1 < 0.0 or e2 < 0.0) and (e0 > 0.0 or e1 > 0.0 or e2 > 0.0):
        return None
    det = e0 + e1 + e2
    if det == 0.0:
        return None
    p0t[2] *= sz
    p1t[2] *= sz
    p2t[2] *= sz
    tscaled = e0 * p0t[2] + e1 * p1t[2] + e2 * p2t[2]
    if (det < 0.0 and tscaled >= 0.0) or (det > 0.0 and tscaled <= 0.0):
        return None
    invdet = 1.0 / det
    b0, b1, b2 = e0 * invdet, e1 * invdet, e2 * invdet
    t = tscaled * invdet
    if t >= best_t:
        return None
    if obj.texture:
        uv = [obj.texture[poly[i][1]] for i in range(3)]
    else:
        uv = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0)]
    duv02 = (uv[0][0] - uv[2][0], uv[0][1] - uv[2][1])
    duv12 = (uv[1][0] - uv[2][0], uv[1][1] - uv[2][1])
    dp02, dp12 = sub(p0, p2), sub(p1, p2)
    determinant = (duv02[0] * duv12[1]) - (duv02[1] * duv12[0])
    if determinant == 0.0:
        dpdu, dpdv = coordinate_system(cross(sub(p2, p1), sub(p1, p0)))
    else:
        inv = 1.0 / determinant
        dpdu = mul(sub(smul(duv12[1], dp02), smul(duv02[1], dp12)), inv)
        dpdv = mul(sub(smul(-duv12[0], dp02), smul(duv02[0], dp12)), inv)
    if obj.normal:
        n0, n1, n2 = obj.normal[poly[0][2]], obj.normal[poly[1][2]], obj.normal[poly[2][2]]
        ns = add(add(smul(b0, n0), smul(b1, n1)), smul(b2, n2))
        ss = dpdu
        ts = cross(ns, ss)
        if dot(ts, ts) > 0.0:
            ss, ts = cross(ts, ns), ts
        else:
            ss, ts = coordinate_system(ns)
        return t, (dpdu, dpdv), (ss, ts), ns
    n = cross(dp02, dp12)
    if dot(n, neg(d)) < 0.0:
        n = neg(n)
    return t, (dpdu, dpdv), (dpdu, dpdv), n


# ---- sphere (shape/sphere.rs:30-123, core/math.rs:7-30) ----------------------------------------
def quad_roots(a, b, c):
    if a == 0.0:
        if b == 0.0:
            return None
        return (-c / b,)
    d = b * b - 4.0 * a * c
    if d < 0.0:
        return None
    q = -(b + signum(b) * math.sqrt(d)) / 2.0
    q_over_a = q / a
    return (q_over_a, q_over_a if q == 0.0 else c / q)


def sphere_t(o, d, cen, rad):
    l = sub(o, cen)
    a = dot(d, d)
    b = 2.0 * dot(d, l)
    c = dot(l, l) - rad * rad
    roots = quad_roots(a, b, c)
    if roots is None:
        return -INF, False
    if len(roots) == 1:
        return roots[0], False
    t0, t1 = fmin(roots[0], roots[1]), fmax(roots[0], roots[1])
    return (t1, True) if t0 < 0.0 else (t0, False)


def sphere_isect(o, d, cen, rad, t, inside):
    p = sub(add(o, mul(d, t)), cen)
    if p[0] == 0.0 and p[1] == 0.0:
        p = (1e-5 * rad, p[1], p[2])
    phi = math.atan2(p[1], p[0])
    if phi < 0.0:
        phi += 2.0 * PI
    theta = math.acos(fmin(fmax(p[2] / rad, -1.0), 1.0))
    dpdu = (-2.0 * PI * p[1], 2.0 * PI * p[0], 0.0)
    sin_phi, cos_phi = sincos(phi)
    dpdv = smul(PI, (p[2] * cos_phi, p[2] * sin_phi, -rad * math.sin(theta)))
    return (dpdu, dpdv) if inside else (dpdv, dpdu)


CUBE_DIFFERENTIALS = (((0.0, 1.0, 0.0), (0.0, 0.0, 1.0)), ((0.0, 0.0, 1.0), (1.0, 0.0, 0.0)), ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0)))  # cuboid.rs:126-130


def cuboid_isect(o, d, dinv, mn, mx, best_t):  # Bounds::intersect, cuboid.rs:55-102 -> (t, dp0, dp1, n) or None
    tnear, tfar = -INF, INF
    near = far = CUBE_DIFFERENTIALS[0]
    for i in range(3):
        dp = CUBE_DIFFERENTIALS[i]
        t1 = (mn[i] - o[i]) * dinv[i]
        t2 = (mx[i] - o[i]) * dinv[i]
        if t1 < t2:
            tmin, tmax, dp0, dp1 = t1, t2, dp[1], dp[0]
        else:
            tmin, tmax, dp0, dp1 = t2, t1, dp[0], dp[1]
        if tmin > tnear:
            near = (dp0, dp1)
        if tmax < tfar:
            far = (dp1, dp0)
        tnear = fmax(tnear, tmin)
        tfar = fmin(tfar, tmax)
    if tnear > tfar or tfar <= 0.0:
        return None
    t, dp = (tfar, far) if tnear <= 0.0 else (tnear, near)
    if t >= best_t:
        return None
    n = cross(dp[0], dp[1])
    if dot(n, neg(d)) < 0.0:
        n = neg(n)
    return t, dp[0], dp[1], n


def intersect(agg, o, d, best_t):
    """BVHAccel::intersect of one aggregate without the BVH (bvh.rs:461-522): every primitive in insertion order, nested
    groups recursively with the running best t.  Returns None or an isect dict in the PARENT's space."""
    tr = agg.transform
    o_l, d_l = transform_point(tr.minv, o), transform_vector(tr.minv, d)
    dinv = (_div(1.0, d_l[0]), _div(1.0, d_l[1]), _div(1.0, d_l[2]))
    hit = None
    for node in agg.contents:
        if node[0] == "sphere":
            _, cen, rad, mat = node
            t, inside = sphere_t(o_l, d_l, cen, rad)
            if t < 0.0 or t >= best_t:
                continue
            dpdu, dpdv = sphere_isect(o_l, d_l, cen, rad, t, inside)
            hit = {"t": t, "g": (dpdu, dpdv), "s": (dpdu, dpdv), "n": None, "mat": mat}
        elif node[0] == "cuboid":
            _, mn, mx, mat = node
            r = cuboid_isect(o_l, d_l, dinv, mn, mx, best_t)
            if r is None:
                continue
            t, dp0, dp1, n = r
            hit = {"t": t, "g": (dp0, dp1), "s": (dp0, dp1), "n": n, "mat": mat}
        elif node[0] == "mesh":
            r = intersect_mesh(node[1], node[2], o_l, d_l, best_t)
            if r is None:
                continue
            hit = r
        else:
            r = intersect(node[1], o_l, d_l, best_t)
            if r is None:
                continue
            hit = r
        best_t = hit["t"]
    if hit is None:
        return None
    # transform_ray_intersection (transform.rs:243-264), then swap_backface (surface.rs:87-99)
    g = (transform_vector(tr.m, hit["g"][0]), transform_vector(tr.m, hit["g"][1]))
    if hit["g"] != hit["s"]:
        sfc = (transform_vector(tr.m, hit["s"][0]), transform_vector(tr.m, hit["s"][1]))
    else:
        sfc = g
    n = None if hit["n"] is None else transform_normal(tr.minv, hit["n"])
    if agg.swap:
        g, sfc = (g[1], g[0]), (sfc[1], sfc[0])
        n = None if n is None else neg(n)
    return {"t": hit["t"], "g": g, "s": sfc, "n": n, "mat": hit["mat"]}


DEFAULT_MATERIAL = Material.matte([0.5, 0.5, 0.5], 0.0)  # material/mod.rs:15-17


def intersect_mesh(obj, mat, o, d, best_t):
    """BVHAccel::from_mesh (bvh.rs:141-147): an accel of its own with the identity transform, the mesh's material (or the
    default) as the accel's default material; triangles in file order."""
    ident = mat_identity()
    o_l, d_l = transform_point(ident, o), transform_vector(ident, d)
    hit = None
    for poly in obj.polys:
        r = triangle_isect(obj, poly, o_l, d_l, best_t)
        if r is None:
            continue
        hit = r
        best_t = r[0]
    if hit is None:
        return None
    t, g, sfc, n = hit
    g2 = (transform_vector(ident, g[0]), transform_vector(ident, g[1]))
    s2 = (transform_vector(ident, sfc[0]), transform_vector(ident, sfc[1])) if g != sfc else g2
    return {"t": t, "g": g2, "s": s2, "n": transform_normal(ident, n), "mat": mat if mat is not None else DEFAULT_MATERIAL}


def closest(scene, o, d):
    if getattr(scene, "_closest", None) is not None:  # tests/pyref_bvh.py: through the reference's BVH instead of by brute force
        return scene._closest(scene, o, d)
    return intersect(scene.root, o, d, INF)


# ---- BxDF utilities (core/bxdf/mod.rs:237-288) --------------------------------------------------
def cos2_theta(w): return w[2] * w[2]
def sin2_theta(w): return fmax(1.0 - cos2_theta(w), 0.0)
def sin_theta(w): return math.sqrt(sin2_theta(w))
def tan_theta(w): return _div(sin_theta(w), w[2])
def tan2_theta(w): return _div(sin2_theta(w), cos2_theta(w))


def _div(a, b):  # IEEE division (Python raises on a zero divisor)
    if b == 0.0:
        if a != a or a == 0.0:
            return float("nan")
        return math.copysign(INF, a) * math.copysign(1.0, b)
    return a / b


def cos_phi(w):
    s = sin_theta(w)
    return 1.0 if s == 0.0 else fmin(fmax(w[0] / s, -1.0), 1.0)


def sin_phi(w):
    s = sin_theta(w)
    return 0.0 if s == 0.0 else fmin(fmax(w[1] / s, -1.0), 1.0)


def reflect(wo, n):  # -1.0 * wo + 2.0 * wo.dot(n) * n
    return add(smul(-1.0, wo), smul(2.0 * dot(wo, n), n))


def refract(wi, n, eta):
    cos_i = dot(n, wi)
    sin2_i = fmax(1.0 - cos_i * cos_i, 0.0)
    sin2_t = eta * eta * sin2_i
    if sin2_t >= 1.0:
        return None
    cos_t = math.sqrt(1.0 - sin2_t)
    return add(smul(eta * -1.0, wi), smul(eta * cos_i - cos_t, n))


# ---- Fresnel (core/bxdf/fresnel.rs:37-91) --------------------------------------------------------
def fr_dielectric(cos_i, eta_i, eta_t):
    cos_i = fmin(fmax(cos_i, -1.0), 1.0)
    if not cos_i > 0.0:
        eta_i, eta_t = eta_t, eta_i
        cos_i = abs(cos_i)
    sin_i = math.sqrt(fmax(1.0 - cos_i * cos_i, 0.0))
    sin_t = eta_i / eta_t * sin_i
    if sin_t >= 1.0:
        return 1.0
    cos_t = math.sqrt(fmax(1.0 - sin_t * sin_t, 0.0))
    r_parl = _div((eta_t * cos_i) - (eta_i * cos_t), (eta_t * cos_i) + (eta_i * cos_t))
    r_perp = _div((eta_i * cos_i) - (eta_t * cos_t), (eta_i * cos_i) + (eta_t * cos_t))
    return (r_parl * r_parl + r_perp * r_perp) * 0.5


def fr_conductor(cos_i, eta_i, eta_t, k):
    cos_i = fmin(fmax(cos_i, -1.0), 1.0)
    out = []
    for c in range(3):
        eta = eta_t[c] / eta_i[c]
        etak = k[c] / eta_i[c]
        cos2 = cos_i * cos_i
        sin2 = 1.0 - cos2
        eta2, etak2 = eta * eta, etak * etak
        t0 = eta2 - etak2 - sin2
        a2plusb2 = math.sqrt(t0 * t0 + 4.0 * (eta2 * etak2))
        t1 = a2plusb2 + cos2
        a = math.sqrt(0.5 * (a2plusb2 + t0))
        t2 = 2.0 * cos_i * a
        rs = _div(t1 - t2, t1 + t2)
        t3 = cos2 * a2plusb2 + sin2 * sin2
        t4 = t2 * sin2
        rp = _div(rs * (t3 - t4), t3 + t4)
        out.append(0.5 * (rp + rs))
    return tuple(out)


def substance_eval(sub_, cos_i):
    if sub_[0] == "dielectric":
        v = fr_dielectric(cos_i, sub_[1], sub_[2])
        return (v, v, v)
    if sub_[0] == "conductor":
        return fr_conductor(cos_i, sub_[1], sub_[2], sub_[3])
    return (1.0, 1.0, 1.0)


# ---- BxDFs: ("lambert", r) ("oren", r, a, b) ("micro", r, substance, ax, ay) ("srefl", r, substance) ("strans", t, ea, eb)
REFLECTION, TRANSMISSION, SPECULAR = 1, 2, 16
BXDF_TYPE = {"lambert": 1 | 4, "oren": 1 | 4, "micro": 1 | 8, "srefl": 1 | 16, "strans": 2 | 16}


def oren_new(r, sigma):  # diffuse.rs:29-35
    sigma = sigma * (PI / 180.0)  # Rad::from(Deg(sigma))
    s2 = sigma * sigma
    a = 1.0 - (s2 / 2.0 * (s2 + 0.33))
    b = 0.45 * s2 / (s2 + 0.09)
    return ("oren", r, a, b)


def tr_d(ax, ay, wh):  # microfacet.rs:31-40
    t2 = tan2_theta(wh)
    if math.isinf(t2):
        return 0.0
    cos4 = cos2_theta(wh) * cos2_theta(wh)
    e = ((cos_phi(wh) * cos_phi(wh)) / (ax * ax) + (sin_phi(wh) * sin_phi(wh)) / (ay * ay)) * t2
    return 1.0 / (PI * ax * ay * cos4 * (1.0 + e) * (1.0 + e))


def tr_lambda(ax, ay, w):  # microfacet.rs:55-66
    att = abs(tan_theta(w))
    if math.isinf(att):
        return 0.0
    alpha = math.sqrt((cos_phi(w) * cos_phi(w)) * ax * ax + (sin_phi(w) * sin_phi(w)) * ay * ay)
    a2t2 = (alpha * att) * (alpha * att)
    return (math.sqrt(1.0 + a2t2) - 1.0) / 2.0


def bxdf_f(b, wo, wi):  # bxdf/mod.rs:152-162
    k = b[0]
    if k == "lambert":
        return mul(b[1], FRAC_1_PI)
    if k == "oren":  # diffuse.rs:37-56
        _, r, a, bb = b
        si, so = sin_theta(wi), sin_theta(wo)
        if si > 1e-4 and so > 1e-4:
            d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo)
            max_cos = fmax(d_cos, 0.0)
        else:
            max_cos = 0.0
        if abs(wi[2]) > abs(wo[2]):
            sin_alpha, tan_beta = so, si / abs(wi[2])
        else:
            sin_alpha, tan_beta = si, _div(so, abs(wo[2]))
        return mul(mul(r, FRAC_1_PI), a + bb * max_cos * sin_alpha * tan_beta)
    if k == "micro":  # microfacet.rs:101-115
        _, r, sub_, ax, ay = b
        cos_o, cos_i = abs(wo[2]), abs(wi[2])
        wh = add(wi, wo)
        if cos_i == 0.0 or cos_o == 0.0:
            return ZERO
        if wh[0] == 0.0 and wh[1] == 0.0 and wh[2] == 0.0:
            return ZERO
        wh = normalize(wh)
        spectrum = substance_eval(sub_, dot(wi, wh))
        g = 1.0 / (1.0 + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi))
        return div(mulv(mul(mul(r, tr_d(ax, ay, wh)), g), spectrum), 4.0 * cos_i * cos_o)
    return ZERO  # specular BxDFs scatter only through sample_f


def bxdf_sample_f(b, wo):  # specular.rs:17-24, 43-63 (the only BxDFs the integrator samples)
    if b[0] == "srefl":
        wi = (-wo[0], -wo[1], wo[2])
        spectrum = div(mulv(substance_eval(b[2], wi[2]), b[1]), abs(wi[2]))
        return spectrum, wi, 1.0
    _, t, eta_a, eta_b = b
    entering = wo[2] > 0.0
    eta_i, eta_t = (eta_a, eta_b) if entering else (eta_b, eta_a)
    wi = refract(wo, (0.0, 0.0, 1.0), eta_i / eta_t)
    if wi is None:
        return ZERO, ZERO, 0.0
    fr = substance_eval(("dielectric", eta_a, eta_b), wi[2])
    spectrum = div(mulv(t, sub((1.0, 1.0, 1.0), fr)), abs(wi[2]))
    return spectrum, wi, 1.0


def scattering(mat):  # material/*.rs
    k, p = mat.kind, mat.p
    if k == "matte":
        return [("lambert", p[0])] if p[1] == 0.0 else [oren_new(p[0], p[1])]
    if k == "plastic":
        out = []
        if p[0] != ZERO:
            out.append(("lambert", p[0]))
        if p[1] != ZERO:
            out.append(("micro", p[1], ("dielectric", 1.0, 1.5), p[2], p[2]))
        return out
    if k == "metal":
        white = (1.0, 1.0, 1.0)
        return [("micro", white, ("conductor", white, p[0], p[1]), p[2], p[3])]
    if k == "mirror":
        return [("srefl", p[0], ("noop",))]
    out = []  # glass with roughness 0 (material/mod.rs:39-40)
    if p[0] != ZERO:
        out.append(("srefl", p[0], ("dielectric", 1.0, p[2])))
    if p[1] != ZERO:
        out.append(("strans", p[1], 1.0, p[2]))
    return out


class BSDF:  # interaction/bsdf.rs
    def __init__(self, ng, ns, ss, bxdfs):
        self.ng, self.ns, self.ss, self.ts, self.bxdfs = ng, ns, ss, cross(ns, ss), bxdfs

    def to_local(self, v): return (dot(v, self.ss), dot(v, self.ts), dot(v, self.ns))

    def to_world(self, v):
        s, t, n = self.ss, self.ts, self.ns
        return (s[0] * v[0] + t[0] * v[1] + n[0] * v[2], s[1] * v[0] + t[1] * v[1] + n[1] * v[2], s[2] * v[0] + t[2] * v[1] + n[2] * v[2])

    def f(self, wo, wi):  # bsdf.rs:73-92
        reflect_ = dot(wi, self.ng) * dot(wo, self.ng) > 0.0
        wo_l, wi_l = self.to_local(wo), self.to_local(wi)
        if wo_l[2] == 0.0:
            return ZERO
        f = ZERO
        for b in self.bxdfs:
            t = BXDF_TYPE[b[0]]
            if (reflect_ and (t & REFLECTION)) or (not reflect_ and (t & TRANSMISSION)):
                f = add(f, bxdf_f(b, wo_l, wi_l))
        return f

    def sample_f(self, wo, flags):  # bsdf.rs:94-145 with the integrator's fixed sample (0.5, 0.5)
        match = [b for b in self.bxdfs if (BXDF_TYPE[b[0]] & flags) == BXDF_TYPE[b[0]]]
        if not match:
            return ZERO, ZERO, 0.0
        comp = min(int(math.floor(0.5 * float(len(match)))), len(match) - 1)
        b = match[comp]
        wo_l = self.to_local(wo)
        if wo_l[2] == 0.0:
            return ZERO, ZERO, 0.0
        spectrum, wi_l, pdf = bxdf_sample_f(b, wo_l)
        if pdf == 0.0:
            return spectrum, wi_l, pdf
        wi = self.to_world(wi_l)
        spectrum = tuple(fmin(fmax(c, 0.0), 1.0) for c in spectrum)  # (every sampled BxDF is specular)
        return spectrum, wi, pdf / float(len(match))


# ---- integrator (integrate/integrate.rs:23-132, interaction/surface.rs:158-183, light/point.rs:42-54) ----
def background(scene, d):  # material/background.rs:25-34
    a = abs(dot((0.0, 0.0, 1.0), d))
    t = fmin(math.sqrt(1.0 - a * a) / scene.bg_scale, 1.0)  # powf(2.) is x * x (LLVM folds it)
    return tuple(scene.bg_inner[i] * (1.0 - t) + scene.bg_outer[i] * t for i in range(3))


def li(scene, o, d, depth):
    hit = closest(scene, o, d)
    if hit is None:
        return background(scene, normalize(d))
    t, mat = hit["t"], hit["mat"]
    wo = neg(normalize(d))
    ng = normalize(cross(hit["g"][0], hit["g"][1]))  # isect.ng() face-forwarded to wo (surface.rs:161-162)
    if dot(ng, wo) < 0.0:
        ng = neg(ng)
    # isect.ns(): the authoritative normal if the shape set one, else from the surface shading; NOT face-forwarded (surface.rs:163)
    ns = normalize(hit["n"]) if hit["n"] is not None else normalize(cross(hit["s"][0], hit["s"][1]))
    dpdu = hit["s"][0]
    err = 2.220446049250313e-16 * 2.0 ** 16
    p0 = add(o, mul(d, t))
    p_err = mul(ng, err)
    ss = normalize(dpdu)
    bsdf = BSDF(ng, ns, ss, scattering(mat))
    n = ns
    p = add(p0, p_err)
    output = ZERO
    for lpos, lint, fall in scene.lights:
        sd = sub(lpos, p)
        occ = closest(scene, p, sd)  # full closest hit, occluded iff t < 1 (point.rs:47-49)
        if occ is not None and occ["t"] < 1.0:
            continue
        wi = sub(lpos, p)
        dist = magnitude(wi)
        f_att = fall[0] + fall[1] * dist + fall[2] * dist * dist
        if f_att == 0.0:
            continue
        wi = normalize(wi)
        wi_dot_n = dot(wi, n)
        f = bsdf.f(wo, wi)
        output = add(output, div(mul(mulv(smul(PI, lint), f), wi_dot_n), f_att))
    output = add(output, mulv(scene.ambient, bsdf.f(wo, n)))
    refracted = reflected = ZERO
    if depth < scene.recursion:
        # specular_transmit (integrate.rs:108-132)
        spectrum, wi, pdf = bsdf.sample_f(wo, TRANSMISSION | SPECULAR)
        if not (pdf <= 0.0 or spectrum == ZERO or abs(dot(wi, ns)) == 0.0):
            sub_li = li(scene, sub(p0, p_err), wi, depth + 1)
            refracted = div(mul(mulv(spectrum, sub_li), abs(dot(wi, ns))), pdf)
        # specular_reflect (integrate.rs:82-106)
        spectrum, wi, pdf = bsdf.sample_f(wo, REFLECTION | SPECULAR)
        if not (pdf <= 0.0 or spectrum == ZERO or dot(wi, ns) <= 0.0):
            wr = reflect(wo, ns)
            reflected = mulv(spectrum, li(scene, add(p0, p_err), wr, depth + 1))
    return add(add(output, reflected), refracted)


def to_byte(c):  # img.rs:65-67
    v = fmin(fmax(c, 0.0), 1.0) * 255.0
    return int(math.floor(v + 0.5)) if v == v else 0  # round half away from zero (v >= 0); NaN as u8 = 0


def render(scene, w, h):
    """(radiance [h][w] of 3-tuples, rgba bytes [h][w] of 4-tuples) -- lib.rs:110-162, integrate.rs:16-20"""
    rad, rgba = [], []
    for y in range(h):
        rrow, brow = [], []
        for x in range(w):
            rays = scene.camera.sample(x, y, w, h)
            weight = 1.0 / float(len(rays))
            c = ZERO
            for o, d in rays:
                c = add(c, li(scene, o, d, 0))
            c = mul(c, weight)
            rrow.append(c)
            brow.append((to_byte(c[0]), to_byte(c[1]), to_byte(c[2]), 255))
        rad.append(rrow)
        rgba.append(brow)
    return rad, rgba
