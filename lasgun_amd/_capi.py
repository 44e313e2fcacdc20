"""ctypes binding over a lasgun C ABI (include/lasgun_hip.h).

Host-side mirror of the reference's public surface for the render path, with the
reference's own names and argument meaning:

  Scene      /root/reference/src/scene.rs:49-143
  Aggregate  /root/reference/src/scene/node.rs:35-115
  Material   /root/reference/src/material/mod.rs:15-46
  Camera     /root/reference/src/camera.rs:75-102
  Film       /root/reference/src/film.rs:22-45
  Accel / capture / capture_subset / render   /root/reference/src/lib.rs:42-56,110

The binding is parametrised by (shared library, symbol prefix): every function
`<prefix>name` must have the signature declared in include/lasgun_hip.h.  The
product instantiates it once, for liblasgun_hip.so with prefix "lg_"
(lasgun_amd/__init__.py).  Nothing here computes anything: every call goes
straight to the C ABI, and a missing library is an ImportError, never a fallback.
"""
import ctypes as C
import numpy as np

_D3 = C.c_double * 3


class CMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_double * 10)]


class CStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "shadow_rays", "secondary_rays", "nodes_tested", "spheres_tested",
        "cuboids_tested", "triangles_tested", "accel_entries", "hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class LasgunError(RuntimeError):
    pass


class ObjError(LasgunError):
    """Mirrors obj::ObjError: the one `Result` on the reference's path (scene.rs:120-130)."""


def _v3(v):
    v = list(v)
    if len(v) != 3:
        raise ValueError("expected 3 components")
    return _D3(float(v[0]), float(v[1]), float(v[2]))


# name -> (restype, argtypes).  These are exactly the entry points of include/lasgun_hip.h
# that have a counterpart in the reference's API (the core ABI).
CORE_SIGNATURES = {
    "last_error": (C.c_char_p, []),
    "material_default": (CMaterial, []),
    "material_matte": (CMaterial, [_D3, C.c_double]),
    "material_plastic": (CMaterial, [_D3, _D3, C.c_double]),
    "material_metal": (CMaterial, [_D3, _D3, C.c_double, C.c_double]),
    "material_glass": (CMaterial, [_D3, _D3, C.c_double]),
    "material_mirror": (CMaterial, [_D3]),
    "scene_new": (C.c_void_p, []),
    "scene_free": (None, [C.c_void_p]),
    "scene_set_perspective_camera": (None, [C.c_void_p, C.c_double]),
    "scene_set_orthographic_camera": (None, [C.c_void_p, C.c_double]),
    "camera_look_at": (None, [C.c_void_p, _D3, _D3, _D3]),
    "camera_set_supersampling": (None, [C.c_void_p, C.c_uint8]),
    "camera_set_aperture_radius": (None, [C.c_void_p, C.c_double]),
    "scene_set_solid_background": (None, [C.c_void_p, _D3]),
    "scene_set_radial_background": (None, [C.c_void_p, _D3, _D3, C.c_double]),
    "scene_set_ambient_light": (None, [C.c_void_p, _D3]),
    "scene_set_mesh_smoothing": (None, [C.c_void_p, C.c_int]),
    "scene_set_max_recursion_depth": (None, [C.c_void_p, C.c_uint32]),
    "scene_set_threads": (None, [C.c_void_p, C.c_size_t]),
    "scene_add_point_light": (None, [C.c_void_p, _D3, _D3, _D3]),
    "scene_parse_obj": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "scene_load_obj": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "scene_root": (C.c_void_p, [C.c_void_p]),
    "scene_set_root": (None, [C.c_void_p, C.c_void_p]),
    "aggregate_new": (C.c_void_p, []),
    "aggregate_free": (None, [C.c_void_p]),
    "aggregate_add_group": (None, [C.c_void_p, C.c_void_p]),
    "aggregate_add_sphere": (None, [C.c_void_p, _D3, C.c_double, C.POINTER(CMaterial)]),
    "aggregate_add_cube": (None, [C.c_void_p, _D3, C.c_double, C.POINTER(CMaterial)]),
    "aggregate_add_box": (None, [C.c_void_p, _D3, _D3, C.POINTER(CMaterial)]),
    "aggregate_add_obj": (None, [C.c_void_p, C.c_uint32]),
    "aggregate_add_obj_of": (None, [C.c_void_p, C.c_uint32, C.POINTER(CMaterial)]),
    "aggregate_swap_backface": (None, [C.c_void_p]),
    "aggregate_translate": (None, [C.c_void_p, _D3]),
    "aggregate_scale": (None, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "aggregate_rotate_x": (None, [C.c_void_p, C.c_double]),
    "aggregate_rotate_y": (None, [C.c_void_p, C.c_double]),
    "aggregate_rotate_z": (None, [C.c_void_p, C.c_double]),
    "aggregate_rotate": (None, [C.c_void_p, C.c_double, _D3]),
    "aggregate_get_transform": (None, [C.c_void_p, C.c_double * 16, C.c_double * 16]),
    "film_new": (C.c_void_p, [C.c_uint32, C.c_uint32]),
    "film_wrap": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_void_p]),
    "film_pixels": (C.c_void_p, [C.c_void_p]),
    "film_width": (C.c_uint32, [C.c_void_p]),
    "film_height": (C.c_uint32, [C.c_void_p]),
    "film_free": (None, [C.c_void_p]),
    "accel_from": (C.c_void_p, [C.c_void_p]),
    "accel_free": (None, [C.c_void_p]),
    "capture": (C.c_int, [C.c_void_p, C.c_void_p]),
    "capture_subset": (C.c_int, [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "render": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32]),
    # build-only extras shared by both sides (no reference counterpart)
    "capture_radiance": None,  # signature differs per side; bound by the owner
    "accel_dump": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t),
                             C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_size_t)]),
    "kat_intersect": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_size_t, _D3, _D3, C.c_double * 8]),
    "kat_surface_interaction": (C.c_int, [_D3, _D3, C.c_double, _D3, _D3, _D3]),
    "math_eval": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
}


class Api:
    """One bound C ABI: `Api(ctypes.CDLL(path), "lg_")`."""

    def __init__(self, lib, prefix, extra_signatures=None):
        self.lib = lib
        self.prefix = prefix
        sigs = dict(CORE_SIGNATURES)
        if extra_signatures:
            sigs.update(extra_signatures)
        self._fn = {}
        for name, sig in sigs.items():
            if sig is None:
                continue
            fn = getattr(lib, prefix + name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype, fn.argtypes = sig
            self._fn[name] = fn
        api = self

        # ------------------------------------------------------------------
        class Material:
            """material/mod.rs:15-46 -- a Copy POD (tag + parameters)."""

            def __init__(self, c):
                self.c = c

            @staticmethod
            def default():
                return Material(api.call("material_default"))

            @staticmethod
            def matte(kd, sigma):
                return Material(api.call("material_matte", _v3(kd), float(sigma)))

            @staticmethod
            def plastic(kd, ks, roughness):
                return Material(api.call("material_plastic", _v3(kd), _v3(ks), float(roughness)))

            @staticmethod
            def metal(eta, k, u_roughness, v_roughness):
                return Material(api.call("material_metal", _v3(eta), _v3(k), float(u_roughness), float(v_roughness)))

            @staticmethod
            def glass(kr, kt, eta):
                return Material(api.call("material_glass", _v3(kr), _v3(kt), float(eta)))

            @staticmethod
            def mirror(kr):
                return Material(api.call("material_mirror", _v3(kr)))

        # ------------------------------------------------------------------
        class Aggregate:
            """scene/node.rs:35-115.  add_group MOVES the child (as in Rust)."""

            def __init__(self, handle=None, owned=True):
                self.h = handle if handle is not None else api.call("aggregate_new")
                self.owned = owned

            @staticmethod
            def new():
                return Aggregate()

            def _alive(self):
                if self.h is None:
                    raise LasgunError("use of a moved Aggregate")
                return self.h

            def __del__(self):
                if getattr(self, "owned", False) and self.h is not None:
                    api.call("aggregate_free", self.h)
                    self.h = None

            def add_group(self, aggregate):
                child = aggregate._alive()
                if not aggregate.owned:
                    raise LasgunError("cannot move a borrowed Aggregate")
                api.call("aggregate_add_group", self._alive(), child)
                aggregate.h = None

            def add_sphere(self, center, radius, material):
                api.call("aggregate_add_sphere", self._alive(), _v3(center), float(radius), C.byref(material.c))

            def add_cube(self, origin, dim, material):
                api.call("aggregate_add_cube", self._alive(), _v3(origin), float(dim), C.byref(material.c))

            def add_box(self, minbound, maxbound, material):
                api.call("aggregate_add_box", self._alive(), _v3(minbound), _v3(maxbound), C.byref(material.c))

            def add_obj(self, mesh):
                api.call("aggregate_add_obj", self._alive(), int(mesh))

            def add_obj_of(self, mesh, material):
                api.call("aggregate_add_obj_of", self._alive(), int(mesh), C.byref(material.c))

            def swap_backface(self):
                api.call("aggregate_swap_backface", self._alive())

            def translate(self, delta):
                api.call("aggregate_translate", self._alive(), _v3(delta)); return self

            def scale(self, x, y, z):
                api.call("aggregate_scale", self._alive(), float(x), float(y), float(z)); return self

            def rotate_x(self, theta):
                api.call("aggregate_rotate_x", self._alive(), float(theta)); return self

            def rotate_y(self, theta):
                api.call("aggregate_rotate_y", self._alive(), float(theta)); return self

            def rotate_z(self, theta):
                api.call("aggregate_rotate_z", self._alive(), float(theta)); return self

            def rotate(self, theta, axis):
                api.call("aggregate_rotate", self._alive(), float(theta), _v3(axis)); return self

            def transform(self):
                """(m, minv) as 4x4 column-major numpy arrays [col][row] (test hook)."""
                m = (C.c_double * 16)(); mi = (C.c_double * 16)()
                api.call("aggregate_get_transform", self._alive(), m, mi)
                return np.array(m).reshape(4, 4), np.array(mi).reshape(4, 4)

        # ------------------------------------------------------------------
        class Camera:
            """camera.rs:75-102; a view of scene.camera (Rust hands out &mut Camera)."""

            def __init__(self, scene):
                self.scene = scene

            def look_at(self, origin, look, up):
                api.call("camera_look_at", self.scene.h, _v3(origin), _v3(look), _v3(up))

            def set_supersampling(self, base):
                api.call("camera_set_supersampling", self.scene.h, int(base))

            def set_aperture_radius(self, radius):
                api.call("camera_set_aperture_radius", self.scene.h, float(radius))

        # ------------------------------------------------------------------
        class Scene:
            """scene.rs:49-143."""

            def __init__(self):
                self.h = api.call("scene_new")

            @staticmethod
            def new():
                return Scene()

            def __del__(self):
                if getattr(self, "h", None) is not None:
                    api.call("scene_free", self.h)
                    self.h = None

            @property
            def root(self):
                return Aggregate(api.call("scene_root", self.h), owned=False)

            @property
            def camera(self):
                return Camera(self)

            def set_perspective_camera(self, fov):
                api.call("scene_set_perspective_camera", self.h, float(fov)); return Camera(self)

            def set_orthographic_camera(self, scale):
                api.call("scene_set_orthographic_camera", self.h, float(scale)); return Camera(self)

            def set_solid_background(self, color):
                api.call("scene_set_solid_background", self.h, _v3(color))

            def set_radial_background(self, inner, outer, scale):
                api.call("scene_set_radial_background", self.h, _v3(inner), _v3(outer), float(scale))

            def set_ambient_light(self, color):
                api.call("scene_set_ambient_light", self.h, _v3(color))

            def set_mesh_smoothing(self, enabled):
                api.call("scene_set_mesh_smoothing", self.h, 1 if enabled else 0)

            def set_max_recursion_depth(self, max_depth):
                api.call("scene_set_max_recursion_depth", self.h, int(max_depth))

            def set_threads(self, threads):
                api.call("scene_set_threads", self.h, int(threads))

            def add_point_light(self, position, intensity, falloff):
                api.call("scene_add_point_light", self.h, _v3(position), _v3(intensity), _v3(falloff))

            def parse_obj(self, text):
                data = text.encode() if isinstance(text, str) else bytes(text)
                ref = C.c_uint32()
                if api.call("scene_parse_obj", self.h, data, len(data), C.byref(ref)):
                    raise ObjError(api.last_error())
                return ref.value

            def load_obj(self, path):
                ref = C.c_uint32()
                if api.call("scene_load_obj", self.h, str(path).encode(), C.byref(ref)):
                    raise ObjError(api.last_error())
                return ref.value

            def set_root(self, node):
                if not node.owned:
                    raise LasgunError("cannot move a borrowed Aggregate")
                api.call("scene_set_root", self.h, node._alive())
                node.h = None

        # ------------------------------------------------------------------
        class Film:
            """film.rs:22-45: w*h RGBA8, row-major, top-left origin, zero-filled."""

            def __init__(self, width, height, handle=None, keep=None):
                self.h = handle if handle is not None else api.call("film_new", int(width), int(height))
                if not self.h:
                    raise LasgunError(api.last_error())
                self.w, self.h_px = int(width), int(height)
                self._keep = keep

            @staticmethod
            def new(width, height):
                return Film(width, height)

            @staticmethod
            def new_with_output(width, height, array):
                """Wrap an externally owned (h, w, 4) uint8 buffer (film.rs:36)."""
                a = np.ascontiguousarray(array, dtype=np.uint8)
                if a.size != width * height * 4 or a is not array and not np.shares_memory(a, array):
                    raise ValueError("need a C-contiguous uint8 buffer of w*h*4 bytes")
                h = api.call("film_wrap", int(width), int(height), a.ctypes.data)
                return Film(width, height, handle=h, keep=a)

            def __del__(self):
                if getattr(self, "h", None):
                    api.call("film_free", self.h)
                    self.h = None

            def pixels(self):
                """(h, w, 4) uint8 numpy copy of the film's buffer."""
                ptr = api.call("film_pixels", self.h)
                buf = (C.c_uint8 * (self.w * self.h_px * 4)).from_address(ptr)
                return np.frombuffer(buf, dtype=np.uint8).reshape(self.h_px, self.w, 4).copy()

        # ------------------------------------------------------------------
        class Accel:
            """lib.rs:42 `Accel::from(&scene)`: borrows the scene for its lifetime."""

            def __init__(self, scene):
                self.scene = scene  # keep the borrow alive
                self.h = api.call("accel_from", scene.h)
                if not self.h:
                    raise LasgunError(api.last_error())

            @staticmethod
            def from_scene(scene):
                return Accel(scene)

            def __del__(self):
                if getattr(self, "h", None):
                    api.call("accel_free", self.h)
                    self.h = None

            def dump(self):
                """(floats, ints) flattening of every BVH in the scene graph (build-parity tests)."""
                pf = C.POINTER(C.c_double)(); nf = C.c_size_t(); pi = C.POINTER(C.c_int64)(); ni = C.c_size_t()
                if api.call("accel_dump", self.h, C.byref(pf), C.byref(nf), C.byref(pi), C.byref(ni)):
                    raise LasgunError(api.last_error())
                f = np.ctypeslib.as_array(pf, shape=(nf.value,)).copy() if nf.value else np.zeros(0)
                i = np.ctypeslib.as_array(pi, shape=(ni.value,)).copy() if ni.value else np.zeros(0, dtype=np.int64)
                return f, i

        self.Material, self.Aggregate, self.Camera, self.Scene, self.Film, self.Accel = (
            Material, Aggregate, Camera, Scene, Film, Accel)

    # -- plumbing ----------------------------------------------------------
    def call(self, name, *args):
        return self._fn[name](*args)

    def last_error(self):
        e = self.call("last_error")
        return e.decode() if e else ""

    # -- lib.rs entry points -------------------------------------------------
    def capture(self, scene, film):
        """lib.rs:55 -- synchronous; on return every pixel of `film` is written.  On the HIP library: split over the devices of
        set_devices / set_device; a process that named none gets EVERY visible GPU for films of 2^18 pixels and more (an accel per
        device, one RCCL gather) and the HIP current device for smaller ones (include/lasgun_hip.h, lg_capture)."""
        if self.call("capture", scene.h, film.h):
            raise LasgunError(self.last_error())

    def capture_subset(self, k, n, accel, film):
        """lib.rs:110 -- writes exactly pixel indices {k + i*n < w*h}."""
        if self.call("capture_subset", int(k), int(n), accel.h, film.h):
            raise LasgunError(self.last_error())

    def render(self, scene, resolution):
        """lib.rs:46 -- Film::new + capture."""
        w, h = resolution
        fh = self.call("render", scene.h, int(w), int(h))
        if not fh:
            raise LasgunError(self.last_error())
        return self.Film(w, h, handle=fh)

    # -- known-answer / math hooks ---------------------------------------------
    def kat_intersect(self, kind, params=None, obj_text=None, origin=(0, 0, 0), d=(0, 0, 1)):
        p = (C.c_double * 8)(*([float(v) for v in (params or [])] + [0.0] * (8 - len(params or []))))
        text = (obj_text or "").encode()
        out = (C.c_double * 8)()
        if self.call("kat_intersect", int(kind), p, text, len(text), _v3(origin), _v3(d), out):
            raise LasgunError(self.last_error())
        o = list(out)
        return {"hit": o[0] != 0.0, "t": o[1], "ng": tuple(o[2:5]), "ns": tuple(o[5:8])}

    def kat_surface_interaction(self, origin, d, t, dpdu, dpdv):
        out = _D3()
        if self.call("kat_surface_interaction", _v3(origin), _v3(d), float(t), _v3(dpdu), _v3(dpdv), out):
            raise LasgunError(self.last_error())
        return tuple(out)

    def math_eval(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float64)
        out = np.empty_like(a)
        if self.call("math_eval", int(op), a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data):
            raise LasgunError(self.last_error() or "math_eval failed")
        return out
