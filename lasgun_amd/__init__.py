"""lasgun_amd -- MI355X-native drop-in for nfrasser/lasgun's per-pixel ray-trace path.

The package is a thin host-side mirror of the reference's `Scene` / `Aggregate` / `Material` /
`Film` / `Accel` / `capture` / `capture_subset` / `render` surface (see `_capi.py`) over the C ABI
of `liblasgun_hip.so` (include/lasgun_hip.h), whose render entry points run hand-written HIP
kernels for gfx950.  There is no CPU render path: importing this package without the built
library raises ImportError, and every render call fails without a HIP device.
"""
import ctypes as _C
import os as _os

import numpy as _np

from ._capi import Api, CStats, LasgunError, ObjError  # noqa: F401
from . import scenes  # noqa: F401

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.environ.get("LASGUN_HIP_LIB") or _os.path.join(_HERE, "liblasgun_hip.so")  # env override: A/B builds

if not _os.path.exists(LIB_PATH):
    raise ImportError(
        "lasgun_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C lasgun_amd/csrc` (there is no CPU fallback)" % LIB_PATH)

_EXTRA = {
    "set_device": (_C.c_int, [_C.c_int]),
    "device_count": (_C.c_int, []),
    "trim_pool": (_C.c_uint64, [_C.c_int]),
    "set_devices": (_C.c_int, [_C.POINTER(_C.c_int), _C.c_int]),
    "capture_rows_device": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32,
                                       _C.c_void_p, _C.c_void_p]),
    "capture_subset_device": (_C.c_int, [_C.c_size_t, _C.c_size_t, _C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_void_p,
                                         _C.c_void_p]),
    "capture_subsets": (_C.c_int, [_C.POINTER(_C.c_size_t), _C.c_size_t, _C.c_size_t, _C.c_void_p, _C.c_void_p]),
    "capture_subsets_device": (_C.c_int, [_C.POINTER(_C.c_size_t), _C.c_size_t, _C.c_size_t, _C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_void_p,
                                          _C.c_void_p]),
    "capture_interleaved_device": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32,
                                              _C.c_void_p, _C.c_void_p]),
    "accel_stream": (_C.c_void_p, [_C.c_void_p]),
    "accel_set_mode": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_set_prune": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_set_streaming": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_get_prune": (_C.c_int, [_C.c_void_p]),
    "accel_last_organisation": (_C.c_int, [_C.c_void_p]),
    "accel_set_tile_order": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_set_sample_order": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_set_tile_parts": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_set_lds_scene": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_set_wf_split": (_C.c_int, [_C.c_void_p, _C.c_int]),
    "accel_synchronize": (_C.c_int, [_C.c_void_p]),
    "capture_radiance": (_C.c_int, [_C.c_size_t, _C.c_size_t, _C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_void_p]),
    "capture_pixels": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_void_p, _C.c_size_t, _C.c_void_p, _C.c_void_p]),
    "capture_rect": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_void_p, _C.c_void_p]),
    "audit_prune": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_void_p]),
    "audit_fast": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_void_p]),
    "capture_stats": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.POINTER(CStats)]),
    "profile_enable": (None, [_C.c_void_p, _C.c_int]),
    "profile_read_kinds": (_C.c_int, [_C.c_void_p, _C.c_double * 5, _C.c_uint64 * 5]),
    "capture_stats_kind": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_int, _C.POINTER(CStats)]),
    "profile_read": (_C.c_int, [_C.c_void_p, _C.POINTER(_C.c_double), _C.POINTER(_C.c_uint64)]),
    "probe_rate": (_C.c_int, [_C.c_int, _C.POINTER(_C.c_double)]),
    "accel_from_on": (_C.c_void_p, [_C.c_void_p, _C.c_int]),
    "tune_export": (_C.c_size_t, [_C.c_void_p, _C.c_size_t]),
    "tune_import": (_C.c_int, [_C.c_void_p, _C.c_size_t]),
    "tune_clear": (None, []),
    "multi_create": (_C.c_void_p, [_C.c_void_p, _C.POINTER(_C.c_int), _C.c_int, _C.c_uint32]),
    "multi_free": (None, [_C.c_void_p]),
    "multi_capture_device": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_void_p]),
    "multi_capture": (_C.c_int, [_C.c_void_p, _C.c_void_p]),
    "multi_capture_device_all": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.POINTER(_C.c_void_p)]),
    "multi_rank_count": (_C.c_int, [_C.c_void_p]),
    "multi_accel": (_C.c_void_p, [_C.c_void_p, _C.c_int]),
    "multi_uses_rccl": (_C.c_int, [_C.c_void_p]),
    "accel_info": (_C.c_int, [_C.c_void_p, _C.c_uint64 * 8]),
    "trace_pixel": (_C.c_int, [_C.c_void_p, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_uint32, _C.c_int, _C.POINTER(_C.c_double), _C.c_size_t]),
    "host_build_dump": (_C.c_int, [_C.c_void_p, _C.POINTER(_C.POINTER(_C.c_double)), _C.POINTER(_C.c_size_t),
                                   _C.POINTER(_C.POINTER(_C.c_int64)), _C.POINTER(_C.c_size_t), _C.c_uint64 * 8]),
    "host_check_wide_records": (_C.c_int, [_C.c_void_p, _C.c_uint64 * 8]),
    "host_check_strips": (_C.c_int, [_C.c_void_p, _C.c_uint64 * 8]),
}


class TuneEntry(_C.Structure):  # lg_tune_entry (include/lasgun_hip.h)
    _fields_ = [("key", _C.c_uint64 * 12), ("choice", _C.c_int32), ("reserved", _C.c_int32)]


class HipApi(Api):
    """The product's binding: core surface + the GPU-only extras of include/lasgun_hip.h."""

    def _stream(self, accel, stream):
        """hipStream_t to enqueue on: an explicit handle (0 = HIP's default stream, e.g. torch's
        `current_stream().cuda_stream`) or, when None, the accel's own stream."""
        if stream is None:
            return _C.c_void_p(self.call("accel_stream", accel.h))
        return _C.c_void_p(int(stream))

    def set_mode(self, accel, fast):
        """False = the reference traversal (parity path, default); True = the opt-in fast mode."""
        if self.call("accel_set_mode", accel.h, 1 if fast else 0):
            raise LasgunError(self.last_error())

    def set_prune(self, accel, enabled):
        """Pruned form of the reference traversal: None / -1 = the accel's default (on for scenes with a mesh of >= 4096
        triangles), False / True = off / on (include/lasgun_hip.h, lg_accel_set_prune)."""
        if self.call("accel_set_prune", accel.h, -1 if enabled is None or enabled == -1 else (1 if enabled else 0)):
            raise LasgunError(self.last_error())

    def last_organisation(self, accel):
        """What the accel's last launch ran as: "megakernel", "wavefront" (level by level), "queue"; None before the first."""
        v = self.call("accel_last_organisation", accel.h)
        name = {0: "megakernel", 1: "wavefront", 2: "queue"}.get(v & 15) if v >= 0 else None
        if name and (v & 16):
            name += ", bottom-up"
        if name and (v & 64):
            name += ", middle-out"
        if name and (v & 32):
            name += ", samples in a row"
        if name and (v & 128):
            name += ", tiles in parts"
        return name

    def set_tile_order(self, accel, order):
        """The direction the megakernel and the queue organisation claim a launch's tiles in: 0 top-down, 1 bottom-up, 2 from the middle row outwards, None / -1 = middle-out unless measured otherwise
        (include/lasgun_hip.h, lg_accel_set_tile_order).  Same bytes either way."""
        if self.call("accel_set_tile_order", accel.h, -1 if order is None else int(order)):
            raise LasgunError(self.last_error())

    def set_tile_parts(self, accel, parts):
        """The work item of the megakernel and of the queue organisation's level 0: a whole tile per wave (1) or a tile in 2 / 4 / 8 parts; None / -1 = whole unless measured otherwise for
        a small launch (include/lasgun_hip.h, lg_accel_set_tile_parts).  Same bytes either way."""
        if self.call("accel_set_tile_parts", accel.h, -1 if parts is None else int(parts)):
            raise LasgunError(self.last_error())

    def set_sample_order(self, accel, order):
        """A supersampled pixel's samples: 0 side by side in one launch chain, 1 one after the other, None / -1 = the default (side by side;
        the megakernel's form measured) (include/lasgun_hip.h, lg_accel_set_sample_order).  Same bytes either way."""
        if self.call("accel_set_sample_order", accel.h, -1 if order is None else int(order)):
            raise LasgunError(self.last_error())

    def get_prune(self, accel):
        """Whether a render of this accel uses the pruned reference walk right now (accel default, LASGUN_PRUNE, set_prune, fast mode)."""
        return bool(self.call("accel_get_prune", accel.h))

    def set_streaming(self, accel, enabled):
        """Kernel organisation (include/lasgun_hip.h, lg_accel_set_streaming): 1 / True = the accel's defaults, 0 / False = the
        megakernel only, 2 = the level-by-level wavefront pipeline wherever possible, 3 = the queue organisation (every recursion
        level in one persistent launch) wherever possible.  Same bytes out in every organisation."""
        if self.call("accel_set_streaming", accel.h, int(enabled) if enabled in (0, 1, 2, 3) else (1 if enabled else 0)):
            raise LasgunError(self.last_error())

    def set_wf_split(self, accel, bands):
        """Bands of a big wavefront launch on internal streams (0 = default; include/lasgun_hip.h, lg_accel_set_wf_split)."""
        self.call("accel_set_wf_split", accel.h, int(bands))

    def set_lds_scene(self, accel, enabled):
        """Scene tables resident in LDS for the streaming traversal kernels (default on); returns
        whether this accel's scene is small enough for that variant to exist."""
        return bool(self.call("accel_set_lds_scene", accel.h, 1 if enabled else 0))

    def set_device(self, device):
        if self.call("set_device", int(device)):
            raise LasgunError(self.last_error())

    def trace_pixel_log(self, accel, w, h, x, y, fast=False, nlights=1):
        """trace_pixel plus the event log of the PRIMARY ray's walk: rows of (code, a, b, c) -- 1.xy fast node pair
        (x/y = child boxes hit; a = accel*1e5 + node, b/c = tnear of the children), 2.x reference node, 3.x primitive
        test (a = primref, b = t, c = accel; .1 = accepted), 4 / 4.5 accel entry, 5 return, 6 triangle accepted, 9 end."""
        need = 7 + 2 * nlights
        n = need + 1 + 4 * 4000
        out = (_C.c_double * n)()
        if self.call("trace_pixel", accel.h, int(w), int(h), int(x), int(y), 1 if fast else 0, out, n):
            raise LasgunError(self.last_error())
        cnt = int(out[need])
        return [tuple(out[need + 1 + 4 * i + k] for k in range(4)) for i in range(cnt)]

    def trace_pixel(self, accel, w, h, x, y, fast=False, max_lights=64):
        """{"t", "ref", "accel", "shadow": [(t, ref), ...]} of pixel (x, y), sample 0 (debugging / test hook)."""
        n = 7 + 2 * max_lights
        out = (_C.c_double * n)()
        if self.call("trace_pixel", accel.h, int(w), int(h), int(x), int(y), 1 if fast else 0, out, n):
            raise LasgunError(self.last_error())
        nl = int(out[3])
        return {"t": out[0], "ref": int(out[1]), "accel": int(out[2]), "shadow": [(out[4 + 2 * l], int(out[5 + 2 * l])) for l in range(nl)],
                "shadow_origin": [out[4 + 2 * nl], out[5 + 2 * nl], out[6 + 2 * nl]]}

    def set_devices(self, ids=None):
        """Devices a host-film `capture` / `render` is split over (None or [] = every visible device)."""
        ids = list(ids or [])
        arr = (_C.c_int * max(len(ids), 1))(*ids)
        if self.call("set_devices", arr, len(ids)):
            raise LasgunError(self.last_error())

    def trim_pool(self, device=-1):
        """Give the device buffers parked in the library's pool back to the driver; returns the bytes freed."""
        return int(self.call("trim_pool", int(device)))

    def device_count(self):
        return int(self.call("device_count"))

    def capture_rows_device(self, accel, width, height, y0, y1, dev_ptr, row0=None, stream=None):
        """Enqueue rows [y0, y1) into device memory at `dev_ptr` (pixel (0,row0) first). No host copy."""
        if self.call("capture_rows_device", accel.h, width, height, y0, y1, y0 if row0 is None else row0,
                     _C.c_void_p(int(dev_ptr)), self._stream(accel, stream)):
            raise LasgunError(self.last_error())

    def capture_interleaved_device(self, accel, width, height, block_rows, n, r, dev_ptr, stream=None):
        """Enqueue the rows {y : (y // block_rows) % n == r} into a compact (height/n)-row device tile."""
        if self.call("capture_interleaved_device", accel.h, width, height, block_rows, n, r, _C.c_void_p(int(dev_ptr)),
                     self._stream(accel, stream)):
            raise LasgunError(self.last_error())

    def capture_subset_device(self, k, n, accel, width, height, dev_ptr, stream=None):
        if self.call("capture_subset_device", k, n, accel.h, width, height, _C.c_void_p(int(dev_ptr)),
                     self._stream(accel, stream)):
            raise LasgunError(self.last_error())

    def capture_subsets(self, ks, n, accel, film):
        """Several subsets of one n as ONE render: writes exactly the pixels of the calls capture_subset(k, n, ...) for k in ks (no
        counterpart in the reference, whose progressive caller makes those calls one by one: www/renderer.ts:103-120)."""
        ks = [int(k) for k in ks]
        arr = (_C.c_size_t * max(len(ks), 1))(*ks)
        if self.call("capture_subsets", arr, len(ks), int(n), accel.h, film.h):
            raise LasgunError(self.last_error())

    def capture_subsets_device(self, ks, n, accel, width, height, dev_ptr, stream=None):
        """Enqueue the subsets {k + i*n}, k in ks, as one render into a full width*height device film."""
        ks = [int(k) for k in ks]
        arr = (_C.c_size_t * max(len(ks), 1))(*ks)
        if self.call("capture_subsets_device", arr, len(ks), int(n), accel.h, width, height, _C.c_void_p(int(dev_ptr)),
                     self._stream(accel, stream)):
            raise LasgunError(self.last_error())

    def synchronize(self, accel):
        if self.call("accel_synchronize", accel.h):
            raise LasgunError(self.last_error())

    def capture_radiance(self, accel, w, h, k=0, n=1, **_):
        rgb = _np.full((h, w, 3), _np.nan, dtype=_np.float64)
        if self.call("capture_radiance", k, n, accel.h, w, h, rgb.ctypes.data):
            raise LasgunError(self.last_error())
        return rgb

    def capture_pixels(self, accel, w, h, offsets, radiance=True):
        """(rgba (n, 4) uint8, radiance (n, 3) float64 or None) of the pixels `offsets` (y * w + x), in list order."""
        off = _np.ascontiguousarray(offsets, dtype=_np.uint64)
        rgba = _np.zeros((off.size, 4), dtype=_np.uint8)
        rad = _np.full((off.size, 3), _np.nan, dtype=_np.float64) if radiance else None
        if self.call("capture_pixels", accel.h, w, h, off.ctypes.data, off.size, rgba.ctypes.data, rad.ctypes.data if radiance else None):
            raise LasgunError(self.last_error())
        return rgba, rad

    def capture_rect(self, accel, w, h, x0, y0, x1, y1, radiance=True):
        """(rgba (y1-y0, x1-x0, 4) uint8, radiance (y1-y0, x1-x0, 3) float64 or None) of a crop of the w x h film."""
        rgba = _np.zeros((y1 - y0, x1 - x0, 4), dtype=_np.uint8)
        rad = _np.full((y1 - y0, x1 - x0, 3), _np.nan, dtype=_np.float64) if radiance else None
        if self.call("capture_rect", accel.h, w, h, x0, y0, x1, y1, rgba.ctypes.data, rad.ctypes.data if radiance else None):
            raise LasgunError(self.last_error())
        return rgba, rad

    def capture_stats(self, accel, w, h, y0=0, y1=None):
        s = CStats()
        if self.call("capture_stats", accel.h, w, h, y0, h if y1 is None else y1, _C.byref(s)):
            raise LasgunError(self.last_error())
        return s.as_dict()

    def audit_prune(self, accel, w, h, y0=0, y1=None):
        """Audit of the pruned reference walk on rows [y0, y1) (include/lasgun_hip.h, lg_audit_prune): every node / run it skips
        is also walked the reference's way; `violations` (skipped primitives the reference would have accepted) must be 0."""
        class _A(_C.Structure):
            _fields_ = [("skipped_nodes", _C.c_uint64), ("skipped_runs", _C.c_uint64), ("primitives", _C.c_uint64), ("violations", _C.c_uint64),
                        ("min_slack_nodes", _C.c_double), ("min_slack_runs", _C.c_double), ("max_margin_used_nodes", _C.c_double)]
        r = _A()
        if self.call("audit_prune", accel.h, w, h, y0, h if y1 is None else y1, _C.byref(r)):
            raise LasgunError(self.last_error())
        return {k: getattr(r, k) for k, _ in _A._fields_}

    def audit_fast(self, accel, w, h, y0=0, y1=None):
        """Audit of the opt-in FAST mode on rows [y0, y1) (lg_audit_fast): every ray also walked the reference's way on the device;
        {"rays", "fallbacks", "violations"} -- `violations` (rays whose answer is not the reference's) must be 0."""
        class _A(_C.Structure):
            _fields_ = [("rays", _C.c_uint64), ("fallbacks", _C.c_uint64), ("violations", _C.c_uint64)]
        r = _A()
        if self.call("audit_fast", accel.h, w, h, y0, h if y1 is None else y1, _C.byref(r)):
            raise LasgunError(self.last_error())
        return {k: int(getattr(r, k)) for k, _ in _A._fields_}

    def capture_stats_kind(self, accel, w, h, kind, y0=0, y1=None):
        """Work counters of one kind of traversal: 1 = closest-hit (primary/secondary), 2 = shadow."""
        s = CStats()
        if self.call("capture_stats_kind", accel.h, w, h, y0, h if y1 is None else y1, int(kind), _C.byref(s)):
            raise LasgunError(self.last_error())
        return s.as_dict()

    def profile_read_kinds(self, accel):
        """Streaming pipeline: {kernel: (total ms, launches)} from HIP events around each kernel."""
        ms = (_C.c_double * 5)(); n = (_C.c_uint64 * 5)()
        if self.call("profile_read_kinds", accel.h, ms, n):
            raise LasgunError(self.last_error())
        names = ("trace<closest>", "combine", "trace<shadow>", "shade", "trace_kernel")
        return {names[i]: (ms[i], int(n[i])) for i in range(5)}

    def profile_enable(self, accel, enabled=True):
        self.call("profile_enable", accel.h, 1 if enabled else 0)

    def profile_read(self, accel):
        ms = _C.c_double(); n = _C.c_uint64()
        if self.call("profile_read", accel.h, _C.byref(ms), _C.byref(n)):
            raise LasgunError(self.last_error())
        return ms.value, int(n.value)

    def host_build_dump(self, scene):
        """Host-only BVH build + flatten (no GPU needed): (floats, ints, info dict)."""
        pf = _C.POINTER(_C.c_double)(); nf = _C.c_size_t(); pi = _C.POINTER(_C.c_int64)(); ni = _C.c_size_t()
        info = (_C.c_uint64 * 8)()
        if self.call("host_build_dump", scene.h, _C.byref(pf), _C.byref(nf), _C.byref(pi), _C.byref(ni), info):
            raise LasgunError(self.last_error())
        f = _np.ctypeslib.as_array(pf, shape=(nf.value,)).copy()
        i = _np.ctypeslib.as_array(pi, shape=(ni.value,)).copy()
        keys = ("nodes", "primrefs", "spheres", "cuboids", "triangles", "accels", "max_stack", "has_specular")
        return f, i, dict(zip(keys, [int(v) for v in info]))

    def host_check_strips(self, scene):
        """Host-only self-check of the triangle strips of the mesh leaves (include/lasgun_hip.h, lg_host_check_strips)."""
        out = (_C.c_uint64 * 8)()
        if self.call("host_check_strips", scene.h, out):
            raise LasgunError(self.last_error())
        return dict(zip(("leaves", "runs", "triangles", "entries", "violations", "records_hash", "strips_hash"), [int(v) for v in out]))

    def host_check_wide_records(self, scene):
        """Host-only self-check of the fast mode's wide node records (no GPU needed): dict of counts; `violations` must be 0."""
        out = (_C.c_uint64 * 8)()
        if self.call("host_check_wide_records", scene.h, out):
            raise LasgunError(self.last_error())
        keys = ("records", "children", "leaves", "deepest_stack", "violations", "reserved_stack")
        return dict(zip(keys, [int(v) for v in out]))

    # ---- the table of measured organisation choices (lg_tune_*): entries are (twelve key words, choice) tuples
    def tune_export(self):
        n = int(self.call("tune_export", None, 0))
        buf = (TuneEntry * max(n, 1))()
        m = int(self.call("tune_export", _C.cast(buf, _C.c_void_p), n))
        return [(tuple(int(x) for x in buf[i].key), int(buf[i].choice)) for i in range(min(n, m))]

    def tune_import(self, entries):
        entries = list(entries)
        buf = (TuneEntry * max(len(entries), 1))()
        for i, (key, choice) in enumerate(entries):
            for j in range(12):
                buf[i].key[j] = int(key[j])
            buf[i].choice = int(choice)
        if self.call("tune_import", _C.cast(buf, _C.c_void_p), len(entries)):
            raise LasgunError(self.last_error())

    def tune_clear(self):
        self.call("tune_clear")

    def Multi(self, scene, devices, block_rows=64):
        """One film on several GPUs of this process, gathered on devices[0] over xGMI with one grouped RCCL exchange
        (lg_multi_*): `.capture_device(w, h, dev_ptr)`, `.capture(film)`, `.accel(rank)`, `.uses_rccl`."""
        api = self

        class _Multi:
            def __init__(self):
                ids = list(devices)
                arr = (_C.c_int * max(len(ids), 1))(*ids)
                self.scene = scene  # borrowed by the C side: keep it alive
                self.h = api.call("multi_create", scene.h, arr, len(ids), int(block_rows))
                if not self.h:
                    raise LasgunError(api.last_error())

            @property
            def uses_rccl(self):
                return bool(api.call("multi_uses_rccl", self.h))

            @property
            def ranks(self):
                return int(api.call("multi_rank_count", self.h))

            def accel(self, rank):
                class _Borrowed:  # the multi owns it
                    pass
                a = _Borrowed()
                a.h = api.call("multi_accel", self.h, int(rank))
                return a

            def capture_device(self, w, h, dev_ptr):
                if api.call("multi_capture_device", self.h, int(w), int(h), _C.c_void_p(int(dev_ptr))):
                    raise LasgunError(api.last_error())

            def capture_device_all(self, w, h, dev_ptrs):
                """All-gather form: dev_ptrs[r] = a w*h*4-byte buffer on rank r's device; every one receives the whole film."""
                arr = (_C.c_void_p * len(dev_ptrs))(*[int(p) for p in dev_ptrs])
                if api.call("multi_capture_device_all", self.h, int(w), int(h), arr):
                    raise LasgunError(api.last_error())

            def capture(self, film):
                if api.call("multi_capture", self.h, film.h):
                    raise LasgunError(api.last_error())

            def close(self):
                if self.h:
                    api.call("multi_free", self.h)
                    self.h = None

            def __del__(self):
                try:
                    self.close()
                except Exception:  # noqa: BLE001
                    pass
        return _Multi()

    def probe_rate(self, what):
        """Measured GB/s of the current device: "hbm_copy" (read + written bytes) or "lds_read"."""
        v = _C.c_double()
        if self.call("probe_rate", {"hbm_copy": 0, "lds_read": 1}[what], _C.byref(v)):
            raise LasgunError(self.last_error())
        return v.value

    def accel_info(self, accel):
        out = (_C.c_uint64 * 8)()
        self.call("accel_info", accel.h, out)
        keys = ("nodes", "primrefs", "spheres", "cuboids", "triangles", "accels", "max_stack", "device_bytes")
        return dict(zip(keys, [int(v) for v in out]))


def device_source_sha16():
    """First 16 hex digits of the SHA-256 over the device sources (csrc/*.h, csrc/k_*.hip, in name order): the provenance
    stamp of profiler counters -- bench.py reports a committed PMC figure only while the sources it was collected on are unchanged."""
    import glob
    import hashlib
    src = _os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for path in sorted(glob.glob(_os.path.join(src, "*.h")) + glob.glob(_os.path.join(src, "k_*.hip"))):
        if _os.path.basename(path) in ("host.h", "internal.h", "tune.h"):  # host-side headers: no kernel is compiled from them
            continue
        h.update(_os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and load it by
    file name; liblasgun_hip.so asks for the soname libamdhip64.so.7.  If this library is loaded first it
    binds to the system runtime and a later `import torch` brings a SECOND runtime into the process, which
    then finds no GPU.  Loading torch's copy first (when a torch wheel with a bundled runtime is installed;
    torch itself is not imported) makes both resolve to the same object, whichever is imported first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = _os.path.join(_os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if _os.path.exists(bundled):
        _C.CDLL(bundled, mode=_C.RTLD_GLOBAL)


_share_torch_hip_runtime()
api = HipApi(_C.CDLL(LIB_PATH), "lg_", _EXTRA)

# reference-shaped names at package level: `from lasgun_amd import Scene, Material, capture`
Scene, Aggregate, Material, Camera, Film, Accel = api.Scene, api.Aggregate, api.Material, api.Camera, api.Film, api.Accel
capture, capture_subset, render = api.capture, api.capture_subset, api.render
