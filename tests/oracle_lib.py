"""Loader for the CPU oracle (oracle/liblasgun_oracle.so) -- test infrastructure only.

Binds the oracle's `orc_*` C API with the same parametrised binding the product uses for
`lg_*`, so one scene-building function can drive both sides.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from lasgun_amd._capi import Api, CStats, LasgunError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblasgun_oracle.so")

_EXTRA = {
    "set_trig_mode": (None, [C.c_int]),
    "capture_radiance": (C.c_int, [C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "capture_subset_mt": (C.c_int, [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "capture_pixels": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "stats_reset": (None, []),
    "stats_read": (None, [C.POINTER(CStats)]),
}


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class OracleApi(Api):
    def set_trig_mode(self, portable):
        """0 = glibc libm (what the Rust reference calls), 1 = the portable algorithm the GPU uses."""
        self.call("set_trig_mode", 1 if portable else 0)

    def capture_radiance(self, accel, w, h, k=0, n=1, nthreads=8):
        rgb = np.full((h, w, 3), np.nan, dtype=np.float64)
        if self.call("capture_radiance", k, n, accel.h, w, h, rgb.ctypes.data, nthreads):
            raise LasgunError(self.last_error())
        return rgb

    def capture_subset_mt(self, k, n, accel, film, nthreads):
        if self.call("capture_subset_mt", k, n, accel.h, film.h, nthreads):
            raise LasgunError(self.last_error())

    def capture_pixels(self, accel, w, h, offsets, radiance=True, nthreads=8):
        """(rgba (n, 4), radiance (n, 3) or None) of the pixels `offsets` (y * w + x), in list order."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        rgba = np.zeros((off.size, 4), dtype=np.uint8)
        rad = np.full((off.size, 3), np.nan, dtype=np.float64) if radiance else None
        if self.call("capture_pixels", accel.h, w, h, off.ctypes.data, off.size, rgba.ctypes.data, rad.ctypes.data if radiance else None, nthreads):
            raise LasgunError(self.last_error())
        return rgba, rad

    def capture_rect(self, accel, w, h, x0, y0, x1, y1, radiance=True, nthreads=8):
        ys, xs = np.mgrid[y0:y1, x0:x1]
        rgba, rad = self.capture_pixels(accel, w, h, (ys * w + xs).ravel(), radiance, nthreads)
        return rgba.reshape(y1 - y0, x1 - x0, 4), (rad.reshape(y1 - y0, x1 - x0, 3) if radiance else None)

    def stats_reset(self):
        self.call("stats_reset")

    def stats_read(self):
        s = CStats()
        self.call("stats_read", C.byref(s))
        return s.as_dict()


_api = None


def oracle():
    global _api
    if _api is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _api = OracleApi(C.CDLL(ORACLE_SO), "orc_", _EXTRA)
    return _api
