"""ctypes binding of the CPU oracle (oracle/_build/librtc_oracle.so) for the tests.

Test infrastructure only: nothing under raytracer_challenge_amd/ imports this.  The oracle
takes the same descriptors as the product's C-ABI, so a test hands identical inputs to both.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_challenge_amd import descs as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "librtc_oracle.so")
KAT_PATH = os.path.join(ORACLE_DIR, "_build", "kat")

LITERAL, STREAMING = 0, 1
_lib = None


def build():
    """(Re)build the oracle with its Makefile (g++, a second or two)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or not os.path.exists(KAT_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        dp, u64, i32 = C.POINTER(C.c_double), C.c_uint64, C.c_int32
        sp, lp, cp, st = (C.POINTER(D.RtcShapeDesc), C.POINTER(D.RtcLightDesc), C.POINTER(D.RtcCameraDesc),
                          C.POINTER(D.RtcStats))
        L.orc_render.argtypes = [sp, u64, lp, cp, i32, i32, u64, u64, u64, dp, st]
        L.orc_render_pixels.argtypes = [sp, u64, lp, cp, i32, i32, C.POINTER(u64), u64, dp, st]
        L.orc_color_at.argtypes = [sp, u64, lp, dp, u64, i32, i32, dp, st]
        L.orc_mat4_inverse.argtypes = [dp, dp, dp]
        L.orc_ray_for_pixel.argtypes = [cp, u64, u64, dp]
        L.orc_draw_sphere.argtypes = [u64, dp, dp]
        L.orc_pow.argtypes = [C.c_double, C.c_double]
        L.orc_pow.restype = C.c_double
        L.orc_quantize_u8.argtypes = [dp, u64, C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def render(world, camera, depth=5, mode=STREAMING, rows=None, stats=None):
    """Full canvas (vsize, hsize, 3); rows=(begin, end, step) restricts the rows rendered
    (others stay NaN)."""
    cd = camera.desc()
    out = np.full((cd.vsize, cd.hsize, 3), np.nan, dtype=np.float64)
    b, e, s = rows if rows is not None else (0, cd.vsize, 1)
    stp = C.byref(stats) if stats is not None else None
    ld = world.light_desc()
    rc = lib().orc_render(world.shape_descs(), len(world.objects), C.byref(ld), C.byref(cd), depth, mode, b, e, s,
                          _dptr(out), stp)
    assert rc == 0
    return out


def render_pixels(world, camera, pixel_indices, depth=5, mode=STREAMING, stats=None):
    cd = camera.desc()
    idx = np.ascontiguousarray(pixel_indices, dtype=np.uint64)
    out = np.empty((idx.size, 3), dtype=np.float64)
    stp = C.byref(stats) if stats is not None else None
    ld = world.light_desc()
    rc = lib().orc_render_pixels(world.shape_descs(), len(world.objects), C.byref(ld), C.byref(cd), depth, mode,
                                 idx.ctypes.data_as(C.POINTER(C.c_uint64)), idx.size, _dptr(out), stp)
    assert rc == 0
    return out


def color_at(world, rays, depth=5, mode=STREAMING, stats=None):
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    out = np.empty((rays.shape[0], 3), dtype=np.float64)
    stp = C.byref(stats) if stats is not None else None
    ld = world.light_desc()
    rc = lib().orc_color_at(world.shape_descs(), len(world.objects), C.byref(ld), _dptr(rays), rays.shape[0], depth,
                            mode, _dptr(out), stp)
    assert rc == 0
    return out


def mat4_inverse(m16):
    m = np.ascontiguousarray(np.asarray(m16, dtype=np.float64).reshape(16))
    out = np.empty(16, dtype=np.float64)
    det = C.c_double(0)
    lib().orc_mat4_inverse(_dptr(m), _dptr(out), C.byref(det))
    return out.reshape(4, 4), det.value


def ray_for_pixel(camera, x, y):
    cd = camera.desc()
    out = np.empty(6, dtype=np.float64)
    lib().orc_ray_for_pixel(C.byref(cd), x, y, _dptr(out))
    return out


def draw_sphere(canvas_size=100):
    """draw_sphere.zig:38-83: (canvas (n, n, 3), rays (n*n, 6))."""
    out = np.empty((canvas_size, canvas_size, 3), dtype=np.float64)
    rays = np.empty((canvas_size * canvas_size, 6), dtype=np.float64)
    assert lib().orc_draw_sphere(canvas_size, _dptr(out), _dptr(rays)) == 0
    return out, rays


def zig_pow(x, y):
    return lib().orc_pow(float(x), float(y))


def quantize_u8(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.empty(rgb.shape, dtype=np.uint8)
    lib().orc_quantize_u8(_dptr(rgb), rgb.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out
