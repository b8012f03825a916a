"""ctypes binding of librtc_hip.so (the C-ABI of include/rtc_hip.h).

There is no CPU fallback: if the shared library is missing, or a call returns a non-zero
status, an exception is raised.  torch is not needed here; callers that keep buffers on
the device pass raw device pointers (e.g. tensor.data_ptr()) and stream handles.
"""
import ctypes as C
import os

import numpy as np

from . import descs as D

# RTC_HIP_LIB lets a developer point at another build of the same library (A/B timing).
_LIB_PATH = os.environ.get("RTC_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librtc_hip.so")
_lib = None


class RtcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtc error {code}: {message}")
        self.code = code
        self.message = message


def lib_path():
    return _LIB_PATH


def lib():
    """Load librtc_hip.so once.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `make -C raytracer_challenge_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback for the render path"
        )
    L = C.CDLL(_LIB_PATH)
    dp, u64, u32, i32, vp = C.POINTER(C.c_double), C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p
    L.rtc_last_error.restype = C.c_char_p
    L.rtc_last_error.argtypes = []
    L.rtc_device_count.argtypes = [C.POINTER(i32)]
    L.rtc_set_device.argtypes = [i32]
    L.rtc_scene_create.argtypes = [C.POINTER(D.RtcShapeDesc), u64, C.POINTER(D.RtcLightDesc), C.POINTER(vp)]
    L.rtc_scene_destroy.argtypes = [vp]
    L.rtc_scene_set_acceleration.argtypes = [vp, i32]
    L.rtc_scene_set_integrator.argtypes = [vp, i32]
    L.rtc_scene_set_canvas_store.argtypes = [vp, i32]
    L.rtc_scene_group_counts.argtypes = [vp, C.POINTER(C.c_uint64 * D.RTC_SWEEP_GROUPS)]
    L.rtc_sweep_group_info.argtypes = [u32, C.POINTER(D.RtcSweepGroupInfo)]
    L.rtc_render.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, dp, C.POINTER(D.RtcStats)]
    L.rtc_render_u8.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, C.POINTER(C.c_uint8), C.POINTER(D.RtcStats)]
    L.rtc_render_device.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, u32, u32, vp, vp, C.POINTER(D.RtcStats)]
    L.rtc_render_device_noalloc.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, u32, u32, vp, vp, C.POINTER(D.RtcStats)]
    L.rtc_render_device_to_canvas.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, u32, u32, vp, vp, C.POINTER(D.RtcStats)]
    L.rtc_scatter_tiles_device.argtypes = [C.POINTER(D.RtcCameraDesc), u32, u32, vp, vp, vp]
    L.rtc_scatter_tiles_u8_device.argtypes = [C.POINTER(D.RtcCameraDesc), u32, u32, vp, vp, vp]
    L.rtc_multi_create.argtypes = [C.POINTER(D.RtcShapeDesc), u64, C.POINTER(D.RtcLightDesc), C.POINTER(i32), u32,
                                   C.POINTER(vp)]
    L.rtc_multi_destroy.argtypes = [vp]
    L.rtc_multi_rank_device.argtypes = [vp, u32, C.POINTER(i32)]
    L.rtc_multi_set_integrator.argtypes = [vp, i32]
    L.rtc_render_multi.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, i32, dp, C.POINTER(D.RtcStats),
                                   C.POINTER(D.RtcMultiTiming)]
    L.rtc_render_multi_u8.argtypes = [vp, C.POINTER(D.RtcCameraDesc), i32, i32, C.POINTER(C.c_uint8),
                                      C.POINTER(D.RtcStats), C.POINTER(D.RtcMultiTiming)]
    L.rtc_ipc_canvas_create.argtypes = [u64, C.POINTER(vp), C.POINTER(C.c_uint8 * D.RTC_IPC_HANDLE_BYTES)]
    L.rtc_ipc_canvas_open.argtypes = [C.POINTER(C.c_uint8 * D.RTC_IPC_HANDLE_BYTES), C.POINTER(vp)]
    L.rtc_ipc_canvas_close.argtypes = [vp, i32]
    L.rtc_tile_buffer_doubles.argtypes = [C.POINTER(D.RtcCameraDesc), u32, u32, C.POINTER(u64)]
    L.rtc_deinterleave_device.argtypes = [C.POINTER(D.RtcCameraDesc), u32, u64, vp, vp, vp]
    L.rtc_deinterleave_u8_device.argtypes = [C.POINTER(D.RtcCameraDesc), u32, u64, vp, vp, vp]
    L.rtc_color_at.argtypes = [vp, dp, u64, i32, dp, C.POINTER(D.RtcStats)]
    L.rtc_quantize_u8_device.argtypes = [vp, u64, vp, vp]
    L.rtc_mat4_inverse.argtypes = [dp, dp, dp]
    L.rtc_arith_probe.argtypes = [dp, dp, u64, dp, dp, dp]
    for name in EXPORTS:
        if name != "rtc_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


# every symbol include/rtc_hip.h declares
EXPORTS = (
    "rtc_last_error",
    "rtc_device_count",
    "rtc_set_device",
    "rtc_scene_create",
    "rtc_scene_destroy",
    "rtc_scene_set_acceleration",
    "rtc_scene_set_integrator",
    "rtc_scene_set_canvas_store",
    "rtc_scene_group_counts",
    "rtc_sweep_group_info",
    "rtc_render",
    "rtc_render_u8",
    "rtc_render_device",
    "rtc_render_device_noalloc",
    "rtc_render_device_to_canvas",
    "rtc_scatter_tiles_device",
    "rtc_scatter_tiles_u8_device",
    "rtc_multi_create",
    "rtc_multi_destroy",
    "rtc_multi_rank_device",
    "rtc_multi_set_integrator",
    "rtc_render_multi",
    "rtc_render_multi_u8",
    "rtc_ipc_canvas_create",
    "rtc_ipc_canvas_open",
    "rtc_ipc_canvas_close",
    "rtc_tile_buffer_doubles",
    "rtc_deinterleave_device",
    "rtc_deinterleave_u8_device",
    "rtc_color_at",
    "rtc_quantize_u8_device",
    "rtc_mat4_inverse",
    "rtc_arith_probe",
)


def check(code):
    if code != D.RTC_OK:
        raise RtcError(code, lib().rtc_last_error().decode("utf-8", "replace"))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def device_count():
    n = C.c_int32(0)
    check(lib().rtc_device_count(C.byref(n)))
    return n.value


def set_device(i):
    check(lib().rtc_set_device(int(i)))


def mat4_inverse(m16):
    m = np.ascontiguousarray(np.asarray(m16, dtype=np.float64).reshape(16))
    out = np.empty(16, dtype=np.float64)
    det = C.c_double(0)
    check(lib().rtc_mat4_inverse(_dptr(m), _dptr(out), C.byref(det)))
    return out.reshape(4, 4), det.value


def arith_probe(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape and a.ndim == 1
    q, s, p = (np.empty_like(a) for _ in range(3))
    check(lib().rtc_arith_probe(_dptr(a), _dptr(b), a.size, _dptr(q), _dptr(s), _dptr(p)))
    return q, s, p


def sweep_group_table():
    """rtc_sweep_group_info for every group: a list of dicts (geo, axis_aligned, two_roots, flops_*, name)."""
    out = []
    for g in range(D.RTC_SWEEP_GROUPS):
        info = D.RtcSweepGroupInfo()
        check(lib().rtc_sweep_group_info(g, C.byref(info)))
        out.append(info.as_dict())
    return out


def tile_buffer_doubles(camera_desc, tile_rank, tile_ranks):
    n = C.c_uint64(0)
    check(lib().rtc_tile_buffer_doubles(C.byref(camera_desc), tile_rank, tile_ranks, C.byref(n)))
    return n.value


def deinterleave_device(camera_desc, tile_ranks, per_rank_doubles, d_gathered, d_out, stream=None):
    check(lib().rtc_deinterleave_device(C.byref(camera_desc), tile_ranks, per_rank_doubles, d_gathered, d_out, stream))


def deinterleave_u8_device(camera_desc, tile_ranks, per_rank_bytes, d_gathered_u8, d_out_u8, stream=None):
    check(lib().rtc_deinterleave_u8_device(C.byref(camera_desc), tile_ranks, per_rank_bytes, d_gathered_u8, d_out_u8,
                                           stream))


def scatter_tiles_device(camera_desc, tile_rank, tile_ranks, d_tiles, d_canvas, stream=None):
    check(lib().rtc_scatter_tiles_device(C.byref(camera_desc), tile_rank, tile_ranks, d_tiles, d_canvas, stream))


def scatter_tiles_u8_device(camera_desc, tile_rank, tile_ranks, d_tiles_u8, d_canvas_u8, stream=None):
    check(lib().rtc_scatter_tiles_u8_device(C.byref(camera_desc), tile_rank, tile_ranks, d_tiles_u8, d_canvas_u8, stream))


def ipc_canvas_create(n_bytes):
    """rtc_ipc_canvas_create: (device pointer, handle bytes) of a canvas other processes can map."""
    p = C.c_void_p(None)
    h = (C.c_uint8 * D.RTC_IPC_HANDLE_BYTES)()
    check(lib().rtc_ipc_canvas_create(int(n_bytes), C.byref(p), C.byref(h)))
    return p.value, bytes(h)


def ipc_canvas_open(handle):
    h = (C.c_uint8 * D.RTC_IPC_HANDLE_BYTES)(*handle)
    p = C.c_void_p(None)
    check(lib().rtc_ipc_canvas_open(C.byref(h), C.byref(p)))
    return p.value


def ipc_canvas_close(d_canvas, is_owner):
    check(lib().rtc_ipc_canvas_close(C.c_void_p(d_canvas), 1 if is_owner else 0))


def quantize_u8_device(d_rgb, n_values, d_out_u8, stream=None):
    check(lib().rtc_quantize_u8_device(d_rgb, n_values, d_out_u8, stream))


class Scene:
    """An uploaded world (rtc_scene_create / rtc_scene_destroy)."""

    def __init__(self, shape_descs, n_shapes, light_desc):
        self._h = C.c_void_p(None)
        check(lib().rtc_scene_create(shape_descs, int(n_shapes), C.byref(light_desc), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rtc_scene_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_integrator(self, integrator):
        """rtc_scene_set_integrator: 0 = the reference's post-order sum (default), 1 = attenuation-weighted."""
        check(lib().rtc_scene_set_integrator(self._h, int(integrator)))

    def set_canvas_store(self, mode):
        """rtc_scene_set_canvas_store: 0 = every finished pixel stored at once (default), 1 = whole 8x8 blocks."""
        check(lib().rtc_scene_set_canvas_store(self._h, int(mode)))

    def set_acceleration(self, mode):
        """rtc_scene_set_acceleration: 0 = test every shape (default), 1 = hierarchy over the spheres."""
        check(lib().rtc_scene_set_acceleration(self._h, int(mode)))

    def group_counts(self):
        """rtc_scene_group_counts: shapes per sweep group, in the order of sweep_group_table()."""
        out = (C.c_uint64 * D.RTC_SWEEP_GROUPS)()
        check(lib().rtc_scene_group_counts(self._h, C.byref(out)))
        return tuple(int(v) for v in out)

    def render(self, camera_desc, depth=5, stats=None):
        """rtc_render: returns the canvas as a (vsize, hsize, 3) float64 array."""
        out = np.empty((camera_desc.vsize, camera_desc.hsize, 3), dtype=np.float64)
        sp = C.byref(stats) if stats is not None else None
        check(lib().rtc_render(self._h, C.byref(camera_desc), int(depth), _dptr(out), sp))
        return out

    def render_u8(self, camera_desc, depth=5, stats=None):
        """rtc_render_u8: the canvas after byteColorValue (ppm.zig:6-9) as a (vsize, hsize, 3) uint8 array."""
        out = np.empty((camera_desc.vsize, camera_desc.hsize, 3), dtype=np.uint8)
        sp = C.byref(stats) if stats is not None else None
        check(
            lib().rtc_render_u8(
                self._h, C.byref(camera_desc), int(depth), out.ctypes.data_as(C.POINTER(C.c_uint8)), sp
            )
        )
        return out

    def render_device(self, camera_desc, depth, d_out, tile_rank=0, tile_ranks=1, stream=None, stats=None):
        """rtc_render_device: d_out is a raw device pointer (int)."""
        sp = C.byref(stats) if stats is not None else None
        check(
            lib().rtc_render_device(
                self._h, C.byref(camera_desc), int(depth), tile_rank, tile_ranks, C.c_void_p(d_out), stream, sp
            )
        )

    def render_device_noalloc(self, camera_desc, depth, d_out, tile_rank=0, tile_ranks=1, stream=None, stats=None):
        """rtc_render_device_noalloc: as render_device, scratch data in the library's fixed device array."""
        sp = C.byref(stats) if stats is not None else None
        check(
            lib().rtc_render_device_noalloc(
                self._h, C.byref(camera_desc), int(depth), tile_rank, tile_ranks, C.c_void_p(d_out), stream, sp
            )
        )

    def render_device_to_canvas(self, camera_desc, depth, d_canvas, tile_rank=0, tile_ranks=1, stream=None, stats=None):
        """rtc_render_device_to_canvas: this rank's tiles straight into the full row-major canvas."""
        sp = C.byref(stats) if stats is not None else None
        check(
            lib().rtc_render_device_to_canvas(
                self._h, C.byref(camera_desc), int(depth), tile_rank, tile_ranks, C.c_void_p(d_canvas), stream, sp
            )
        )

    def color_at(self, rays, depth=5, stats=None):
        """rtc_color_at: rays is (n, 6) [origin xyz, direction xyz]; returns (n, 3)."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.empty((rays.shape[0], 3), dtype=np.float64)
        sp = C.byref(stats) if stats is not None else None
        check(lib().rtc_color_at(self._h, _dptr(rays), rays.shape[0], int(depth), _dptr(out), sp))
        return out


class Multi:
    """A world replicated over the ranks of one process (rtc_multi_create / rtc_multi_destroy)."""

    def __init__(self, shape_descs, n_shapes, light_desc, n_ranks, devices=None):
        self._h = C.c_void_p(None)
        self.n_ranks = int(n_ranks)
        dev = (C.c_int32 * self.n_ranks)(*devices) if devices is not None else None
        check(lib().rtc_multi_create(shape_descs, int(n_shapes), C.byref(light_desc), dev, self.n_ranks, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rtc_multi_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_integrator(self, integrator):
        check(lib().rtc_multi_set_integrator(self._h, int(integrator)))

    def rank_device(self, rank):
        d = C.c_int32(-1)
        check(lib().rtc_multi_rank_device(self._h, rank, C.byref(d)))
        return d.value

    def render(self, camera_desc, depth=5, mode=D.RTC_MULTI_PEER_STORES, stats=None, timing=None, to_host=True):
        """rtc_render_multi: the (vsize, hsize, 3) float64 canvas, or None with to_host=False."""
        out = np.empty((camera_desc.vsize, camera_desc.hsize, 3), dtype=np.float64) if to_host else None
        check(lib().rtc_render_multi(self._h, C.byref(camera_desc), int(depth), int(mode),
                                     _dptr(out) if to_host else None,
                                     C.byref(stats) if stats is not None else None,
                                     C.byref(timing) if timing is not None else None))
        return out

    def render_u8(self, camera_desc, depth=5, mode=D.RTC_MULTI_PEER_STORES, stats=None, timing=None, to_host=True):
        out = np.empty((camera_desc.vsize, camera_desc.hsize, 3), dtype=np.uint8) if to_host else None
        check(lib().rtc_render_multi_u8(self._h, C.byref(camera_desc), int(depth), int(mode),
                                        out.ctypes.data_as(C.POINTER(C.c_uint8)) if to_host else None,
                                        C.byref(stats) if stats is not None else None,
                                        C.byref(timing) if timing is not None else None))
        return out
