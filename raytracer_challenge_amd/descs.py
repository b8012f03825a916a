"""ctypes mirrors of the plain-old-data descriptors declared in include/rtc_hip.h.

Field order, types and padding follow the header exactly; tests/test_cabi.py checks the
sizes against the values the C compiler reports.
"""
import ctypes as C

RTC_OK = 0
RTC_ERR_INVALID_ARGUMENT = 1
RTC_ERR_SINGULAR_TRANSFORM = 2
RTC_ERR_NOT_AFFINE = 3
RTC_ERR_DUPLICATE_SHAPE = 4
RTC_ERR_OUT_OF_MEMORY = 5
RTC_ERR_DEVICE = 6
RTC_ERR_TOO_LARGE = 7

RTC_GEO_SPHERE = 0
RTC_GEO_PLANE = 1
RTC_GEO_CUBE = 2  # extension beyond the reference
RTC_SWEEP_GROUPS = 5

RTC_PATTERN_POINT = 0
RTC_PATTERN_STRIPE = 1
RTC_PATTERN_GRADIENT = 2
RTC_PATTERN_RING = 3
RTC_PATTERN_CHECKERS = 4

RTC_ACCEL_NONE = 0
RTC_ACCEL_BVH = 1
RTC_INTEGRATOR_POSTORDER = 0
RTC_INTEGRATOR_WEIGHTED = 1
RTC_CANVAS_STORE_PIXELS = 0
RTC_CANVAS_STORE_BLOCKS = 1
RTC_NOALLOC_MAX_TILES = 262144
RTC_MULTI_MAX_RANKS = 64
RTC_IPC_HANDLE_BYTES = 64
RTC_MULTI_PEER_STORES = 0
RTC_MULTI_PEER_SCATTER = 1
RTC_MULTI_RCCL_GATHER = 2
RTC_MULTI_HOST_DIRECT = 3
RTC_MULTI_PEER_BLOCK_STORES = 4
RTC_TILE_W = 64
RTC_TILE_H = 8


class RtcPatternDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("_pad0", C.c_int32),
        ("a", C.c_double * 3),
        ("b", C.c_double * 3),
        ("transform", C.c_double * 16),
    ]


class RtcMaterialDesc(C.Structure):
    _fields_ = [
        ("color", C.c_double * 3),
        ("ambient", C.c_double),
        ("diffuse", C.c_double),
        ("specular", C.c_double),
        ("shininess", C.c_double),
        ("reflective", C.c_double),
        ("transparency", C.c_double),
        ("refractive_index", C.c_double),
        ("has_pattern", C.c_int32),
        ("_pad0", C.c_int32),
        ("pattern", RtcPatternDesc),
    ]


class RtcShapeDesc(C.Structure):
    _fields_ = [
        ("geo", C.c_int32),
        ("_pad0", C.c_int32),
        ("transform", C.c_double * 16),
        ("material", RtcMaterialDesc),
    ]


class RtcLightDesc(C.Structure):
    _fields_ = [
        ("position", C.c_double * 4),
        ("intensity", C.c_double * 3),
    ]


class RtcCameraDesc(C.Structure):
    _fields_ = [
        ("hsize", C.c_uint64),
        ("vsize", C.c_uint64),
        ("field_of_view", C.c_double),
        ("half_width", C.c_double),
        ("half_height", C.c_double),
        ("pixel_size", C.c_double),
        ("transform", C.c_double * 16),
    ]


class RtcStats(C.Structure):
    _fields_ = [
        ("radiance_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("plane_tests", C.c_uint64),
        ("sphere_hits", C.c_uint64),
        ("plane_hits", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("pixels", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("sphere_bound_skips", C.c_uint64),
        ("plane_side_skips", C.c_uint64),
        ("cube_tests", C.c_uint64),
        ("cube_hits", C.c_uint64),
        ("sphere_prefilter_skips", C.c_uint64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class RtcMultiTiming(C.Structure):
    _fields_ = [
        ("render_and_transfer_ms", C.c_double),
        ("assemble_ms", C.c_double),
        ("device_ms", C.c_double),
        ("to_host_ms", C.c_double),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class RtcSweepGroupInfo(C.Structure):
    _fields_ = [
        ("geo", C.c_int32),
        ("axis_aligned", C.c_int32),
        ("two_roots", C.c_int32),
        ("flops_reference", C.c_int32),
        ("flops_executed", C.c_int32),
        ("flops_roots", C.c_int32),
        ("name", C.c_char * 40),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["name"] = d["name"].decode()
        return d
