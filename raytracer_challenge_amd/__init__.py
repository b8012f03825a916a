"""MI355X-native render path for the Zig ray tracer stevschmid/raytracer-challenge.

`host` mirrors the reference's scene API (World, Shape, Material, Pattern, Camera, Canvas,
canvasToPPM, viewTransform); `Camera.render` goes through `native`, the ctypes binding of
librtc_hip.so (HIP kernels + C-ABI built from csrc/).  `scenes` builds the benchmark worlds.
"""
from . import descs, host, native, scenes  # noqa: F401
from .host import (  # noqa: F401
    Camera,
    Canvas,
    Color,
    Mat4,
    Material,
    Pattern,
    PointLight,
    Shape,
    Vec4,
    World,
    byte_color_values,
    canvas_to_ppm,
    init_glass_sphere,
    init_point,
    init_vector,
    view_transform,
)
from .native import RtcError  # noqa: F401
