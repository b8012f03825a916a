"""Host-side mirror of the reference's scene API for the render path.

The reference is Zig and no Zig toolchain exists in this image, so the layer that sits
above the C-ABI (the part that stays Zig in a real integration) is mirrored here with the
same names, argument meaning and defaults: Vec4/initPoint/initVector (vector.zig:4-118),
Mat4 and its pre-multiplying fluent builders (matrix.zig:99-236), Color (color.zig:4-59),
Pattern (pattern.zig:64-75), Material (material.zig:6-18), Shape (shape.zig:73-82),
PointLight (light.zig:9-14), World (world.zig:15-59), Camera (camera.zig:17-90),
Canvas (canvas.zig:4-39), canvasToPPM (ppm.zig:6-38) and viewTransform (calc.zig:721-738).

Python floats are IEEE binary64 and CPython evaluates expressions in source order
without fusing, so the set-up arithmetic here keeps the reference's operation order.
The trigonometric functions come from libm where the reference uses Zig's std.math
(last-bit parity of those is unpinned; they are set-up only).

Everything below Camera.render is NOT mirrored in Python: render() flattens the world
into the C-ABI descriptors and calls the HIP library, which is the product.
"""
import math

import numpy as np

from . import descs as D


# ----------------------------------------------------------------------------- vector.zig
class Vec4:
    __slots__ = ("x", "y", "z", "w")

    def __init__(self, x, y, z, w):
        self.x, self.y, self.z, self.w = float(x), float(y), float(z), float(w)

    def add(self, o):
        return Vec4(self.x + o.x, self.y + o.y, self.z + o.z, self.w + o.w)

    def sub(self, o):
        return Vec4(self.x - o.x, self.y - o.y, self.z - o.z, self.w - o.w)

    def negate(self):
        return Vec4(-self.x, -self.y, -self.z, -self.w)

    def scale(self, s):
        return Vec4(s * self.x, s * self.y, s * self.z, s * self.w)

    def length(self):
        return math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z + self.w * self.w)

    def normalize(self):
        ln = self.length()
        return Vec4(self.x / ln, self.y / ln, self.z / ln, self.w / ln)

    def dot(self, o):
        return self.x * o.x + self.y * o.y + self.z * o.z + self.w * o.w

    def cross(self, o):
        return init_vector(
            self.y * o.z - self.z * o.y,
            self.z * o.x - self.x * o.z,
            self.x * o.y - self.y * o.x,
        )

    def as_tuple(self):
        return (self.x, self.y, self.z, self.w)

    def __repr__(self):
        return f"Vec4({self.x}, {self.y}, {self.z}, {self.w})"


def init_point(x, y, z):
    return Vec4(x, y, z, 1.0)


def init_vector(x, y, z):
    return Vec4(x, y, z, 0.0)


# ----------------------------------------------------------------------------- color.zig
class Color:
    __slots__ = ("red", "green", "blue")

    def __init__(self, red, green, blue):
        self.red, self.green, self.blue = float(red), float(green), float(blue)

    def as_tuple(self):
        return (self.red, self.green, self.blue)

    def __repr__(self):
        return f"Color({self.red}, {self.green}, {self.blue})"


Color.Black = Color(0, 0, 0)
Color.White = Color(1, 1, 1)


# ----------------------------------------------------------------------------- matrix.zig
class Mat4:
    """Row-major 4x4 (matrix.zig:99-103).  The fluent builders pre-multiply:
    m.translate(..) is T*m (matrix.zig:116-125), so identity().scale().translate() is T*S."""

    __slots__ = ("mat",)

    def __init__(self, rows):
        self.mat = [[float(v) for v in row] for row in rows]

    @staticmethod
    def identity():
        return Mat4([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])

    def mult(self, other):  # matrix.zig:197-213
        a, b = self.mat, other.mat
        return Mat4(
            [
                [
                    a[r][0] * b[0][c] + a[r][1] * b[1][c] + a[r][2] * b[2][c] + a[r][3] * b[3][c]
                    for c in range(4)
                ]
                for r in range(4)
            ]
        )

    def mult_vec(self, v):  # matrix.zig:215-224
        m = self.mat
        return Vec4(
            m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + m[0][3] * v.w,
            m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + m[1][3] * v.w,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + m[2][3] * v.w,
            m[3][0] * v.x + m[3][1] * v.y + m[3][2] * v.z + m[3][3] * v.w,
        )

    def transpose(self):
        return Mat4([[self.mat[c][r] for c in range(4)] for r in range(4)])

    def translate(self, x, y, z):
        return Mat4([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]]).mult(self)

    def scale(self, x, y, z):
        return Mat4([[x, 0, 0, 0], [0, y, 0, 0], [0, 0, z, 0], [0, 0, 0, 1]]).mult(self)

    def rotate_x(self, rad):
        c, s = math.cos(rad), math.sin(rad)
        return Mat4([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]]).mult(self)

    def rotate_y(self, rad):
        c, s = math.cos(rad), math.sin(rad)
        return Mat4([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]]).mult(self)

    def rotate_z(self, rad):
        c, s = math.cos(rad), math.sin(rad)
        return Mat4([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).mult(self)

    def shear(self, xy, xz, yx, yz, zx, zy):
        return Mat4([[1, xy, xz, 0], [yx, 1, yz, 0], [zx, zy, 1, 0], [0, 0, 0, 1]]).mult(self)

    def flat(self):
        return [v for row in self.mat for v in row]

    def __repr__(self):
        return f"Mat4({self.mat})"


# ----------------------------------------------------------------------------- pattern.zig
class Pattern:
    """Pattern (pattern.zig:64-75).  kind is one of 'point', 'stripe', 'gradient', 'ring',
    'checkers' in the declaration order of the reference's union."""

    KINDS = {
        "point": D.RTC_PATTERN_POINT,
        "stripe": D.RTC_PATTERN_STRIPE,
        "gradient": D.RTC_PATTERN_GRADIENT,
        "ring": D.RTC_PATTERN_RING,
        "checkers": D.RTC_PATTERN_CHECKERS,
    }

    def __init__(self, kind="point", a=None, b=None, transform=None):
        if kind not in self.KINDS:
            raise ValueError(f"unknown pattern kind {kind!r}")
        self.kind = kind
        self.a = a if a is not None else Color(0, 0, 0)
        self.b = b if b is not None else Color(0, 0, 0)
        self.transform = transform if transform is not None else Mat4.identity()


# ----------------------------------------------------------------------------- material.zig
class Material:
    def __init__(
        self,
        color=None,
        ambient=0.1,
        diffuse=0.9,
        specular=0.9,
        shininess=200.0,
        reflective=0.0,
        transparency=0.0,
        refractive_index=1.0,
        pattern=None,
    ):
        self.color = color if color is not None else Color(1, 1, 1)
        self.ambient = float(ambient)
        self.diffuse = float(diffuse)
        self.specular = float(specular)
        self.shininess = float(shininess)
        self.reflective = float(reflective)
        self.transparency = float(transparency)
        self.refractive_index = float(refractive_index)
        self.pattern = pattern


# ----------------------------------------------------------------------------- shape.zig
class Shape:
    """Shape (shape.zig:73-82): geo is 'sphere' (the default) or 'plane'; 'cube' is the library's
    extension beyond the reference (RTC_GEO_CUBE)."""

    GEOS = {"sphere": D.RTC_GEO_SPHERE, "plane": D.RTC_GEO_PLANE, "cube": D.RTC_GEO_CUBE}

    def __init__(self, geo="sphere", transform=None, material=None):
        if geo not in self.GEOS:
            raise ValueError(f"unknown geo {geo!r}")
        self.geo = geo
        self.transform = transform if transform is not None else Mat4.identity()
        self.material = material if material is not None else Material()


def init_glass_sphere():  # calc.zig:379-384
    return Shape("sphere", material=Material(transparency=1.0, refractive_index=1.5))


# ----------------------------------------------------------------------------- light.zig
class PointLight:
    def __init__(self, position=None, intensity=None):
        self.position = position if position is not None else init_point(0, 0, 0)
        self.intensity = intensity if intensity is not None else Color(1, 1, 1)


# ----------------------------------------------------------------------------- world.zig
class World:
    def __init__(self):
        self.objects = []
        self.light = PointLight()

    @staticmethod
    def init_default():  # world.zig:30-54
        w = World()
        w.light = PointLight(init_point(-10, 10, -10), Color(1, 1, 1))
        w.objects.append(Shape("sphere", material=Material(color=Color(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)))
        w.objects.append(Shape("sphere", transform=Mat4.identity().scale(0.5, 0.5, 0.5)))
        return w

    # ---- flattening into the C-ABI descriptors (what the Zig glue does with @cImport)
    def shape_descs(self):
        arr = (D.RtcShapeDesc * max(1, len(self.objects)))()
        for i, s in enumerate(self.objects):
            _fill_shape(arr[i], s)
        return arr

    def light_desc(self):
        d = D.RtcLightDesc()
        d.position[:] = self.light.position.as_tuple()
        d.intensity[:] = self.light.intensity.as_tuple()
        return d


def _fill_shape(d, s):
    d.geo = Shape.GEOS[s.geo]
    d.transform[:] = s.transform.flat()
    m, dm = s.material, d.material
    dm.color[:] = m.color.as_tuple()
    dm.ambient = m.ambient
    dm.diffuse = m.diffuse
    dm.specular = m.specular
    dm.shininess = m.shininess
    dm.reflective = m.reflective
    dm.transparency = m.transparency
    dm.refractive_index = m.refractive_index
    if m.pattern is None:
        dm.has_pattern = 0
        dm.pattern.transform[:] = Mat4.identity().flat()
    else:
        dm.has_pattern = 1
        dm.pattern.kind = Pattern.KINDS[m.pattern.kind]
        dm.pattern.a[:] = m.pattern.a.as_tuple()
        dm.pattern.b[:] = m.pattern.b.as_tuple()
        dm.pattern.transform[:] = m.pattern.transform.flat()


# ----------------------------------------------------------------------------- calc.zig:721-738
def view_transform(frm, to, up):
    forward = to.sub(frm).normalize()
    left = forward.cross(up.normalize())
    true_up = left.cross(forward)
    orientation = Mat4(
        [
            [left.x, left.y, left.z, 0],
            [true_up.x, true_up.y, true_up.z, 0],
            [-forward.x, -forward.y, -forward.z, 0],
            [0, 0, 0, 1],
        ]
    )
    translation = Mat4.identity().translate(-frm.x, -frm.y, -frm.z)
    return orientation.mult(translation)


# ----------------------------------------------------------------------------- canvas.zig
class Canvas:
    """Row-major framebuffer of f64 RGB (canvas.zig:4-39): pixels[y, x] = (red, green, blue)."""

    def __init__(self, width, height, pixels=None):
        self.width, self.height = int(width), int(height)
        if pixels is None:
            pixels = np.zeros((self.height, self.width, 3), dtype=np.float64)
        assert pixels.shape == (self.height, self.width, 3) and pixels.dtype == np.float64
        self.pixels = pixels

    def set(self, x, y, color):
        assert 0 <= x < self.width and 0 <= y < self.height
        self.pixels[y, x] = color.as_tuple()

    def at(self, x, y):
        assert 0 <= x < self.width and 0 <= y < self.height
        return Color(*self.pixels[y, x])


# ----------------------------------------------------------------------------- ppm.zig
def byte_color_values(values):
    """byteColorValue (ppm.zig:6-9) over an array: trunc(clamp(v*256, 0, 255)) as u8.
    NaN takes the clamp's upper bound, as Zig's min/max do."""
    v = np.asarray(values, dtype=np.float64) * 256.0
    v = np.where(v < 255.0, v, 255.0)
    v = np.where(v > 0.0, v, 0.0)
    return v.astype(np.uint8)


def canvas_to_ppm(canvas, writer):
    """canvasToPPM (ppm.zig:11-38): ASCII P3, five pixels per line, newline after each row."""
    q = byte_color_values(canvas.pixels)
    out = ["P3\n", f"{canvas.width} {canvas.height}\n", "255\n"]
    for y in range(canvas.height):
        row = q[y]
        parts = []
        for x in range(canvas.width):
            if x > 0:
                parts.append("\n" if x % 5 == 0 else " ")
            parts.append(f"{row[x, 0]} {row[x, 1]} {row[x, 2]}")
        parts.append("\n")
        out.append("".join(parts))
    writer.write("".join(out))


# ----------------------------------------------------------------------------- camera.zig
class Camera:
    def __init__(self, hsize, vsize, field_of_view):  # Camera.init, camera.zig:29-46
        self.hsize, self.vsize = int(hsize), int(vsize)
        self.field_of_view = float(field_of_view)
        half_view = math.tan(self.field_of_view * 0.5)
        aspect = float(self.hsize) / float(self.vsize)
        self.half_width = half_view if aspect >= 1.0 else half_view * aspect
        self.half_height = half_view if aspect < 1.0 else half_view / aspect
        self.pixel_size = (2.0 * self.half_width) / float(self.hsize)
        self.transform = Mat4.identity()

    def desc(self):
        d = D.RtcCameraDesc()
        d.hsize, d.vsize = self.hsize, self.vsize
        d.field_of_view = self.field_of_view
        d.half_width, d.half_height, d.pixel_size = self.half_width, self.half_height, self.pixel_size
        d.transform[:] = self.transform.flat()
        return d

    def render(self, world, depth=5, stats=None, integrator=0):
        """Camera.render (camera.zig:74-89) through the HIP library.  The reference's
        hard-coded depth is 5 (camera.zig:83).  Raises RtcError on any non-zero status,
        as the Zig glue maps it to an error from `render`'s `!Canvas`.  `integrator`:
        RTC_INTEGRATOR_POSTORDER (0, the reference's order of summation) or RTC_INTEGRATOR_WEIGHTED (1)."""
        from . import native

        with native.Scene(world.shape_descs(), len(world.objects), world.light_desc()) as scene:
            if integrator:
                scene.set_integrator(integrator)
            pixels = scene.render(self.desc(), depth, stats)
        return Canvas(self.hsize, self.vsize, pixels)
