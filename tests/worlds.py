"""Seeded random worlds and hand-built edge-case worlds shared by the parity tests."""
import math

import numpy as np

from raytracer_challenge_amd import (Camera, Color, Mat4, Material, Pattern, PointLight, Shape, World, init_point,
                                     init_vector, view_transform)


def random_world(seed, n_spheres=6, n_planes=2, glass=True, n_cubes=0, axis_only=False):
    """axis_only: the spheres are ellipsoids along the axes (scale and translation only: diagonal inverses) and every
    shininess is an integer, as in the reference's own scenes; such worlds of at most 32 shapes are rendered by the
    kernels without the general kinds."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    w = World()
    w.light = PointLight(init_point(u(-8, 8), u(3, 9), u(-8, -2)), Color(u(0.5, 1.2), u(0.5, 1.2), u(0.5, 1.2)))

    def rand_pattern():
        kind = ("point", "stripe", "gradient", "ring", "checkers")[int(rng.integers(0, 5))]
        t = Mat4.identity().scale(u(0.1, 0.9), u(0.1, 0.9), u(0.1, 0.9)).rotate_z(u(0, 3)).translate(u(-1, 1), 0, 0)
        return Pattern(kind, Color(u(0, 1), u(0, 1), u(0, 1)), Color(u(0, 1), u(0, 1), u(0, 1)), t)

    def rand_material(i):
        m = Material(color=Color(u(0, 1), u(0, 1), u(0, 1)), ambient=u(0, 0.3), diffuse=u(0.2, 1), specular=u(0, 1),
                     shininess=float(rng.choice([5, 10, 50, 200, 300, 12.5])))
        r = rng.uniform()
        if r < 0.3:
            m.reflective = u(0.1, 1.0)
        elif r < 0.6 and glass:
            m.reflective = u(0.1, 0.9)
            m.transparency = u(0.3, 1.0)
            m.refractive_index = u(1.0, 2.5)
        elif r < 0.7 and glass:
            m.transparency = u(0.3, 1.0)
            m.refractive_index = u(1.0, 2.5)
        if rng.uniform() < 0.5:
            m.pattern = rand_pattern()
        return m

    for i in range(n_planes):
        t = Mat4.identity().rotate_x(u(-0.3, 0.3)).rotate_z(u(-0.3, 0.3)).translate(0, u(-3, -1) if i == 0 else u(6, 9), 0)
        w.objects.append(Shape("plane", t, rand_material(i)))
    for i in range(n_spheres):
        if axis_only:
            t = Mat4.identity().scale(u(0.3, 1.5), u(0.3, 1.5), u(0.3, 1.5)).translate(u(-3, 3), u(-1, 3), u(-2, 4))
        else:
            t = (Mat4.identity().scale(u(0.3, 1.5), u(0.3, 1.5), u(0.3, 1.5)).rotate_y(u(0, 3))
                 .shear(u(-0.2, 0.2), 0, 0, u(-0.2, 0.2), 0, 0).translate(u(-3, 3), u(-1, 3), u(-2, 4)))
        w.objects.append(Shape("sphere", t, rand_material(i)))
    for i in range(n_cubes):  # the library's extension beyond the reference's two kinds
        t = (Mat4.identity().scale(u(0.2, 1.2), u(0.2, 1.2), u(0.2, 1.2)).rotate_y(u(0, 3)).rotate_x(u(-0.5, 0.5))
             .shear(u(-0.2, 0.2), 0, 0, u(-0.2, 0.2), 0, 0).translate(u(-3, 3), u(-1, 3), u(-2, 4)))
        w.objects.append(Shape("cube", t, rand_material(i)))
    cam = Camera(40, 24, u(0.7, 1.4))
    cam.transform = view_transform(init_point(u(-1, 1), u(0.5, 2), u(-9, -6)), init_point(0, 1, 0), init_vector(0, 1, 0))
    return w, cam


def cube_world():
    """Cubes (the extension beyond the reference) where they meet the rest of the path: a glass cube
    around a glass sphere, a glass cube inside a glass sphere, two coincident cubes and a cube face
    coincident with a plane (ties decided by list order), a mirror cube, a patterned cube, and a huge
    glass cube that contains the camera (an open container from the first ray on)."""
    w = World()
    w.light = PointLight(init_point(-4, 7, -9), Color(1, 1, 1))
    glass = lambda ri, c: Material(color=Color(*c), diffuse=0.2, transparency=0.85, reflective=0.4, refractive_index=ri)
    w.objects.append(Shape("plane", Mat4.identity().translate(0, -1, 0),
                           Material(pattern=Pattern("checkers", Color(0.9, 0.9, 0.9), Color(0.2, 0.2, 0.3)), reflective=0.2)))
    w.objects.append(Shape("cube", Mat4.identity().scale(30, 30, 30), glass(1.05, (0.02, 0.02, 0.05))))
    w.objects.append(Shape("cube", Mat4.identity().rotate_y(0.5).translate(-2, 0.2, 1), glass(1.5, (0.1, 0.05, 0.05))))
    w.objects.append(Shape("sphere", Mat4.identity().scale(0.6, 0.6, 0.6).translate(-2, 0.2, 1), glass(2.0, (0.05, 0.1, 0.05))))
    w.objects.append(Shape("sphere", Mat4.identity().scale(1.4, 1.4, 1.4).translate(2, 0.6, 1), glass(1.3, (0.05, 0.05, 0.1))))
    w.objects.append(Shape("cube", Mat4.identity().scale(0.5, 0.5, 0.5).rotate_z(0.4).translate(2, 0.6, 1), glass(2.2, (0.2, 0.1, 0.0))))
    # the bottom face of this cube lies in the floor plane; the second cube is the same cube again
    w.objects.append(Shape("cube", Mat4.identity().scale(0.5, 0.5, 0.5).translate(0, -0.5, -1),
                           Material(color=Color(0.8, 0.3, 0.2), transparency=0.5, refractive_index=1.4)))
    w.objects.append(Shape("cube", Mat4.identity().scale(0.5, 0.5, 0.5).translate(0, -0.5, -1),
                           Material(color=Color(0.2, 0.3, 0.8), reflective=0.6)))
    w.objects.append(Shape("cube", Mat4.identity().scale(0.4, 1.2, 0.4).rotate_y(0.8).translate(0.3, 0.2, 3),
                           Material(color=Color(1, 1, 1), reflective=0.9, diffuse=0.1)))
    w.objects.append(Shape("cube", Mat4.identity().scale(0.7, 0.3, 0.7).rotate_x(0.3).translate(-0.5, 1.5, 2),
                           Material(pattern=Pattern("stripe", Color(1, 0.8, 0.1), Color(0.1, 0.5, 0.3),
                                                    Mat4.identity().scale(0.2, 0.2, 0.2)), shininess=50.0)))
    cam = Camera(64, 40, 1.0)
    cam.transform = view_transform(init_point(0.3, 1.2, -7), init_point(0, 0.3, 0), init_vector(0, 1, 0))
    return w, cam


def nested_glass_world():
    """Three nested / overlapping glass spheres and a glass plane (the n1/n2 set-up of
    calc.zig:400-446 plus a plane container) in front of a patterned backdrop."""
    w = World()
    w.light = PointLight(init_point(-6, 8, -10), Color(1, 1, 1))
    a = Shape("sphere", Mat4.identity().scale(2, 2, 2),
              Material(color=Color(0.1, 0.1, 0.1), transparency=0.9, reflective=0.5, refractive_index=1.5, diffuse=0.2))
    b = Shape("sphere", Mat4.identity().translate(0, 0, -0.25),
              Material(color=Color(0.1, 0.2, 0.1), transparency=0.8, reflective=0.3, refractive_index=2.0, diffuse=0.2))
    c = Shape("sphere", Mat4.identity().translate(0, 0, 0.25),
              Material(color=Color(0.2, 0.1, 0.1), transparency=1.0, refractive_index=2.5, diffuse=0.2))
    glass_plane = Shape("plane", Mat4.identity().rotate_x(math.pi / 2).translate(0, 0, -3),
                        Material(color=Color(0.05, 0.05, 0.1), transparency=0.9, reflective=0.2, refractive_index=1.3,
                                 diffuse=0.1))
    backdrop = Shape("plane", Mat4.identity().rotate_x(math.pi / 2).translate(0, 0, 6),
                     Material(pattern=Pattern("checkers", Color(0.9, 0.9, 0.9), Color(0.1, 0.1, 0.4)), ambient=0.4,
                              specular=0))
    floor = Shape("plane", Mat4.identity().translate(0, -2.5, 0),
                  Material(pattern=Pattern("ring", Color(0.8, 0.3, 0.3), Color(0.3, 0.8, 0.3)), reflective=0.3))
    w.objects += [a, b, c, glass_plane, backdrop, floor]
    cam = Camera(48, 32, 0.9)
    cam.transform = view_transform(init_point(0.3, 0.4, -7), init_point(0, 0, 0), init_vector(0, 1, 0))
    return w, cam


def tie_world():
    """Coincident surfaces: every root of one shape ties with a root of another, so the hit
    and the container order are decided by list order alone (stable sort, ray.zig:75-92).
    Mixed kinds (sheared / axis-aligned, spheres / planes) so that shapes that tie sit in
    different groups of the device's sweep table."""
    w = World()
    w.light = PointLight(init_point(-5, 6, -8), Color(1, 1, 1))
    big = Mat4.identity().scale(1.5, 1.5, 1.5)
    # same geometry, different materials, glass in between opaque: list order decides
    w.objects.append(Shape("sphere", big, Material(color=Color(0.9, 0.1, 0.1), transparency=0.8, reflective=0.3,
                                                     refractive_index=1.4)))
    w.objects.append(Shape("sphere", big, Material(color=Color(0.1, 0.9, 0.1), transparency=0.6, refractive_index=2.0)))
    # the same sphere again as a general (rotated) matrix: rotation of a sphere leaves it in place
    w.objects.append(Shape("sphere", Mat4.identity().scale(1.5, 1.5, 1.5).rotate_y(0.7),
                           Material(color=Color(0.1, 0.1, 0.9), transparency=0.7, reflective=0.2, refractive_index=1.2,
                                    pattern=Pattern("stripe", Color(1, 1, 0), Color(0, 1, 1)))))
    # coincident planes: axis-aligned and (numerically) the same plane through a general matrix
    w.objects.append(Shape("plane", Mat4.identity().translate(0, -2, 0),
                           Material(color=Color(0.8, 0.8, 0.2), reflective=0.5)))
    w.objects.append(Shape("plane", Mat4.identity().shear(0, 0, 0, 0, 0, 0).rotate_y(0.3).translate(0, -2, 0),
                           Material(color=Color(0.2, 0.8, 0.8), transparency=0.5, refractive_index=1.1)))
    # a glass plane behind the camera (a container that never closes) tied with another one
    w.objects.append(Shape("plane", Mat4.identity().rotate_x(math.pi / 2).translate(0, 0, -9),
                           Material(color=Color(0.5, 0.5, 0.5), transparency=0.9, refractive_index=1.7)))
    w.objects.append(Shape("plane", Mat4.identity().rotate_x(math.pi / 2).translate(0, 0, -9),
                           Material(color=Color(0.5, 0.2, 0.5), transparency=0.9, refractive_index=1.9)))
    w.objects.append(Shape("sphere", Mat4.identity().scale(0.5, 0.5, 0.5).translate(0.2, 0.1, 0),
                           Material(color=Color(1, 1, 1), transparency=0.9, reflective=0.9, refractive_index=1.5)))
    cam = Camera(48, 32, 0.8)
    cam.transform = view_transform(init_point(0.2, 0.5, -7), init_point(0, 0, 0), init_vector(0, 1, 0))
    return w, cam


def grazing_world_and_rays(seed, n_rays=6000):
    """Spheres of very different size, stretch and distance from the origin, and rays that graze
    them: through surface points along tangent directions, displaced along the normal by
    0, +-1e-15 ... +-1e-6 of the shape's size, starting near and far.  Whether such a line hits
    is decided in the last bits of the reference's discriminant; the device's bounding-sphere
    early-out must never get ahead of it."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    w = World()
    w.light = PointLight(init_point(-20, 30, -25), Color(1, 1, 1))
    transforms = []
    for kind in range(10):
        size = 10.0 ** u(-3, 3) if kind % 3 else u(0.5, 2)
        far = 10.0 ** u(0, 5) if kind % 2 else u(0, 5)
        c = rng.normal(size=3)
        c = c / np.linalg.norm(c) * far
        m = Mat4.identity()
        if kind < 4:      # uniform scale + translation (diagonal inverse)
            m = m.scale(size, size, size)
        elif kind < 7:    # non-uniform scale
            m = m.scale(size * u(0.2, 1), size * u(0.2, 1), size * u(0.2, 1))
        else:             # rotated, sheared ellipsoid
            m = (m.scale(size * u(0.3, 1), size * u(0.3, 1), size * u(0.3, 1)).rotate_x(u(0, 3)).rotate_z(u(0, 3))
                 .shear(u(-0.3, 0.3), 0, 0, 0, u(-0.3, 0.3), 0))
        m = m.translate(float(c[0]), float(c[1]), float(c[2]))
        transforms.append((m, size))
        w.objects.append(Shape("sphere", m, Material(color=Color(u(0.2, 1), u(0.2, 1), u(0.2, 1)), ambient=0.5,
                                                       reflective=0.3 if kind % 4 == 0 else 0.0)))
    rays = np.empty((n_rays, 6))
    for k in range(n_rays):
        m, size = transforms[k % len(transforms)]
        a = np.array(m.mat, dtype=np.float64).reshape(4, 4)
        p_local = rng.normal(size=3)
        p_local /= np.linalg.norm(p_local)
        p = a[:3, :3] @ p_local + a[:3, 3]
        n = np.linalg.inv(a[:3, :3]).T @ p_local   # surface normal in world space
        n /= np.linalg.norm(n)
        t = np.cross(n, rng.normal(size=3))
        t /= np.linalg.norm(t)
        eps = (0.0, 1e-15, -1e-15, 1e-13, -1e-13, 1e-11, -1e-11, 1e-9, -1e-9, 1e-6, -1e-6)[int(rng.integers(0, 11))]
        back = size * 10.0 ** u(-1, 3)
        o = p + n * (eps * size) - t * back
        d = t * (1.0 if k % 2 else 10.0 ** u(-2, 2))   # refracted rays are not normalised either
        rays[k, :3] = o
        rays[k, 3:] = d
    return w, rays


def shadow_grazing_world_and_rays(seed, n_rays=2500):
    """Shadow rays that graze spheres: rays aimed at floor points q chosen so that the line from q
    to the light touches a sphere (tangency found by bisection on the discriminant in double
    precision), then turned away from or into the sphere by 0, +-1e-15 ... +-1e-6 of the sphere's
    angular size.  Whether such a point is in shadow is decided in the last bits of the
    reference's discriminant; the device's test from the light must never get ahead of it.
    Tiny, large, stretched and sheared spheres, near and far from the light."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    w = World()
    light = np.array([u(-3, 3), u(20, 30), u(-3, 3)])
    w.light = PointLight(init_point(*light), Color(1, 1, 1))
    floor_y = -3.0
    w.objects.append(Shape("plane", Mat4.identity().translate(0, floor_y, 0),
                           Material(color=Color(0.9, 0.9, 0.9), specular=0.0, reflective=0.2 if seed % 2 else 0.0)))
    mats = []
    for kind in range(9):
        size = 10.0 ** u(-3, 0.7) if kind % 3 else u(0.5, 2)
        m = Mat4.identity()
        if kind < 3:
            m = m.scale(size, size, size)
        elif kind < 6:
            m = m.scale(size * u(0.2, 1), size * u(0.2, 1), size * u(0.2, 1))
        else:
            m = (m.scale(size * u(0.3, 1), size * u(0.3, 1), size * u(0.3, 1)).rotate_x(u(0, 3)).rotate_z(u(0, 3))
                 .shear(u(-0.3, 0.3), 0, 0, 0, u(-0.3, 0.3), 0))
        c = np.array([u(-12, 12), u(0, 15) if kind % 2 else u(18, 19.9), u(-12, 12)])
        m = m.translate(float(c[0]), float(c[1]), float(c[2]))
        w.objects.append(Shape("sphere", m, Material(color=Color(u(0.2, 1), u(0.2, 1), u(0.2, 1)), ambient=0.3)))
        a = np.array(m.mat, dtype=np.float64).reshape(4, 4)
        mats.append((np.linalg.inv(a), c))

    def disc(inv, direction):
        lo = inv[:3, :3] @ light + inv[:3, 3]
        ld = inv[:3, :3] @ direction
        b = 2 * ld.dot(lo)
        return b * b - 4 * ld.dot(ld) * (lo.dot(lo) - 1)

    rays = np.empty((n_rays, 6))
    k = 0
    while k < n_rays:
        inv, c = mats[int(rng.integers(0, len(mats)))]
        axis0 = c - light
        axis0 /= np.linalg.norm(axis0)
        side = np.cross(axis0, rng.normal(size=3))
        side /= np.linalg.norm(side)
        turned = lambda alpha: np.cos(alpha) * axis0 + np.sin(alpha) * side
        lo_a, hi_a = 0.0, 1.0
        if disc(inv, turned(lo_a)) <= 0 or disc(inv, turned(hi_a)) >= 0:
            continue
        for _ in range(64):
            mid = 0.5 * (lo_a + hi_a)
            if disc(inv, turned(mid)) > 0:
                lo_a = mid
            else:
                hi_a = mid
        eps = (0.0, 1e-15, -1e-15, 1e-13, -1e-13, 1e-11, -1e-11, 1e-9, -1e-9, 1e-6, -1e-6)[int(rng.integers(0, 11))]
        direction = turned(lo_a * (1.0 + eps))
        if direction[1] > -1e-3:
            continue
        q = light + direction * ((floor_y - light[1]) / direction[1])
        eye = np.array([u(-5, 5), u(1, 12), u(-40, -25)])
        rays[k, :3] = eye
        rays[k, 3:] = (q - eye) / np.linalg.norm(q - eye)
        k += 1
    return w, rays


def touching_spheres_world_and_rays(seed, n_rays=4000):
    """Rows of spheres that touch, nearly touch or slightly overlap their neighbours (gaps of 0, +-1e-15 ... +-1e-6
    of the radius), more than 32 of them, with glass, mirrors and diffuse surfaces mixed; the light sits on the
    axis of the first row.  Rays run along the rows and start outside, inside, exactly on touching points and on
    surface points facing away from a neighbour, so that the hit of one sphere coincides with (or lies within
    rounding of) the entry of the next, spheres end exactly at a ray's origin, and shadow rays leave a surface
    point with an occluder touching it from behind.  Whatever a renderer leaves out because it lies "behind the
    origin" or "beyond the hit" is decided here in the last bits."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    w = World()
    w.light = PointLight(init_point(-9, 0, 0), Color(1, 1, 1))
    gaps = (0.0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3)
    centres = []
    for row in range(3):
        r = (1.0, 0.5, 2.0)[row]
        x = 0.0
        for k in range(13):
            gap = gaps[(k + row) % len(gaps)]
            c = np.array([x, 3.0 * row * (row + 1) / 2.0 * 1.5, 0.25 * row])
            centres.append((c, r))
            kind = (k + row) % 3
            if kind == 0:
                mat = Material(color=Color(u(0, 0.3), u(0, 0.3), u(0, 0.3)), diffuse=0.4, reflective=0.6, transparency=0.8,
                               refractive_index=1.0 + 0.1 * (k % 6))
            elif kind == 1:
                mat = Material(color=Color(u(0.2, 1), u(0.2, 1), u(0.2, 1)), reflective=u(0.2, 0.8))
            else:
                mat = Material(color=Color(u(0.2, 1), u(0.2, 1), u(0.2, 1)), ambient=0.3)
            m = Mat4.identity().scale(r, r, r)
            if k % 4 == 3:
                m = m.rotate_y(0.3 * k)  # the same sphere through a general matrix
            w.objects.append(Shape("sphere", m.translate(float(c[0]), float(c[1]), float(c[2])), mat))
            x += 2.0 * r * (1.0 + gap)
    w.objects.append(Shape("plane", Mat4.identity().translate(0, -3, 0), Material(color=Color(0.5, 0.5, 0.5), reflective=0.2)))
    rays = np.empty((n_rays, 6))
    axis = np.array([1.0, 0.0, 0.0])
    for k in range(n_rays):
        c, r = centres[int(rng.integers(0, len(centres)))]
        off = (0.0, 1e-15, -1e-15, 1e-12, 1e-9, -1e-9, 1e-6, 1e-3, 0.3)[int(rng.integers(0, 9))] * r
        side = np.array([0.0, 1.0, 0.0]) if k % 2 else np.array([0.0, 0.0, 1.0])
        mode = k % 5
        sign = 1.0 if (k // 5) % 2 else -1.0
        if mode == 0:    # from far outside, along the row
            o = c - sign * axis * (r * u(3, 40)) + side * off
        elif mode == 1:  # from the touching point with the neighbour
            o = c + sign * axis * r + side * off
        elif mode == 2:  # from the centre of a sphere
            o = c + side * off
        elif mode == 3:  # from the surface point facing away from where the ray goes (the sphere ends at the origin)
            o = c - sign * axis * r * (1.0 + off / r) + side * 0.0
        else:            # from a random surface point in a random direction
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            o = c + n * r * (1.0 + off / r)
        d = sign * axis
        if mode == 4:
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
        elif k % 3 == 0:
            d = d + side * u(-1e-3, 1e-3)
        rays[k, :3] = o
        rays[k, 3:] = d * (1.0 if k % 2 else 10.0 ** u(-2, 2))
    return w, rays
