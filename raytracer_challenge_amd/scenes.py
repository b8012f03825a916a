"""Scene builders for the configurations the render path is measured on.

  draw_world_scene()  — the reference's production scene (draw_world.zig:29-192): the
                        Ch. 11 reflection/refraction room, 6 planes + 7 spheres.
  default_world()     — World.initDefault (world.zig:30-54), fixture of most reference tests.
  synthetic_scene()   — 10 k random spheres + 10 planes, mixed reflective / glass / patterned,
                        from a fixed seed (no counterpart in the reference; it exists to
                        load the brute-force sweep).
"""
import math

from .host import (
    Camera,
    Color,
    Mat4,
    Material,
    Pattern,
    PointLight,
    Shape,
    World,
    init_point,
    init_vector,
    view_transform,
)


def default_world():
    return World.init_default()


def draw_sphere_world():
    """The world of draw_sphere.zig:30-48 (the Ch. 6 plumbing program): one unit sphere of colour
    (1, 0.2, 1) with the default material, one white light at (-10, 10, -10)."""
    world = World()
    world.light = PointLight(init_point(-10, 10, -10), Color(1, 1, 1))
    world.objects.append(Shape("sphere", Mat4.identity(), Material(color=Color(1.0, 0.2, 1.0))))
    return world


def draw_world_scene(hsize=900, vsize=450):
    """World and camera of draw_world.zig:29-192.  The reference renders 900x450
    (draw_world.zig:32); other resolutions only change hsize/vsize, the aspect rule of
    Camera.init (camera.zig:33-34) does the rest."""
    world = World()
    camera = Camera(hsize, vsize, 1.152)
    # the reference builds from/to with initVector (w = 0); only differences of them are used
    frm = init_vector(-2.6, 1.5, -3.9)
    to = init_vector(-0.6, 1, -0.8)
    up = init_vector(0, 1, 0)
    camera.transform = view_transform(frm, to, up)

    world.light = PointLight(init_point(-4.9, 4.9, -1), Color(1, 1, 1))

    wall_pattern = Pattern(
        "stripe",
        Color(0.45, 0.45, 0.45),
        Color(0.55, 0.55, 0.55),
        Mat4.identity().scale(0.25, 0.25, 0.25).rotate_y(1.5708),
    )

    def wall_material():
        return Material(pattern=wall_pattern, ambient=0, diffuse=0.4, specular=0, reflective=0.3)

    floor_pattern = Pattern(
        "checkers", Color(0.35, 0.35, 0.35), Color(0.65, 0.65, 0.65), Mat4.identity().rotate_y(0.31415)
    )
    # floor
    world.objects.append(Shape("plane", material=Material(pattern=floor_pattern, specular=0, reflective=0.4)))
    # ceiling
    world.objects.append(
        Shape(
            "plane",
            transform=Mat4.identity().translate(0, 5, 0),
            material=Material(color=Color(0.8, 0.8, 0.8), ambient=0.3, specular=0),
        )
    )
    # west, east, north, south walls
    world.objects.append(
        Shape("plane", Mat4.identity().rotate_y(1.5708).rotate_z(1.5708).translate(-5, 0, 0), wall_material())
    )
    world.objects.append(
        Shape("plane", Mat4.identity().rotate_y(1.5708).rotate_z(1.5708).translate(5, 0, 0), wall_material())
    )
    world.objects.append(Shape("plane", Mat4.identity().rotate_x(1.5708).translate(0, 0, 5), wall_material()))
    world.objects.append(Shape("plane", Mat4.identity().rotate_x(1.5708).translate(0, 0, -5), wall_material()))

    # background balls
    for scale, pos, col in (
        (0.4, (4.6, 0.4, 1.0), (0.8, 0.5, 0.3)),
        (0.3, (4.7, 0.3, 0.4), (0.9, 0.4, 0.5)),
        (0.5, (-1, 0.5, 4.5), (0.4, 0.9, 0.6)),
        (0.3, (-1.7, 0.3, 4.7), (0.4, 0.6, 0.9)),
    ):
        world.objects.append(
            Shape(
                "sphere",
                Mat4.identity().scale(scale, scale, scale).translate(*pos),
                Material(color=Color(*col), shininess=50),
            )
        )
    # red sphere
    world.objects.append(
        Shape(
            "sphere",
            Mat4.identity().translate(-0.6, 1, 0.6),
            Material(color=Color(1.0, 0.3, 0.2), specular=0.4, shininess=5),
        )
    )
    # blue glass sphere, green glass sphere
    for scale, pos, col in ((0.7, (0.6, 0.7, -0.6), (0, 0, 0.2)), (0.5, (-0.7, 0.5, -0.8), (0, 0.2, 0))):
        world.objects.append(
            Shape(
                "sphere",
                Mat4.identity().scale(scale, scale, scale).translate(*pos),
                Material(
                    color=Color(*col),
                    ambient=0,
                    diffuse=0.4,
                    specular=0.9,
                    shininess=300,
                    reflective=0.9,
                    transparency=0.9,
                    refractive_index=1.5,
                ),
            )
        )
    return world, camera


class SplitMix64:
    """64-bit SplitMix generator (Steele, Lea, Flood 2014); integers only, so the stream is
    identical on every machine."""

    def __init__(self, seed):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        # 53 random bits -> [0, 1)
        u = (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def choice(self, seq):
        return seq[self.next_u64() % len(seq)]


def synthetic_scene(hsize=3840, vsize=2160, n_spheres=9990, seed=0x5EED):
    """Synthetic load: n_spheres random spheres (centres uniform in [-40,40]x[0.2,20]x[-40,40],
    radius uniform in [0.2,1.0], transform = T*S) plus 10 planes (a floor at y = 0 and nine far
    bounding planes).  25 % of the spheres are mirrors (reflective ~ U[0.2,0.9]), 15 % glass
    (reflective = transparency = 0.9, refractive index 1.5), the rest diffuse with a random
    colour and shininess in {5, 50, 200, 300}; 30 % carry a random stripe / gradient / ring /
    checkers pattern with a scale + rotateY transform.  Camera at (0,12,-60) looking at
    (0,4,0), field of view pi/3; one point light at (-30,50,-30)."""
    rng = SplitMix64(seed)
    world = World()
    world.light = PointLight(init_point(-30, 50, -30), Color(1, 1, 1))

    def rand_color():
        return Color(rng.uniform(0.1, 1.0), rng.uniform(0.1, 1.0), rng.uniform(0.1, 1.0))

    def rand_pattern():
        kind = rng.choice(("stripe", "gradient", "ring", "checkers"))
        s = rng.uniform(0.1, 0.6)
        return Pattern(kind, rand_color(), rand_color(), Mat4.identity().scale(s, s, s).rotate_y(rng.uniform(0, math.pi)))

    # planes first: floor, ceiling far above, eight far walls in a ring (all distinct)
    world.objects.append(
        Shape(
            "plane",
            material=Material(
                pattern=Pattern("checkers", Color(0.3, 0.3, 0.3), Color(0.7, 0.7, 0.7), Mat4.identity().scale(2, 2, 2)),
                specular=0,
                reflective=0.2,
            ),
        )
    )
    world.objects.append(
        Shape("plane", Mat4.identity().translate(0, 200, 0), Material(color=Color(0.5, 0.7, 1.0), ambient=0.6, specular=0))
    )
    for k in range(8):
        ang = k * (math.pi / 4)
        world.objects.append(
            Shape(
                "plane",
                Mat4.identity().rotate_x(math.pi / 2).translate(0, 0, 150 + k).rotate_y(ang),
                Material(color=Color(0.4 + 0.05 * k, 0.5, 0.6), ambient=0.4, diffuse=0.5, specular=0),
            )
        )

    for _ in range(n_spheres):
        cx, cy, cz = rng.uniform(-40, 40), rng.uniform(0.2, 20), rng.uniform(-40, 40)
        r = rng.uniform(0.2, 1.0)
        transform = Mat4.identity().scale(r, r, r).translate(cx, cy, cz)
        u = rng.uniform()
        if u < 0.25:
            mat = Material(color=rand_color(), reflective=rng.uniform(0.2, 0.9), shininess=rng.choice((5, 50, 200, 300)))
        elif u < 0.40:
            mat = Material(
                color=Color(rng.uniform(0, 0.2), rng.uniform(0, 0.2), rng.uniform(0, 0.2)),
                ambient=0,
                diffuse=0.4,
                specular=0.9,
                shininess=300,
                reflective=0.9,
                transparency=0.9,
                refractive_index=1.5,
            )
        else:
            mat = Material(color=rand_color(), shininess=rng.choice((5, 50, 200, 300)))
        if rng.uniform() < 0.30:
            mat.pattern = rand_pattern()
        world.objects.append(Shape("sphere", transform, mat))

    camera = Camera(hsize, vsize, math.pi / 3)
    camera.transform = view_transform(init_point(0, 12, -60), init_point(0, 4, 0), init_vector(0, 1, 0))
    return world, camera
