"""The plumbing configuration: draw_sphere.zig:38-83 (one Phong-lit sphere, 100x100, pixel-corner
sampling, no shadows) restated in the oracle.  The reference holds no known answer for this
program (and the file no longer compiles against its own revision), so the restatement is checked
through what its pieces are pinned by: the rays against hand arithmetic, the hit mask against the
sphere's silhouette, the colours against calc.worldColorAt of the same rays (which differs only in
lighting at over_point, 1e-4 off the surface, instead of the hit point).  Parity of the loop as
a whole against a Zig binary: unpinned."""
import math

import numpy as np

import oracle_lib as O
from raytracer_challenge_amd import scenes


def test_draw_sphere_rays_and_silhouette():
    n = 100
    canvas, rays = O.draw_sphere(n)
    assert canvas.shape == (n, n, 3) and rays.shape == (n * n, 6)
    # draw_sphere.zig:42-43,52,58: pixel corners, wall of size 7 at z = 10, eye at (0, 0, -5)
    px = 7.0 / n
    for x, y in ((0, 0), (50, 50), (99, 3), (17, 88)):
        wx, wy = -3.5 + x * px, 3.5 - y * px
        d = np.array([wx, wy, 15.0])
        d = d / math.sqrt((wx * wx + wy * wy) + 15.0 * 15.0)
        assert np.array_equal(rays[y * n + x, :3], (0, 0, -5))
        assert np.allclose(rays[y * n + x, 3:], d, rtol=0, atol=1e-15)
    # the unit sphere seen from distance 5: rays with |d x o| < 1, i.e. tan(angle) < 1/sqrt(24)
    xs = -3.5 + np.arange(n) * px
    ys = 3.5 - np.arange(n) * px
    r2 = (xs[None, :] ** 2 + ys[:, None] ** 2) / 15.0 ** 2
    inside = r2 < 1.0 / 24.0
    lit = np.any(canvas != 0, axis=2)
    edge = np.abs(r2 - 1.0 / 24.0) < 1e-12
    assert np.array_equal(lit[~edge], inside[~edge])
    # pixel (50, 50) looks down the axis: ambient + diffuse + a weak highlight of colour (1, 0.2, 1)
    c = canvas[50, 50]
    assert c[0] == c[2] and c[1] < c[0] and 0.1 < c[0] < 1.9


def test_draw_sphere_agrees_with_world_color_at():
    canvas, rays = O.draw_sphere(100)
    w = scenes.draw_sphere_world()
    for mode in (O.LITERAL, O.STREAMING):
        got = O.color_at(w, rays, 0, mode).reshape(100, 100, 3)
        assert np.array_equal(np.any(got != 0, axis=2), np.any(canvas != 0, axis=2))
        # lighting at over_point instead of the point: the light vector turns by ~1e-5 rad
        assert np.abs(got - canvas).max() < 2e-3
