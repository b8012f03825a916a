// TEST INFRASTRUCTURE ONLY — known-answer tests of the CPU oracle.
//
// Every case restates one `test "..."` block of the reference (file:line in the case
// name) with the reference's own inputs, expected values and tolerances
// (utils.zig:7-31: colours 1e-4, vectors/matrices 1e-5, scalars f64 epsilon).
// Prints one line per case ("ok <name>" / "FAIL <name>: ...") and exits non-zero on
// any failure.  Cases that walk a hand-built intersection list use the LITERAL
// integrator, as only it takes such a list.
#include "rtc_oracle.hpp"

#include <cstdio>
#include <functional>
#include <string>

using namespace orc;

static int g_fail = 0;
static std::string g_case;
static bool g_case_failed = false;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::printf("FAIL %s: %s (line %d)\n", g_case.c_str(), #cond, __LINE__); \
            g_case_failed = true;                                                    \
        }                                                                            \
    } while (0)

static bool approx(double e, double a, double tol) { return std::fabs(e - a) <= tol; }
#define CHECK_EPS(e, a) CHECK(approx((e), (a), F64_EPSILON))
#define CHECK_COLOR(e, a) \
    CHECK(approx((e).red, (a).red, 1e-4) && approx((e).green, (a).green, 1e-4) && approx((e).blue, (a).blue, 1e-4))
#define CHECK_VEC(e, a)                                                                              \
    CHECK(approx((e).x, (a).x, 1e-5) && approx((e).y, (a).y, 1e-5) && approx((e).z, (a).z, 1e-5) && \
          approx((e).w, (a).w, 1e-5))

static bool matApprox(const Mat4& e, const Mat4& a) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            if (!approx(e.mat[r][c], a.mat[r][c], 1e-5)) return false;
    return true;
}

static void run(const char* name, const std::function<void()>& fn) {
    g_case = name;
    g_case_failed = false;
    fn();
    if (g_case_failed)
        ++g_fail;
    else
        std::printf("ok %s\n", name);
}

static const double S2 = std::sqrt(2.0);
static const double PI = 3.14159265358979323846;

static Shape glassSphere() {  // calc.zig:379-384
    Shape s;
    s.material.transparency = 1.0;
    s.material.refractive_index = 1.5;
    return s;
}
static Shape planeShape() {
    Shape s;
    s.geo = GEO_PLANE;
    return s;
}

int main() {
    // ------------------------------------------------------------ vector.zig / color.zig
    run("vector.zig:208-259 length, normalize, dot, cross, reflect", [] {
        CHECK_VEC(initVector(1, 0, 0), initVector(4, 0, 0).normalize());
        CHECK(initVector(1, 2, 3).normalize().eql(initVector(1.0 / std::sqrt(14.0), 2.0 / std::sqrt(14.0), 3.0 / std::sqrt(14.0))));
        CHECK_EPS(1.0, initVector(1, 2, 3).normalize().length());
        CHECK_EPS(20.0, initVector(1, 2, 3).dot(initVector(2, 3, 4)));
        CHECK_VEC(initVector(-1, 2, -1), initVector(1, 2, 3).cross(initVector(2, 3, 4)));
        CHECK_VEC(initVector(1, -2, 1), initVector(2, 3, 4).cross(initVector(1, 2, 3)));
        CHECK_VEC(initVector(1, 1, 0), initVector(1, -1, 0).reflect(initVector(0, 1, 0)));
        CHECK(initVector(0, -1, 0).reflect(initVector(1, 1, 0).normalize()).eql(initVector(1, 0, 0)));
        CHECK_EPS(std::sqrt(14.0), initVector(1, 2, 3).length());
    });
    run("color.zig:61-80 add/sub/mult", [] {
        const Color c1{0.9, 0.6, 0.75}, c2{0.7, 0.1, 0.25};
        CHECK_COLOR((Color{1.6, 0.7, 1.0}), c1.add(c2));
        CHECK_COLOR((Color{0.2, 0.5, 0.5}), c1.sub(c2));
        CHECK_COLOR((Color{0.9, 0.2, 0.04}), (Color{1, 0.2, 0.4}).mult(Color{0.9, 1, 0.1}));
    });

    // ------------------------------------------------------------ matrix.zig
    run("matrix.zig:560-590 determinants and cofactors", [] {
        const Mat3 a{{{1, 2, 6}, {-5, 8, -4}, {2, 6, 4}}};
        CHECK_EPS(a.cofactor(0, 0), 56);
        CHECK_EPS(a.cofactor(0, 1), 12);
        CHECK_EPS(a.cofactor(0, 2), -46);
        CHECK_EPS(a.determinant(), -196);
        const Mat4 b{{{-2, -8, 3, 5}, {-3, 1, 7, 3}, {1, 2, -9, 6}, {-6, 7, 7, -9}}};
        CHECK_EPS(b.cofactor(0, 0), 690);
        CHECK_EPS(b.cofactor(0, 1), 447);
        CHECK_EPS(b.cofactor(0, 2), 210);
        CHECK_EPS(b.cofactor(0, 3), 51);
        CHECK_EPS(b.determinant(), -4071);
    });
    run("matrix.zig:592-618 invertibility", [] {
        const Mat4 a{{{6, 4, 4, 4}, {5, 5, 7, 6}, {4, -9, 3, -7}, {9, 1, 7, -6}}};
        CHECK_EPS(a.determinant(), -2120);
        CHECK(a.isInvertible());
        const Mat4 b{{{-4, 2, -2, -3}, {9, 6, 2, 6}, {0, -5, 1, -5}, {0, 0, 0, 0}}};
        CHECK_EPS(b.determinant(), 0);
        CHECK(!b.isInvertible());
    });
    run("matrix.zig:620-648 inverse of a matrix", [] {
        const Mat4 a{{{-5, 2, 6, -8}, {1, -5, 1, 8}, {7, 7, -6, -7}, {1, -3, 7, 4}}};
        const Mat4 b = a.inverse();
        const Mat4 e{{{0.21805, 0.45113, 0.24060, -0.04511},
                      {-0.80827, -1.45677, -0.44361, 0.52068},
                      {-0.07895, -0.22368, -0.05263, 0.19737},
                      {-0.52256, -0.81391, -0.30075, 0.30639}}};
        CHECK_EPS(a.determinant(), 532);
        CHECK_EPS(a.cofactor(2, 3), -160);
        CHECK_EPS(b.mat[3][2], -160.0 / 532.0);
        CHECK_EPS(a.cofactor(3, 2), 105);
        CHECK_EPS(b.mat[2][3], 105.0 / 532.0);
        CHECK(matApprox(b, e));
    });
    run("matrix.zig:650-692 two more inverses", [] {
        const Mat4 a{{{8, -5, 9, 2}, {7, 5, 6, 1}, {-6, 0, 9, 6}, {-3, 0, -9, -4}}};
        const Mat4 ea{{{-0.15385, -0.15385, -0.28205, -0.53846},
                       {-0.07692, 0.12308, 0.02564, 0.03077},
                       {0.35897, 0.35897, 0.43590, 0.92308},
                       {-0.69231, -0.69231, -0.76923, -1.92308}}};
        CHECK(matApprox(a.inverse(), ea));
        const Mat4 b{{{9, 3, 0, 9}, {-5, -2, -6, -3}, {-4, 9, 6, 4}, {-7, 6, 6, 2}}};
        const Mat4 eb{{{-0.04074, -0.07778, 0.14444, -0.22222},
                       {-0.07778, 0.03333, 0.36667, -0.33333},
                       {-0.02901, -0.14630, -0.10926, 0.12963},
                       {0.17778, 0.06667, -0.26667, 0.33333}}};
        CHECK(matApprox(b.inverse(), eb));
    });
    run("matrix.zig:694-717 product times inverse", [] {
        const Mat4 a{{{3, -9, 7, 3}, {3, -8, 2, -9}, {-4, 4, 4, 1}, {-6, 5, -1, 1}}};
        const Mat4 b{{{8, 2, 2, 2}, {3, -1, 7, 0}, {7, 0, 5, 4}, {6, -2, 0, 5}}};
        CHECK(matApprox(a, a.mult(b).mult(b.inverse())));
    });
    run("matrix.zig:719-806 translate/scale/rotate on points and vectors", [] {
        const Mat4 t = Mat4::identity().translate(5, -3, 2);
        CHECK(t.multVec(initPoint(-3, 4, 5)).eql(initPoint(2, 1, 7)));
        CHECK(t.inverse().multVec(initPoint(-3, 4, 5)).eql(initPoint(-8, 7, 3)));
        CHECK(t.multVec(initVector(-3, 4, 5)).eql(initVector(-3, 4, 5)));
        const Mat4 s = Mat4::identity().scale(2, 3, 4);
        CHECK(s.multVec(initPoint(-4, 6, 8)).eql(initPoint(-8, 18, 32)));
        CHECK(s.inverse().multVec(initVector(-4, 6, 8)).eql(initVector(-2, 2, 2)));
        CHECK_VEC(initPoint(0, S2 / 2, S2 / 2), Mat4::identity().rotateX(PI / 4).multVec(initPoint(0, 1, 0)));
        CHECK_VEC(initPoint(0, 0, 1), Mat4::identity().rotateX(PI / 2).multVec(initPoint(0, 1, 0)));
        CHECK_VEC(initPoint(0, S2 / 2, -S2 / 2),
                  Mat4::identity().rotateX(PI / 4).inverse().multVec(initPoint(0, 1, 0)));
        CHECK_VEC(initPoint(S2 / 2, 0, S2 / 2), Mat4::identity().rotateY(PI / 4).multVec(initPoint(0, 0, 1)));
        CHECK_VEC(initPoint(1, 0, 0), Mat4::identity().rotateY(PI / 2).multVec(initPoint(0, 0, 1)));
        CHECK_VEC(initPoint(-S2 / 2, S2 / 2, 0), Mat4::identity().rotateZ(PI / 4).multVec(initPoint(0, 1, 0)));
        CHECK_VEC(initPoint(-1, 0, 0), Mat4::identity().rotateZ(PI / 2).multVec(initPoint(0, 1, 0)));
    });
    run("matrix.zig:808-882 shear and chaining order", [] {
        const Vec4 p = initPoint(2, 3, 4);
        CHECK(Mat4::identity().shear(1, 0, 0, 0, 0, 0).multVec(p).eql(initPoint(5, 3, 4)));
        CHECK(Mat4::identity().shear(0, 1, 0, 0, 0, 0).multVec(p).eql(initPoint(6, 3, 4)));
        CHECK(Mat4::identity().shear(0, 0, 1, 0, 0, 0).multVec(p).eql(initPoint(2, 5, 4)));
        CHECK(Mat4::identity().shear(0, 0, 0, 1, 0, 0).multVec(p).eql(initPoint(2, 7, 4)));
        CHECK(Mat4::identity().shear(0, 0, 0, 0, 1, 0).multVec(p).eql(initPoint(2, 3, 6)));
        CHECK(Mat4::identity().shear(0, 0, 0, 0, 0, 1).multVec(p).eql(initPoint(2, 3, 7)));
        const Mat4 t = Mat4::identity().rotateX(PI / 2).scale(5, 5, 5).translate(10, 5, 7);
        CHECK_VEC(initPoint(15, 0, 7), t.multVec(initPoint(1, 0, 1)));
    });

    // ------------------------------------------------------------ ray.zig
    run("ray.zig:48-55 position", [] {
        const Ray r{initPoint(2, 3, 4), initVector(1, 0, 0)};
        CHECK(r.position(0).eql(initPoint(2, 3, 4)));
        CHECK(r.position(1).eql(initPoint(3, 3, 4)));
        CHECK(r.position(-1).eql(initPoint(1, 3, 4)));
        CHECK(r.position(2.5).eql(initPoint(4.5, 3, 4)));
    });
    run("ray.zig:97-161 hit selection", [] {
        const Shape s;
        {
            Intersections xs;
            xs.list = {{1, s}, {2, s}};
            CHECK(xs.hit().has_value() && xs.hit()->t == 1);
        }
        {
            Intersections xs;
            xs.list = {{-1, s}, {1, s}};
            CHECK(xs.hit().has_value() && xs.hit()->t == 1);
        }
        {
            Intersections xs;
            xs.list = {{-2, s}, {-1, s}};
            CHECK(!xs.hit().has_value());
        }
        {
            Intersections xs;
            xs.list = {{5, s}, {7, s}, {-3, s}, {2, s}};
            CHECK(xs.hit().has_value() && xs.hit()->t == 2);
        }
    });
    run("ray.zig:163-181 transform", [] {
        const Ray r{initPoint(1, 2, 3), initVector(0, 1, 0)};
        const Ray r2 = r.transform(Mat4::identity().translate(3, 4, 5));
        CHECK(r2.origin.eql(initPoint(4, 6, 8)) && r2.direction.eql(initVector(0, 1, 0)));
        const Ray r3 = r.transform(Mat4::identity().scale(2, 3, 4));
        CHECK(r3.origin.eql(initPoint(2, 6, 12)) && r3.direction.eql(initVector(0, 3, 0)));
    });

    // ------------------------------------------------------------ shape.zig
    run("shape.zig:134-159 sphere normals", [] {
        const Shape s;
        CHECK(normalAt(s, initPoint(1, 0, 0)).eql(initVector(1, 0, 0)));
        CHECK(normalAt(s, initPoint(0, 1, 0)).eql(initVector(0, 1, 0)));
        CHECK(normalAt(s, initPoint(0, 0, 1)).eql(initVector(0, 0, 1)));
        const double k = std::sqrt(3.0) / 3.0;
        const Vec4 n = normalAt(s, initPoint(k, k, k));
        CHECK(n.eql(initVector(k, k, k)));
        CHECK(n.eql(n.normalize()));
    });
    run("shape.zig:163-183 normals of transformed spheres", [] {
        Shape s;
        s.transform = Mat4::identity().translate(0, 1, 0);
        CHECK_VEC(initVector(0, 0.70711, -0.70711), normalAt(s, initPoint(0, 1.70711, -0.70711)));
        Shape s2;
        s2.transform = Mat4::identity().rotateZ(PI / 5.0).scale(1, 0.5, 1);
        CHECK_VEC(initVector(0, 0.97014, -0.24254), normalAt(s2, initPoint(0, S2 / 2, -S2 / 2)));
    });
    run("shape.zig:185-257 sphere intersections", [] {
        const Shape s;
        double t[2];
        CHECK(intersectShape(s, {initPoint(0, 0, -5), initVector(0, 0, 1)}, t) == 2);
        CHECK_EPS(4.0, t[0]);
        CHECK_EPS(6.0, t[1]);
        CHECK(intersectShape(s, {initPoint(0, 1, -5), initVector(0, 0, 1)}, t) == 2);
        CHECK_EPS(5.0, t[0]);
        CHECK_EPS(5.0, t[1]);
        CHECK(intersectShape(s, {initPoint(0, 2, -5), initVector(0, 0, 1)}, t) == 0);
        CHECK(intersectShape(s, {initPoint(0, 0, 0), initVector(0, 0, 1)}, t) == 2);
        CHECK_EPS(-1.0, t[0]);
        CHECK_EPS(1.0, t[1]);
        CHECK(intersectShape(s, {initPoint(0, 0, 5), initVector(0, 0, 1)}, t) == 2);
        CHECK_EPS(-6.0, t[0]);
        CHECK_EPS(-4.0, t[1]);
    });
    run("shape.zig:259-286 scaled and translated spheres", [] {
        Shape s;
        s.transform = Mat4::identity().scale(2, 2, 2);
        double t[2];
        CHECK(intersectShape(s, {initPoint(0, 0, -5), initVector(0, 0, 1)}, t) == 2);
        CHECK_EPS(3.0, t[0]);
        CHECK_EPS(7.0, t[1]);
        Shape s2;
        s2.transform = Mat4::identity().translate(5, 0, 0);
        CHECK(intersectShape(s2, {initPoint(0, 0, -5), initVector(0, 0, 1)}, t) == 0);
    });
    run("shape.zig:288-350 plane normal and intersections", [] {
        CHECK_VEC(initVector(0, 1, 0), localNormalAt(GEO_PLANE, initPoint(0, 0, 0)));
        CHECK_VEC(initVector(0, 1, 0), localNormalAt(GEO_PLANE, initPoint(10, 0, -10)));
        CHECK_VEC(initVector(0, 1, 0), localNormalAt(GEO_PLANE, initPoint(-5, 0, 150)));
        double t[2];
        CHECK(planeLocalIntersect({initPoint(0, 10, 0), initVector(0, 0, 1)}, t) == 0);
        CHECK(planeLocalIntersect({initPoint(0, 0, 0), initVector(0, 0, 1)}, t) == 0);
        CHECK(planeLocalIntersect({initPoint(0, 1, 0), initVector(0, -1, 0)}, t) == 1);
        CHECK_EPS(1.0, t[0]);
        CHECK(planeLocalIntersect({initPoint(0, -1, 0), initVector(0, 1, 0)}, t) == 1);
        CHECK_EPS(1.0, t[0]);
    });

    // ------------------------------------------------------------ pattern.zig
    run("pattern.zig:104-112 object and pattern transformation (point pattern)", [] {
        Pattern p;
        p.transform = Mat4::identity().translate(0.5, 1, 1.5);
        Shape s;
        s.transform = Mat4::identity().scale(2, 2, 2);
        CHECK_COLOR((Color{0.75, 0.5, 0.25}), patternAt(p, s, initPoint(2.5, 3, 3.5)));
    });
    run("pattern.zig:114-149 stripes", [] {
        Pattern p;
        p.kind = PAT_STRIPE;
        p.a = White;
        p.b = Black;
        CHECK_COLOR(White, patternLocal(p, initPoint(0, 0, 0)));
        CHECK_COLOR(White, patternLocal(p, initPoint(0, 1, 0)));
        CHECK_COLOR(White, patternLocal(p, initPoint(0, 2, 0)));
        CHECK_COLOR(White, patternLocal(p, initPoint(0, 0, 1)));
        CHECK_COLOR(White, patternLocal(p, initPoint(0, 0, 2)));
        CHECK_COLOR(White, patternLocal(p, initPoint(0.9, 0, 0)));
        CHECK_COLOR(Black, patternLocal(p, initPoint(1, 0, 0)));
        CHECK_COLOR(Black, patternLocal(p, initPoint(-0.1, 0, 0)));
        CHECK_COLOR(Black, patternLocal(p, initPoint(-1, 0, 0)));
        CHECK_COLOR(White, patternLocal(p, initPoint(-1.1, 0, 0)));
        p.transform = Mat4::identity().translate(0.5, 0, 0);
        Shape s;
        s.transform = Mat4::identity().scale(2, 2, 2);
        CHECK_COLOR(White, patternAt(p, s, initPoint(2.5, 0, 0)));
    });
    run("pattern.zig:151-190 gradient, ring, checkers", [] {
        const Shape s;
        Pattern g;
        g.kind = PAT_GRADIENT;
        g.a = White;
        g.b = Black;
        CHECK_COLOR(White, patternAt(g, s, initPoint(0, 0, 0)));
        CHECK_COLOR((Color{0.75, 0.75, 0.75}), patternAt(g, s, initPoint(0.25, 0, 0)));
        CHECK_COLOR((Color{0.5, 0.5, 0.5}), patternAt(g, s, initPoint(0.5, 0, 0)));
        CHECK_COLOR((Color{0.25, 0.25, 0.25}), patternAt(g, s, initPoint(0.75, 0, 0)));
        Pattern r = g;
        r.kind = PAT_RING;
        CHECK_COLOR(White, patternAt(r, s, initPoint(0, 0, 0)));
        CHECK_COLOR(Black, patternAt(r, s, initPoint(1, 0, 0)));
        CHECK_COLOR(Black, patternAt(r, s, initPoint(0, 0, 1)));
        CHECK_COLOR(Black, patternAt(r, s, initPoint(0.708, 0, 0.708)));
        Pattern c = g;
        c.kind = PAT_CHECKERS;
        CHECK_COLOR(White, patternAt(c, s, initPoint(0, 0, 0)));
        CHECK_COLOR(White, patternAt(c, s, initPoint(0.99, 0, 0)));
        CHECK_COLOR(Black, patternAt(c, s, initPoint(1.01, 0, 0)));
        CHECK_COLOR(White, patternAt(c, s, initPoint(0, 0.99, 0)));
        CHECK_COLOR(Black, patternAt(c, s, initPoint(0, 1.01, 0)));
        CHECK_COLOR(White, patternAt(c, s, initPoint(0, 0, 0.99)));
        CHECK_COLOR(Black, patternAt(c, s, initPoint(0, 0, 1.01)));
    });

    // ------------------------------------------------------------ material/light/world defaults
    run("material.zig:20-31, light.zig:9-14, world.zig:63-71 defaults", [] {
        const Material m;
        CHECK(m.color.red == 1 && m.color.green == 1 && m.color.blue == 1);
        CHECK(m.ambient == 0.1 && m.diffuse == 0.9 && m.specular == 0.9 && m.shininess == 200.0);
        CHECK(m.reflective == 0.0 && m.transparency == 0.0 && m.refractive_index == 1.0 && !m.has_pattern);
        const World w = defaultWorld();
        CHECK(eqlColor(w.objects[0].material.color, Color{0.8, 1.0, 0.6}));
        CHECK(eqlMat4(w.objects[1].transform, Mat4::identity().scale(0.5, 0.5, 0.5)));
        CHECK(w.light.position.eql(initPoint(-10, 10, -10)) && eqlColor(w.light.intensity, White));
    });

    // ------------------------------------------------------------ calc.zig lighting
    run("calc.zig:75-170 lighting, six geometric cases", [] {
        const Shape obj;
        const Vec4 position = initPoint(0, 0, 0);
        const Vec4 n = initVector(0, 0, -1);
        const PointLight l0{initPoint(0, 0, -10), White};
        CHECK_COLOR((Color{1.9, 1.9, 1.9}), lighting(obj, l0, position, initVector(0, 0, -1), n, false));
        CHECK_COLOR((Color{1.0, 1.0, 1.0}), lighting(obj, l0, position, initVector(0, 1, -1).normalize(), n, false));
        const PointLight l45{initPoint(0, 10, -10), White};
        CHECK_COLOR((Color{0.7364, 0.7364, 0.7364}), lighting(obj, l45, position, initVector(0, 0, -1), n, false));
        CHECK_COLOR((Color{1.6364, 1.6364, 1.6364}),
                    lighting(obj, l45, position, initVector(0, -S2 / 2, -S2 / 2), n, false));
        const PointLight behind{initPoint(0, 0, 10), White};
        CHECK_COLOR((Color{0.1, 0.1, 0.1}), lighting(obj, behind, position, initVector(0, 0, -1), n, false));
        CHECK_COLOR((Color{0.1, 0.1, 0.1}), lighting(obj, l0, position, initVector(0, 0, -1), n, true));
    });
    run("calc.zig:172-198 lighting with a pattern", [] {
        Shape obj;
        obj.material.has_pattern = true;
        obj.material.pattern.kind = PAT_STRIPE;
        obj.material.pattern.a = {0.5, 0.2, 0.9};
        obj.material.pattern.b = {0.2, 0.8, 1.0};
        obj.material.ambient = 1;
        obj.material.diffuse = 0;
        obj.material.specular = 0;
        const PointLight l{initPoint(0, 0, -10), White};
        const Vec4 e = initVector(0, 0, -1), n = initVector(0, 0, -1);
        CHECK_COLOR(obj.material.pattern.a, lighting(obj, l, initPoint(0.9, 0, 0), e, n, false));
        CHECK_COLOR(obj.material.pattern.b, lighting(obj, l, initPoint(1.1, 0, 0), e, n, false));
    });

    // ------------------------------------------------------------ calc.zig intersectWorld / prepareComputations
    run("calc.zig:216-231 intersect a world with a ray", [] {
        const World w = defaultWorld();
        Intersections xs = intersectWorld(w, {initPoint(0, 0, -5), initVector(0, 0, 1)});
        CHECK(xs.list.size() == 4);
        CHECK_EPS(4.0, xs.list[0].t);
        CHECK_EPS(4.5, xs.list[1].t);
        CHECK_EPS(5.5, xs.list[2].t);
        CHECK_EPS(6.0, xs.list[3].t);
    });
    run("calc.zig:308-350 precomputing the state of an intersection", [] {
        const Ray r{initPoint(0, 0, -5), initVector(0, 0, 1)};
        const Shape shape;
        const Computations c = prepareComputations({4, shape}, r, nullptr);
        CHECK(c.t == 4.0);
        CHECK(eqlShape(shape, c.object));
        CHECK(c.point.x == 0 && c.point.y == 0 && c.point.z == -1 && c.point.w == 1);
        CHECK(c.eyev.x == 0 && c.eyev.y == 0 && c.eyev.z == -1);
        CHECK(c.normalv.x == 0 && c.normalv.y == 0 && c.normalv.z == -1);
        CHECK(!c.inside);
        const Computations ci = prepareComputations({1, shape}, {initPoint(0, 0, 0), initVector(0, 0, 1)}, nullptr);
        CHECK(ci.point.z == 1 && ci.eyev.z == -1 && ci.inside && ci.normalv.z == -1);
    });
    run("calc.zig:352-398 over_point, under_point, reflectv", [] {
        Shape shape;
        shape.transform = Mat4::identity().translate(0, 0, 1);
        const Ray r{initPoint(0, 0, -5), initVector(0, 0, 1)};
        const Computations c = prepareComputations({5, shape}, r, nullptr);
        CHECK(c.over_point.z == -0.0001);
        CHECK(c.point.z > c.over_point.z);
        Shape g = glassSphere();
        g.transform = Mat4::identity().translate(0, 0, 1);
        const Computations cu = prepareComputations({5.0, g}, r, nullptr);
        CHECK(cu.under_point.z == 0.0001);
        CHECK(cu.point.z < cu.under_point.z);
        const Computations cr =
            prepareComputations({S2, planeShape()}, {initPoint(0, 0, -1), initVector(-0.0, -S2 / 2, S2 / 2)}, nullptr);
        CHECK_VEC(initVector(0, S2 / 2, S2 / 2), cr.reflectv);
    });
    run("calc.zig:400-446 n1 and n2 at various intersections", [] {
        Shape a = glassSphere();
        a.transform = Mat4::identity().scale(2, 2, 2);
        a.material.refractive_index = 1.5;
        Shape b = glassSphere();
        b.transform = Mat4::identity().translate(0, 0, -0.25);
        b.material.refractive_index = 2.0;
        Shape c = glassSphere();
        c.transform = Mat4::identity().translate(0, 0, 0.25);
        c.material.refractive_index = 2.5;
        const Ray r{initPoint(0, 0, -4), initVector(0, 0, 1)};
        Intersections xs;
        xs.list = {{2, a}, {2.75, b}, {3.25, c}, {4.75, b}, {5.25, c}, {6, a}};
        const double n1[6] = {1.0, 1.5, 2.0, 2.5, 2.5, 1.5};
        const double n2[6] = {1.5, 2.0, 2.5, 2.5, 1.5, 1.0};
        for (int i = 0; i < 6; ++i) {
            const Computations comps = prepareComputations(xs.list[(size_t)i], r, &xs);
            CHECK_EPS(n1[i], comps.n1);
            CHECK_EPS(n2[i], comps.n2);
        }
    });

    run("calc.zig:269-291 with two value-identical shapes: they toggle ONE container (std.meta.eql, calc.zig:274,280)", [] {
        // The reference holds no test for this; it is what its container walk does, replayed by the LITERAL restatement:
        // two coincident glass spheres that are equal field by field are "the same object" to the walk, so the second
        // entry closes the container the first one opened.  Seen from inside both (t1 < 0 < t2 for each) the walk
        // reaches the exit hit with NO open container: n1 = 1.0, and the hit sphere is then "entered": n2 = 1.5.
        // Two spheres that differ in anything (here: the refractive index in its last bit) are two containers:
        // n1 = the second's index, n2 = what remains after the first is removed.
        Shape a = glassSphere();
        const Shape twin = a;
        const Ray r{initPoint(0, 0, 0), initVector(0, 0, 1)};
        {
            Intersections xs;
            xs.list = {{-1, a}, {-1, twin}, {1, a}, {1, twin}};  // sorted by t, insertion order kept (ray.zig:89-92)
            const Computations c = prepareComputations(xs.list[2], r, &xs);  // xs.hit(): the first t >= 0
            CHECK_EPS(1.0, c.n1);
            CHECK_EPS(1.5, c.n2);
        }
        {
            Shape other = a;
            other.material.refractive_index = std::nextafter(1.5, 2.0);
            Intersections xs;
            xs.list = {{-1, a}, {-1, other}, {1, a}, {1, other}};
            const Computations c = prepareComputations(xs.list[2], r, &xs);
            CHECK(c.n1 == other.material.refractive_index);
            CHECK(c.n2 == other.material.refractive_index);
        }
        // through the integrators: the LITERAL one compares by value like the reference, the STREAMING one (the
        // GPU's algorithm) identifies shapes by list index - which is why the library refuses such a world
        // (RTC_ERR_DUPLICATE_SHAPE) instead of rendering it differently from the reference
        World w;
        w.light = PointLight{initPoint(-10, 10, -10), Color{1, 1, 1}};
        w.objects.push_back(a);
        w.objects.push_back(twin);
        Shape floor = planeShape();
        floor.transform = Mat4::identity().translate(0, -1.5, 0);
        w.objects.push_back(floor);
        w.prepare();  // (the hoisted inverses of the STREAMING integrator)
        const Ray from_inside{initPoint(0, 0, 0.2), initVector(0.1, -0.3, 1).normalize()};
        const Color lit = worldColorAt(w, from_inside, 5, LITERAL);
        const Color str = worldColorAt(w, from_inside, 5, STREAMING);
        CHECK(!(lit.red == str.red && lit.green == str.green && lit.blue == str.blue));
        // with a single sphere the two integrators agree, as everywhere else
        w.objects.erase(w.objects.begin() + 1);
        w.prepare();
        const Color lit1 = worldColorAt(w, from_inside, 5, LITERAL);
        const Color str1 = worldColorAt(w, from_inside, 5, STREAMING);
        CHECK(lit1.red == str1.red && lit1.green == str1.green && lit1.blue == str1.blue);
    });

    // ------------------------------------------------------------ calc.zig shadeHit
    run("calc.zig:472-487 shading an intersection", [] {
        const World w = defaultWorld();
        const Ray r{initPoint(0, 0, -5), initVector(0, 0, 1)};
        const Computations c = prepareComputations({4, w.objects[0]}, r, nullptr);
        CHECK_COLOR((Color{0.38066, 0.47583, 0.2855}), shadeHit(w, c, 5, LITERAL));
    });
    run("calc.zig:489-509 shading from the inside", [] {
        World w = defaultWorld();
        w.light = {initPoint(0, 0.25, 0), White};
        const Ray r{initPoint(0, 0, 0), initVector(0, 0, 1)};
        const Computations c = prepareComputations({0.5, w.objects[1]}, r, nullptr);
        CHECK_COLOR((Color{0.90498, 0.90498, 0.90498}), shadeHit(w, c, 5, LITERAL));
    });
    run("calc.zig:511-539 shade_hit in shadow", [] {
        World w;
        w.light = {initPoint(0, 0, -10), White};
        w.objects.push_back(Shape{});
        Shape s2;
        s2.transform = Mat4::identity().translate(0, 0, 10);
        w.objects.push_back(s2);
        const Ray r{initPoint(0, 0, 5), initVector(0, 0, 1)};
        const Computations c = prepareComputations({4, s2}, r, nullptr);
        CHECK_COLOR((Color{0.1, 0.1, 0.1}), shadeHit(w, c, 5, LITERAL));
    });
    run("calc.zig:541-562 shade_hit with a reflective material", [] {
        World w = defaultWorld();
        Shape shape = planeShape();
        shape.transform = Mat4::identity().translate(0, -1, 0);
        shape.material.reflective = 0.5;
        w.objects.push_back(shape);
        const Ray r{initPoint(0, 0, -3), initVector(0, -S2 / 2, S2 / 2)};
        const Computations c = prepareComputations({S2, shape}, r, nullptr);
        CHECK_COLOR((Color{0.87677, 0.92436, 0.82918}), shadeHit(w, c, 5, LITERAL));
    });
    auto floorBallWorld = [](double reflective, Shape* floor_out) {
        World w = defaultWorld();
        Shape floor = planeShape();
        floor.transform = Mat4::identity().translate(0, -1, 0);
        floor.material.reflective = reflective;
        floor.material.transparency = 0.5;
        floor.material.refractive_index = 1.5;
        w.objects.push_back(floor);
        Shape ball;
        ball.transform = Mat4::identity().translate(0, -3.5, -0.5);
        ball.material.color = {1, 0, 0};
        ball.material.ambient = 0.5;
        w.objects.push_back(ball);
        *floor_out = floor;
        return w;
    };
    run("calc.zig:564-600 shade_hit with a transparent material", [&] {
        Shape floor;
        const World w = floorBallWorld(0.0, &floor);
        const Ray r{initPoint(0, 0, -3), initVector(0, -S2 / 2, S2 / 2)};
        Intersections xs;
        xs.list = {{S2, floor}};
        const Computations c = prepareComputations(xs.list[0], r, &xs);
        CHECK_COLOR((Color{0.93642, 0.68642, 0.68642}), shadeHit(w, c, 5, LITERAL));
    });
    run("calc.zig:602-639 shade_hit with a reflective, transparent material", [&] {
        Shape floor;
        const World w = floorBallWorld(0.5, &floor);
        const Ray r{initPoint(0, 0, -3), initVector(0, -S2 / 2, S2 / 2)};
        Intersections xs;
        xs.list = {{S2, floor}};
        const Computations c = prepareComputations(xs.list[0], r, &xs);
        CHECK_COLOR((Color{0.93391, 0.69643, 0.69243}), shadeHit(w, c, 5, LITERAL));
    });

    // ------------------------------------------------------------ calc.zig worldColorAt (both integrators)
    for (Mode mode : {LITERAL, STREAMING}) {
        const std::string tag = mode == LITERAL ? " [literal]" : " [streaming]";
        run(("calc.zig:654-672 colour when a ray misses / hits" + tag).c_str(), [mode] {
            World w = defaultWorld();
            w.prepare();
            CHECK_COLOR(Black, worldColorAt(w, {initPoint(0, 0, -5), initVector(0, 1, 0)}, 5, mode));
            CHECK_COLOR((Color{0.38066, 0.47583, 0.2855}),
                        worldColorAt(w, {initPoint(0, 0, -5), initVector(0, 0, 1)}, 5, mode));
        });
        run(("calc.zig:674-690 intersection behind the ray" + tag).c_str(), [mode] {
            World w = defaultWorld();
            w.objects[0].material.color = {0.3, 0.3, 1.0};
            w.objects[0].material.ambient = 1.0;
            w.objects[1].material.color = {0.5, 1.0, 0.2};
            w.objects[1].material.ambient = 1.0;
            w.prepare();
            CHECK_COLOR(w.objects[1].material.color,
                        worldColorAt(w, {initPoint(0, 0, 0.75), initVector(0, 0, -1)}, 5, mode));
        });
        run(("calc.zig:692-719 mutually reflective surfaces terminate" + tag).c_str(), [mode] {
            World w;
            w.light = {initPoint(0, 0, 0), White};
            Shape lower = planeShape();
            lower.transform = Mat4::identity().translate(0, -1, 0);
            lower.material.reflective = 1;
            w.objects.push_back(lower);
            Shape upper = planeShape();
            upper.transform = Mat4::identity().translate(0, 1, 0);
            upper.material.reflective = 1;
            w.objects.push_back(upper);
            w.prepare();
            const Color c = worldColorAt(w, {initPoint(0, 0, 0), initVector(0, 1, 1)}, 5, mode);
            CHECK(std::isfinite(c.red));
        });
        run(("camera.zig:137-152 rendering a world with a camera" + tag).c_str(), [mode] {
            World w = defaultWorld();
            Camera c = Camera::init(11, 11, PI * 0.5);
            c.transform = viewTransform(initPoint(0, 0, -5), initPoint(0, 0, 0), initVector(0, 1, 0));
            std::vector<double> img(11 * 11 * 3, 0.0);
            render(c, w, 5, mode, img.data(), 0, 11, 1);
            const double* p = &img[(5 * 11 + 5) * 3];
            CHECK_COLOR((Color{0.38066, 0.47583, 0.2855}), (Color{p[0], p[1], p[2]}));
        });
    }

    // ------------------------------------------------------------ calc.zig viewTransform
    run("calc.zig:740-775 view transforms", [] {
        CHECK(matApprox(viewTransform(initPoint(0, 0, 0), initPoint(0, 0, -1), initVector(0, 1, 0)), Mat4::identity()));
        CHECK(matApprox(viewTransform(initPoint(0, 0, 8), initPoint(0, 0, 0), initVector(0, 1, 0)),
                        Mat4::identity().translate(0, 0, -8)));
        const Mat4 e{{{-0.50709, 0.50709, 0.67612, -2.36643},
                      {0.76772, 0.60609, 0.12122, -2.82843},
                      {-0.35857, 0.59761, -0.71714, 0.00000},
                      {0.00000, 0.00000, 0.00000, 1.00000}}};
        CHECK(matApprox(viewTransform(initPoint(1, 3, 2), initPoint(4, -2, 8), initVector(1, 1, 0)), e));
    });

    // ------------------------------------------------------------ calc.zig isShadowed
    for (Mode mode : {LITERAL, STREAMING}) {
        run(mode == LITERAL ? "calc.zig:792-826 isShadowed, four cases [literal]"
                            : "calc.zig:792-826 isShadowed, four cases [streaming]",
            [mode] {
                World w = defaultWorld();
                w.prepare();
                auto sh = [&](const Vec4& p) {
                    return mode == LITERAL ? isShadowedLiteral(w, p) : isShadowedStreaming(w, p);
                };
                CHECK(sh(initPoint(0, 10, 0)) == false);
                CHECK(sh(initPoint(10, -10, 10)) == true);
                CHECK(sh(initPoint(-20, 20, -20)) == false);
                CHECK(sh(initPoint(-2, 2, -2)) == false);
            });
    }

    // ------------------------------------------------------------ calc.zig refractedColor
    run("calc.zig:855-894 refracted colour: opaque surface, depth 0", [] {
        World w = defaultWorld();
        const Ray r{initPoint(0, 0, -5), initVector(0, 0, 1)};
        w.objects[1].material.transparency = 0.0;
        const Computations c = prepareComputations({4, w.objects[1]}, r, nullptr);
        CHECK_COLOR(Black, refractedColor(w, c, 5, LITERAL));
        World w2 = defaultWorld();
        w2.objects[0].material.transparency = 1.0;
        w2.objects[0].material.refractive_index = 1.5;
        const Computations c2 = prepareComputations({4, w2.objects[0]}, r, nullptr);
        CHECK_COLOR(Black, refractedColor(w2, c2, 0, LITERAL));
    });
    run("calc.zig:896-916 refracted colour under total internal reflection", [] {
        World w = defaultWorld();
        w.objects[0].material.transparency = 1.0;
        w.objects[0].material.refractive_index = 1.5;
        const Ray r{initPoint(0, 0, S2 / 2), initVector(0, 1, 0)};
        Intersections xs;
        xs.list = {{-S2 / 2, w.objects[0]}, {S2 / 2, w.objects[0]}};
        const Computations c = prepareComputations(xs.list[1], r, &xs);
        CHECK_COLOR(Black, refractedColor(w, c, 5, LITERAL));
    });
    run("calc.zig:918-943 refracted colour with a refracted ray", [] {
        World w = defaultWorld();
        w.objects[0].material.ambient = 1.0;
        w.objects[0].material.has_pattern = true;
        w.objects[0].material.pattern = Pattern{};
        w.objects[1].material.transparency = 1.0;
        w.objects[1].material.refractive_index = 1.5;
        const Ray r{initPoint(0, 0, 0.1), initVector(0, 1, 0)};
        Intersections xs;
        xs.list = {{-0.9899, w.objects[0]}, {-0.4899, w.objects[1]}, {0.4899, w.objects[1]}, {0.9899, w.objects[0]}};
        const Computations c = prepareComputations(xs.list[2], r, &xs);
        CHECK_COLOR((Color{0, 0.99888, 0.04725}), refractedColor(w, c, 5, LITERAL));
    });

    // ------------------------------------------------------------ calc.zig reflectedColor
    run("calc.zig:961-1002 reflected colour", [] {
        World w = defaultWorld();
        w.objects[1].material.ambient = 1.0;
        const Computations c =
            prepareComputations({1, w.objects[1]}, {initPoint(0, 0, 0), initVector(0, 0, 1)}, nullptr);
        CHECK_COLOR(Black, reflectedColor(w, c, 5, LITERAL));
        World w2 = defaultWorld();
        Shape shape = planeShape();
        shape.transform = Mat4::identity().translate(0, -1, 0);
        shape.material.reflective = 0.5;
        w2.objects.push_back(shape);
        const Computations c2 =
            prepareComputations({S2, shape}, {initPoint(0, 0, -3), initVector(0, -S2 / 2, S2 / 2)}, nullptr);
        CHECK_COLOR((Color{0.19032, 0.2379, 0.14274}), reflectedColor(w2, c2, 5, LITERAL));
    });

    // ------------------------------------------------------------ calc.zig schlick
    run("calc.zig:1028-1067 Schlick approximation", [] {
        const Shape shape = glassSphere();
        {
            Intersections xs;
            xs.list = {{-S2 / 2, shape}, {S2 / 2, shape}};
            const Computations c = prepareComputations(xs.list[1], {initPoint(0, 0, S2 / 2), initVector(0, 1, 0)}, &xs);
            CHECK(approx(1.0, schlick(c), 1e-5));
        }
        {
            Intersections xs;
            xs.list = {{-1, shape}, {1, shape}};
            const Computations c = prepareComputations(xs.list[1], {initPoint(0, 0, 0), initVector(0, 1, 0)}, &xs);
            CHECK(approx(0.04, schlick(c), 1e-5));
        }
        {
            Intersections xs;
            xs.list = {{1.8589, shape}};
            const Computations c = prepareComputations(xs.list[0], {initPoint(0, 0.99, -2), initVector(0, 0, 1)}, &xs);
            CHECK(approx(0.48873, schlick(c), 1e-5));
        }
    });

    // ------------------------------------------------------------ camera.zig
    run("camera.zig:92-107 construction and pixel size", [] {
        const Camera c = Camera::init(160, 120, 0.5 * PI);
        CHECK(c.hsize == 160 && c.vsize == 120 && c.field_of_view == 0.5 * PI);
        CHECK_EPS(0.01, Camera::init(200, 125, 0.5 * PI).pixel_size);
        CHECK_EPS(0.01, Camera::init(125, 200, 0.5 * PI).pixel_size);
    });
    run("camera.zig:109-133 rays through the canvas", [] {
        Camera c = Camera::init(201, 101, PI * 0.5);
        Ray r = c.rayForPixel(100, 50);
        CHECK_VEC(initPoint(0, 0, 0), r.origin);
        CHECK_VEC(initVector(0, 0, -1), r.direction);
        r = c.rayForPixel(0, 0);
        CHECK_VEC(initPoint(0, 0, 0), r.origin);
        CHECK_VEC(initVector(0.66519, 0.33259, -0.66851), r.direction);
        c.transform = Mat4::identity().translate(0, -2, 5).rotateY(PI / 4.0);
        r = c.rayForPixel(100, 50);
        CHECK_VEC(initPoint(0, 2, -5), r.origin);
        CHECK_VEC(initVector(S2 / 2, 0, -S2 / 2), r.direction);
    });

    // ------------------------------------------------------------ ppm.zig quantisation
    run("ppm.zig:63-110 byteColorValue", [] {
        CHECK(byteColorValue(1.5) == 255 && byteColorValue(0.0) == 0 && byteColorValue(0.5) == 128);
        CHECK(byteColorValue(-0.5) == 0 && byteColorValue(1.0) == 255);
        CHECK(byteColorValue(0.8) == 204 && byteColorValue(0.6) == 153);
    });

    // ------------------------------------------------------------ std.math.pow restatement
    run("zig pow restatement: integer exponents equal square-and-multiply", [] {
        const double xs[] = {0.5, 0.9999, 0.123456789, 1e-3, 0.999999, -0.2, 0.7071067811865476};
        for (double x : xs) {
            CHECK(zigmath::pow(x, 2.0) == x * x);
            CHECK(zigmath::pow(x, 5.0) == x * ((x * x) * (x * x)));
            // 200 = 8 + 64 + 128
            double p = x, a = 1;
            for (int i = 200, k = 0; i; i >>= 1, ++k) {
                if (i & 1) a *= p;
                p *= p;
            }
            CHECK(zigmath::pow(x, 200.0) == a || std::fabs(a) < 1e-290);
        }
        CHECK(zigmath::pow(0.0, 2.0) == 0.0 && zigmath::pow(3.0, 0.0) == 1.0 && zigmath::pow(1.0, 77.5) == 1.0);
        CHECK(approx(std::pow(0.7, 10.5), zigmath::pow(0.7, 10.5), 1e-15));
        CHECK(approx(std::pow(0.3, 2.25), zigmath::pow(0.3, 2.25), 1e-15));
    });

    // ------------------------------------------------------------ extension: the cube (not in the reference)
    // The reference has no cube; these are the scenarios of the book's Ch. 12 ("Cubes"), the chapter after the
    // last one the reference implements, with the book's inputs and answers, at the reference's tolerances.
    run("extension (book Ch. 12, no reference counterpart): a ray intersects a cube", [] {
        const double cases[7][8] = {{5, 0.5, 0, -1, 0, 0, 4, 6},  {-5, 0.5, 0, 1, 0, 0, 4, 6}, {0.5, 5, 0, 0, -1, 0, 4, 6},
                                    {0.5, -5, 0, 0, 1, 0, 4, 6},  {0.5, 0, 5, 0, 0, -1, 4, 6}, {0.5, 0, -5, 0, 0, 1, 4, 6},
                                    {0, 0.5, 0, 0, 0, 1, -1, 1}};
        for (const auto& c : cases) {
            double ts[2] = {0, 0};
            const Ray r{initPoint(c[0], c[1], c[2]), initVector(c[3], c[4], c[5])};
            CHECK(cubeLocalIntersect(r, ts) == 2);
            CHECK_EPS(c[6], ts[0]);
            CHECK_EPS(c[7], ts[1]);
        }
    });
    run("extension (book Ch. 12, no reference counterpart): a ray misses a cube", [] {
        const double cases[6][6] = {{-2, 0, 0, 0.2673, 0.5345, 0.8018}, {0, -2, 0, 0.8018, 0.2673, 0.5345},
                                    {0, 0, -2, 0.5345, 0.8018, 0.2673}, {2, 0, 2, 0, 0, -1},
                                    {0, 2, 2, 0, -1, 0},                {2, 2, 0, -1, 0, 0}};
        for (const auto& c : cases) {
            double ts[2];
            CHECK(cubeLocalIntersect(Ray{initPoint(c[0], c[1], c[2]), initVector(c[3], c[4], c[5])}, ts) == 0);
        }
    });
    run("extension: rays that lie in a face plane of the cube (parallel to an axis, origin on the slab's plane)", [] {
        // the decision of cubeCheckAxis: the face plane belongs to the cube, whichever axis it is across.  Columns:
        // origin, direction, expected number of t, t0, t1.
        const double cases[12][9] = {
            {-1, 0.5, -5, 0, 0, 1, 2, 4, 6},   {1, 0.5, -5, 0, 0, 1, 2, 4, 6},    // in the planes x = -1 / x = 1, along z
            {0.5, -1, -5, 0, 0, 1, 2, 4, 6},   {0.5, 1, -5, 0, 0, 1, 2, 4, 6},    // y = -1 / y = 1, along z
            {-5, 0.2, -1, 1, 0, 0, 2, 4, 6},   {-5, 0.2, 1, 1, 0, 0, 2, 4, 6},    // z = -1 / z = 1, along x
            {0.3, -5, 1, 0, 1, 0, 2, 4, 6},    {1, -5, 0.3, 0, 1, 0, 2, 4, 6},    // z = 1 / x = 1, along y
            {-1, -1, -5, 0, 0, 1, 2, 4, 6},                                        // along an edge
            {-1, 1.5, -5, 0, 0, 1, 0, 0, 0},   {0.5, -1, 3, 1, 0, 0, 0, 0, 0},    // in a face plane but beside the cube
            {1, 1, 0, 0, 0, -1, 2, -1, 1}};                                        // on an edge, from the middle of it
        for (const auto& c : cases) {
            double ts[2] = {0, 0};
            const int n = cubeLocalIntersect(Ray{initPoint(c[0], c[1], c[2]), initVector(c[3], c[4], c[5])}, ts);
            CHECK(n == (int)c[6]);
            if (n == 2) {
                CHECK_EPS(c[7], ts[0]);
                CHECK_EPS(c[8], ts[1]);
            }
        }
    });
    run("extension (book Ch. 12, no reference counterpart): the normal on the surface of a cube", [] {
        const double cases[8][6] = {{1, 0.5, -0.8, 1, 0, 0},  {-1, -0.2, 0.9, -1, 0, 0}, {-0.4, 1, -0.1, 0, 1, 0},
                                    {0.3, -1, -0.7, 0, -1, 0}, {-0.6, 0.3, 1, 0, 0, 1},   {0.4, 0.4, -1, 0, 0, -1},
                                    {1, 1, 1, 1, 0, 0},        {-1, -1, -1, -1, 0, 0}};
        for (const auto& c : cases)
            CHECK_VEC(initVector(c[3], c[4], c[5]), localNormalAt(GEO_CUBE, initPoint(c[0], c[1], c[2])));
    });
    run("extension: a transformed cube through Shape.intersect / Shape.normalAt, both integrators", [] {
        World w;
        w.light = PointLight{initPoint(-10, 10, -10), Color{1, 1, 1}};
        Shape cube;
        cube.geo = GEO_CUBE;
        cube.transform = Mat4::identity().scale(1, 2, 3).translate(0, 0, 4);
        cube.material.transparency = 0.5;
        cube.material.reflective = 0.3;
        cube.material.refractive_index = 1.5;
        w.objects.push_back(cube);
        Shape floor = planeShape();
        floor.transform = Mat4::identity().translate(0, -2.5, 0);
        w.objects.push_back(floor);
        double ts[2] = {0, 0};
        CHECK(intersectShape(cube, Ray{initPoint(0, 0, -5), initVector(0, 0, 1)}, ts) == 2);
        CHECK_EPS(6.0, ts[0]);
        CHECK_EPS(12.0, ts[1]);
        CHECK_VEC(initVector(0, 0, -1), normalAt(cube, initPoint(0.2, 0.3, 1)));
        CHECK_VEC(initVector(0, 1, 0), normalAt(cube, initPoint(0.2, 2, 3)));
        w.prepare();
        for (double dx = -0.4; dx < 0.5; dx += 0.1)
            for (double oz : {-5.0, 4.0}) {  // from outside and from inside the cube
                const Ray r{initPoint(0.1, 0.2, oz), initVector(dx, -0.3, 1).normalize()};
                const Color a = worldColorAt(w, r, 5, LITERAL), b = worldColorAt(w, r, 5, STREAMING);
                CHECK(a.red == b.red && a.green == b.green && a.blue == b.blue);
            }
    });

    std::printf("%s: %d failed\n", g_fail ? "KAT FAILED" : "KAT PASSED", g_fail);
    return g_fail ? 1 : 0;
}
