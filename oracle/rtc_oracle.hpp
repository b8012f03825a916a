// TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's render path.
//
// This file is the parity oracle and the single-thread CPU baseline.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it;
// the product library (librtc_hip.so) never includes or links anything from oracle/.
//
// It restates, in scalar C++17 (compile with -O2 -ffp-contract=off, no -ffast-math,
// no -march flags so that no FMA is ever formed), the arithmetic of the Zig sources of
// stevschmid/raytracer-challenge, keeping the order of every floating-point operation.
// Each function cites the reference lines it follows.
//
// Two interchangeable integrators are provided:
//   LITERAL   — the reference's own cost model: a cofactor inverse on every call, a
//               growing intersection list that stores whole shapes by value, a stable
//               sort, and the container walk with structural equality (calc.zig:269-291).
//   STREAMING — inverses hoisted to world set-up, a running minimum instead of the
//               sorted list and a closed form for n1/n2.  Same arithmetic, same results.
//
// Parity pinning: the restatement is checked against every known-answer test the
// reference holds for this path (oracle/kat.cpp ports them).  The reference commits no
// image, and the Zig toolchain is absent, so image-level output is self-generated;
// std.math.pow/tan/sin/cos of the unpinned Zig standard library are "parity unpinned"
// at the last-bit level (only the reference's 1e-4/1e-5 known answers cover them).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <vector>

namespace orc {

constexpr double F64_EPSILON = std::numeric_limits<double>::epsilon();

// utils.zig:3-5 — approxEqAbs(a, b, f64 epsilon): |a-b| <= eps
inline bool epsilonEq(double a, double b) { return std::fabs(a - b) <= F64_EPSILON; }

// ---------------------------------------------------------------- vector.zig:20-118
struct Vec4 {
    double x, y, z, w;
    Vec4 add(const Vec4& o) const { return {x + o.x, y + o.y, z + o.z, w + o.w}; }
    Vec4 sub(const Vec4& o) const { return {x - o.x, y - o.y, z - o.z, w - o.w}; }
    Vec4 negate() const { return {-x, -y, -z, -w}; }
    // vector.zig:71-78: scalar * component
    Vec4 scale(double s) const { return {s * x, s * y, s * z, s * w}; }
    // vector.zig:84-86: left-to-right sum of four squares, hardware sqrt
    double length() const { return std::sqrt(x * x + y * y + z * z + w * w); }
    // vector.zig:88-97: four divisions by the length (no reciprocal)
    Vec4 normalize() const {
        const double len = length();
        return {x / len, y / len, z / len, w / len};
    }
    // vector.zig:99-104
    double dot(const Vec4& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
    // vector.zig:106-112
    Vec4 cross(const Vec4& o) const {
        return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x, 0.0};
    }
    // vector.zig:114-117
    Vec4 reflect(const Vec4& n) const {
        const double dp = dot(n);
        return sub(n.scale(2.0 * dp));
    }
    bool eql(const Vec4& o) const {
        return epsilonEq(x, o.x) && epsilonEq(y, o.y) && epsilonEq(z, o.z) && epsilonEq(w, o.w);
    }
};
inline Vec4 initPoint(double x, double y, double z) { return {x, y, z, 1.0}; }   // vector.zig:4
inline Vec4 initVector(double x, double y, double z) { return {x, y, z, 0.0}; }  // vector.zig:8

// ---------------------------------------------------------------- color.zig:4-59
struct Color {
    double red, green, blue;
    Color add(const Color& o) const { return {red + o.red, green + o.green, blue + o.blue}; }
    Color sub(const Color& o) const { return {red - o.red, green - o.green, blue - o.blue}; }
    Color scale(double s) const { return {s * red, s * green, s * blue}; }
    Color mult(const Color& o) const { return {red * o.red, green * o.green, blue * o.blue}; }
};
constexpr Color Black{0, 0, 0};
constexpr Color White{1, 1, 1};

// ---------------------------------------------------------------- matrix.zig:7-301
struct Mat2 {
    double mat[2][2];
    double determinant() const { return mat[0][0] * mat[1][1] - mat[0][1] * mat[1][0]; }  // :23-26
};

struct Mat3 {
    double mat[3][3];
    Mat2 submatrix(int row, int col) const {  // matrix.zig:54-75
        Mat2 r{};
        int dy = 0;
        for (int sy = 0; sy < 3; ++sy) {
            if (sy == row) continue;
            int dx = 0;
            for (int sx = 0; sx < 3; ++sx) {
                if (sx == col) continue;
                r.mat[dy][dx] = mat[sy][sx];
                ++dx;
            }
            ++dy;
        }
        return r;
    }
    double minor(int row, int col) const { return submatrix(row, col).determinant(); }
    double cofactor(int row, int col) const {  // matrix.zig:82-86
        const double d = minor(row, col);
        return ((row + col) % 2 != 0) ? -d : d;
    }
    double determinant() const {  // matrix.zig:88-96: accumulate from 0 over row 0
        double det = 0;
        for (int col = 0; col < 3; ++col) det += mat[0][col] * cofactor(0, col);
        return det;
    }
};

struct Mat4 {
    double mat[4][4];

    static Mat4 identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }

    // matrix.zig:197-213: four products summed left to right
    Mat4 mult(const Mat4& o) const {
        Mat4 r{};
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col)
                r.mat[row][col] = mat[row][0] * o.mat[0][col] + mat[row][1] * o.mat[1][col] +
                                  mat[row][2] * o.mat[2][col] + mat[row][3] * o.mat[3][col];
        return r;
    }
    // matrix.zig:215-224
    Vec4 multVec(const Vec4& v) const {
        return {mat[0][0] * v.x + mat[0][1] * v.y + mat[0][2] * v.z + mat[0][3] * v.w,
                mat[1][0] * v.x + mat[1][1] * v.y + mat[1][2] * v.z + mat[1][3] * v.w,
                mat[2][0] * v.x + mat[2][1] * v.y + mat[2][2] * v.z + mat[2][3] * v.w,
                mat[3][0] * v.x + mat[3][1] * v.y + mat[3][2] * v.z + mat[3][3] * v.w};
    }
    Mat4 transpose() const {  // matrix.zig:226-236
        Mat4 r{};
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col) r.mat[row][col] = mat[col][row];
        return r;
    }
    Mat3 submatrix(int row, int col) const {  // matrix.zig:238-259
        Mat3 r{};
        int dy = 0;
        for (int sy = 0; sy < 4; ++sy) {
            if (sy == row) continue;
            int dx = 0;
            for (int sx = 0; sx < 4; ++sx) {
                if (sx == col) continue;
                r.mat[dy][dx] = mat[sy][sx];
                ++dx;
            }
            ++dy;
        }
        return r;
    }
    double minor(int row, int col) const { return submatrix(row, col).determinant(); }
    double cofactor(int row, int col) const {  // matrix.zig:266-270
        const double d = minor(row, col);
        return ((row + col) % 2 != 0) ? -d : d;
    }
    double determinant() const {  // matrix.zig:272-280
        double det = 0;
        for (int col = 0; col < 4; ++col) det += mat[0][col] * cofactor(0, col);
        return det;
    }
    bool isInvertible() const { return !epsilonEq(determinant(), 0.0); }  // matrix.zig:282
    // matrix.zig:286-300: inv[row][col] = cofactor(col,row) / det.  The reference asserts
    // invertibility (a panic in safe builds); callers here check isInvertible() first.
    Mat4 inverse() const {
        const double det = determinant();
        Mat4 r{};
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col) r.mat[row][col] = cofactor(col, row) / det;
        return r;
    }

    // matrix.zig:116-180: the fluent builders PRE-multiply (m.translate(..) == T * m)
    Mat4 translate(double x, double y, double z) const {
        return Mat4{{{1, 0, 0, x}, {0, 1, 0, y}, {0, 0, 1, z}, {0, 0, 0, 1}}}.mult(*this);
    }
    Mat4 scale(double x, double y, double z) const {
        return Mat4{{{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}}}.mult(*this);
    }
    // The reference calls Zig's std.math.sin/cos here; libm stands in (parity unpinned at
    // the last bit, host-side set-up only).
    Mat4 rotateX(double r) const {
        return Mat4{{{1, 0, 0, 0}, {0, std::cos(r), -std::sin(r), 0}, {0, std::sin(r), std::cos(r), 0}, {0, 0, 0, 1}}}
            .mult(*this);
    }
    Mat4 rotateY(double r) const {
        return Mat4{{{std::cos(r), 0, std::sin(r), 0}, {0, 1, 0, 0}, {-std::sin(r), 0, std::cos(r), 0}, {0, 0, 0, 1}}}
            .mult(*this);
    }
    Mat4 rotateZ(double r) const {
        return Mat4{{{std::cos(r), -std::sin(r), 0, 0}, {std::sin(r), std::cos(r), 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}
            .mult(*this);
    }
    Mat4 shear(double xy, double xz, double yx, double yz, double zx, double zy) const {
        return Mat4{{{1, xy, xz, 0}, {yx, 1, yz, 0}, {zx, zy, 1, 0}, {0, 0, 0, 1}}}.mult(*this);
    }
    bool eql(const Mat4& o) const {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c)
                if (!epsilonEq(mat[r][c], o.mat[r][c])) return false;
        return true;
    }
};

// ---------------------------------------------------------------- pattern/material/shape/light/world
enum PatternKind { PAT_POINT = 0, PAT_STRIPE = 1, PAT_GRADIENT = 2, PAT_RING = 3, PAT_CHECKERS = 4 };
// GEO_CUBE is NOT in the reference (shape.zig:76-79 has sphere and plane): it is the next primitive
// the `geo` union is the hook for, written the way the reference writes its two (see cubeLocalIntersect).
enum GeoKind { GEO_SPHERE = 0, GEO_PLANE = 1, GEO_CUBE = 2 };

struct Pattern {  // pattern.zig:64-75
    int kind = PAT_POINT;
    Color a{0, 0, 0}, b{0, 0, 0};
    Mat4 transform = Mat4::identity();
};

struct Material {  // material.zig:6-18
    Color color{1, 1, 1};
    double ambient = 0.1;
    double diffuse = 0.9;
    double specular = 0.9;
    double shininess = 200.0;
    double reflective = 0.0;
    double transparency = 0.0;
    double refractive_index = 1.0;
    bool has_pattern = false;
    Pattern pattern{};
};

struct Shape {  // shape.zig:73-82
    int geo = GEO_SPHERE;
    Mat4 transform = Mat4::identity();
    Material material{};
};

struct PointLight {  // light.zig:9-14
    Vec4 position = initPoint(0, 0, 0);
    Color intensity = White;
};

// std.meta.eql on the reference's structs: field-wise ==, floats with == (so -0 == 0 and
// NaN != NaN), optionals and tagged unions by tag then payload.  (calc.zig:274,280)
inline bool eqlMat4(const Mat4& a, const Mat4& b) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            if (!(a.mat[r][c] == b.mat[r][c])) return false;
    return true;
}
inline bool eqlColor(const Color& a, const Color& b) {
    return a.red == b.red && a.green == b.green && a.blue == b.blue;
}
inline bool eqlPattern(const Pattern& a, const Pattern& b) {
    if (a.kind != b.kind) return false;
    if (a.kind != PAT_POINT && !(eqlColor(a.a, b.a) && eqlColor(a.b, b.b))) return false;
    return eqlMat4(a.transform, b.transform);
}
inline bool eqlMaterial(const Material& a, const Material& b) {
    if (!(eqlColor(a.color, b.color) && a.ambient == b.ambient && a.diffuse == b.diffuse &&
          a.specular == b.specular && a.shininess == b.shininess && a.reflective == b.reflective &&
          a.transparency == b.transparency && a.refractive_index == b.refractive_index))
        return false;
    if (a.has_pattern != b.has_pattern) return false;
    return !a.has_pattern || eqlPattern(a.pattern, b.pattern);
}
inline bool eqlShape(const Shape& a, const Shape& b) {
    return a.geo == b.geo && eqlMat4(a.transform, b.transform) && eqlMaterial(a.material, b.material);
}

// ---------------------------------------------------------------- ray.zig:14-37
struct Ray {
    Vec4 origin, direction;
    Vec4 position(double t) const { return origin.add(direction.scale(t)); }  // ray.zig:27-29
    Ray transform(const Mat4& m) const { return {m.multVec(origin), m.multVec(direction)}; }  // :31-36
};

// ray.zig:57-60 — the reference stores the whole Shape by value in every intersection.
struct Intersection {
    double t;
    Shape object;
};
inline bool eqlIntersection(const Intersection& a, const Intersection& b) {
    return a.t == b.t && eqlShape(a.object, b.object);
}

struct Intersections {  // ray.zig:62-93
    std::vector<Intersection> list;
    static bool lessThan(const Intersection& a, const Intersection& b) { return a.t < b.t; }  // :89-92
    // ray.zig:75-83: stable sort, then the first element with t >= 0
    std::optional<Intersection> hit() {
        std::stable_sort(list.begin(), list.end(), lessThan);
        for (const auto& i : list)
            if (i.t >= 0) return i;
        return std::nullopt;
    }
};

// Counters shared by both integrators (see RtcStats in include/rtc_hip.h).
struct Stats {
    uint64_t radiance_rays = 0, shadow_rays = 0, sphere_tests = 0, plane_tests = 0;
    uint64_t sphere_hits = 0, plane_hits = 0, shaded_hits = 0;
    uint64_t cube_tests = 0, cube_hits = 0;
};

struct World {  // world.zig:15-28
    std::vector<Shape> objects;
    PointLight light{};
    // STREAMING mode only: inverses computed once (same bits as the per-call recomputation,
    // because Mat4::inverse is a deterministic function of the transform).
    std::vector<Mat4> inv;
    std::vector<Mat4> pattern_inv;
    mutable Stats stats;
    void prepare() {
        inv.clear();
        pattern_inv.clear();
        for (const auto& o : objects) {
            inv.push_back(o.transform.inverse());
            pattern_inv.push_back(o.material.has_pattern ? o.material.pattern.transform.inverse()
                                                         : Mat4::identity());
        }
    }
};

// world.zig:30-54
inline World defaultWorld() {
    World w;
    w.light = {initPoint(-10, 10, -10), White};
    Shape s1;
    s1.material.color = {0.8, 1.0, 0.6};
    s1.material.diffuse = 0.7;
    s1.material.specular = 0.2;
    w.objects.push_back(s1);
    Shape s2;
    s2.transform = Mat4::identity().scale(0.5, 0.5, 0.5);
    w.objects.push_back(s2);
    return w;
}

// ---------------------------------------------------------------- Zig std.math.pow
// The reference calls std.math.pow(f64, x, y) at calc.zig:67,1024,1025.  Zig's standard
// library is not part of the reference tree and its version is unpinned (0.9.x inferred);
// its pow is a port of Go's math.Pow: special cases, frexp of x, binary exponentiation
// over the integer part of y on the mantissa with renormalisation, exp(yf*ln(x)) for a
// fractional part, then scalbn.  Restated here from that published algorithm.  For the
// integer exponents this path meets (2, 5 and the shininess values) it equals plain
// square-and-multiply with the multiplications in the same order.
namespace zigmath {

// musl/FreeBSD exp (Zig std/math/exp.zig is a port of it)
inline double exp64(double x) {
    static const double half[2] = {0.5, -0.5};
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    uint32_t hx = (uint32_t)(bits >> 32);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx >= 0x4086232b) {
        if (std::isnan(x)) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023;
        if (x < -708.39641853226410622 && x < -745.13321910194110842) return 0;
    }
    double hi, lo;
    int k;
    if (hx > 0x3fd62e42) {
        if (hx >= 0x3ff0a2b2)
            k = (int)(invln2 * x + half[sign]);
        else
            k = 1 - sign - sign;
        hi = x - k * ln2hi;
        lo = k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000) {
        k = 0;
        hi = x;
        lo = 0;
    } else {
        return 1 + x;
    }
    const double xx = x * x;
    const double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    const double y = 1 + (x * c / (2 - c) - lo + hi);
    if (k == 0) return y;
    return std::ldexp(y, k);
}

// musl/FreeBSD log (Zig std/math/ln.zig is a port of it)
inline double ln64(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ix;
    std::memcpy(&ix, &x, 8);
    uint32_t hx = (uint32_t)(ix >> 32);
    int k = 0;
    if (hx < 0x00100000 || (hx >> 31)) {
        if ((ix << 1) == 0) return -1 / (x * x);
        if (hx >> 31) return (x - x) / 0.0;
        k -= 54;
        x *= 0x1p54;
        std::memcpy(&ix, &x, 8);
        hx = (uint32_t)(ix >> 32);
    } else if (hx >= 0x7ff00000) {
        return x;
    } else if (hx == 0x3ff00000 && (ix << 32) == 0) {
        return 0;
    }
    hx += 0x3ff00000 - 0x3fe6a09e;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffff) + 0x3fe6a09e;
    ix = ((uint64_t)hx << 32) | (ix & 0xffffffff);
    std::memcpy(&x, &ix, 8);
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

inline bool isOddInteger(double x) {
    double ip;
    const double fp = std::modf(x, &ip);
    if (fp != 0) return false;
    return std::fmod(ip, 2.0) != 0;  // |ip| < 2^53 assumed odd/even decidable; beyond that even
}

inline double pow(double x, double y) {
    if (y == 0 || x == 1) return 1;
    if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
    if (y == 1) return x;
    if (x == 0) {
        if (y < 0) {
            if (isOddInteger(y)) return std::copysign(std::numeric_limits<double>::infinity(), x);
            return std::numeric_limits<double>::infinity();
        }
        if (isOddInteger(y)) return x;
        return 0;
    }
    if (std::isinf(y)) {
        if (x == -1) return 1;
        if ((std::fabs(x) < 1) == (y > 0)) return 0;
        return std::numeric_limits<double>::infinity();
    }
    if (std::isinf(x)) {
        if (x < 0) {
            if (y < 0) return isOddInteger(y) ? -0.0 : 0.0;
            return isOddInteger(y) ? -std::numeric_limits<double>::infinity()
                                   : std::numeric_limits<double>::infinity();
        }
        return y < 0 ? 0.0 : std::numeric_limits<double>::infinity();
    }
    if (y == 0.5) return std::sqrt(x);
    if (y == -0.5) return 1 / std::sqrt(x);

    double yi;
    double yf = std::modf(std::fabs(y), &yi);
    if (yf != 0 && x < 0) return std::numeric_limits<double>::quiet_NaN();
    if (yi >= 9223372036854775808.0) return exp64(y * ln64(x));

    double a1 = 1.0;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5) {
            yf -= 1;
            yi += 1;
        }
        a1 = exp64(yf * ln64(x));
    }
    int xe;
    double x1 = std::frexp(x, &xe);
    int64_t i = (int64_t)yi;
    while (i != 0) {
        if (xe < -(1 << 12) || (1 << 12) < xe) {
            ae += xe;
            break;
        }
        if (i & 1) {
            a1 *= x1;
            ae += xe;
        }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5) {
            x1 += x1;
            xe -= 1;
        }
        i >>= 1;
    }
    if (y < 0) {
        a1 = 1 / a1;
        ae = -ae;
    }
    return std::ldexp(a1, ae);
}
}  // namespace zigmath

// ---------------------------------------------------------------- shape.zig:17-108
// Sphere.localIntersect, shape.zig:23-45.  Both roots are always appended, t1 first.
inline int sphereLocalIntersect(const Ray& ray, double out_t[2]) {
    const Vec4 shape_to_ray = ray.origin.sub(initPoint(0, 0, 0));
    const double a = ray.direction.dot(ray.direction);
    const double b = 2.0 * ray.direction.dot(shape_to_ray);
    const double c = shape_to_ray.dot(shape_to_ray) - 1.0;
    const double discriminant = b * b - 4 * a * c;
    if (discriminant < 0) return 0;
    out_t[0] = (-b - std::sqrt(discriminant)) / (2 * a);
    out_t[1] = (-b + std::sqrt(discriminant)) / (2 * a);
    return 2;
}
// Plane.localIntersect, shape.zig:55-70
inline int planeLocalIntersect(const Ray& ray, double out_t[2]) {
    if (std::fabs(ray.direction.y) < F64_EPSILON) return 0;
    out_t[0] = -ray.origin.y / ray.direction.y;
    return 1;
}
// ---- extension beyond the reference: the axis-aligned cube [-1, 1]^3 --------------------------------
// The reference stops at Ch. 11 and has no cube; `Shape.geo` (shape.zig:76-79) is where the book's next
// primitive would be added, with a localIntersect / localNormalAt pair like the two existing ones
// (shape.zig:17-70).  This restates the book's Ch. 12 algorithm in the reference's idiom: f64, the
// parallel test against std.math.epsilon(f64) as Plane.localIntersect does (shape.zig:62), std.math.max /
// std.math.min as plain comparisons (a > b ? a : b), both t always appended, smaller first.  There is no
// reference output to compare with: PARITY UNPINNED; the oracle defines the behaviour the device must match.
inline void cubeCheckAxis(double origin, double direction, double* tmin, double* tmax) {
    const double tmin_numerator = -1.0 - origin;
    const double tmax_numerator = 1.0 - origin;
    double lo, hi;
    if (std::fabs(direction) >= F64_EPSILON) {
        lo = tmin_numerator / direction;
        hi = tmax_numerator / direction;
    } else {
        // A ray parallel to the slab: inside it for good (-inf, +inf) or outside for good.  With the origin exactly
        // ON one of the two planes a numerator is zero and 0 * inf is not a number, which the comparisons below
        // would keep or drop depending on the axis; the decision taken here: the face plane belongs to the cube, a
        // zero numerator gets the bound it has for origins just inside (the same on all three axes).
        const double inf = std::numeric_limits<double>::infinity();
        lo = tmin_numerator == 0 ? -inf : tmin_numerator * inf;
        hi = tmax_numerator == 0 ? inf : tmax_numerator * inf;
    }
    if (lo > hi) {
        const double t = lo;
        lo = hi;
        hi = t;
    }
    *tmin = lo;
    *tmax = hi;
}
inline double max2(double a, double b) { return a > b ? a : b; }  // std.math.max
inline double min2(double a, double b) { return a < b ? a : b; }  // std.math.min
inline int cubeLocalIntersect(const Ray& ray, double out_t[2]) {
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    cubeCheckAxis(ray.origin.x, ray.direction.x, &xtmin, &xtmax);
    cubeCheckAxis(ray.origin.y, ray.direction.y, &ytmin, &ytmax);
    cubeCheckAxis(ray.origin.z, ray.direction.z, &ztmin, &ztmax);
    const double tmin = max2(max2(xtmin, ytmin), ztmin);
    const double tmax = min2(min2(xtmax, ytmax), ztmax);
    if (tmin > tmax) return 0;
    out_t[0] = tmin;
    out_t[1] = tmax;
    return 2;
}
inline Vec4 cubeLocalNormalAt(const Vec4& p) {
    const double ax = std::fabs(p.x), ay = std::fabs(p.y), az = std::fabs(p.z);
    const double maxc = max2(max2(ax, ay), az);
    if (maxc == ax) return initVector(p.x, 0, 0);
    if (maxc == ay) return initVector(0, p.y, 0);
    return initVector(0, 0, p.z);
}

inline Vec4 localNormalAt(int geo, const Vec4& p) {  // shape.zig:18-21, 49-53
    if (geo == GEO_SPHERE) return p.sub(initPoint(0, 0, 0)).normalize();
    if (geo == GEO_CUBE) return cubeLocalNormalAt(p);
    return initVector(0, 1, 0);
}
// Shape.normalAt, shape.zig:84-97, given the object's inverse
inline Vec4 normalAtInv(int geo, const Mat4& object_space, const Vec4& world_point) {
    const Vec4 local_point = object_space.multVec(world_point);
    const Vec4 local_normal = localNormalAt(geo, local_point);
    Vec4 world_normal = object_space.transpose().multVec(local_normal);
    world_normal.w = 0;
    return world_normal.normalize();
}
inline Vec4 normalAt(const Shape& s, const Vec4& world_point) {
    return normalAtInv(s.geo, s.transform.inverse(), world_point);
}
// Shape.intersect, shape.zig:99-107 (inverse recomputed on every call)
inline int intersectShape(const Shape& s, const Ray& world_ray, double out_t[2]) {
    const Mat4 object_space = s.transform.inverse();
    const Ray local_ray = world_ray.transform(object_space);
    if (s.geo == GEO_CUBE) return cubeLocalIntersect(local_ray, out_t);
    return s.geo == GEO_SPHERE ? sphereLocalIntersect(local_ray, out_t) : planeLocalIntersect(local_ray, out_t);
}

// ---------------------------------------------------------------- pattern.zig:15-92
inline Color patternLocal(const Pattern& p, const Vec4& pp) {
    switch (p.kind) {
        case PAT_POINT: return {pp.x, pp.y, pp.z};  // pattern.zig:85
        case PAT_STRIPE: {                          // pattern.zig:21-24
            const double c = std::floor(pp.x);
            return std::fmod(c, 2.0) == 0 ? p.a : p.b;
        }
        case PAT_GRADIENT: {  // pattern.zig:33-37
            const Color distance = p.b.sub(p.a);
            const double fraction = pp.x - std::floor(pp.x);
            return p.a.add(distance.scale(fraction));
        }
        case PAT_RING: {  // pattern.zig:46-49
            const double c = std::floor(std::sqrt(pp.x * pp.x + pp.z * pp.z));
            return std::fmod(c, 2.0) == 0 ? p.a : p.b;
        }
        default: {  // checkers, pattern.zig:58-61
            const double c = std::floor(pp.x) + std::floor(pp.y) + std::floor(pp.z);
            return std::fmod(c, 2.0) == 0 ? p.a : p.b;
        }
    }
}
// Pattern.patternAt, pattern.zig:77-91
inline Color patternAtInv(const Pattern& p, const Mat4& object_space, const Mat4& pattern_space,
                          const Vec4& world_point) {
    const Vec4 object_point = object_space.multVec(world_point);
    const Vec4 pattern_point = pattern_space.multVec(object_point);
    return patternLocal(p, pattern_point);
}
inline Color patternAt(const Pattern& p, const Shape& object, const Vec4& world_point) {
    return patternAtInv(p, object.transform.inverse(), p.transform.inverse(), world_point);
}

// ---------------------------------------------------------------- calc.zig:26-73
// `color` is the pattern colour or material.color, decided by the caller.
inline Color lightingWithColor(const Material& material, const Color& color, const PointLight& light,
                               const Vec4& position, const Vec4& eyev, const Vec4& normalv, bool in_shadow) {
    const Color effective_color = color.mult(light.intensity);
    const Vec4 lightv = light.position.sub(position).normalize();
    const Color ambient = effective_color.scale(material.ambient);
    const double light_dot_normal = lightv.dot(normalv);
    Color diffuse = Black, specular = Black;
    if (light_dot_normal > 0.0 && !in_shadow) {
        diffuse = effective_color.scale(material.diffuse * light_dot_normal);
        const Vec4 reflectv = lightv.negate().reflect(normalv);
        const double reflect_dot_eye = reflectv.dot(eyev);
        if (reflect_dot_eye > 0.0) {
            const double factor = zigmath::pow(reflect_dot_eye, material.shininess);
            specular = light.intensity.scale(material.specular * factor);
        }
    }
    return ambient.add(diffuse).add(specular);
}
inline Color lighting(const Shape& object, const PointLight& light, const Vec4& position, const Vec4& eyev,
                      const Vec4& normalv, bool in_shadow) {
    const Material& m = object.material;
    const Color color = m.has_pattern ? patternAt(m.pattern, object, position) : m.color;
    return lightingWithColor(m, color, light, position, eyev, normalv, in_shadow);
}

// ---------------------------------------------------------------- calc.zig:233-306
struct Computations {
    double t;
    Shape object;
    int object_index;  // STREAMING mode bookkeeping (-1 in LITERAL mode)
    Vec4 point, over_point, under_point, eyev, normalv, reflectv;
    double n1, n2;
    bool inside;
};

// calc.zig:200-214 — LITERAL
inline Intersections intersectWorld(const World& world, const Ray& world_ray) {
    Intersections total;
    for (const Shape& object : world.objects) {
        double ts[2];
        const int n = intersectShape(object, world_ray, ts);
        if (object.geo == GEO_SPHERE) {
            world.stats.sphere_tests++;
            if (n) world.stats.sphere_hits++;
        } else if (object.geo == GEO_CUBE) {
            world.stats.cube_tests++;
            if (n) world.stats.cube_hits++;
        } else {
            world.stats.plane_tests++;
            if (n) world.stats.plane_hits++;
        }
        Intersections object_intersections;  // one heap list per object, as the reference
        for (int k = 0; k < n; ++k) object_intersections.list.push_back({ts[k], object});
        total.list.insert(total.list.end(), object_intersections.list.begin(), object_intersections.list.end());
    }
    std::stable_sort(total.list.begin(), total.list.end(), Intersections::lessThan);
    return total;
}

// calc.zig:247-306 — LITERAL (xs == nullptr is the reference's `xs: ?Intersections = null`)
inline Computations prepareComputations(const Intersection& intersection, const Ray& ray, const Intersections* xs) {
    Computations c{};
    c.t = intersection.t;
    c.object = intersection.object;
    c.object_index = -1;
    c.point = ray.position(intersection.t);
    c.eyev = ray.direction.negate();
    c.inside = false;
    c.normalv = normalAt(intersection.object, c.point);
    if (c.normalv.dot(c.eyev) < 0) {
        c.inside = true;
        c.normalv = c.normalv.negate();
    }
    c.reflectv = ray.direction.reflect(c.normalv);
    const double epsilon = 0.0001;
    c.over_point = c.point.add(c.normalv.scale(epsilon));
    c.under_point = c.point.sub(c.normalv.scale(epsilon));
    c.n1 = 1.0;
    c.n2 = 1.0;
    if (xs != nullptr) {
        std::vector<Shape> containers;
        for (const Intersection& i : xs->list) {
            const bool is_hit = eqlIntersection(i, intersection);
            if (is_hit) c.n1 = containers.empty() ? 1.0 : containers.back().material.refractive_index;
            bool removed = false;
            for (size_t idx = 0; idx < containers.size(); ++idx) {
                if (eqlShape(containers[idx], i.object)) {
                    containers.erase(containers.begin() + (long)idx);
                    removed = true;
                    break;
                }
            }
            if (!removed) containers.push_back(i.object);
            if (is_hit) {
                c.n2 = containers.empty() ? 1.0 : containers.back().material.refractive_index;
                break;
            }
        }
    }
    return c;
}

// calc.zig:1004-1026
inline double schlick(const Computations& comps) {
    double cos = comps.eyev.dot(comps.normalv);
    if (comps.n1 > comps.n2) {
        const double n = comps.n1 / comps.n2;
        const double sin2_t = n * n * (1.0 - cos * cos);
        if (sin2_t > 1.0) return 1.0;
        const double cos_t = std::sqrt(1.0 - sin2_t);
        cos = cos_t;
    }
    const double r0 = zigmath::pow((comps.n1 - comps.n2) / (comps.n1 + comps.n2), 2.0);
    return r0 + (1 - r0) * zigmath::pow(1 - cos, 5.0);
}

enum Mode { LITERAL = 0, STREAMING = 1 };

inline Color worldColorAt(const World& world, const Ray& ray, int remaining, Mode mode);

// calc.zig:777-790 — LITERAL
inline bool isShadowedLiteral(const World& world, const Vec4& point) {
    world.stats.shadow_rays++;
    const Vec4 v = world.light.position.sub(point);
    const double distance = v.length();
    const Vec4 direction = v.normalize();
    const Ray ray{point, direction};
    Intersections xs = intersectWorld(world, ray);
    const auto hit = xs.hit();
    return hit.has_value() && hit->t < distance;
}

// STREAMING sweep: nearest t >= 0 in the reference's order (ascending object index,
// t1 before t2, strict <: ray.zig:75-92 and calc.zig:204-211) and, optionally, the two
// innermost open containers for the n1/n2 closed form.
struct SweepResult {
    int hit_index = -1;
    double hit_t = 0;
    bool hit_is_exit_of_open_sphere = false;  // hit is t2 of a sphere with t1 < 0
    int top1 = -1, top2 = -1;                 // open containers, innermost first
    double top1_t = 0, top2_t = 0;
};
inline SweepResult sweepStreaming(const World& world, const Ray& ray) {
    SweepResult r;
    const int n = (int)world.objects.size();
    auto push_container = [&](double t, int idx) {
        // ordering of the reference's sorted list among t < 0: by t ascending, ties by
        // insertion order; the LAST one is the innermost.  Visiting ascending idx, >= keeps
        // the later object on ties.
        if (r.top1 < 0 || t >= r.top1_t) {
            r.top2 = r.top1;
            r.top2_t = r.top1_t;
            r.top1 = idx;
            r.top1_t = t;
        } else if (r.top2 < 0 || t >= r.top2_t) {
            r.top2 = idx;
            r.top2_t = t;
        }
    };
    for (int i = 0; i < n; ++i) {
        const Shape& o = world.objects[(size_t)i];
        const Ray local = ray.transform(world.inv[(size_t)i]);
        double ts[2];
        if (o.geo == GEO_SPHERE || o.geo == GEO_CUBE) {
            // two t per shape, the smaller first: a convex solid the origin is inside of is an open container
            if (o.geo == GEO_SPHERE) {
                world.stats.sphere_tests++;
                if (sphereLocalIntersect(local, ts) == 0) continue;
                world.stats.sphere_hits++;
            } else {
                world.stats.cube_tests++;
                if (cubeLocalIntersect(local, ts) == 0) continue;
                world.stats.cube_hits++;
            }
            const double t1 = ts[0], t2 = ts[1];
            if (t1 >= 0) {
                if (r.hit_index < 0 || t1 < r.hit_t) {
                    r.hit_index = i;
                    r.hit_t = t1;
                    r.hit_is_exit_of_open_sphere = false;
                }
                // t2 >= t1 can never beat t1 under strict <
            } else if (t2 >= 0) {
                push_container(t1, i);
                if (r.hit_index < 0 || t2 < r.hit_t) {
                    r.hit_index = i;
                    r.hit_t = t2;
                    r.hit_is_exit_of_open_sphere = true;
                }
            }
            // both negative: the sphere opens and closes before the hit — no effect
        } else {
            world.stats.plane_tests++;
            if (planeLocalIntersect(local, ts) == 0) continue;
            world.stats.plane_hits++;
            const double t = ts[0];
            if (t >= 0) {
                if (r.hit_index < 0 || t < r.hit_t) {
                    r.hit_index = i;
                    r.hit_t = t;
                    r.hit_is_exit_of_open_sphere = false;
                }
            } else if (t < 0) {
                push_container(t, i);  // a plane behind the origin never closes
            }
        }
    }
    return r;
}

inline bool isShadowedStreaming(const World& world, const Vec4& point) {
    world.stats.shadow_rays++;
    const Vec4 v = world.light.position.sub(point);
    const double distance = v.length();
    const Vec4 direction = v.normalize();
    const SweepResult s = sweepStreaming(world, Ray{point, direction});
    return s.hit_index >= 0 && s.hit_t < distance;
}

// calc.zig:945-959
inline Color reflectedColor(const World& world, const Computations& comps, int remaining, Mode mode) {
    if (remaining <= 0) return Black;
    const double reflective = comps.object.material.reflective;
    if (std::fabs(reflective - 0.0) <= F64_EPSILON) return Black;
    const Ray reflect_ray{comps.over_point, comps.reflectv};
    const Color color = worldColorAt(world, reflect_ray, remaining - 1, mode);
    return color.scale(reflective);
}

// calc.zig:828-853
inline Color refractedColor(const World& world, const Computations& comps, int remaining, Mode mode) {
    if (remaining <= 0) return Black;
    if (std::fabs(comps.object.material.transparency - 0.0) <= F64_EPSILON) return Black;
    const double n_ratio = comps.n1 / comps.n2;
    const double cos_i = comps.eyev.dot(comps.normalv);
    const double sin2_t = n_ratio * n_ratio * (1 - cos_i * cos_i);
    if (sin2_t > 1.0) return Black;
    const double cos_t = std::sqrt(1.0 - sin2_t);
    const Vec4 direction = comps.normalv.scale(n_ratio * cos_i - cos_t).sub(comps.eyev.scale(n_ratio));
    const Ray refract_ray{comps.under_point, direction};
    const Color color = worldColorAt(world, refract_ray, remaining - 1, mode);
    return color.scale(comps.object.material.transparency);
}

// calc.zig:448-470
inline Color shadeHit(const World& world, const Computations& comps, int remaining, Mode mode) {
    world.stats.shaded_hits++;
    const bool in_shadow =
        mode == LITERAL ? isShadowedLiteral(world, comps.over_point) : isShadowedStreaming(world, comps.over_point);
    Color surface;
    if (mode == LITERAL || comps.object_index < 0) {
        surface = lighting(comps.object, world.light, comps.over_point, comps.eyev, comps.normalv, in_shadow);
    } else {
        const Material& m = comps.object.material;
        const size_t oi = (size_t)comps.object_index;
        const Color color =
            m.has_pattern ? patternAtInv(m.pattern, world.inv[oi], world.pattern_inv[oi], comps.over_point) : m.color;
        surface = lightingWithColor(m, color, world.light, comps.over_point, comps.eyev, comps.normalv, in_shadow);
    }
    const Color reflected = reflectedColor(world, comps, remaining, mode);
    const Color refracted = refractedColor(world, comps, remaining, mode);
    const Material& material = comps.object.material;
    if (material.reflective > 0.0 && material.transparency > 0.0) {
        const double reflectance = schlick(comps);
        return surface.add(reflected.scale(reflectance)).add(refracted.scale(1.0 - reflectance));
    }
    return surface.add(reflected).add(refracted);
}

// calc.zig:641-652
inline Color worldColorAt(const World& world, const Ray& ray, int remaining, Mode mode) {
    world.stats.radiance_rays++;
    if (mode == LITERAL) {
        Intersections xs = intersectWorld(world, ray);
        const auto hit = xs.hit();
        if (!hit.has_value()) return Black;
        const Computations comps = prepareComputations(*hit, ray, &xs);
        return shadeHit(world, comps, remaining, mode);
    }
    const SweepResult s = sweepStreaming(world, ray);
    if (s.hit_index < 0) return Black;
    const size_t hi = (size_t)s.hit_index;
    const Shape& obj = world.objects[hi];
    Computations c{};
    c.t = s.hit_t;
    c.object = obj;
    c.object_index = s.hit_index;
    c.point = ray.position(s.hit_t);
    c.eyev = ray.direction.negate();
    c.inside = false;
    c.normalv = normalAtInv(obj.geo, world.inv[hi], c.point);
    if (c.normalv.dot(c.eyev) < 0) {
        c.inside = true;
        c.normalv = c.normalv.negate();
    }
    c.reflectv = ray.direction.reflect(c.normalv);
    const double epsilon = 0.0001;
    c.over_point = c.point.add(c.normalv.scale(epsilon));
    c.under_point = c.point.sub(c.normalv.scale(epsilon));
    // n1/n2 closed form: every list entry before the hit has t < 0, so the open containers
    // at the hit are the spheres with t1 < 0 <= t2 and the planes with t < 0.
    auto ri = [&](int idx) { return idx < 0 ? 1.0 : world.objects[(size_t)idx].material.refractive_index; };
    c.n1 = ri(s.top1);
    if (s.hit_is_exit_of_open_sphere)
        c.n2 = (s.top1 == s.hit_index) ? ri(s.top2) : ri(s.top1);
    else
        c.n2 = obj.material.refractive_index;
    return shadeHit(world, c, remaining, mode);
}

// calc.zig:721-738
inline Mat4 viewTransform(const Vec4& from, const Vec4& to, const Vec4& up) {
    const Vec4 forward = to.sub(from).normalize();
    const Vec4 left = forward.cross(up.normalize());
    const Vec4 true_up = left.cross(forward);
    const Mat4 orientation{{{left.x, left.y, left.z, 0},
                            {true_up.x, true_up.y, true_up.z, 0},
                            {-forward.x, -forward.y, -forward.z, 0},
                            {0, 0, 0, 1}}};
    const Mat4 translation = Mat4::identity().translate(-from.x, -from.y, -from.z);
    return orientation.mult(translation);
}

// ---------------------------------------------------------------- camera.zig:17-90
struct Camera {
    uint64_t hsize = 0, vsize = 0;
    double field_of_view = 0, half_width = 0, half_height = 0, pixel_size = 0;
    Mat4 transform = Mat4::identity();

    // camera.zig:29-46 (Zig's std.math.tan; libm stands in, parity unpinned at the last bit)
    static Camera init(uint64_t hsize, uint64_t vsize, double field_of_view) {
        Camera c;
        const double half_view = std::tan(field_of_view * 0.5);
        const double aspect = (double)hsize / (double)vsize;
        c.half_width = (aspect >= 1.0) ? half_view : half_view * aspect;
        c.half_height = (aspect < 1.0) ? half_view : half_view / aspect;
        c.pixel_size = (2.0 * c.half_width) / (double)hsize;
        c.hsize = hsize;
        c.vsize = vsize;
        c.field_of_view = field_of_view;
        return c;
    }
    // camera.zig:48-72 (the inverse is recomputed per pixel in the reference)
    Ray rayForPixelInv(const Mat4& inv, uint64_t x, uint64_t y) const {
        const double x_offset = ((double)x + 0.5) * pixel_size;
        const double y_offset = ((double)y + 0.5) * pixel_size;
        const double world_x = half_width - x_offset;
        const double world_y = half_height - y_offset;
        const Vec4 pixel = inv.multVec(initPoint(world_x, world_y, -1));
        const Vec4 origin = inv.multVec(initPoint(0, 0, 0));
        const Vec4 direction = pixel.sub(origin).normalize();
        return {origin, direction};
    }
    Ray rayForPixel(uint64_t x, uint64_t y) const { return rayForPixelInv(transform.inverse(), x, y); }
};

// camera.zig:74-89 over rows [row_begin, row_end) stepping row_step; out is the full
// row-major canvas (canvas.zig:25), rows not visited are left untouched.
inline void render(const Camera& cam, World& world, int depth, Mode mode, double* out_rgb, uint64_t row_begin,
                   uint64_t row_end, uint64_t row_step) {
    if (mode == STREAMING) world.prepare();
    const Mat4 cam_inv = cam.transform.inverse();
    for (uint64_t y = row_begin; y < row_end; y += row_step) {
        for (uint64_t x = 0; x < cam.hsize; ++x) {
            const Ray ray = (mode == LITERAL) ? cam.rayForPixel(x, y) : cam.rayForPixelInv(cam_inv, x, y);
            const Color color = worldColorAt(world, ray, depth, mode);
            double* p = out_rgb + (y * cam.hsize + x) * 3;
            p[0] = color.red;
            p[1] = color.green;
            p[2] = color.blue;
        }
    }
}

// draw_sphere.zig:22-83 — the Ch. 6 "plumbing" program: one unit sphere of colour (1, 0.2, 1), a white
// light at (-10, 10, -10), the eye at (0, 0, -5), a wall of size 7 at z = 10 sampled at pixel CORNERS
// (world_x = -half + x * pixel_size, draw_sphere.zig:52,58), no shadows, lighting at the hit point itself.
// The file is stale against the mounted revision (it imports sphere.zig and calls calc.intersectSphere,
// calc.sphereNormalAt and a five-argument lighting, none of which exist any more); its loop is restated
// with what those became: Shape.intersect + Intersections.hit (shape.zig:99-107, ray.zig:75-83),
// Shape.normalAt (shape.zig:84-97) and calc.lighting with in_shadow = false (calc.zig:26-73).
// out_rgb: canvas_size^2 * 3 doubles, row-major, black where the ray misses (Canvas.init zero-fills,
// canvas.zig:12-22); out_rays (optional): the ray of every pixel, 6 doubles each.
inline void drawSphere(uint64_t canvas_size, double* out_rgb, double* out_rays) {
    Shape shape;
    shape.geo = GEO_SPHERE;
    shape.transform = Mat4::identity();
    shape.material.color = {1.0, 0.2, 1.0};  // draw_sphere.zig:30-32
    const Vec4 origin = initPoint(0, 0, -5);
    const double wall_z = 10.0;
    const double wall_size = 7.0;
    const double pixel_size = wall_size / (double)canvas_size;
    const double half = 0.5 * wall_size;
    PointLight light;
    light.intensity = {1, 1, 1};
    light.position = initPoint(-10, 10, -10);
    for (uint64_t y = 0; y < canvas_size; ++y) {
        const double world_y = half - (double)y * pixel_size;
        for (uint64_t x = 0; x < canvas_size; ++x) {
            const double world_x = -half + (double)x * pixel_size;
            const Vec4 wall_pos = initPoint(world_x, world_y, wall_z);
            const Vec4 ray_dir = wall_pos.sub(origin).normalize();
            const Ray ray{origin, ray_dir};
            double* px = out_rgb + (y * canvas_size + x) * 3;
            px[0] = px[1] = px[2] = 0.0;
            if (out_rays) {
                double* r = out_rays + (y * canvas_size + x) * 6;
                r[0] = origin.x, r[1] = origin.y, r[2] = origin.z;
                r[3] = ray_dir.x, r[4] = ray_dir.y, r[5] = ray_dir.z;
            }
            double ts[2];
            const int n = intersectShape(shape, ray, ts);
            Intersections xs;
            for (int k = 0; k < n; ++k) xs.list.push_back({ts[k], shape});
            const auto hit = xs.hit();
            if (hit.has_value()) {
                const Vec4 point = ray.position(hit->t);
                const Vec4 normal = normalAt(hit->object, point);
                const Vec4 eye = ray_dir.negate();
                const Color color = lighting(hit->object, light, point, eye, normal, false);
                px[0] = color.red, px[1] = color.green, px[2] = color.blue;
            }
        }
    }
}

// ppm.zig:6-9
inline uint8_t byteColorValue(double val) {
    double v = val * 256.0;
    // std.math.clamp(v, 0, 255): max(lower, min(v, upper)); NaN falls through both compares
    v = v < 255.0 ? v : 255.0;
    v = v > 0.0 ? v : 0.0;
    return (uint8_t)v;
}

}  // namespace orc
