// Device side of librtc_hip.so: the per-pixel render path of the reference
// (Camera.render -> calc.worldColorAt -> intersectWorld / prepareComputations / shadeHit /
// isShadowed / lighting / reflectedColor / refractedColor / schlick) as one persistent
// FP64 kernel for gfx950.
//
// Numerical contract: every floating-point expression below keeps the operation order of
// the Zig source it cites, no multiply-add is ever fused (this file is compiled with
// -ffp-contract=off and carries the pragma below), divisions and square roots are the
// correctly rounded IEEE ones.  Vectors are carried as three components: on this path
// points have w == 1 and directions w == 0 exactly, so the reference's fourth terms are
// an exact "+ m3" or an added zero, and results are value-identical (only the sign of a
// zero can differ).
//
// Execution model: one lane owns one pixel at a time.  The reference's recursion (a binary
// tree of at most 2^(depth+1)-1 radiance rays) becomes an explicit per-lane frame stack
// that is unwound in the reference's post-order, so the colour sums associate exactly as
// in calc.zig:464-469.  Every loop iteration each live lane processes one tree node: a
// radiance sweep over all shapes, then (on a hit) a shadow sweep and the shading.  The
// sweep visits every shape, as calc.intersectWorld does (a pair is cut short only where a
// conservative test with a stated margin proves that the reference's own arithmetic finds
// nothing that matters: the bounding-sphere tests, the light-side test of the shadow sweep,
// the root filters); the shapes are visited group by
// group (four kinds, one specialised loop each) rather than in list order, and equal t are
// resolved by list index, which selects what the reference's stable sort selects.  It is the
// same work for every lane whatever ray it holds, so a wavefront does not diverge over the
// shape list; the records are read with wave-uniform addresses into scalar registers.
// Lanes that finish a pixel take the next one from the 8x8 block their wave is dealing out;
// waves draw blocks from sixteen striped queues, costliest rows of tiles first; at the end of
// the frame idle lanes take over parked subtrees of busy lanes of their wave.  What a hit
// gathers per lane and the top of the frame stacks live in LDS.
#include "rtc_internal.h"

#pragma clang fp contract(off)

namespace rtc {
namespace {

constexpr double kF64Eps = 2.220446049250313080847263336181640625e-16;  // std.math.epsilon(f64)
constexpr double kSurfaceEps = 0.0001;                                   // calc.zig:262
constexpr int kBlock = 256;
#ifndef RTC_WAVES_PER_EU
#define RTC_WAVES_PER_EU 4  // minimum waves per SIMD the register allocation must allow (<= 128 VGPRs)
#endif
#ifndef RTC_WAVES_PER_EU_LARGE
#define RTC_WAVES_PER_EU_LARGE 4  // the same for the kernels of worlds beyond the on-chip tables (TABLES == 0)
#endif

struct V3 {
    double x, y, z;
};

// vector.zig:99-104 with the w*w' term known to be zero
__device__ __forceinline__ double dot3(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// vector.zig:84-97 (w == 0): length, then one division per component
__device__ __forceinline__ V3 normalize3(const V3& a) {
    const double len = sqrt((a.x * a.x + a.y * a.y) + a.z * a.z);
    return {a.x / len, a.y / len, a.z / len};
}
// matrix.zig:215-224 for a point (w == 1): ((m0*x + m1*y) + m2*z) + m3
__device__ __forceinline__ V3 xform_point(const double* __restrict__ m, const V3& p) {
    return {((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
            ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]};
}
// matrix.zig:215-224 for a direction (w == 0)
__device__ __forceinline__ V3 xform_dir(const double* __restrict__ m, const V3& d) {
    return {(m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z,
            (m[8] * d.x + m[9] * d.y) + m[10] * d.z};
}

// ---------------------------------------------------------------------------------------
// std.math.pow of the Zig standard library (called at calc.zig:67,1024,1025).  That library
// is outside the reference tree; its pow is a port of Go's math.Pow: special cases, frexp,
// binary exponentiation over the integer part of y on the mantissa with renormalisation,
// exp(yf*ln x) for a fractional part, scalbn.  Restated from that published algorithm.
__device__ double zig_exp(double x) {  // musl/FreeBSD e_exp.c, which Zig's exp ports
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    uint32_t hx = (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx >= 0x4086232b) {
        if (x != x) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023;
        if (x < -708.39641853226410622 && x < -745.13321910194110842) return 0;
    }
    double hi, lo;
    int k;
    if (hx > 0x3fd62e42) {
        if (hx >= 0x3ff0a2b2)
            k = (int)(invln2 * x + (sign ? -0.5 : 0.5));
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
    return ldexp(y, k);
}

__device__ double zig_ln(double x) {  // musl/FreeBSD e_log.c, which Zig's ln ports
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    int k = 0;
    if (hx < 0x00100000 || (hx >> 31)) {
        if ((ix << 1) == 0) return -1 / (x * x);
        if (hx >> 31) return (x - x) / 0.0;
        k -= 54;
        x *= 0x1p54;
        ix = (unsigned long long)__double_as_longlong(x);
        hx = (uint32_t)(ix >> 32);
    } else if (hx >= 0x7ff00000) {
        return x;
    } else if (hx == 0x3ff00000 && (ix << 32) == 0) {
        return 0;
    }
    hx += 0x3ff00000 - 0x3fe6a09e;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffff) + 0x3fe6a09e;
    ix = ((unsigned long long)hx << 32) | (ix & 0xffffffffull);
    x = __longlong_as_double((long long)ix);
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

__device__ __forceinline__ bool is_odd_integer(double v) {
    const double ip = trunc(v);
    if (v - ip != 0) return false;
    return fmod(ip, 2.0) != 0;
}

__device__ double zig_pow(double x, double y) {
    const double inf = __builtin_huge_val();
    if (y == 0 || x == 1) return 1;
    if (x != x || y != y) return __builtin_nan("");
    if (y == 1) return x;
    if (x == 0) {
        if (y < 0) return is_odd_integer(y) ? copysign(inf, x) : inf;
        return is_odd_integer(y) ? x : 0.0;
    }
    if (isinf(y)) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return inf;
    }
    if (isinf(x)) {
        if (x < 0) {
            if (y < 0) return is_odd_integer(y) ? -0.0 : 0.0;
            return is_odd_integer(y) ? -inf : inf;
        }
        return y < 0 ? 0.0 : inf;
    }
    if (y == 0.5) return sqrt(x);
    if (y == -0.5) return 1 / sqrt(x);

    const double ay = fabs(y);
    double yi = trunc(ay);
    double yf = ay - yi;  // exact: modf's fractional part
    if (yf != 0 && x < 0) return __builtin_nan("");
    if (yi >= 9223372036854775808.0) return zig_exp(y * zig_ln(x));

    double a1 = 1.0;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5) {
            yf -= 1;
            yi += 1;
        }
        a1 = zig_exp(yf * zig_ln(x));
    }
    int xe;
    double x1 = frexp(x, &xe);
    long long i = (long long)yi;
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
    return ldexp(a1, ae);
}

// zig_pow(x, n) for an integer 1 <= n < 500 known in advance (the shininess of a material, the
// 2 and 5 of Schlick), by the multiplications its loop amounts to; x of either sign, zero or
// not finite (frexp then returns x itself with exponent 0, and the products are the signed
// zeros, infinities or NaN zig_pow returns).  That loop squares and multiplies the mantissa of x,
// doubles a squared mantissa that has dropped below 0.5, and adds up the exponents on the side;
// doubling is exact, so its products are the plain products of the mantissa m of x scaled by
// powers of two, and round the same way as long as the plain ones stay in the normal range:
// every power formed here is at least 0.5^512.  The result is the same real number
// m^n * 2^(e n) before the one rounding of the final ldexp.
__device__ __forceinline__ double pow_small_integer(double x, int n) {
    int e;
    const double m = frexp(x, &e);
    double a = 1.0, p = m;
    int bits = n;
    while (bits != 0) {
        a = (bits & 1) ? a * p : a;
        p *= p;
        bits >>= 1;
    }
    return ldexp(a, e * n);
}
constexpr int kPowSmallLimit = 500;

// ---------------------------------------------------------------------------------------
// One sweep of a world ray over every shape: calc.intersectWorld (calc.zig:200-214) +
// Intersections.hit (ray.zig:75-83) as a running minimum.
//
// The reference sorts all t values (stable) and takes the first with t >= 0; visiting the
// shapes in list order, t1 before t2, and replacing the best only on strict < selects the
// same element.  For n1/n2 (calc.zig:269-291) everything the reference's container walk
// meets before the hit has t < 0, so the containers still open at the hit are the spheres
// with t1 < 0 <= t2 and the planes with t < 0 (a plane behind the origin never closes),
// ordered by entry t, ties by list order; only the innermost two can matter.
//
// The quadratic's coefficients (54 flops per sphere) and the plane's two row products (11)
// are always evaluated exactly as the reference does.  The roots (a square root and two
// divisions; one division for a plane) are only needed when they can change the outcome,
// so they are skipped when cheap bounds PROVE that the correctly rounded roots the reference
// would compute lie entirely behind the origin or beyond the best hit so far:
//   sphere behind   b > 0 and disc < b*b*(1-2^-40)  =>  sqrt(disc) < b  =>  t1 <= t2 < 0
//   sphere beyond   L = (-b - best*den) - 2^-40*(|b| + best*den) > 0 and L*L > disc*(1+2^-40)
//                   =>  (-b - sqrt(disc))/den > best by a margin of 2^-41 relative, far more
//                   than the few-ulp rounding of the reference's expression  =>  t2 >= t1 > best
//   plane beyond    |oy| > best*|dy|*(1+2^-40)  =>  t > best
//   plane behind    oy and dy of equal sign  =>  t < 0 (only a container, irrelevant for a
//                   shadow ray)
// A skipped root can neither become the hit nor (being > best >= 0, or closed again) an open
// container, so the result is unchanged.  The bounds assume t does not underflow to zero.
struct Sweep {
    double t;         // nearest t >= 0 (valid when idx >= 0); for a shadow sweep: starts at the light distance
    int idx;          // table position of the hit shape, -1 = miss; bit 30 set: the hit is t2 of a sphere
                      // whose t1 < 0 (that sphere is an open container)
    int top1, top2;   // innermost and second innermost open container (table positions), -1 = none
};
constexpr int kExitOpenBit = 1 << 30;

constexpr int kRecDoubles = rtc::kRecordDoubles;  // one shape record: 12 matrix entries, the list index, bounding sphere
#ifdef RTC_MUTATE_BOUND
constexpr double kBoundShrink = 1.0 + 0x1p-20;  // test of the tests: skips too eagerly
#else
constexpr double kBoundShrink = 1.0 - 0x1p-27;
#endif  // see the bounding-sphere test in sweep_shape and its host side
constexpr double kOneMinus = 1.0 - 0x1p-40;
constexpr double kOnePlus = 1.0 + 0x1p-40;

// STATS builds only: (ray, shape) pairs whose local intersect produced roots, and pairs the
// bounding-sphere test dismissed.
struct SweepCounters {
    uint32_t sphere_hits, plane_hits, sphere_skips, plane_skips, cube_hits, sphere_prefilter_skips;
};

struct ShapeRec {
    double m[12];
    long long order;  // index of the shape in the reference's list (world.objects)
    double bc[3];     // spheres: centre of a world-space bounding sphere ...
    double br2;       // ... and its squared radius, enlarged by the error margin (+inf: no bound)
    // planes: on which side of the plane the light lies (+1 / -1; 0 = unknown), how far (the
    // local y of the light, absolute value), and the guard factor of the same-side test below
    double light_side, light_height, side_guard;
    // spheres: the bounding-sphere test anchored at the light, for shadow rays (see sphere_from_light
    // on the host): centre minus light, and the ray-independent part of the test's left side
    double lc[3];
    double lk;
};
// Wave-uniform read of the record at table position i (scalar loads): what the sweep of the record's group
// reads, from where the group's layout keeps it (rtc_internal.h); the table ends with padding records so that
// reading ahead is always in bounds.
enum : int { kSphereAxis = 0, kSphereGeneral = 1, kPlaneAxis = 2, kPlaneGeneral = 3, kCube = 4 };
template <int KIND, bool SHADOW>
__device__ __forceinline__ ShapeRec load_rec(const double* __restrict__ tab, int i) {
    ShapeRec r;
    const double* __restrict__ p = tab + (size_t)i * kRecDoubles;
    if (KIND == kSphereAxis) {
        const double* __restrict__ m = p + (SHADOW ? kRecAxisSphereShadowM : kRecAxisSphereM);
        r.m[0] = m[0], r.m[3] = m[1], r.m[5] = m[2], r.m[7] = m[3], r.m[10] = m[4], r.m[11] = m[5];
        if (SHADOW) {
            r.lc[0] = p[kRecAxisSphereLight], r.lc[1] = p[kRecAxisSphereLight + 1], r.lc[2] = p[kRecAxisSphereLight + 2];
            r.lk = p[kRecAxisSphereLight + 3];
        } else {
            r.order = __double_as_longlong(p[kRecAxisSphereOrder]);
            r.br2 = p[kRecAxisSphereBr2];
            r.bc[0] = p[kRecAxisSphereBc], r.bc[1] = p[kRecAxisSphereBc + 1], r.bc[2] = p[kRecAxisSphereBc + 2];
        }
    } else if (KIND == kPlaneAxis) {
        r.m[5] = p[kRecAxisPlaneM], r.m[7] = p[kRecAxisPlaneM + 1];
        r.order = __double_as_longlong(p[kRecAxisPlaneOrder]);
        r.light_side = p[kRecAxisPlaneLight], r.light_height = p[kRecAxisPlaneLight + 1], r.side_guard = p[kRecAxisPlaneLight + 2];
    } else if (KIND == kPlaneGeneral) {
        r.m[4] = p[kRecPlaneM], r.m[5] = p[kRecPlaneM + 1], r.m[6] = p[kRecPlaneM + 2], r.m[7] = p[kRecPlaneM + 3];
        r.order = __double_as_longlong(p[kRecPlaneOrder]);
        r.light_side = p[kRecPlaneLight], r.light_height = p[kRecPlaneLight + 1], r.side_guard = p[kRecPlaneLight + 2];
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) r.m[k] = p[k];
        r.order = __double_as_longlong(p[kRecFullOrder]);
        r.bc[0] = p[kRecFullBound], r.bc[1] = p[kRecFullBound + 1], r.bc[2] = p[kRecFullBound + 2];
        r.br2 = p[kRecFullBound + 3];
        r.lc[0] = p[kRecFullLight], r.lc[1] = p[kRecFullLight + 1], r.lc[2] = p[kRecFullLight + 2];
        r.lk = p[kRecFullLight + 3];
    }
    return r;
}

// Per-lane list of the two innermost open containers, kept in LDS (24 B per lane).  It is
// touched only when a lane actually meets a container (a plane behind the ray origin, or a
// sphere the origin is inside), so it does not belong in registers: loop-carried registers
// that are modified under a divergent branch cost a register copy per variable per shape.
struct ContainerSlots {
    double t[2][kBlock];
    int i[2][kBlock];
};

// The table holds the shapes in groups (kSweepGroupTable), each in list order, so that each group is
// swept by a loop specialised for it: no per-shape dispatch, and the zero entries of
// axis-aligned inverses are never multiplied.
static_assert(kSweepGroupTable[kSphereAxis].geo == RTC_GEO_SPHERE && kSweepGroupTable[kSphereAxis].axis_aligned &&
                  kSweepGroupTable[kSphereGeneral].geo == RTC_GEO_SPHERE && kSweepGroupTable[kPlaneAxis].geo == RTC_GEO_PLANE &&
                  kSweepGroupTable[kPlaneAxis].axis_aligned && kSweepGroupTable[kPlaneGeneral].geo == RTC_GEO_PLANE &&
                  kSweepGroupTable[kCube].geo == RTC_GEO_CUBE && kSweepGroups == 5,
              "the specialised loops follow the group table");

// std.math.max / std.math.min as the plain comparisons they are (a NaN operand is kept or dropped
// exactly as the comparison decides, which fmax/fmin would not reproduce)
__device__ __forceinline__ double max_cmp(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double min_cmp(double a, double b) { return a < b ? a : b; }
// One slab of the cube [-1, 1]^3 (the book's check_axis, in the reference's idiom: the parallel test
// against the f64 epsilon as shape.zig:62 does, the swap by comparison)
__device__ __forceinline__ void cube_axis(double origin, double direction, double& lo, double& hi) {
    const double n0 = -1.0 - origin, n1 = 1.0 - origin;
    const bool parallel = !(fabs(direction) >= kF64Eps);
    const double inf = __builtin_huge_val();
    // (parallel and the origin exactly on a face plane: 0 * inf would be a NaN; the face belongs to the cube, so
    // a zero numerator gets the bound of an origin just inside - the oracle's cubeCheckAxis takes the same decision)
    const double a = parallel ? (n0 == 0 ? -inf : n0 * inf) : n0 / direction;
    const double b = parallel ? (n1 == 0 ? inf : n1 * inf) : n1 / direction;
    const bool swap = a > b;
    lo = swap ? b : a;
    hi = swap ? a : b;
}

// The bounding-sphere tests (see sweep_shape).  bc/br2: centre and enlarged squared radius of the world-space
// bounding sphere; lc/lk: the same anchored at the light, for shadow rays (sphere_from_light, rtc_api.hip).
__device__ __forceinline__ bool sphere_clear_of_ray(const double* bc, double br2, const V3& o, const V3& d, double dds) {
    const V3 oc = {bc[0] - o.x, bc[1] - o.y, bc[2] - o.z};
    const double ocd = fma(oc.z, d.z, fma(oc.y, d.y, oc.x * d.x));
    const double oc2 = fma(oc.z, oc.z, fma(oc.y, oc.y, oc.x * oc.x));
    // br2 holds the enlarged squared radius already divided by kBoundShrink and dds = |d|^2 kBoundShrink:
    // (oc2 kBoundShrink - r2) |d|^2 with one scalar operand per instruction
    const double apart = (oc2 - br2) * dds;
    return apart > ocd * ocd;
}
__device__ __forceinline__ bool sphere_clear_of_shadow_ray(const double* lc, double lk, const V3& d, double shadow_margin) {
    // a shadow ray ends at the light: the same test from there, with the sphere's part of it evaluated at upload
    const double lcd = fma(lc[2], d.z, fma(lc[1], d.y, lc[0] * d.x));
    return lk - shadow_margin > lcd * lcd;
}

// One shape against one ray.  The reference's list order decides ties (stable sort by t,
// ray.zig:75-92): among equal t the shape earlier in the list is the hit, and the one later
// in the list is the inner container.  The groups are not visited in list order, so ties
// are resolved explicitly with the list indices (`order`), in the rare branch where two t
// are equal.
template <int KIND, bool SHADOW, bool STATS, bool BOUND_DONE = false>
__device__ __forceinline__ void sweep_shape(const ShapeRec& rec, int pos, const int* __restrict__ order_of,
                                            const V3& o, const V3& d, double dd, double side_scale, double shadow_margin, Sweep& s,
                                            ContainerSlots& cs, SweepCounters& ctr) {
    const double* m = rec.m;
    const int tid = (int)threadIdx.x;
    // What this shape contributes, decided inside the branches below; the sweep state itself
    // is only modified in straight-line code after them.
    bool better = false;  // `cand` replaces the best hit
    double cand = 0.0;
    int cand_idx = pos;
    bool push = false;    // (push_t, pos) is an open container
    double push_t = 0.0;
    const bool have = SHADOW || s.idx >= 0;
    if (KIND == kSphereAxis || KIND == kSphereGeneral) {
        // A line that passes the shape's bounding sphere by more than the error margin gives
        // disc < 0 in the reference's arithmetic: nothing below would change any state.
        // (distance^2 - r^2) * |d|^2 > (oc.d)^2, evaluated with fused operations: the margin
        // folded into br2 and kBoundShrink is thousands of times their rounding error.
        bool clear;
        if (BOUND_DONE) {
            clear = false;  // the caller ran the test on the streamed bounds and calls for the lanes that failed it
        } else if (SHADOW) {
            clear = sphere_clear_of_shadow_ray(rec.lc, rec.lk, d, shadow_margin);
        } else {
            clear = sphere_clear_of_ray(rec.bc, rec.br2, o, d, dd);
        }
        if (STATS && clear) ++ctr.sphere_skips;
        if (!clear) {
        // Ray.transform (ray.zig:31-36), Sphere.localIntersect (shape.zig:23-45)
        V3 lo, ld;
        if (KIND == kSphereAxis) {
            // the off-diagonal entries are exact zeros: their products are zeros and adding
            // them changes no value, so they are not evaluated
            lo = {m[0] * o.x + m[3], m[5] * o.y + m[7], m[10] * o.z + m[11]};
            ld = {m[0] * d.x, m[5] * d.y, m[10] * d.z};
        } else {
            lo = xform_point(m, o);
            ld = xform_dir(m, d);
        }
        const double a = dot3(ld, ld);
        const double b = 2.0 * dot3(ld, lo);
        const double c = dot3(lo, lo) - 1.0;
        const double bb = b * b;
        const double disc = bb - 4.0 * a * c;
        if (!(disc < 0)) {
            if (STATS) ++ctr.sphere_hits;
            const double den = 2.0 * a;
            bool need = !(b > 0 && disc < bb * kOneMinus);
            if (have) {
                const double bd = s.t * den;
                const double L = (-b - bd) - 0x1p-40 * (fabs(b) + bd);
                need = need && !(L > 0 && L * L > disc * kOnePlus);
            }
            if (need) {
                const double sq = sqrt(disc);
                const double t1 = (-b - sq) / den;
                const double t2 = (-b + sq) / den;
                const bool in1 = t1 >= 0;
                const bool in2 = !in1 && t2 >= 0;  // t2 >= t1 never beats t1 under strict <
                cand = in1 ? t1 : t2;
                better = (in1 || in2) && (!have || cand < s.t);
                if (!SHADOW && (in1 || in2) && s.idx >= 0 && cand == s.t)
                    better = (int)rec.order < order_of[s.idx & ~kExitOpenBit];
                cand_idx = in2 ? (pos | kExitOpenBit) : pos;
                push = in2;
                push_t = t1;
            }
        }
        }
    } else if (KIND == kCube) {
        // extension beyond the reference: Ray.transform (ray.zig:31-36), then the three slabs
        const V3 lo = xform_point(m, o);
        const V3 ld = xform_dir(m, d);
        double x0, x1, y0, y1, z0, z1;
        cube_axis(lo.x, ld.x, x0, x1);
        cube_axis(lo.y, ld.y, y0, y1);
        cube_axis(lo.z, ld.z, z0, z1);
        const double t1 = max_cmp(max_cmp(x0, y0), z0);
        const double t2 = min_cmp(min_cmp(x1, y1), z1);
        if (!(t1 > t2)) {
            if (STATS) ++ctr.cube_hits;
            const bool in1 = t1 >= 0;
            const bool in2 = !in1 && t2 >= 0;  // t2 >= t1 never beats t1 under strict <
            cand = in1 ? t1 : t2;
            better = (in1 || in2) && (!have || cand < s.t);
            if (!SHADOW && (in1 || in2) && s.idx >= 0 && cand == s.t)
                better = (int)rec.order < order_of[s.idx & ~kExitOpenBit];
            cand_idx = in2 ? (pos | kExitOpenBit) : pos;
            push = in2;
            push_t = t1;
        }
    } else {
        // Plane.localIntersect (shape.zig:55-70): only row 1 of the local ray is used
        double loy, ldy;
        if (KIND == kPlaneAxis) {  // m[4] and m[6] are exact zeros
            loy = m[5] * o.y + m[7];
        } else {
            loy = ((m[4] * o.x + m[5] * o.y) + m[6] * o.z) + m[7];
        }
        // A shadow ray ends at the light.  When its origin and the light lie on the same side of
        // the plane, both farther from it than 2^-32 of the magnitudes involved, the line meets
        // the plane behind the origin or beyond the light: with heights h0 (origin) and h1 (light)
        // of equal sign, t = h0 * dist / (h0 - h1) exceeds dist by the factor 1 + h1/(h0 - h1) when
        // the ray descends, and is negative when it climbs; the reference's rounding moves t and
        // dist by parts in 2^50, the guard is parts in 2^32 (and a direction whose rounded local
        // y has the wrong sign is so flat that |t| is 2^16 times the distance; the surface offset
        // of 1e-4 that lifts a ray's origin off the plane it starts on stays well above the guard).  Such a plane
        // cannot make calc.isShadowed true and nothing more of it is evaluated.  The statistics
        // build counts the reference's intersections and evaluates every plane.
        if (SHADOW) {
            const double guard = rec.side_guard * side_scale;
            if (loy * rec.light_side > guard && rec.light_height > guard) {
                if (!STATS) return;
#ifndef RTC_ITEM_STATS
                ++ctr.plane_skips;
#endif
            }
        }
        if (KIND == kPlaneAxis) {
            ldy = m[5] * d.y;
        } else {
            ldy = (m[4] * d.x + m[5] * d.y) + m[6] * d.z;
        }
        const bool hits = !(fabs(ldy) < kF64Eps);
#ifndef RTC_ITEM_STATS
        if (STATS && hits) ++ctr.plane_hits;
#endif
        // A radiance sweep needs t of nearly every plane anyway (a plane behind the origin is a
        // container, and some lane of the wave usually has it behind), so only the shadow sweep
        // pays for deciding whether the division can be left out.
        bool need = hits;
        if (SHADOW) {
            // sign of t = -loy/ldy from the sign of the product (a zero or underflowed product
            // goes to the exact path); beyond the light when |loy| > dist*|ldy|*(1+2^-40)
            const double p = -loy * ldy;
            const double lim = (s.t * kOnePlus) * (ldy * ldy);
            need = hits && !(p < 0) && !(p > lim);
        }
        if (SHADOW) {
            if (need) {
                const double t = -loy / ldy;
                cand = t;
                better = t >= 0 && t < s.t;
            }
        } else {
            // no branch around the division: nearly every lane needs it (a quotient that the
            // reference does not form, |ldy| < epsilon, is computed and not used)
            const double t = -loy / ldy;
            const bool ahead = hits && t >= 0;
            cand = t;
            better = ahead && (s.idx < 0 || t < s.t);
            if (ahead && s.idx >= 0 && t == s.t) better = (int)rec.order < order_of[s.idx & ~kExitOpenBit];
            push = hits && t < 0;
            push_t = t;
        }
    }
    s.t = better ? cand : s.t;
    s.idx = better ? cand_idx : s.idx;
    if (!SHADOW && push) {
        // the container later in the sorted list (larger t; on equal t later in the list) is the inner one
        const int i1 = cs.i[0][tid], i2 = cs.i[1][tid];
        const double c1 = cs.t[0][tid], c2 = cs.t[1][tid];
        const int my_order = (int)rec.order;
        const bool after1 = i1 < 0 || push_t > c1 || (push_t == c1 && my_order > order_of[i1]);
        if (after1) {
            cs.i[1][tid] = i1;
            cs.t[1][tid] = c1;
            cs.i[0][tid] = pos;
            cs.t[0][tid] = push_t;
        } else if (i2 < 0 || push_t > c2 || (push_t == c2 && my_order > order_of[i2])) {
            cs.i[1][tid] = pos;
            cs.t[1][tid] = push_t;
        }
    }
}

// Sweep of one group, two shapes per loop trip.  Scalar loads return out of order, so the
// only wait is "all of them": each record is consumed first (that wait covers only its own
// loads), then the record after the next is requested and stays in flight during the
// arithmetic.  The two record buffers alternate, so no register copies are needed.
#ifdef RTC_LDS_SWEEP
#define RTC_RECORD_ARRIVED(x) asm volatile("" ::"v"(x) : "memory")
#else
#define RTC_RECORD_ARRIVED(x) asm volatile("" ::"s"(x) : "memory")
#endif
template <int KIND, bool SHADOW, bool STATS>
__device__ __forceinline__ void sweep_group(const double* __restrict__ tab, const int* __restrict__ order_of, int begin,
                                            int end, const V3& o, const V3& d, double dd, double side_scale, double shadow_margin, Sweep& s,
                                            ContainerSlots& cs, SweepCounters& ctr) {
    if (begin >= end) return;
    ShapeRec A = load_rec<KIND, SHADOW>(tab, begin);
    for (int i = begin; i < end; i += 2) {
        RTC_RECORD_ARRIVED(A.m[5]);  // (an entry every layout loads)
        const ShapeRec B = load_rec<KIND, SHADOW>(tab, i + 1);
        sweep_shape<KIND, SHADOW, STATS>(A, i, order_of, o, d, dd, side_scale, shadow_margin, s, cs, ctr);
        RTC_RECORD_ARRIVED(B.m[5]);
        A = load_rec<KIND, SHADOW>(tab, i + 2);
        if (i + 1 < end) sweep_shape<KIND, SHADOW, STATS>(B, i + 1, order_of, o, d, dd, side_scale, shadow_margin, s, cs, ctr);
    }
}

// The spheres of a large world (more shapes than the LDS tables hold).  Nearly every (ray, sphere) pair of
// such a world ends at the bounding test, which needs four numbers of the sphere (centre and radius term;
// for a shadow ray the light-anchored pair), not the 24 of its record.  Those four are kept in two dense
// side tables (32 bytes per sphere, table order, one for each kind of ray) that are streamed through the
// scalar cache four spheres per request, two requests ahead; the full record is fetched (a scalar load
// whose latency is then exposed) only for a sphere some lane of the wave cannot dismiss.  Per sphere the
// sweep moves 32 instead of 192 bytes from the L2 and issues a quarter of a load instruction instead of
// three; what is evaluated, and in which order, is unchanged.
//
// Ahead of that test runs the same test in packed single precision, two spheres per instruction: a third
// pair of side tables holds, per pair of spheres, the centres and radius terms as floats.  It only ever
// dismisses a pair: its margins cover every rounding of the float evaluation (derivation at
// sphere_prefilter_tables, rtc_api.hip), so "clear" there implies the inequality the double test evaluates,
// and a sphere it cannot dismiss goes on to the double test (operands fetched on demand) and from there as
// before.  In a world of thousands of spheres nearly every pair ends in the float test: 7 instead of 13
// vector instructions per radiance pair, 3.5 instead of 6 per shadow pair.
typedef float float2_t __attribute__((ext_vector_type(2)));
struct Pairs4 {
    float v[32];  // four pairs of spheres; per pair: cx0 cx1 cy0 cy1 cz0 cz1 r0 r1
};
__device__ __forceinline__ Pairs4 load_pairs4(const float* __restrict__ p) {
    Pairs4 r;
#pragma unroll
    for (int k = 0; k < 32; ++k) r.v[k] = p[k];
    return r;
}
// The ray in single precision, each component in both halves of a register pair.  Radiance rays: a = 2^-19.5
// |o|^2 rounded up (the ray's share of the margin), b = |d|^2 (1 - 2^-27)(1 - 2^-18) rounded down.  Shadow
// rays: a = 2^-25 dist^2 rounded up (the per-ray term of the light-anchored test).
struct Ray32 {
    float2_t ox, oy, oz, dx, dy, dz, a, b;
    // For the segment form of the test (pair_clear<.., true>): the end of the stretch of the ray that matters, in
    // units of the ray-centre product.  Radiance rays: best hit so far times |d|^2, rounded up (+inf: none yet); the
    // stretch is [0, lim].  Shadow rays: minus the distance to the light, rounded away from zero; the product is
    // taken from the light, and the stretch is [lim, 0].
    float lim;
};
__device__ __forceinline__ float float_up(float x) { return __uint_as_float(__float_as_uint(x) + 1u); }    // x > 0
__device__ __forceinline__ float float_down(float x) { return __uint_as_float(__float_as_uint(x) - 1u); }  // x > 0, normal
template <bool SHADOW>
__device__ __forceinline__ Ray32 ray_in_floats(const V3& o, const V3& d, double dd, double shadow_margin) {
    Ray32 r;
    r.lim = 0.0f;  // (set by the sweep that uses the segment form)
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z, dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    r.ox = {ox, ox}, r.oy = {oy, oy}, r.oz = {oz, oz}, r.dx = {dx, dx}, r.dy = {dy, dy}, r.dz = {dz, dz};
    if (SHADOW) {
        const float m = float_up(fmaxf((float)shadow_margin, 0x1p-100f));
        r.a = {m, m};
        r.b = {0.0f, 0.0f};
    } else {
        const double oo = fma(o.z, o.z, fma(o.y, o.y, o.x * o.x));
        const float rho = float_up(fmaxf((float)(0x1.6a09e667f3bcdp-20 * oo), 0x1p-100f));
#ifdef RTC_MUTATE_PREFILTER
        const float ds = (float)(dd * (1.0 + 0x1p-16));  // test of the tests: the packed test dismisses too eagerly
#else
        const float ds = (float)(dd * (1.0 - 0x1p-18));
#endif
        const float dsd = ds > 0x1p-100f ? float_down(ds) : 0.0f;  // 0: nothing is dismissed for this ray
        r.a = {rho, rho};
        r.b = {dsd, dsd};
    }
    return r;
}
// two spheres against the ray: c0 / c1 = the pair's first / second sphere is certainly clear of the ray.
//
// SEGMENT = false: clear of the ray's LINE, (|c - o|^2 - bound) |d|^2 > ((c - o).d)^2 with the margins derived at
// sphere_prefilter_tables (rtc_api.hip): the reference's discriminant is then negative.
//
// SEGMENT = true: clear of the stretch of the ray that can matter, t in [0, T] with T the best hit so far (a
// shadow ray: from its origin to the light).  With q = (c - o).d and tq = q clamped to [0, T |d|^2], the point of
// that stretch nearest the centre is at t = tq / |d|^2 and its squared distance from the centre is
// |c - o|^2 - tq (2 q - tq) / |d|^2: the same inequality with tq (2 q - tq) in place of q^2 (they are equal where
// q lies inside the stretch).  A sphere the stretch clears by the test's margin has both roots behind the origin or
// both beyond T (one on either side would put the whole stretch inside it), by a relative 2^-22 of the root's
// scale where the reference's roots are good to 2^-40: the full test would find roots that are neither the hit
// nor, the origin being outside, an open container; nor can they tie with the hit (a sphere that touches the
// stretch's end is not clear).  T only ever shrinks during a sweep, so what it lets go stays irrelevant.
// Roundings: the right side as a function of q has slope 2 tq and tq^2 <= tq (2 q - tq), so the split that covers
// the error of the computed q (see there) holds as it stands; 2 q - tq lies in [q, 2 q], no cancellation; one more
// rounded operation than q^2, still inside the factor 1 - 2^-18 ((1 - 2^-22.4)/(1 - 2^-18) > 1 + 2^-19); the limit
// is rounded outwards; a NaN q gives a NaN right side, which dismisses nothing.  The same holds for a block's sphere:
// its derivation (sphere_block_bound) uses distances from a set of points to the centres, whichever the set.
// The statistics build keeps the line form: it counts the pairs the reference finds roots for.
template <bool SHADOW, bool SEGMENT>
__device__ __forceinline__ void pair_clear(const float* __restrict__ p, const Ray32& r, bool& c0, bool& c1) {
    const float2_t cx = {p[0], p[1]}, cy = {p[2], p[3]}, cz = {p[4], p[5]}, k = {p[6], p[7]};
    const float2_t two = {2.0f, 2.0f};
    float2_t lhs, rhs;
    if (SHADOW) {
        const float2_t lcd = __builtin_elementwise_fma(cz, r.dz, __builtin_elementwise_fma(cy, r.dy, cx * r.dx));
        lhs = k - r.a;
        if (SEGMENT) {
            const float2_t tq = {__builtin_amdgcn_fmed3f(lcd.x, r.lim, 0.0f), __builtin_amdgcn_fmed3f(lcd.y, r.lim, 0.0f)};
            rhs = tq * __builtin_elementwise_fma(lcd, two, -tq);
        } else {
            rhs = lcd * lcd;
        }
    } else {
        const float2_t x = cx - r.ox, y = cy - r.oy, z = cz - r.oz;
        const float2_t oc2 = __builtin_elementwise_fma(z, z, __builtin_elementwise_fma(y, y, x * x));
        const float2_t ocd = __builtin_elementwise_fma(z, r.dz, __builtin_elementwise_fma(y, r.dy, x * r.dx));
        lhs = ((oc2 - r.a) - k) * r.b;
        if (SEGMENT) {
            const float2_t tq = {__builtin_amdgcn_fmed3f(ocd.x, 0.0f, r.lim), __builtin_amdgcn_fmed3f(ocd.y, 0.0f, r.lim)};
            rhs = tq * __builtin_elementwise_fma(ocd, two, -tq);
        } else {
            rhs = ocd * ocd;
        }
    }
    c0 = lhs.x > rhs.x;
    c1 = lhs.y > rhs.y;
}
// the limit of the segment form from a non-negative double, rounded up (+inf stays +inf)
__device__ __forceinline__ float segment_limit(double v) {
    const float f = (float)v;
    return f < __builtin_huge_valf() ? float_up(f) : __builtin_huge_valf();
}

template <bool SHADOW, bool STATS>
__device__ __forceinline__ void sweep_spheres_streamed(const float* __restrict__ pairs, const double* __restrict__ bounds,
                                                       const double* __restrict__ tab, const int* __restrict__ order_of,
                                                       int begin, int split, int end, const V3& o, const V3& d, double dd,
                                                       double shadow_margin, uint32_t block_pairs_offset, double block_shrink,
                                                       double block_margin_scale, Sweep& s, ContainerSlots& cs,
                                                       SweepCounters& ctr) {
    if (begin >= end) return;
    const Ray32 r32 = ray_in_floats<SHADOW>(o, d, dd, shadow_margin);
    // the same ray for the block tests: their shrink in the |d|^2 term, their larger per-ray margin for shadow rays
    Ray32 r32b = r32;
    {
        const Ray32 t = ray_in_floats<SHADOW>(o, d, dd * block_shrink, shadow_margin * block_margin_scale);
        r32b.a = t.a;
        r32b.b = t.b;
        if (SHADOW && !(block_shrink > 0.0)) r32b.a = {__builtin_huge_valf(), __builtin_huge_valf()};  // dismisses nothing
    }
    // what the float test could not dismiss: the double test on the sphere's four operands (fetched now), then,
    // for the lanes that fail it too, the full test on the sphere's record
    auto rest = [&](int i, bool clear32) {
        if (i >= end) return;  // (the tables are padded: the padding is not a shape)
        if (STATS && clear32) ++ctr.sphere_prefilter_skips;
        if (!clear32) {
            const double* __restrict__ b = bounds + (size_t)i * 4;
            const double bb[4] = {b[0], b[1], b[2], b[3]};
            const bool clear = SHADOW ? sphere_clear_of_shadow_ray(bb, bb[3], d, shadow_margin)
                                      : sphere_clear_of_ray(bb, bb[3], o, d, dd);
            if (STATS && clear) ++ctr.sphere_skips;
            if (!clear) {
                if (i < split)  // the spheres with a diagonal inverse come first
                    sweep_shape<kSphereAxis, SHADOW, STATS, true>(load_rec<kSphereAxis, SHADOW>(tab, i), i, order_of, o, d, dd,
                                                                  0.0, shadow_margin, s, cs, ctr);
                else
                    sweep_shape<kSphereGeneral, SHADOW, STATS, true>(load_rec<kSphereGeneral, SHADOW>(tab, i), i, order_of, o,
                                                                     d, dd, 0.0, shadow_margin, s, cs, ctr);
            }
        }
    };
    // eight spheres: four packed tests back to back, one branch for the eight (named flags: an array of
    // flags would be kept as bytes in a vector register and cost instructions to pack and unpack)
    auto eight = [&](const float* p, int i) {
        bool c0, c1, c2, c3, c4, c5, c6, c7;
        pair_clear<SHADOW, false>(p, r32, c0, c1);
        pair_clear<SHADOW, false>(p + 8, r32, c2, c3);
        pair_clear<SHADOW, false>(p + 16, r32, c4, c5);
        pair_clear<SHADOW, false>(p + 24, r32, c6, c7);
        if (STATS || !(c0 && c1 && c2 && c3 && c4 && c5 && c6 && c7)) {
            rest(i, c0);
            rest(i + 1, c1);
            rest(i + 2, c2);
            rest(i + 3, c3);
            rest(i + 4, c4);
            rest(i + 5, c5);
            rest(i + 6, c6);
            rest(i + 7, c7);
        }
    };
    // Blocks first: 64 spheres = eight blocks = four records of the block table per trip, one trip read ahead; the
    // spheres of a block are looked at (their own records fetched then) only by lanes that cannot dismiss the block.
    // (begin is the start of the table: blocks are counted from there.)
    static_assert(kSphereBlock == 8, "one call of eight() per block");
    const float* __restrict__ p = pairs + (size_t)begin * 4;
    const float* __restrict__ g = pairs + block_pairs_offset;
    Pairs4 G = load_pairs4(g);
    for (int i = begin; i < end; i += 8 * kSphereBlock, p += 32 * kSphereBlock, g += 32) {
        asm volatile("" ::"s"(G.v[31]) : "memory");
        const Pairs4 N = load_pairs4(g + 32);
        bool b0, b1, b2, b3, b4, b5, b6, b7;
        pair_clear<SHADOW, false>(G.v, r32b, b0, b1);
        pair_clear<SHADOW, false>(G.v + 8, r32b, b2, b3);
        pair_clear<SHADOW, false>(G.v + 16, r32b, b4, b5);
        pair_clear<SHADOW, false>(G.v + 24, r32b, b6, b7);
        // which blocks this lane cannot dismiss, as a bit mask; the loop below visits the blocks some lane
        // needs (one copy of the per-sphere code, not eight)
        const uint32_t need = (b0 ? 0u : 1u) | (b1 ? 0u : 2u) | (b2 ? 0u : 4u) | (b3 ? 0u : 8u) | (b4 ? 0u : 16u) |
                              (b5 ? 0u : 32u) | (b6 ? 0u : 64u) | (b7 ? 0u : 128u);
        uint32_t wave_needs = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) wave_needs |= __ballot((need >> q) & 1u) != 0 ? 1u << q : 0u;
        if (STATS) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int first = i + q * kSphereBlock;
                if (first < end && !((need >> q) & 1u))
                    ctr.sphere_prefilter_skips += (uint32_t)(end - first < kSphereBlock ? end - first : kSphereBlock);
            }
        }
#pragma unroll 1
        for (uint32_t left = wave_needs; left != 0; left &= left - 1u) {
            const int q = __ffs((int)left) - 1;
            const int first = i + q * kSphereBlock;
            if (first >= end) break;
            const Pairs4 S = load_pairs4(p + q * 4 * kSphereBlock);
            if ((need >> q) & 1u) eight(S.v, first);
        }
        G = N;
    }
}

// The same sweep with per-lane work lists.  After a few bounces the rays of a wave have little in common: nearly
// every block is needed by SOME lane, so testing a needed block's members wave-wide (above) runs the member tests
// for the union of what 64 rays need, most of the wave masked off.  Here only the coarsest tests stay wave-uniform:
// the table position order is a spatial one (rtc_scene_create), so eight blocks that follow each other - a
// "super-block" of 64 spheres - are a cluster too and have a bounding sphere of their own in a third table behind
// the blocks' (same format, same test, same derivation with 64 members).  The super-blocks are tested with scalar
// loads, one packed test per two; each lane notes the ones it cannot dismiss as bits of a 64-bit word (a window of
// 64 super-blocks = 4096 spheres).  At the end of a window every lane works off its own list, one item per trip
// of the drain loop: either a super-block (its eight blocks fetched from the block table with per-lane vector
// loads and tested; the survivors become the lane's current blocks) or a block (its eight spheres fetched from the
// pair table and tested; the survivors go on to the double test and then to the sphere's full test on rows 0..2 of
// its inverse - the per-hit inverse table holds them for every sphere; a diagonal inverse's products with its exact
// zeros add nothing).  Both kinds of item are the same loads and the same four packed tests on different
// addresses and margins, so lanes at different items run the same instructions.  The loop runs as long as the
// busiest lane's list, not as long as the union of all lists.  What is evaluated for a (ray, sphere) pair that
// reaches the full test is what sweep_shape evaluates; every dismissal before it implies the double test's (see
// sphere_block_bound and sphere_prefilter_tables, rtc_api.hip).  A shadow ray that has found an occluder needs
// nothing more (calc.zig:789 only asks whether one exists); the statistics build goes on, to count what the
// reference intersects.
constexpr int kSuperBlock = 8 * kSphereBlock;  // spheres per super-block
// Bit k of x moved to bit k XOR m (m < 8, x an 8-bit mask; m < 64, x a 64-bit mask): conditional swaps of neighbouring
// bits, pairs, nibbles ...  The spheres' table order makes `position XOR (where the ray's direction is negative)` a
// front-to-back order (rtc_scene_create), so the lowest set bit of the result is the nearest entry left.
__device__ __forceinline__ uint32_t in_visiting_order8(uint32_t x, uint32_t m) {
    uint32_t t = ((x & 0x55u) << 1) | ((x >> 1) & 0x55u);
    x = (m & 1u) ? t : x;
    t = ((x & 0x33u) << 2) | ((x >> 2) & 0x33u);
    x = (m & 2u) ? t : x;
    t = ((x & 0x0fu) << 4) | ((x >> 4) & 0x0fu);
    return (m & 4u) ? t : x;
}
__device__ __forceinline__ unsigned long long in_visiting_order64(unsigned long long x, uint32_t m) {
    unsigned long long t = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
    x = (m & 1u) ? t : x;
    t = ((x & 0x3333333333333333ull) << 2) | ((x >> 2) & 0x3333333333333333ull);
    x = (m & 2u) ? t : x;
    t = ((x & 0x0f0f0f0f0f0f0f0full) << 4) | ((x >> 4) & 0x0f0f0f0f0f0f0f0full);
    x = (m & 4u) ? t : x;
    t = ((x & 0x00ff00ff00ff00ffull) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffull);
    x = (m & 8u) ? t : x;
    t = ((x & 0x0000ffff0000ffffull) << 16) | ((x >> 16) & 0x0000ffff0000ffffull);
    x = (m & 16u) ? t : x;
    t = (x << 32) | (x >> 32);
    return (m & 32u) ? t : x;
}
#ifndef RTC_VOTE_NUM
#define RTC_VOTE_NUM 1  // the sphere tests get the trip when (lanes with a sphere waiting) x DEN >= (lanes with a block to do) x NUM
#define RTC_VOTE_DEN 2  // (measured: 1:4, 1:2, 1:1, 2:1 within 4 % of each other, 1:2 the fastest)
#endif
// (the statistics build keeps the line form of the packed tests, see pair_clear; the developer build RTC_ITEM_STATS
// counts the items of the work loop as the frame kernels see them, with the segment form, in two statistics slots)
#ifdef RTC_ITEM_STATS
template <bool STATS>
constexpr bool kSegmentForm = true;
#else
template <bool STATS>
constexpr bool kSegmentForm = !STATS;
#endif
template <bool SHADOW, bool STATS>
__device__ __forceinline__ void sweep_spheres_lanes(const float* __restrict__ pairs, const double* __restrict__ bounds,
                                                    const double* __restrict__ inv, const int* __restrict__ order_of,
                                                    int begin, int end, const V3& o, const V3& d, double dd,
                                                    double shadow_margin, uint32_t block_pairs_offset,
                                                    uint32_t super_pairs_offset, uint32_t group_pairs_offset, int top_log2,
                                                    uint32_t order_mask, bool all_diagonal, double block_shrink, double block_margin_scale, Sweep& s,
                                                    ContainerSlots& cs, SweepCounters& ctr) {
    if (begin >= end) return;
    Ray32 r32 = ray_in_floats<SHADOW>(o, d, dd, shadow_margin);
    // the stretch of the ray that matters (pair_clear): up to the best hit so far - on entry the nearest plane's,
    // brought up to date after every sphere test - or, for a shadow ray, from the light back to the origin; dd
    // carries the line test's shrink 1 - 2^-27, the factor makes the product an upper bound of T |d|^2
    auto reach = [&]() -> float {
#ifdef RTC_MUTATE_SEGMENT  // test of the tests: a stretch that ends early by more than the tests' own slack
        if (SHADOW) return -segment_limit(s.t * (1.0 - 0x1p-6));
        return s.idx >= 0 ? segment_limit(s.t * dd * (1.0 - 0x1p-6)) : __builtin_huge_valf();
#endif
        if (SHADOW) return -segment_limit(s.t);
        return s.idx >= 0 ? segment_limit(s.t * dd * (1.0 + 0x1p-20)) : __builtin_huge_valf();
    };
    r32.lim = reach();
    // the same ray for the block and super-block tests: their shrink in the |d|^2 term, their larger per-ray
    // margin for shadow rays (only the two margin terms differ)
    float2_t a_blocks, b_blocks;
    {
        const Ray32 t = ray_in_floats<SHADOW>(o, d, dd * block_shrink, shadow_margin * block_margin_scale);
        a_blocks = t.a;
        b_blocks = t.b;
        if (SHADOW && !(block_shrink > 0.0)) a_blocks = {__builtin_huge_valf(), __builtin_huge_valf()};  // dismisses nothing
    }
    static_assert(kSphereBlock == 8 && kSuperBlock == 64, "eight member bits per block, eight block bits per super-block; shifts by 3 and 6");
    // The level that is tested wave-wide (wave-uniform, chosen at upload): super-blocks (top_log2 == 6), or in worlds
    // of thousands of spheres GROUPS of eight super-blocks (512 spheres, top_log2 == 9; a fourth table of the same
    // kind), whose super-blocks are then one more kind of item of the lanes' work loop.
    const bool top_is_group = top_log2 == 9;
    const int top_level = top_is_group ? 2 : 1;  // what an entry of `pending` is, as a level of the work loop
    unsigned long long pending = 0ull;  // per lane: top-level entries of the current window this ray cannot dismiss
    uint32_t sub = 0u;                  // per lane: bits 8-15 the super-blocks of the group in hand this ray cannot dismiss,
                                        // bits 0-7 the blocks of the super-block in hand ...
    int supers_first = 0;               // ... the table position of that group's first sphere ...
    int blocks_first = 0;               // ... and of that super-block's first sphere
    uint32_t cands = 0u;                // per lane: spheres of the block in hand the packed test could not dismiss ...
    int cands_first = 0;                // ... and the table position of that block's first sphere
    // eight entries (four records of two) at `src` against the ray with margins a, b: bit k set = entry k not dismissed
    auto eight = [&](const float* __restrict__ src, const float2_t a, const float2_t b) -> uint32_t {
        const float4* __restrict__ q4 = reinterpret_cast<const float4*>(src);
        float v[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = q4[k];
            v[4 * k] = q.x, v[4 * k + 1] = q.y, v[4 * k + 2] = q.z, v[4 * k + 3] = q.w;
        }
        Ray32 r = r32;
        r.a = a;
        r.b = b;
        bool c0, c1, c2, c3, c4, c5, c6, c7;
        pair_clear<SHADOW, kSegmentForm<STATS>>(v, r, c0, c1);
        pair_clear<SHADOW, kSegmentForm<STATS>>(v + 8, r, c2, c3);
        pair_clear<SHADOW, kSegmentForm<STATS>>(v + 16, r, c4, c5);
        pair_clear<SHADOW, kSegmentForm<STATS>>(v + 24, r, c6, c7);
        return (c0 ? 0u : 1u) | (c1 ? 0u : 2u) | (c2 ? 0u : 4u) | (c3 ? 0u : 8u) | (c4 ? 0u : 16u) | (c5 ? 0u : 32u) |
               (c6 ? 0u : 64u) | (c7 ? 0u : 128u);
    };
    // (units are powers of two, given by their logarithm: a division by a per-lane value is a long instruction sequence;
    // first < end for every item)
    auto real_bits = [&](int first, int unit_log2) -> uint32_t {
        const uint32_t n = ((uint32_t)(end - first) + (1u << unit_log2) - 1u) >> unit_log2;
        return n >= 8u ? 0xffu : (1u << n) - 1u;
    };
    auto count_dismissed = [&](uint32_t dismissed, int first, int unit_log2) {  // statistics build only
        const int unit = 1 << unit_log2;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int at = first + q * unit;
            if (((dismissed >> q) & 1u) && at < end) ctr.sphere_prefilter_skips += (uint32_t)(end - at < unit ? end - at : unit);
        }
    };
    // The work loop.  A lane holds three kinds of work: spheres waiting for the double and the full test (the long,
    // double-precision part), blocks, and super-blocks.  A trip of the loop does ONE kind for the whole wave, chosen
    // by a vote, so that the expensive sphere tests run when many lanes have one to do instead of whenever a single
    // lane does: a lane with spheres waiting takes no further block until they are done.
    auto drain = [&](int window_first) {
        for (;;) {
            if (SHADOW && !STATS && s.idx >= 0) pending = 0ull, sub = 0u, cands = 0u;
            const bool has_sphere = cands != 0u;
            const bool has_eight = !has_sphere && (pending | sub) != 0ull;
            const unsigned long long with_sphere = __ballot(has_sphere), with_eight = __ballot(has_eight);
            if ((with_sphere | with_eight) == 0ull) break;
#ifdef RTC_NO_SPHERE_VOTE
            if (has_sphere) {  // (for comparison: every lane does its spheres as soon as it has them)
                while (cands != 0u) {
#else
            if (__popcll(with_sphere) * RTC_VOTE_DEN >= __popcll(with_eight) * RTC_VOTE_NUM) {
                if (has_sphere) {
#endif
                    const int i = cands_first + (__ffs((int)cands) - 1);
                    cands &= cands - 1u;
#ifdef RTC_ITEM_STATS
                    if (STATS) {
                        ++ctr.sphere_prefilter_skips;  // lanes in a trip of sphere tests ...
                        if (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u)
                            ++ctr.plane_skips;  // ... and such trips
                    }
#endif
                    // The double-precision bounding test dismisses one in sixty of the spheres that get here (the
                    // packed test is nearly as sharp) and costs a round trip to memory ahead of the record's: only
                    // the statistics build runs it, to count as the other sweeps do.  The full test finds the same
                    // spheres without roots.
                    bool clear = false;
#ifndef RTC_ITEM_STATS
                    if (STATS) {
                        const double* __restrict__ b = bounds + (size_t)(uint32_t)i * 4;
                        const double bb[4] = {b[0], b[1], b[2], b[3]};
                        clear = SHADOW ? sphere_clear_of_shadow_ray(bb, bb[3], d, shadow_margin) : sphere_clear_of_ray(bb, bb[3], o, d, dd);
                        if (clear) ++ctr.sphere_skips;
                    }
#endif
                    if (!clear) {
                        ShapeRec rec;
                        const double* __restrict__ m = inv + (size_t)(uint32_t)i * 12;
                        rec.order = order_of[i];
                        if (all_diagonal) {
                            // (wave-uniform: no sphere of the world has an off-diagonal entry; six loads instead of twelve)
                            rec.m[0] = m[0], rec.m[3] = m[3], rec.m[5] = m[5], rec.m[7] = m[7], rec.m[10] = m[10], rec.m[11] = m[11];
                            sweep_shape<kSphereAxis, SHADOW, STATS, true>(rec, i, order_of, o, d, dd, 0.0, shadow_margin, s, cs, ctr);
                        } else {
#pragma unroll
                            for (int k = 0; k < 12; ++k) rec.m[k] = m[k];
                            sweep_shape<kSphereGeneral, SHADOW, STATS, true>(rec, i, order_of, o, d, dd, 0.0, shadow_margin, s, cs, ctr);
                        }
                        if (!SHADOW) r32.lim = reach();
                    }
                }
            }
#ifdef RTC_NO_SPHERE_VOTE
            if (has_eight) {
#else
            else if (has_eight) {
#endif
                // The deepest kind of item the lane holds: a block (level 0: its entries are spheres), a super-block
                // (level 1: blocks) or a group (level 2: super-blocks).  `sub` keeps the blocks in hand in bits 0-7 and
                // the super-blocks in hand in bits 8-15, so its lowest set bit is the deepest item.  Everything below
                // is selects and shifts on that: one copy of the loads and tests serves all levels.
                int level, first;  // `first`: table position of the first sphere of the item
                if (sub != 0u) {
                    const int bit = __ffs((int)sub) - 1;
                    sub &= sub - 1u;
                    level = bit >> 3;
                    // (the masks are kept in visiting order: entry = place XOR the ray's order mask at that level)
                    const int entry = (bit & 7) ^ (int)((order_mask >> (3 + 3 * level)) & 7u);
                    first = (level ? supers_first : blocks_first) + (entry << (3 + 3 * level));
                } else {
                    level = top_level;
                    const int entry = (__ffsll((long long)pending) - 1) ^ (int)((order_mask >> top_log2) & 63u);
                    first = window_first + (entry << top_log2);
                    pending &= pending - 1ull;
                }
                // the item's eight entries: 4 floats each, from the table of the level below (all tables are parts of `pairs`)
                const int unit_log2 = 3 * level;
                const uint32_t table = level == 0 ? 0u : (level == 1 ? block_pairs_offset : super_pairs_offset);
                const float* __restrict__ src = pairs + ((size_t)table + (size_t)((uint32_t)first >> unit_log2) * 4);
                const uint32_t real = real_bits(first, unit_log2);
                const uint32_t open = eight(src, level == 0 ? r32.a : a_blocks, level == 0 ? r32.b : b_blocks) & real;
#ifndef RTC_ITEM_STATS
                if (STATS) count_dismissed(real & ~open, first, unit_log2);
#endif
#ifdef RTC_ITEM_STATS
                if (STATS) {
                    ++ctr.cube_hits;                   // items of the work loop
                    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u)
                        ++ctr.plane_hits;              // trips of the loop that work on items (one count per wave)
                    if (open == 0u) ++ctr.sphere_hits;  // ... none of whose eight entries survived
                    // would the item's OWN bound, tested again with the stretch as it is now, have dismissed it?
                    const uint32_t own_table = level == 0 ? block_pairs_offset : (level == 1 ? super_pairs_offset : group_pairs_offset);
                    const uint32_t own = (uint32_t)first >> (3 * (level + 1));
                    const float* __restrict__ e = pairs + ((size_t)own_table + (size_t)(own >> 1) * 8 + (own & 1u));
                    const float one[8] = {e[0], e[0], e[2], e[2], e[4], e[4], e[6], e[6]};
                    Ray32 rr = r32;
                    rr.a = a_blocks;
                    rr.b = b_blocks;
                    bool d0, d1;
                    pair_clear<SHADOW, true>(one, rr, d0, d1);
                    if (d0) ++ctr.sphere_skips;
                }
#endif
                cands = level == 0 ? open : 0u;
                cands_first = level == 0 ? first : cands_first;
                sub |= level == 0 ? 0u : in_visiting_order8(open, (order_mask >> (3 * level)) & 7u) << (8 * (level - 1));
                blocks_first = level == 1 ? first : blocks_first;
                supers_first = level == 2 ? first : supers_first;
            }
        }
    };
    Ray32 r32s = r32;  // (super-blocks are tested before any sphere of the window: the limit is the one on entry, or of the last drain)
    r32s.a = a_blocks;
    r32s.b = b_blocks;
    const float* __restrict__ g = pairs + (top_is_group ? group_pairs_offset : super_pairs_offset);
    Pairs4 G = load_pairs4(g);
    int trip = 0;
    const int per_trip = 8 << top_log2;  // spheres covered by the eight top-level entries of a trip
    for (int i = begin; i < end; i += per_trip, g += 32, ++trip) {
        asm volatile("" ::"s"(G.v[31]) : "memory");
        const Pairs4 N = load_pairs4(g + 32);
        bool b0, b1, b2, b3, b4, b5, b6, b7;
        pair_clear<SHADOW, kSegmentForm<STATS>>(G.v, r32s, b0, b1);
        pair_clear<SHADOW, kSegmentForm<STATS>>(G.v + 8, r32s, b2, b3);
        pair_clear<SHADOW, kSegmentForm<STATS>>(G.v + 16, r32s, b4, b5);
        pair_clear<SHADOW, kSegmentForm<STATS>>(G.v + 24, r32s, b6, b7);
        const uint32_t real = real_bits(i, top_log2);  // wave-uniform
        const uint32_t need = ((b0 ? 0u : 1u) | (b1 ? 0u : 2u) | (b2 ? 0u : 4u) | (b3 ? 0u : 8u) | (b4 ? 0u : 16u) |
                               (b5 ? 0u : 32u) | (b6 ? 0u : 64u) | (b7 ? 0u : 128u)) & real;
#ifndef RTC_ITEM_STATS
        if (STATS) count_dismissed(real & ~need, i, top_log2);
#endif
        const int slot = trip & 7;
        pending |= (unsigned long long)need << (slot * 8);
        if (slot == 7 || i + per_trip >= end) {
            if (!SHADOW) pending = in_visiting_order64(pending, (order_mask >> top_log2) & 63u);
            drain(i - slot * per_trip);
            if (SHADOW && !STATS && __ballot(s.idx < 0) == 0ull) break;  // every ray of the wave has its occluder
            r32s.lim = r32.lim;
        }
        G = N;
    }
}

// The spheres through the optional hierarchy (rtc_scene_set_acceleration) instead of the two
// sphere loops.  Each lane walks the tree with its own ray: a node carries the (enlarged) boxes
// of both children; a child is entered when the slab test cannot rule out that the ray crosses
// its box between the origin and the best hit so far (the light, for a shadow ray), nearer
// child first.  What is left out are pairs the full test would compute and not use: no roots,
// both roots behind the origin, or the nearer root beyond the hit (a sphere around the origin
// has the origin in its box and is always reached, so the containers are complete).  The
// order of the visits does not matter: sweep_shape resolves equal t by list index.
constexpr int kBvhStack = 64;  // the builder keeps the depth below 40 + ceil(log2(2^21 spheres)) = 61
template <bool SHADOW, bool STATS>
__device__ __forceinline__ void sweep_spheres_bvh(const double* __restrict__ tab, const int* __restrict__ order_of,
                                                  const DevWorld& w, const V3& o, const V3& d, double dd,
                                                  double shadow_margin, Sweep& s, ContainerSlots& cs,
                                                  SweepCounters& ctr) {
    const double inf = __builtin_huge_val();
    const double lim = w.bvh_origin_limit;
    // rays that start far outside the scene, or whose direction is degenerate, are not pruned
    const bool prune = fabs(o.x) <= lim && fabs(o.y) <= lim && fabs(o.z) <= lim && dd > 0.0 && dd < inf;
    const V3 inv = {1.0 / d.x, 1.0 / d.y, 1.0 / d.z};
    int stack[kBvhStack];
    double stack_near[kBvhStack];  // where the ray enters the deferred child's box
    int sp = 0;
    int cur = w.bvh_root;
    for (;;) {
        if (cur < 0) {
            const int pos = ~cur;
            // the table keeps the spheres with a diagonal inverse first: they need six of the twelve entries
            if (pos < w.group[1])
                sweep_shape<kSphereAxis, SHADOW, STATS>(load_rec<kSphereAxis, SHADOW>(tab, pos), pos, order_of, o, d, dd, 0.0,
                                                        shadow_margin, s, cs, ctr);
            else
                sweep_shape<kSphereGeneral, SHADOW, STATS>(load_rec<kSphereGeneral, SHADOW>(tab, pos), pos, order_of, o, d, dd, 0.0,
                                                           shadow_margin, s, cs, ctr);
            if (SHADOW && s.idx >= 0) break;  // calc.isShadowed only asks whether anything is in the way
        } else {
            const BvhNode* __restrict__ n = w.bvh + cur;
            const double reach = (SHADOW || s.idx >= 0) ? s.t * kOnePlus : inf;
            bool enter[2];
            double tnear[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // fmin/fmax drop a NaN operand (0 * inf when the origin lies on a slab of an axis the
                // ray is parallel to): that axis then constrains nothing
                const double x1 = (n->lo[k][0] - o.x) * inv.x, x2 = (n->hi[k][0] - o.x) * inv.x;
                const double y1 = (n->lo[k][1] - o.y) * inv.y, y2 = (n->hi[k][1] - o.y) * inv.y;
                const double z1 = (n->lo[k][2] - o.z) * inv.z, z2 = (n->hi[k][2] - o.z) * inv.z;
                const double t0 = fmax(fmax(fmin(x1, x2), fmin(y1, y2)), fmin(z1, z2));
                const double t1 = fmin(fmin(fmax(x1, x2), fmax(y1, y2)), fmax(z1, z2));
                const bool missed = t0 > t1 || t1 < 0.0 || t0 > reach;
                enter[k] = !(prune && missed);
                tnear[k] = t0;
            }
            const int c0 = n->child[0], c1 = n->child[1];
            if (enter[0] && enter[1]) {
                const bool first0 = !(tnear[1] < tnear[0]);
                stack[sp] = first0 ? c1 : c0;
                stack_near[sp] = first0 ? tnear[1] : tnear[0];
                ++sp;
                cur = first0 ? c0 : c1;
                continue;
            }
            if (enter[0] || enter[1]) {
                cur = enter[0] ? c0 : c1;
                continue;
            }
        }
        // next deferred child whose box the ray still reaches before the best hit found meanwhile
        const double reach = (SHADOW || s.idx >= 0) ? s.t * kOnePlus : inf;
        bool found = false;
        while (sp > 0) {
            --sp;
            if (!(prune && stack_near[sp] > reach)) {
                cur = stack[sp];
                found = true;
                break;
            }
        }
        if (!found) break;
    }
}

// SHADOW: s.t holds the distance to the light on entry and only roots in [0, s.t) are
// accepted (calc.zig:789: hit.t < distance); containers are not tracked.
template <bool SHADOW, bool STATS, bool ACCEL, bool STREAM, bool SLIM = false>
__device__ __forceinline__ void sweep_world(const double* __restrict__ tab, const int* __restrict__ order_of,
                                            const double* __restrict__ sphere_bounds,
                                            const float* __restrict__ sphere_pairs, const DevWorld& w, const V3& o,
                                            const V3& d, Sweep& s, ContainerSlots& cs, SweepCounters& ctr) {
    const int tid = (int)threadIdx.x;
    // for the bounding-sphere tests: |d|^2, scaled by the shrink factor of the test (see sphere_clear_of_ray)
    const double dd = fma(d.z, d.z, fma(d.y, d.y, d.x * d.x)) * kBoundShrink;
    // for the same-side test of the planes: 1 + |o|_1 + |light|_1 bounds every magnitude in it
    const double side_scale = SHADOW ? ((fabs(o.x) + fabs(o.y)) + fabs(o.z)) + w.light_reach : 0.0;
    // the per-ray term of the spheres' test from the light: 2^-25 dist^2 (s.t is the distance on entry)
    const double shadow_margin = SHADOW ? 0x1p-25 * s.t * s.t : 0.0;
    s.idx = -1;
    if (!SHADOW) {
        s.t = 0.0;
        cs.i[0][tid] = -1;
        cs.i[1][tid] = -1;
    }
    sweep_group<kPlaneAxis, SHADOW, STATS>(tab, order_of, w.group[2], w.group[3], o, d, dd, side_scale, shadow_margin, s, cs, ctr);
    sweep_group<kPlaneGeneral, SHADOW, STATS>(tab, order_of, w.group[3], w.group[4], o, d, dd, side_scale, shadow_margin, s, cs, ctr);
    // (SLIM: a world without cubes and without spheres whose inverse has off-diagonal entries; their loops are left out)
    if (!SLIM) sweep_group<kCube, SHADOW, STATS>(tab, order_of, w.group[4], w.group[5], o, d, dd, side_scale, shadow_margin, s, cs, ctr);
    if (ACCEL) {
        if (!(SHADOW && s.idx >= 0)) sweep_spheres_bvh<SHADOW, STATS>(tab, order_of, w, o, d, dd, shadow_margin, s, cs, ctr);
    } else if (STREAM) {
#ifndef RTC_NO_LANE_LISTS
        sweep_spheres_lanes<SHADOW, STATS>(sphere_pairs, sphere_bounds, w.inv, order_of, w.group[0], w.group[2], o, d, dd, shadow_margin,
                                           SHADOW ? w.block_pairs_offset_shadow : w.block_pairs_offset_radiance,
                                           SHADOW ? w.super_pairs_offset_shadow : w.super_pairs_offset_radiance,
                                           SHADOW ? w.group_pairs_offset_shadow : w.group_pairs_offset_radiance, w.sweep_top_log2,
                                           SHADOW ? 0u
                                                  : (d.x < 0 ? w.sweep_axis_mask[0] : 0u) | (d.y < 0 ? w.sweep_axis_mask[1] : 0u) |
                                                        (d.z < 0 ? w.sweep_axis_mask[2] : 0u),
                                           w.group[1] == w.group[2], w.block_shrink, w.block_margin_scale, s, cs, ctr);
#else
        sweep_spheres_streamed<SHADOW, STATS>(sphere_pairs, sphere_bounds, tab, order_of, w.group[0], w.group[1], w.group[2], o, d, dd,
                                              shadow_margin, SHADOW ? w.block_pairs_offset_shadow : w.block_pairs_offset_radiance,
                                              w.block_shrink, w.block_margin_scale, s, cs, ctr);
#endif
    } else {
        sweep_group<kSphereAxis, SHADOW, STATS>(tab, order_of, w.group[0], w.group[1], o, d, dd, side_scale, shadow_margin, s, cs, ctr);
        if (!SLIM) sweep_group<kSphereGeneral, SHADOW, STATS>(tab, order_of, w.group[1], w.group[2], o, d, dd, side_scale, shadow_margin, s, cs, ctr);
    }
    if (!SHADOW) {
        s.top1 = cs.i[0][tid];
        s.top2 = cs.i[1][tid];
    }
}

// Pattern.patternAt (pattern.zig:77-91) and the four pattern functions (pattern.zig:15-62)
__device__ __forceinline__ V3 pattern_color(const DevShade& sh, const double* __restrict__ obj_inv, const V3& wp) {
    const V3 op = xform_point(obj_inv, wp);
    const V3 pp = xform_point(sh.pat_inv, op);
    const V3 a = {sh.pat_a[0], sh.pat_a[1], sh.pat_a[2]};
    const V3 b = {sh.pat_b[0], sh.pat_b[1], sh.pat_b[2]};
    double c;
    switch (sh.pattern) {
        case RTC_PATTERN_POINT:
            return pp;
        case RTC_PATTERN_GRADIENT: {
            const double fraction = pp.x - floor(pp.x);
            // a.add(b.sub(a).scale(fraction))
            return {a.x + fraction * (b.x - a.x), a.y + fraction * (b.y - a.y), a.z + fraction * (b.z - a.z)};
        }
        case RTC_PATTERN_STRIPE:
            c = floor(pp.x);
            break;
        case RTC_PATTERN_RING:
            c = floor(sqrt(pp.x * pp.x + pp.z * pp.z));
            break;
        default:  // checkers
            c = (floor(pp.x) + floor(pp.y)) + floor(pp.z);
            break;
    }
    // @mod(c, 2) == 0 on an integral value: even/odd test
    return fmod(c, 2.0) == 0 ? a : b;
}

// Frame of a node whose children are still being traced.
//   acc      surface colour, then surface + reflected term (calc.zig:466,468)
//   meta     flags, the lane tracing an outsourced subtree, and the table position of the hit
//            shape (its material gives the factor of the reflected child, calc.zig:958)
//   R        Schlick reflectance when flags & kSchlick (calc.zig:464-466)
//   ray      the refracted ray, parked while the reflected subtree is traced
// flags: kHasRefract — a refracted child exists; kWaitRefract — the reflected term is already
// in acc and the frame waits for the refracted colour; kOutsourced — another lane of the wave
// traces the refracted subtree (its lane id sits in the partner bits).
// acc and meta are what every frame uses; for the first kLdsFrames levels they live in LDS.
// In private memory the frame stacks of all resident lanes together are several times the
// L2, and every push and pop would travel to HBM and back.
enum : int { kHasRefract = 1, kSchlick = 2, kWaitRefract = 4, kOutsourced = 8 };
constexpr int kMetaPartnerShift = 4, kMetaShapeShift = 10;  // meta = flags | lane << 4 | shape << 10
#ifndef RTC_LDS_FRAMES
#define RTC_LDS_FRAMES 3
#endif
constexpr int kLdsFrames = RTC_LDS_FRAMES;
struct FrameSlots {
    double acc[kLdsFrames][3][kBlock];
    int meta[kLdsFrames][kBlock];
};
enum : int { kIdle = 0, kRun = 1, kHelperDone = 2 };
enum : int { kAuxOwnerShift = 6, kAuxRemShift = 12, kAuxHelper = 1 << 18, kAuxWaiting = 1 << 19 };
constexpr int kChunk = 64;  // queue positions a wave deals out at a time: one 8x8 block of pixels

// Number of set bits of a wave-wide mask below the calling lane (v_mbcnt).
__device__ __forceinline__ uint32_t set_bits_below_lane(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Lane id of the n-th (0-based) set bit of mask; n < popcount(mask).
__device__ __forceinline__ int nth_set_bit(unsigned long long mask, int n) {
    int pos = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const int below = __popcll(mask & (((1ull << step) - 1ull) << pos));
        const bool up = below <= n;
        n -= up ? below : 0;
        pos += up ? step : 0;
    }
    return pos;
}

// Worlds of at most kLdsShapes shapes keep what a hit gathers per lane (the shape's inverse
// and its material/pattern record) in LDS, copied once per workgroup: a gather then costs an
// LDS access instead of a chain of dependent trips to the L2, which four waves per SIMD
// cannot hide.  Larger worlds spend their time in the sweeps and gather from global memory.
constexpr int kLdsShapes = kLdsShapeLimit;
template <bool ON>
struct HotTables {};
template <>
struct HotTables<true> {
    double mat[kLdsShapes][12];
    DevShade shade[kLdsShapes];
#ifdef RTC_LDS_SWEEP
    // developer switch, for comparison (build with -DRTC_LDS_SHAPES=16 so that the workgroup's LDS still allows four per
    // CU): the sweep table in LDS too, read with wave-uniform LDS addresses - a broadcast into VECTOR registers - instead
    // of scalar loads into scalar registers
    double tab[(kLdsShapes + 2) * kRecDoubles];
#endif
};
static_assert(sizeof(DevShade) % 8 == 0, "DevShade is copied as 8-byte words");

// The by-value kernel arguments hold some sixty scalars (camera, light) that the loop needs at a
// few places only.  Kept in registers over the whole loop they do not fit the scalar file, and
// the allocator parks them in lanes of vector registers: every use then costs a vector
// instruction (v_readlane) per 32 bits.  Reading them from the kernel-argument segment where
// they are used costs a few scalar loads instead (the segment stays in the scalar cache); the
// empty asm makes the address opaque, so that the loads are not hoisted out of the loop again.
// Layout of the segment: the thirteen pointers, then DevWorld, then DevJob (checked against the
// code object's metadata by the build's assembly listing).
constexpr int kWorldArgOffset = 13 * 8;
constexpr int kJobArgOffset = kWorldArgOffset + (int)sizeof(DevWorld);
static_assert(sizeof(DevWorld) % 8 == 0 && alignof(DevJob) == 8 && alignof(DevWorld) == 8, "kernel-argument layout");
typedef const __attribute__((address_space(4))) DevJob* JobArg;
typedef const __attribute__((address_space(4))) DevWorld* WorldArg;
__device__ __forceinline__ JobArg job_argument() {
    const __attribute__((address_space(4))) char* base =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    JobArg j = (JobArg)(base + kJobArgOffset);
    asm volatile("" : "+s"(j));
    return j;
}
__device__ __forceinline__ WorldArg world_argument() {
    const __attribute__((address_space(4))) char* base =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    WorldArg w = (WorldArg)(base + kWorldArgOffset);
    asm volatile("" : "+s"(w));
    return w;
}

// Block stores (rtc_scene_set_canvas_store: RTC_CANVAS_STORE_BLOCKS).  A lane that finishes a pixel normally stores its
// 24 bytes at once, alone or with the few lanes that finish in the same iteration: fine for local memory, whose L2
// merges them, but a poor transaction size when the canvas is another GPU's memory.  Every pixel of an 8x8 block is
// traced by lanes of the ONE wave that drew the block, so that wave can tell when the block is complete without any
// synchronisation: it parks finished colours in a staging area of its own (global memory; kFlushSlots blocks in
// flight per wave, a block that finds no free slot falls back to per-pixel stores) and, when the last pixel of a
// block is done, copies the block out with all 64 lanes in one go: eight runs of 192 contiguous bytes.
constexpr uint32_t kOutStaged = 0x80000000u;  // out_pixel holds (slot << 6 | position in the block), not a canvas index
template <bool ON>
struct FlushTable {};
template <>
struct FlushTable<true> {
    int left[kBlock / 64][kFlushSlots];
    uint32_t out0[kBlock / 64][kFlushSlots], nx[kBlock / 64][kFlushSlots], ny[kBlock / 64][kFlushSlots];
};

// The read-only tables and the output are separate __restrict__ kernel arguments so that
// the compiler can prove the wave-uniform table reads are never clobbered by the kernel's
// own stores and keep them on the scalar path; the structs carry the remaining scalars.
// TABLES: 3 = as 1, for worlds without cubes and general spheres (launch_variant); 0 = per-hit tables gathered from global memory, 1 = from their LDS copy (small
// worlds), 2 = global memory and the spheres swept through the hierarchy.
// FLUSH: the finished pixels of an 8x8 block are stored together when the block's last pixel is done (see kFlushSlots).
template <int MAXD, bool STATS, bool FROM_RAYS, int TABLES, bool FLUSH = false>
__global__ __launch_bounds__(kBlock, TABLES == 0 ? RTC_WAVES_PER_EU_LARGE : RTC_WAVES_PER_EU) void render_kernel(const double* __restrict__ g_tab,
                                                        const DevShade* __restrict__ g_shade,
                                                        const int* __restrict__ g_order,
                                                        const double* __restrict__ g_rays,
                                                        const uint32_t* __restrict__ g_tile_order,
                                                        double* __restrict__ g_out, uint32_t* __restrict__ g_counter,
                                                        unsigned long long* __restrict__ g_stats,
                                                        const double* __restrict__ g_bounds_r,
                                                        const double* __restrict__ g_bounds_s,
                                                        const double* __restrict__ g_inv,
                                                        const float* __restrict__ g_pairs_r,
                                                        const float* __restrict__ g_pairs_s, const DevWorld wc,
                                                        const DevJob job) {
    constexpr bool LDS = TABLES == 1 || TABLES == 3;
    constexpr bool SLIM = TABLES == 3;  // (see launch_variant)
    constexpr bool ACCEL = TABLES == 2;
    constexpr bool STREAM = TABLES == 0;  // large world, list walk: the spheres' bounds are streamed from their side tables
    constexpr int kDrawThreshold = STREAM ? 0 : 1;  // whether job.draw_threshold applies (rtc_refill.inc)
    __shared__ ContainerSlots containers;
    __shared__ HotTables<LDS> hot;
    __shared__ FrameSlots frames;
    const int tid = (int)threadIdx.x;
    if constexpr (LDS) {
        const int n = wc.n;
        for (int k = (int)threadIdx.x; k < n * 12; k += kBlock) hot.mat[k / 12][k % 12] = g_inv[k];
        const double* __restrict__ src = reinterpret_cast<const double*>(g_shade);
        double* dst = reinterpret_cast<double*>(hot.shade);
        for (int k = (int)threadIdx.x; k < n * (int)(sizeof(DevShade) / 8); k += kBlock) dst[k] = src[k];
        __syncthreads();
    }
    auto shade_of = [&](int i) -> const DevShade& {
        if constexpr (LDS)
            return hot.shade[i];
        else
            return g_shade[i];
    };
    auto inverse_of = [&](int i) -> const double* {
        if constexpr (LDS)
            return hot.mat[i];
        else
            return g_inv + (size_t)i * 12;
    };
#ifdef RTC_LDS_SWEEP
    const double* sweep_tab = g_tab;
    if constexpr (LDS) {
        for (int k = (int)threadIdx.x; k < (wc.n + 2) * kRecDoubles; k += kBlock) hot.tab[k] = g_tab[k];
        __syncthreads();
        sweep_tab = hot.tab;
    }
#else
    const double* __restrict__ sweep_tab = g_tab;
#endif
    const int lane = (int)(threadIdx.x & 63);
    DevWorld w = wc;
    w.tab = g_tab;
    w.shade = g_shade;

    // per-lane frame stack: levels below kLdsFrames keep acc/meta in LDS, the rest is private memory
    constexpr int kDeep = MAXD > kLdsFrames ? MAXD - kLdsFrames : 1;
    double f_acc[kDeep][3];
    int f_meta[kDeep];
    double f_ray[MAXD][6];
    double f_R[MAXD];
#pragma unroll
    for (int f = 0; f < MAXD; ++f) f_R[f] = 0.0;  // read (and not used) for frames that never set it
    auto frame_meta = [&](int f) -> int { return f < kLdsFrames ? frames.meta[f][tid] : f_meta[f - kLdsFrames]; };
    auto set_frame_meta = [&](int f, int v) {
        if (f < kLdsFrames)
            frames.meta[f][tid] = v;
        else
            f_meta[f - kLdsFrames] = v;
    };
    auto frame_acc = [&](int f) -> V3 {
        if (f < kLdsFrames) return V3{frames.acc[f][0][tid], frames.acc[f][1][tid], frames.acc[f][2][tid]};
        return V3{f_acc[f - kLdsFrames][0], f_acc[f - kLdsFrames][1], f_acc[f - kLdsFrames][2]};
    };
    auto set_frame_acc = [&](int f, const V3& v) {
        if (f < kLdsFrames) {
            frames.acc[f][0][tid] = v.x;
            frames.acc[f][1][tid] = v.y;
            frames.acc[f][2][tid] = v.z;
        } else {
            f_acc[f - kLdsFrames][0] = v.x;
            f_acc[f - kLdsFrames][1] = v.y;
            f_acc[f - kLdsFrames][2] = v.z;
        }
    };

    uint32_t n_radiance = 0, n_shadow = 0, n_shaded = 0;
    SweepCounters ctr = {};
#ifdef RTC_COSTMAP
    int cost_nodes = 0;
#endif

    // Task state of the lane.  A task is a subtree of the reference's recursion: either a
    // whole pixel (root) or, once the pixel queue is empty, a refracted subtree taken over
    // from a busy lane of the same wave.
    int mode = kIdle;
    // The rest of the task state in one word (registers are the scarce resource of this kernel):
    //   bits 0-5   wait lane: the lane tracing the subtree this lane's top frame waits for (kAuxWaiting)
    //   bits 6-11  owner: for a helper, the lane whose frame consumes this task's colour
    //   bits 12-17 `remaining` (calc.zig:641) of the task's root ray
    //   kAuxHelper  the task is a taken-over subtree: its colour goes to the owner, not to the canvas
    //   kAuxWaiting the top frame waits for a subtree another lane is tracing
    int aux = 0;
    uint32_t out_pixel = 0;    // root: the pixel's slot in the output (three doubles each)
    int sp = 0;
    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    bool exhausted = false;    // wave-uniform: the global pixel queue is empty
    uint32_t loc_next = 0, loc_end = 0;  // wave-uniform: the chunk of queue positions the wave is dealing out
    // One queue head would be an atomic on one address for the whole GPU: about 85 M of them per
    // second go through, which a frame of cheap pixels exceeds.  The blocks are dealt round-robin
    // to kQueues heads on separate cache lines; a wave starts at the queue its index selects.
    uint32_t cur_queue = ((blockIdx.x * (kBlock / 64)) + (threadIdx.x >> 6)) % kQueues;

    // wave-uniform: the 8x8 block the current draw covers
    uint32_t blk_base = 0, blk_px0 = 0, blk_py0 = 0;
    bool blk_valid = false;
    uint32_t blk_out0 = 0;
    // FLUSH: per wave and staging slot, in LDS (touched when a block is drawn or completed, so not worth registers):
    // the pixels of the slot's block still in flight (0 = free), the canvas index of the block's first pixel and the
    // block's extent inside the canvas; in a register the slot of the block being dealt (-1 = none), wave-uniform
    __shared__ FlushTable<FLUSH> fl;
    int blk_slot = -1;
    if constexpr (FLUSH) {
        if (lane < kFlushSlots) fl.left[threadIdx.x >> 6][lane] = 0;  // (read by the same wave only)
    }
    // (job.cam_origin is inverse(camera.transform) * point(0,0,0) (camera.zig:60): the same value
    // for every pixel, evaluated once on the host with the same expression)

#ifdef RTC_PROBE
    const unsigned long long probe_t0 = wall_clock64();
    unsigned long long probe_tx = 0;
    unsigned long long pcv[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pc_mark = clock64();
    enum { pc_refill, pc_sweep1, pc_prep, pc_sweep2, pc_gather, pc_pattern, pc_light, pc_children, pc_push, pc_unwind, pc_iters = 15 };
#define RTC_STAMP(acc)                          \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        const unsigned long long _now = clock64(); \
        pcv[acc] += _now - pc_mark;             \
        pc_mark = _now;                         \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
#else
#define RTC_STAMP(acc) \
    do {               \
    } while (0)
#endif
#ifdef RTC_LANE_MAP
    // developer build (with RTC_PROBE): per region of the loop, how many lanes enter it (lm) and how many times a wave
    // does (lw: counted by the first active lane) - the regions' lane utilisation
    uint32_t lm[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, lw[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RTC_LANE(k)                                                                                                         \
    do {                                                                                                                    \
        ++lm[k];                                                                                                            \
        if (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u) \
            ++lw[k];                                                                                                        \
    } while (0)
#else
#define RTC_LANE(k) \
    do {            \
    } while (0)
#endif
    for (;;) {
        // ---- hand new pixels to idle lanes ----------------------------------------------
        // The wave draws chunks of kChunk consecutive queue positions (one 8x8 pixel block)
        // from the global queue with one atomic, and deals them to its lanes as they fall
        // idle.  A wave therefore works on one or two neighbouring blocks at a time, which
        // keeps the rays of its lanes alike (same shapes hit, same shading branches).
#ifdef RTC_COSTMAP
#define RTC_COSTMAP_RESET() cost_nodes = 0
#else
#define RTC_COSTMAP_RESET() (void)0
#endif
#define RTC_START_TASK()                  \
    do {                                  \
        RTC_COSTMAP_RESET();              \
        mode = kRun;                      \
        aux = job.depth << kAuxRemShift;  \
        sp = 0;                           \
    } while (0)
#define RTC_REFILL_BLOCK_SLOTS
#include "rtc_refill.inc"
#undef RTC_REFILL_BLOCK_SLOTS
#undef RTC_START_TASK
        const bool drained = exhausted && loc_next >= loc_end;  // wave-uniform: no pixel left to start

        bool have_ret = false;     // a finished subtree colour is in `ret`, to be unwound into the frames
        V3 ret = {0.0, 0.0, 0.0};  // Color.Black on a miss (calc.zig:650)
        int fl_done = -1;          // FLUSH: the staging slot of the pixel this lane finished in this iteration

#ifndef RTC_NO_STEAL
        // ---- queue empty: idle lanes take over parked refracted subtrees -------------------
        // Only the order of evaluation changes: the subtree's colour is the same value whoever
        // traces it, and the owner combines it at the same place in its post-order unwinding.
        if (drained) {
            // (1) owners waiting for a taken-over subtree read the helper's colour
            const bool waiting = (aux & kAuxWaiting) != 0;
            const int hl = waiting ? (aux & 63) : lane;
            const int h_done = __shfl((int)(mode == kHelperDone), hl);
            // a finished helper keeps its colour in `ro`, which it no longer needs, until the owner has read it
            const double hx = __shfl(ro.x, hl), hy = __shfl(ro.y, hl), hz = __shfl(ro.z, hl);
            const bool got = waiting && h_done != 0;
            const int ack = __shfl(got ? hl : -1, (aux >> kAuxOwnerShift) & 63);
            if (mode == kHelperDone && ack == lane) mode = kIdle;
            if (got) {
                // refracted child returned: color.scale(transparency) [.scale(1 - reflectance)]
                const int f = sp - 1;
                const int fm = frame_meta(f);
                const double kt = shade_of(fm >> kMetaShapeShift).transparency;
                V3 c = {kt * hx, kt * hy, kt * hz};
                if (fm & kSchlick) {
                    const double T = 1.0 - f_R[f];
                    c = {T * c.x, T * c.y, T * c.z};
                }
                const V3 a0 = frame_acc(f);
                ret = {a0.x + c.x, a0.y + c.y, a0.z + c.z};
                --sp;
                aux &= ~kAuxWaiting;
                have_ret = true;
            }
            // (2) match parked refracted subtrees (shallowest first) with idle lanes
            int offer_f = -1;
            if (mode == kRun) {
                for (int f = sp - 1; f >= 0; --f) {
                    const int fl = frame_meta(f);
                    if ((fl & kHasRefract) && !(fl & (kWaitRefract | kOutsourced))) offer_f = f;
                }
            }
            const unsigned long long offers = __ballot(offer_f >= 0);
            const unsigned long long idles = __ballot(mode == kIdle);
            if (offers != 0 && idles != 0) {
                const int n_off = __popcll(offers), n_idle = __popcll(idles);
                const int n_pairs = n_off < n_idle ? n_off : n_idle;
                const int my_off = (int)set_bits_below_lane(offers), my_idle = (int)set_bits_below_lane(idles);
                const bool give = offer_f >= 0 && my_off < n_pairs;
                const bool take = mode == kIdle && my_idle < n_pairs;
                int partner = lane;
                if (give) partner = nth_set_bit(idles, my_off);
                if (take) partner = nth_set_bit(offers, my_idle);
                double p[6] = {0, 0, 0, 0, 0, 0};
                if (give) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) p[q] = f_ray[offer_f][q];
                    set_frame_meta(offer_f, frame_meta(offer_f) | kOutsourced | (partner << kMetaPartnerShift));
                }
                const int p_rem = ((aux >> kAuxRemShift) & 63) - offer_f - 1;
                double t6[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) t6[q] = __shfl(p[q], partner);
                const int t_rem = __shfl(p_rem, partner);
                if (take) {
                    ro = {t6[0], t6[1], t6[2]};
                    rd = {t6[3], t6[4], t6[5]};
                    aux = (t_rem << kAuxRemShift) | kAuxHelper | (partner << kAuxOwnerShift);
                    sp = 0;
                    mode = kRun;
                }
            }
        }
#endif
        if (__ballot(mode != kIdle) == 0) {
            if (drained) break;
            continue;
        }

        RTC_STAMP(pc_refill);
#ifdef RTC_PROBE
        ++pcv[pc_iters];
#endif
        if (mode != kIdle) RTC_LANE(0);
        if (mode == kRun && !(aux & kAuxWaiting) && !have_ret) {
            RTC_LANE(1);
            // ---- one tree node: calc.worldColorAt (calc.zig:641-652) ---------------------
            if (STATS) ++n_radiance;
#ifdef RTC_COSTMAP
            ++cost_nodes;
#endif
            const int remaining = ((aux >> kAuxRemShift) & 63) - sp;
            Sweep hit;
            sweep_world<false, STATS, ACCEL, STREAM, SLIM>(sweep_tab, g_order, g_bounds_r, g_pairs_r, w, ro, rd, hit, containers, ctr);
            RTC_STAMP(pc_sweep1);

            have_ret = true;
            if (hit.idx >= 0) {
                RTC_LANE(2);
#include "rtc_node.inc"
                if (has_reflect || has_refract) {
                    RTC_LANE(8);
                    // a frame holds only what its unwinding will read: most frames (a reflective,
                    // opaque surface) are the colour so far and the meta word
                    set_frame_acc(sp, surface);
                    if (flags & kSchlick) f_R[sp] = R;
                    if (has_reflect) {
                        if (has_refract) {
                            f_ray[sp][0] = under_point.x;
                            f_ray[sp][1] = under_point.y;
                            f_ray[sp][2] = under_point.z;
                            f_ray[sp][3] = refr_dir.x;
                            f_ray[sp][4] = refr_dir.y;
                            f_ray[sp][5] = refr_dir.z;
                        }
                        set_frame_meta(sp, flags | (hit.idx << kMetaShapeShift));
                        ro = over_point;
                        rd = reflectv;
                    } else {
                        // no reflected child: its term is Color.Black and adds nothing
                        set_frame_meta(sp, flags | kWaitRefract | (hit.idx << kMetaShapeShift));
                        ro = under_point;
                        rd = refr_dir;
                    }
                    ++sp;
                    have_ret = false;  // descend into the child in the next iteration
                } else if (flags & kSchlick) {
                    // surface + Black.scale(R) + Black.scale(1 - R): a non-finite R must poison the colour
                    const double zr = R * 0.0, zt = (1.0 - R) * 0.0;
                    ret = {(surface.x + zr) + zt, (surface.y + zr) + zt, (surface.z + zr) + zt};
                } else {
                    ret = surface;  // surface + Black + Black
                }
            }
        }

        RTC_STAMP(pc_push);
        // ---- unwind finished subtrees in the reference's post-order -----------------------
        if (have_ret) {
            RTC_LANE(9);
            bool settled = false;  // the lane descended into a child or waits for a helper
            // One pass over the levels from the deepest to the root instead of a loop over the
            // lane's own frames: a lane pops level L when that is its top frame, so every lane is
            // served whatever its depth, and the level is a compile-time constant (LDS or private
            // memory is decided statically, addresses are fixed).
#pragma unroll
            for (int L = MAXD - 1; L >= 0; --L) {
                if (!settled && sp == L + 1) {
                    RTC_LANE(10 + (L < 4 ? L : 4));
                    const int flags = frame_meta(L);
                    const V3 acc0 = frame_acc(L);
                    // The returned child is the reflected one (color.scale(reflective), then
                    // .scale(reflectance) under Schlick: calc.zig:958,466) or, when the frame was
                    // waiting for it, the refracted one (color.scale(transparency), then
                    // .scale(1 - reflectance): calc.zig:852,466): one factor or two, then the sum.
                    const bool refracted_back = (flags & kWaitRefract) != 0;
                    const DevShade& fs = shade_of(flags >> kMetaShapeShift);
                    const double k_child = refracted_back ? fs.transparency : fs.reflective;
                    V3 c = {k_child * ret.x, k_child * ret.y, k_child * ret.z};
                    if (flags & kSchlick) {
                        const double fR = f_R[L];
                        const double share = refracted_back ? 1.0 - fR : fR;
                        c = {share * c.x, share * c.y, share * c.z};
                    }
                    const V3 acc = {acc0.x + c.x, acc0.y + c.y, acc0.z + c.z};
                    if (!refracted_back && (flags & kHasRefract)) {
                        // the refracted subtree is next
                        set_frame_acc(L, acc);
                        set_frame_meta(L, flags | kWaitRefract);
                        if (flags & kOutsourced) {
                            aux = (aux & ~63) | kAuxWaiting | ((flags >> kMetaPartnerShift) & 63);
                        } else {
                            ro = {f_ray[L][0], f_ray[L][1], f_ray[L][2]};
                            rd = {f_ray[L][3], f_ray[L][4], f_ray[L][5]};
                        }
                        settled = true;
                    } else {
                        ret = acc;  // the frame is complete (a missing refracted term is Black)
                        sp = L;
                    }
                }
            }
            if (!settled) {
                if (aux & kAuxHelper) {
                    ro = ret;
                    mode = kHelperDone;
                } else {
                    // canvas.set (canvas.zig:24-28)
                    RTC_LANE(15);
#ifdef RTC_COSTMAP
                    ret = {(double)cost_nodes, 0.0, 0.0};  // developer build: nodes traced per pixel instead of the colour
#endif
                    if (FLUSH && (out_pixel & kOutStaged)) {
                        // parked in the wave's staging area until its block is complete (plain stores: the vector L1 is
                        // write-through, so they land in the L2, where the copy-out below reads them past the L1,
                        // whichever lane does; the slot is reused at once and mostly never leaves the L2)
                        const uint32_t fl_wave = (blockIdx.x * (kBlock / 64)) + (threadIdx.x >> 6);
                        double* st = job.flush_stage + ((size_t)fl_wave * (kFlushSlots * 64) + (out_pixel & 0xffu)) * 3;
                        st[0] = ret.x;
                        st[1] = ret.y;
                        st[2] = ret.z;
                        fl_done = (int)((out_pixel >> 6) & 3u);
                    } else {
                        double* __restrict__ px_out = g_out + (size_t)out_pixel * 3;
                        px_out[0] = ret.x;
                        px_out[1] = ret.y;
                        px_out[2] = ret.z;
                    }
                    mode = kIdle;
                }
            }
        }
        RTC_STAMP(pc_unwind);
        if constexpr (FLUSH) {
            // blocks completed by this iteration's pixels leave the staging area: lane i copies pixel i of the block
            if (__ballot(fl_done >= 0) != 0) {
                const int wv = (int)(threadIdx.x >> 6);
                const uint32_t fl_wave = (blockIdx.x * (kBlock / 64)) + (threadIdx.x >> 6);
#pragma unroll 1
                for (int k = 0; k < kFlushSlots; ++k) {
                    const unsigned long long m = __ballot(fl_done == k);
                    if (m == 0) continue;
                    const int left = fl.left[wv][k] - __popcll(m);  // (every lane computes and writes the same value)
                    fl.left[wv][k] = left;
                    if (left > 0) continue;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the wave's staged stores are complete
                    const uint32_t x = (uint32_t)lane & 7u, y = (uint32_t)lane >> 3;
                    if (x < fl.nx[wv][k] && y < fl.ny[wv][k]) {
                        const double* st = job.flush_stage + ((size_t)fl_wave * (kFlushSlots * 64) + (size_t)(k * 64 + lane)) * 3;
                        const double r = __hip_atomic_load(st + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double g = __hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double b = __hip_atomic_load(st + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        double* __restrict__ px_out = g_out + ((size_t)fl.out0[wv][k] + (size_t)y * job.hsize + x) * 3;
                        px_out[0] = r;
                        px_out[1] = g;
                        px_out[2] = b;
                    }
                }
            }
        }
    }

#ifdef RTC_PROBE
    if (lane == 0) {
        unsigned long long* pr = (unsigned long long*)((char*)g_counter + kProbeOffsetBytes);
        const unsigned wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (wid < 8192) {
            pr[wid * 3 + 0] = probe_t0;
            pr[wid * 3 + 1] = probe_tx;
            pr[wid * 3 + 2] = wall_clock64();
            unsigned long long* pc = pr + 3 * 8192 + wid * 16;
            for (int q = 0; q < 16; ++q) pc[q] = pcv[q];
        }
    }
#ifdef RTC_LANE_MAP
    for (int q = 0; q < 16; ++q) {
        unsigned long long v = (unsigned long long)lm[q] | ((unsigned long long)lw[q] << 32);
        unsigned long long lo = v & 0xffffffffull, hi = v >> 32;
        for (int off = 32; off > 0; off >>= 1) lo += __shfl_down(lo, off), hi += __shfl_down(hi, off);
        const unsigned wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (lane == 0 && wid < 8192) {
            unsigned long long* pr = (unsigned long long*)((char*)g_counter + kProbeOffsetBytes);
            // (iterations stay in slot 15's high half: lw[0] is the wave's iteration count as well)
            pr[3 * 8192 + wid * 16 + q] = (lo & 0xffffffffull) | (hi << 32);
        }
    }
#endif
#endif
    if (STATS) {
        // one atomic per counter per wave
        uint32_t v[9] = {n_radiance, n_shadow, ctr.sphere_hits, ctr.plane_hits, n_shaded, ctr.sphere_skips, ctr.plane_skips,
                         ctr.cube_hits, ctr.sphere_prefilter_skips};
        for (int c = 0; c < 9; ++c) {
            unsigned long long sum = v[c];
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
            if (lane == 0) atomicAdd(&g_stats[c], sum);
        }
    }
}

#undef RTC_STAMP
#define RTC_STAMP(acc) \
    do {               \
    } while (0)
#undef RTC_LANE
#define RTC_LANE(k) \
    do {            \
    } while (0)

// The attenuation-weighted integrator (rtc_scene_set_integrator: RTC_INTEGRATOR_WEIGHTED).
//
// The reference's recursion (calc.zig:460-469) multiplies a child's colour by the factors of every
// ancestor on the way back up: reflective or transparency (calc.zig:958,852), and the Schlick share
// (calc.zig:466).  The same sum, taken the other way round: a ray carries the product of the
// factors on its path from the root, every hit adds weight x surface colour to the pixel, and a
// child starts with weight x its factor.  Same rays, same hits, same surface colours as the
// post-order kernel; what differs is the association of the products and of the sum, i.e. the
// last bits of a colour (a few 1e-16 relative; the parity tests allow 1e-9).  What it saves: a hit
// with one child pushes nothing (the post-order kernel parks a frame for it), there is no pass
// that unwinds finished subtrees, and a parked refracted ray is an independent piece of work: any
// idle lane can trace it and add its share to the pixel on its own.
//   per lane: the ray, its weight and `remaining`, the colour gathered so far for the pixel, and a
//   stack of parked refracted rays (ray, weight, remaining), pushed only by hits with two children.
//   Once the pixel queue is empty, idle lanes take over parked rays of busy lanes of the wave (the
//   shallowest first).  The pixel then has several contributors; each adds its share with atomic
//   additions, the owner having stored zeros when it first gave a ray away.  Sums arriving in a
//   different order are why this integrator's last bits can differ from run to run.
// A non-finite Schlick reflectance does not poison a childless hit's colour as calc.zig:466 would
// (weights only reach children).
template <int MAXD, bool STATS, bool FROM_RAYS, int TABLES>
__global__ __launch_bounds__(kBlock, TABLES == 0 ? RTC_WAVES_PER_EU_LARGE : RTC_WAVES_PER_EU) void render_weighted_kernel(
    const double* __restrict__ g_tab, const DevShade* __restrict__ g_shade, const int* __restrict__ g_order,
    const double* __restrict__ g_rays, const uint32_t* __restrict__ g_tile_order, double* __restrict__ g_out,
    uint32_t* __restrict__ g_counter, unsigned long long* __restrict__ g_stats, const double* __restrict__ g_bounds_r,
    const double* __restrict__ g_bounds_s, const double* __restrict__ g_inv, const float* __restrict__ g_pairs_r,
    const float* __restrict__ g_pairs_s, const DevWorld wc, const DevJob job) {
    constexpr bool LDS = TABLES == 1 || TABLES == 3;
    constexpr bool SLIM = TABLES == 3;  // (see launch_variant)
    constexpr bool ACCEL = TABLES == 2;
    constexpr bool STREAM = TABLES == 0;
    constexpr int kDrawThreshold = STREAM ? 0 : 1;  // whether job.draw_threshold applies (rtc_refill.inc)
    __shared__ ContainerSlots containers;
    __shared__ HotTables<LDS> hot;
    if constexpr (LDS) {
        const int n = wc.n;
        for (int k = (int)threadIdx.x; k < n * 12; k += kBlock) hot.mat[k / 12][k % 12] = g_inv[k];
        const double* __restrict__ src = reinterpret_cast<const double*>(g_shade);
        double* dst = reinterpret_cast<double*>(hot.shade);
        for (int k = (int)threadIdx.x; k < n * (int)(sizeof(DevShade) / 8); k += kBlock) dst[k] = src[k];
        __syncthreads();
    }
    auto shade_of = [&](int i) -> const DevShade& {
        if constexpr (LDS)
            return hot.shade[i];
        else
            return g_shade[i];
    };
    auto inverse_of = [&](int i) -> const double* {
        if constexpr (LDS)
            return hot.mat[i];
        else
            return g_inv + (size_t)i * 12;
    };
    const double* __restrict__ sweep_tab = g_tab;
    const int lane = (int)(threadIdx.x & 63);
    DevWorld w = wc;
    w.tab = g_tab;
    w.shade = g_shade;

    // parked refracted rays of the lane (private memory): ray, weight, remaining
    double p_ray[MAXD][6];
    double p_w[MAXD];
    int p_rem[MAXD];

    uint32_t n_radiance = 0, n_shadow = 0, n_shaded = 0;
    SweepCounters ctr = {};

    int mode = kIdle;
    uint32_t out_pixel = 0;
    int sp = 0;             // parked rays on the lane's stack
    uint32_t stolen = 0;    // bit f: entry f of the stack was taken over by another lane
    bool shared = false;    // the pixel has (or may get) other contributors: add, do not store
    int rem = 0;            // `remaining` (calc.zig:641) of the ray being traced
    double wgt = 0.0;       // product of the factors on the path from the pixel's root to this ray
    V3 acc = {0, 0, 0};     // the task's share of the pixel so far
    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    bool exhausted = false;
    uint32_t loc_next = 0, loc_end = 0;
    uint32_t cur_queue = ((blockIdx.x * (kBlock / 64)) + (threadIdx.x >> 6)) % kQueues;
    uint32_t blk_base = 0, blk_px0 = 0, blk_py0 = 0;
    bool blk_valid = false;
    uint32_t blk_out0 = 0;
#ifdef RTC_PROBE
    unsigned long long probe_tx = 0;
#endif

    for (;;) {
#define RTC_START_TASK()      \
    do {                      \
        mode = kRun;          \
        rem = job.depth;      \
        wgt = 1.0;            \
        acc = V3{0, 0, 0};    \
        sp = 0;               \
        stolen = 0;           \
        shared = false;       \
    } while (0)
#include "rtc_refill.inc"
#undef RTC_START_TASK
        const bool drained = exhausted && loc_next >= loc_end;

        // ---- queue empty: idle lanes take over parked rays -------------------------------------
        if (drained) {
            int offer_f = -1;
            if (mode == kRun) {
#pragma unroll
                for (int f = MAXD - 1; f >= 0; --f)
                    if (f < sp && !((stolen >> f) & 1u)) offer_f = f;  // ends at the shallowest
            }
            const unsigned long long offers = __ballot(offer_f >= 0);
            const unsigned long long idles = __ballot(mode == kIdle);
            if (offers != 0 && idles != 0) {
                const int n_off = __popcll(offers), n_idle = __popcll(idles);
                const int n_pairs = n_off < n_idle ? n_off : n_idle;
                const int my_off = (int)set_bits_below_lane(offers), my_idle = (int)set_bits_below_lane(idles);
                const bool give = offer_f >= 0 && my_off < n_pairs;
                const bool take = mode == kIdle && my_idle < n_pairs;
                int partner = lane;
                if (give) partner = nth_set_bit(idles, my_off);
                if (take) partner = nth_set_bit(offers, my_idle);
                double p[7] = {0, 0, 0, 0, 0, 0, 0};
                int p_r = 0;
                if (give) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) p[q] = p_ray[offer_f][q];
                    p[6] = p_w[offer_f];
                    p_r = p_rem[offer_f];
                    stolen |= 1u << offer_f;
                    if (!shared) {
                        // the pixel gets a second contributor: from here on everybody adds, to zeros
                        double* __restrict__ px_out = g_out + (size_t)out_pixel * 3;
                        px_out[0] = 0.0;
                        px_out[1] = 0.0;
                        px_out[2] = 0.0;
                        shared = true;
                    }
                }
                // the zeros are in place before any lane of this wave can add to them
                __threadfence();
                double t7[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) t7[q] = __shfl(p[q], partner);
                const int t_rem = __shfl(p_r, partner);
                const uint32_t t_pixel = (uint32_t)__shfl((int)out_pixel, partner);
                if (take) {
                    ro = {t7[0], t7[1], t7[2]};
                    rd = {t7[3], t7[4], t7[5]};
                    wgt = t7[6];
                    rem = t_rem;
                    out_pixel = t_pixel;
                    acc = V3{0, 0, 0};
                    sp = 0;
                    stolen = 0;
                    shared = true;
                    mode = kRun;
                }
            }
        }
        if (__ballot(mode != kIdle) == 0) {
            if (drained) break;
            continue;
        }

        if (mode == kRun) {
            // ---- one tree node: calc.worldColorAt (calc.zig:641-652) ---------------------
            if (STATS) ++n_radiance;
            const int remaining = rem;
            Sweep hit;
            sweep_world<false, STATS, ACCEL, STREAM, SLIM>(sweep_tab, g_order, g_bounds_r, g_pairs_r, w, ro, rd, hit, containers, ctr);
            bool chain_ends = true;  // a miss adds Color.Black (calc.zig:650)
            if (hit.idx >= 0) {
#include "rtc_node.inc"
                acc = {acc.x + wgt * surface.x, acc.y + wgt * surface.y, acc.z + wgt * surface.z};
                // color.scale(reflective) [.scale(reflectance)] and color.scale(transparency)
                // [.scale(1 - reflectance)] (calc.zig:958,852,466) as factors of the children's weights
                double w_reflect = wgt * m_reflective, w_refract = wgt * m_transparency;
                if (flags & kSchlick) {
                    w_reflect = w_reflect * R;
                    w_refract = w_refract * (1.0 - R);
                }
                if (has_reflect || has_refract) {
                    if (has_reflect && has_refract) {
                        p_ray[sp][0] = under_point.x;
                        p_ray[sp][1] = under_point.y;
                        p_ray[sp][2] = under_point.z;
                        p_ray[sp][3] = refr_dir.x;
                        p_ray[sp][4] = refr_dir.y;
                        p_ray[sp][5] = refr_dir.z;
                        p_w[sp] = w_refract;
                        p_rem[sp] = rem - 1;
                        ++sp;
                    }
                    if (has_reflect) {
                        ro = over_point;
                        rd = reflectv;
                        wgt = w_reflect;
                    } else {
                        ro = under_point;
                        rd = refr_dir;
                        wgt = w_refract;
                    }
                    rem = rem - 1;
                    chain_ends = false;
                }
            }
            if (chain_ends) {
                // the next parked ray nobody has taken, or the task is complete
                bool resumed = false;
#pragma unroll
                for (int L = MAXD - 1; L >= 0; --L) {
                    if (!resumed && sp == L + 1) {
                        sp = L;
                        if (!((stolen >> L) & 1u)) {
                            ro = {p_ray[L][0], p_ray[L][1], p_ray[L][2]};
                            rd = {p_ray[L][3], p_ray[L][4], p_ray[L][5]};
                            wgt = p_w[L];
                            rem = p_rem[L];
                            resumed = true;
                        }
                        stolen &= ~(1u << L);
                    }
                }
                if (!resumed) {
                    // canvas.set (canvas.zig:24-28), or this task's share of it
                    double* __restrict__ px_out = g_out + (size_t)out_pixel * 3;
                    if (shared) {
                        __hip_atomic_fetch_add(px_out + 0, acc.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(px_out + 1, acc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(px_out + 2, acc.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        px_out[0] = acc.x;
                        px_out[1] = acc.y;
                        px_out[2] = acc.z;
                    }
                    mode = kIdle;
                }
            }
        }
    }

    if (STATS) {
        uint32_t v[9] = {n_radiance, n_shadow, ctr.sphere_hits, ctr.plane_hits, n_shaded, ctr.sphere_skips, ctr.plane_skips,
                         ctr.cube_hits, ctr.sphere_prefilter_skips};
        for (int c = 0; c < 9; ++c) {
            unsigned long long sum = v[c];
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
            if (lane == 0) atomicAdd(&g_stats[c], sum);
        }
    }
}

// Scatter gathered per-rank tile buffers into the row-major canvas.
template <typename T>
__global__ __launch_bounds__(kBlock) void deinterleave_kernel(const T* __restrict__ gathered,
                                                              uint64_t per_rank_doubles, uint32_t tile_ranks,
                                                              uint32_t hsize, uint32_t vsize, uint32_t tiles_x,
                                                              uint32_t tiles_total, T* __restrict__ out) {
    const uint32_t tile_px = RTC_TILE_W * RTC_TILE_H;
    const uint64_t total = (uint64_t)tiles_total * tile_px;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t gt = (uint32_t)(g / tile_px), within = (uint32_t)(g % tile_px);
        const uint32_t px = (gt % tiles_x) * RTC_TILE_W + within % RTC_TILE_W;
        const uint32_t py = (gt / tiles_x) * RTC_TILE_H + within / RTC_TILE_W;
        if (px >= hsize || py >= vsize) continue;
        const uint32_t rank = gt % tile_ranks, lt = gt / tile_ranks;
        const T* src = gathered + (uint64_t)rank * per_rank_doubles + ((uint64_t)lt * tile_px + within) * 3;
        T* dst = out + ((uint64_t)py * hsize + px) * 3;
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
    }
}

// One rank's dense tile buffer into the row-major canvas (which may be a peer GPU's memory: the
// sender scatters, so the receiver needs no gather buffer and no pass of its own).
template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_tiles_kernel(const T* __restrict__ tiles, uint32_t tile_rank,
                                                               uint32_t tile_ranks, uint32_t hsize, uint32_t vsize,
                                                               uint32_t tiles_x, uint32_t local_tiles,
                                                               T* __restrict__ canvas) {
    const uint32_t tile_px = RTC_TILE_W * RTC_TILE_H;
    const uint64_t total = (uint64_t)local_tiles * tile_px;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lt = (uint32_t)(g / tile_px), within = (uint32_t)(g % tile_px);
        const uint32_t gt = lt * tile_ranks + tile_rank;
        const uint32_t px = (gt % tiles_x) * RTC_TILE_W + within % RTC_TILE_W;
        const uint32_t py = (gt / tiles_x) * RTC_TILE_H + within / RTC_TILE_W;
        if (px >= hsize || py >= vsize) continue;
        const T* src = tiles + g * 3;
        T* dst = canvas + ((uint64_t)py * hsize + px) * 3;
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
    }
}

// byteColorValue (ppm.zig:6-9): clamp(v*256, 0, 255) with Zig's min/max, then truncate.
__global__ __launch_bounds__(kBlock) void quantize_kernel(const double* __restrict__ rgb, uint64_t n,
                                                          uint8_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        double v = rgb[i] * 256.0;
        v = v < 255.0 ? v : 255.0;
        v = v > 0.0 ? v : 0.0;
        out[i] = (uint8_t)v;
    }
}

// a/b, sqrt(a) and pow(a,b) element-wise, for bit-level comparison against the host.
__global__ __launch_bounds__(kBlock) void arith_probe_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                             uint64_t n, double* __restrict__ out_div,
                                                             double* __restrict__ out_sqrt, double* __restrict__ out_pow) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        out_div[i] = a[i] / b[i];
        out_sqrt[i] = sqrt(a[i]);
        // as the render kernel's calls are dispatched: an integer exponent in [1, 500) takes the
        // plain-multiplication path
        const double x = a[i], y = b[i];
        const bool small_integer = y >= 1.0 && y < (double)kPowSmallLimit && y == floor(y);
        out_pow[i] = small_integer ? pow_small_integer(x, (int)y) : zig_pow(x, y);
    }
}

template <int MAXD, bool STATS, bool FROM_RAYS, int TABLES>
hipError_t launch_tables(const DevWorld& world, const DevJob& job, hipStream_t stream) {
    auto kernel = job.weighted ? render_weighted_kernel<MAXD, STATS, FROM_RAYS, TABLES> : render_kernel<MAXD, STATS, FROM_RAYS, TABLES>;
    // block stores: frame kernels of the post-order integrator only (no statistics, camera rays)
    if constexpr (!STATS && !FROM_RAYS) {
        if (!job.weighted && job.flush_stage != nullptr && job.out_row_major) kernel = render_kernel<MAXD, STATS, FROM_RAYS, TABLES, true>;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    uint64_t blocks = (uint64_t)cus * (uint64_t)per_cu;
    const uint64_t needed = ((uint64_t)job.total + kBlock - 1) / kBlock;
    if (blocks > needed) blocks = needed;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, world.tab, world.shade, world.order, job.rays,
                       job.tile_order, job.out, job.counter, (unsigned long long*)job.stats, world.sphere_bounds_radiance,
                       world.sphere_bounds_shadow, world.inv, world.sphere_pairs_radiance, world.sphere_pairs_shadow, world, job);
    return hipGetLastError();
}

template <int MAXD, bool STATS, bool FROM_RAYS>
hipError_t launch_variant(const DevWorld& world, const DevJob& job, hipStream_t stream) {
    if (world.bvh) return launch_tables<MAXD, STATS, FROM_RAYS, 2>(world, job, stream);
    if (world.n <= kLdsShapes) {
        // A world whose spheres all have diagonal inverses and that holds no cube (the reference's own scenes) gets
        // kernels without the loops and the normal arithmetic of the general kinds: TABLES == 3.  Only the frame
        // kernels (no statistics, camera rays) exist in that form.
        if constexpr (!STATS && !FROM_RAYS) {
            if (world.group[1] == world.group[2] && world.group[4] == world.group[5])
                return launch_tables<MAXD, STATS, FROM_RAYS, 3>(world, job, stream);
        }
        return launch_tables<MAXD, STATS, FROM_RAYS, 1>(world, job, stream);
    }
    return launch_tables<MAXD, STATS, FROM_RAYS, 0>(world, job, stream);
}

}  // namespace

hipError_t launch_render(const DevWorld& world, const DevJob& job, bool with_stats, hipStream_t stream) {
    const bool rays = job.rays != nullptr;
    const bool small = job.depth <= kMaxDepthSmall;
    if (small) {
        if (with_stats) return rays ? launch_variant<kMaxDepthSmall, true, true>(world, job, stream)
                                    : launch_variant<kMaxDepthSmall, true, false>(world, job, stream);
        return rays ? launch_variant<kMaxDepthSmall, false, true>(world, job, stream)
                    : launch_variant<kMaxDepthSmall, false, false>(world, job, stream);
    }
    if (with_stats) return rays ? launch_variant<kMaxDepthLarge, true, true>(world, job, stream)
                                : launch_variant<kMaxDepthLarge, true, false>(world, job, stream);
    return rays ? launch_variant<kMaxDepthLarge, false, true>(world, job, stream)
                : launch_variant<kMaxDepthLarge, false, false>(world, job, stream);
}

namespace {
template <typename T>
hipError_t launch_deinterleave_t(const T* gathered, uint64_t per_rank_values, uint32_t tile_ranks, uint32_t hsize,
                                 uint32_t vsize, T* out, hipStream_t stream) {
    const uint32_t tiles_x = (hsize + RTC_TILE_W - 1) / RTC_TILE_W;
    const uint32_t tiles_y = (vsize + RTC_TILE_H - 1) / RTC_TILE_H;
    const uint32_t tiles_total = tiles_x * tiles_y;
    const uint64_t total = (uint64_t)tiles_total * RTC_TILE_W * RTC_TILE_H;
    uint64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(deinterleave_kernel<T>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, gathered, per_rank_values,
                       tile_ranks, hsize, vsize, tiles_x, tiles_total, out);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_deinterleave(const double* gathered, uint64_t per_rank_doubles, uint32_t tile_ranks, uint32_t hsize,
                               uint32_t vsize, double* out, hipStream_t stream) {
    return launch_deinterleave_t<double>(gathered, per_rank_doubles, tile_ranks, hsize, vsize, out, stream);
}

hipError_t launch_deinterleave_u8(const uint8_t* gathered, uint64_t per_rank_bytes, uint32_t tile_ranks, uint32_t hsize,
                                  uint32_t vsize, uint8_t* out, hipStream_t stream) {
    return launch_deinterleave_t<uint8_t>(gathered, per_rank_bytes, tile_ranks, hsize, vsize, out, stream);
}

namespace {
template <typename T>
hipError_t launch_scatter_tiles_t(const T* tiles, uint32_t tile_rank, uint32_t tile_ranks, uint32_t hsize, uint32_t vsize,
                                  T* canvas, hipStream_t stream) {
    const uint32_t tiles_x = (hsize + RTC_TILE_W - 1) / RTC_TILE_W;
    const uint32_t tiles_y = (vsize + RTC_TILE_H - 1) / RTC_TILE_H;
    const uint64_t tiles_total = (uint64_t)tiles_x * tiles_y;
    const uint64_t local_tiles = tiles_total > tile_rank ? (tiles_total - tile_rank + tile_ranks - 1) / tile_ranks : 0;
    if (local_tiles == 0) return hipSuccess;
    uint64_t blocks = (local_tiles * RTC_TILE_W * RTC_TILE_H + kBlock - 1) / kBlock;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(scatter_tiles_kernel<T>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, tiles, tile_rank, tile_ranks,
                       hsize, vsize, tiles_x, (uint32_t)local_tiles, canvas);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_scatter_tiles(const double* tiles, uint32_t tile_rank, uint32_t tile_ranks, uint32_t hsize,
                                uint32_t vsize, double* canvas, hipStream_t stream) {
    return launch_scatter_tiles_t<double>(tiles, tile_rank, tile_ranks, hsize, vsize, canvas, stream);
}

hipError_t launch_scatter_tiles_u8(const uint8_t* tiles, uint32_t tile_rank, uint32_t tile_ranks, uint32_t hsize,
                                   uint32_t vsize, uint8_t* canvas, hipStream_t stream) {
    return launch_scatter_tiles_t<uint8_t>(tiles, tile_rank, tile_ranks, hsize, vsize, canvas, stream);
}

hipError_t launch_quantize(const double* rgb, uint64_t n, uint8_t* out, hipStream_t stream) {
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, rgb, n, out);
    return hipGetLastError();
}

// Zeroes `words` 32-bit words (the control block of a launch).
__global__ __launch_bounds__(kBlock) void zero_words_kernel(uint32_t* __restrict__ p, uint32_t words) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) p[i] = 0u;
}
hipError_t launch_zero_words(uint32_t* p, uint32_t words, hipStream_t stream) {
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(kBlock), 0, stream, p, words);
    return hipGetLastError();
}

hipError_t launch_arith_probe(const double* a, const double* b, uint64_t n, double* out_div, double* out_sqrt,
                              double* out_pow, hipStream_t stream) {
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(arith_probe_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, b, n, out_div, out_sqrt,
                       out_pow);
    return hipGetLastError();
}

}  // namespace rtc
