// Host side of librtc_hip.so: the C-ABI declared in include/rtc_hip.h.
//
// Scene upload flattens the reference's World (world.zig:15-28) into device tables and
// precomputes, once per shape / pattern / camera, the inverse the reference recomputes on
// every call (shape.zig:85,100; pattern.zig:78-79; camera.zig:62) with the reference's own
// cofactor algorithm, so the bits are the same.
#include "rtc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#pragma clang fp contract(off)

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int fail_hip(const char* what, hipError_t e) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    // the failure is reported through the status code; the runtime's own sticky "last error" is cleared, so
    // that it does not resurface in the caller's next, unrelated HIP call (a framework checking hipGetLastError)
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? RTC_ERR_OUT_OF_MEMORY : RTC_ERR_DEVICE;
}

#define HIP_TRY(expr)                                 \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return fail_hip(#expr, _e); \
    } while (0)

constexpr double kF64Eps = 2.220446049250313080847263336181640625e-16;

// ---- Mat4.inverse, matrix.zig:238-300 -------------------------------------------------
// Cofactor expansion exactly as the reference evaluates it: Mat2 determinant a*d - b*c
// (matrix.zig:23-26); Mat3 and Mat4 determinants accumulate `det += m[0][col] * cofactor`
// from zero over the first row (matrix.zig:88-96, 272-280); cofactor = +-minor by parity of
// row+col; inverse[row][col] = cofactor(col,row) / det.  Index lists replace the
// reference's sub-matrix copies; the arithmetic and its order are unchanged.
double det3_of(const double m[16], const int rows[3], const int cols[3]) {
    auto at = [&](int r, int c) { return m[rows[r] * 4 + cols[c]]; };
    const double minor0 = at(1, 1) * at(2, 2) - at(1, 2) * at(2, 1);
    const double minor1 = at(1, 0) * at(2, 2) - at(1, 2) * at(2, 0);
    const double minor2 = at(1, 0) * at(2, 1) - at(1, 1) * at(2, 0);
    double det = 0;
    det += at(0, 0) * minor0;
    det += at(0, 1) * -minor1;
    det += at(0, 2) * minor2;
    return det;
}

double cofactor4(const double m[16], int row, int col) {
    int rows[3], cols[3];
    for (int i = 0, k = 0; i < 4; ++i)
        if (i != row) rows[k++] = i;
    for (int i = 0, k = 0; i < 4; ++i)
        if (i != col) cols[k++] = i;
    const double minor = det3_of(m, rows, cols);
    return ((row + col) % 2 != 0) ? -minor : minor;
}

double det4(const double m[16]) {
    double det = 0;
    for (int col = 0; col < 4; ++col) det += m[col] * cofactor4(m, 0, col);
    return det;
}

// Returns RTC_OK and the full 4x4 inverse, or the reason the transform is unusable.
int invert_affine(const double m[16], double inv[16], const char* what, uint64_t index) {
    const double det = det4(m);
    // matrix.zig:288 asserts !epsilonEq(det, 0): |det - 0| <= f64 epsilon is a panic there
    if (!(std::fabs(det - 0.0) > kF64Eps)) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s %llu: transform is not invertible (det = %g)", what,
                      (unsigned long long)index, det);
        return fail(RTC_ERR_SINGULAR_TRANSFORM, buf);
    }
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) inv[row * 4 + col] = cofactor4(m, col, row) / det;
    // The kernels carry points as (x,y,z,1) and directions as (x,y,z,0); that is exact only
    // when the inverse maps w to w, i.e. its last row is (0,0,0,1).
    if (!(inv[12] == 0.0 && inv[13] == 0.0 && inv[14] == 0.0 && inv[15] == 1.0)) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s %llu: transform is not affine (inverse last row %g %g %g %g)", what,
                      (unsigned long long)index, inv[12], inv[13], inv[14], inv[15]);
        return fail(RTC_ERR_NOT_AFFINE, buf);
    }
    return RTC_OK;
}

// ---- value equality of two shapes, std.meta.eql (calc.zig:274,280) ---------------------
// Field-wise ==; floats compare with == (so -0 == 0, NaN != NaN); the optional pattern by
// presence, then tag, then payload (the `point` tag has no colours).
bool eq_doubles(const double* a, const double* b, int n) {
    for (int i = 0; i < n; ++i)
        if (!(a[i] == b[i])) return false;
    return true;
}
bool shapes_equal(const RtcShapeDesc& a, const RtcShapeDesc& b) {
    if (a.geo != b.geo || !eq_doubles(a.transform, b.transform, 16)) return false;
    const RtcMaterialDesc &m = a.material, &n = b.material;
    if (!eq_doubles(m.color, n.color, 3) || !(m.ambient == n.ambient) || !(m.diffuse == n.diffuse) ||
        !(m.specular == n.specular) || !(m.shininess == n.shininess) || !(m.reflective == n.reflective) ||
        !(m.transparency == n.transparency) || !(m.refractive_index == n.refractive_index))
        return false;
    if ((m.has_pattern != 0) != (n.has_pattern != 0)) return false;
    if (!m.has_pattern) return true;
    if (m.pattern.kind != n.pattern.kind) return false;
    if (m.pattern.kind != RTC_PATTERN_POINT &&
        !(eq_doubles(m.pattern.a, n.pattern.a, 3) && eq_doubles(m.pattern.b, n.pattern.b, 3)))
        return false;
    return eq_doubles(m.pattern.transform, n.pattern.transform, 16);
}
// Hash consistent with shapes_equal (zeros canonicalised; a NaN anywhere makes the shape
// unequal to everything, itself included, so such shapes are never reported).
struct ShapeHasher {
    uint64_t h = 1469598103934665603ull;
    bool has_nan = false;
    void word(uint64_t v) {
        for (int i = 0; i < 8; ++i) {
            h ^= (v >> (8 * i)) & 0xff;
            h *= 1099511628211ull;
        }
    }
    void f64(double d) {
        if (d != d) has_nan = true;
        if (d == 0.0) d = 0.0;
        uint64_t u;
        std::memcpy(&u, &d, 8);
        word(u);
    }
    void f64s(const double* p, int n) {
        for (int i = 0; i < n; ++i) f64(p[i]);
    }
};
uint64_t hash_shape(const RtcShapeDesc& s, bool* has_nan) {
    ShapeHasher H;
    H.word((uint64_t)s.geo);
    H.f64s(s.transform, 16);
    const RtcMaterialDesc& m = s.material;
    H.f64s(m.color, 3);
    H.f64(m.ambient);
    H.f64(m.diffuse);
    H.f64(m.specular);
    H.f64(m.shininess);
    H.f64(m.reflective);
    H.f64(m.transparency);
    H.f64(m.refractive_index);
    H.word(m.has_pattern ? 1 : 0);
    if (m.has_pattern) {
        H.word((uint64_t)m.pattern.kind);
        if (m.pattern.kind != RTC_PATTERN_POINT) {
            H.f64s(m.pattern.a, 3);
            H.f64s(m.pattern.b, 3);
        }
        H.f64s(m.pattern.transform, 16);
    }
    *has_nan = H.has_nan;
    return H.h;
}

bool valid_stream_and_ptr(const void* p) { return p != nullptr; }

}  // namespace

// ---- the opaque scene ---------------------------------------------------------------------
struct RtcScene {
    int device = 0;
    uint64_t serial = 0;  // unique per rtc_scene_create in this process (an address can be reused, a serial is not)
    uint64_t n = 0;
    uint64_t n_spheres = 0, n_planes = 0, n_cubes = 0;
    double* d_tab = nullptr;
    double* d_inv = nullptr;            // rows 0..2 of inverse(shape.transform), 12 doubles per shape, table order
    float* d_sphere_pairs = nullptr;    // single-precision pre-test tables: radiance, then shadow (see DevWorld)
    size_t sphere_pairs_floats = 0;     // floats per table
    uint32_t block_pairs_offset_radiance = 0, block_pairs_offset_shadow = 0;  // see DevWorld
    uint32_t super_pairs_offset_radiance = 0, super_pairs_offset_shadow = 0;
    uint32_t group_pairs_offset_radiance = 0, group_pairs_offset_shadow = 0;
    int32_t sweep_top_log2 = 6;
    uint32_t sweep_axis_mask[3] = {0, 0, 0};  // see DevWorld
    double block_shrink = 0.0, block_margin_scale = 1.0;
    double* d_sphere_bounds = nullptr;  // radiance table, then shadow table (see DevWorld)
    size_t sphere_bounds_doubles = 0;   // doubles per table
    int32_t* d_order = nullptr;
    int32_t group[rtc::kSweepGroups + 1] = {};
    rtc::DevShade* d_shade = nullptr;
    uint32_t* d_counter = nullptr;  // control block: work-queue heads, statistics counters (see rtc_internal.h)
    uint64_t* d_stats = nullptr;
    double light_pos[3] = {0, 0, 0};
    double light_int[3] = {0, 0, 0};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // Shapes whose hits branch into two subtrees (transparent materials): world-space bounding
    // spheres, used to hand out the tiles that show them first (see plan_tile_order).
    struct Costly {
        double c[3], r;
    };
    std::vector<Costly> costly;
    bool costly_everywhere = false;  // a transparent plane: no useful bound
    // the plan of the last camera / partition, reused while they do not change
    // optional hierarchy over the spheres (rtc_scene_set_acceleration)
    struct SphereBox {
        double lo[3], hi[3];
        int32_t pos;  // table position
    };
    struct SphereBound {
        double b[4];  // centre and enlarged squared radius, as in the sweep table
        double from_light[4];
        int32_t pos;
    };
    std::vector<SphereBound> sphere_bounds;  // kept from rtc_scene_create
    rtc::BvhNode* d_bvh = nullptr;
    int32_t bvh_root = 0;
    double bvh_origin_limit = 0;
    int32_t integrator = RTC_INTEGRATOR_POSTORDER;  // rtc_scene_set_integrator
    int32_t canvas_store = RTC_CANVAS_STORE_PIXELS;  // rtc_scene_set_canvas_store
    double* d_flush_stage = nullptr;                 // staging area of the block stores (see DevJob::flush_stage)
    uint32_t flush_waves = 0;
    void* h_staging = nullptr;  // pinned staging buffer of rtc_render's copy to the host, kept between calls
    size_t staging_bytes = 0;
    std::vector<hipEvent_t> slice_events;
    double* d_canvas = nullptr;  // device canvas of rtc_render, kept between calls
    size_t canvas_bytes = 0;
    // rtc_render of a large canvas renders in bands of tile rows (render_to_host_in_bands): its two streams, one event
    // per band, and the bands' tile lists of the last camera
    hipStream_t band_render_stream = nullptr, band_copy_stream = nullptr;
    std::vector<hipEvent_t> band_events;
    uint32_t* d_band_order = nullptr;
    uint64_t band_order_capacity = 0;
    bool band_plan_valid = false;
    RtcCameraDesc band_camera{};
    std::vector<uint32_t> band_first;  // per band: where its tiles start in d_band_order (one more entry: the end)
    std::vector<uint32_t> band_row;    // per band: its first pixel row (one more entry: vsize)
    uint32_t* d_tile_order = nullptr;
    uint64_t tile_order_capacity = 0;
    unsigned long long* d_strip_order = nullptr;  // the queue in strips (build_strip_order), when the plan has one
    uint64_t strip_order_capacity = 0;
    bool plan_valid = false, plan_identity = true, plan_has_strips = false, plan_row_major = false;
    int plan_strips = 0;
    RtcCameraDesc plan_camera{};
    uint32_t plan_rank = 0, plan_ranks = 0;
};

namespace {

void free_scene(RtcScene* s) {
    if (!s) return;
    if (s->d_tab) (void)hipFree(s->d_tab);
    if (s->d_order) (void)hipFree(s->d_order);
    if (s->d_sphere_bounds) (void)hipFree(s->d_sphere_bounds);
    if (s->d_sphere_pairs) (void)hipFree(s->d_sphere_pairs);
    if (s->d_inv) (void)hipFree(s->d_inv);
    if (s->d_shade) (void)hipFree(s->d_shade);
    if (s->d_counter) (void)hipFree(s->d_counter);
    if (s->d_tile_order) (void)hipFree(s->d_tile_order);
    if (s->d_strip_order) (void)hipFree(s->d_strip_order);
    if (s->d_canvas) (void)hipFree(s->d_canvas);
    if (s->h_staging) (void)hipHostFree(s->h_staging);
    for (hipEvent_t ev : s->slice_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s->band_events) (void)hipEventDestroy(ev);
    if (s->band_render_stream) (void)hipStreamDestroy(s->band_render_stream);
    if (s->band_copy_stream) (void)hipStreamDestroy(s->band_copy_stream);
    if (s->d_band_order) (void)hipFree(s->d_band_order);
    if (s->d_bvh) (void)hipFree(s->d_bvh);
    if (s->d_flush_stage) (void)hipFree(s->d_flush_stage);
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
    delete s;
}

rtc::DevWorld dev_world(const RtcScene* s) {
    rtc::DevWorld w{};
    w.tab = s->d_tab;
    w.order = s->d_order;
    w.inv = s->d_inv;
    w.sphere_pairs_radiance = s->d_sphere_pairs;
    w.sphere_pairs_shadow = s->d_sphere_pairs ? s->d_sphere_pairs + s->sphere_pairs_floats : nullptr;
    w.block_pairs_offset_radiance = s->block_pairs_offset_radiance;
    w.block_pairs_offset_shadow = s->block_pairs_offset_shadow;
    w.super_pairs_offset_radiance = s->super_pairs_offset_radiance;
    w.super_pairs_offset_shadow = s->super_pairs_offset_shadow;
    w.group_pairs_offset_radiance = s->group_pairs_offset_radiance;
    w.group_pairs_offset_shadow = s->group_pairs_offset_shadow;
    w.sweep_top_log2 = s->sweep_top_log2;
    for (int a = 0; a < 3; ++a) w.sweep_axis_mask[a] = s->sweep_axis_mask[a];
    w.block_shrink = s->block_shrink;
    w.block_margin_scale = s->block_margin_scale;
    w.sphere_bounds_radiance = s->d_sphere_bounds;
    w.sphere_bounds_shadow = s->d_sphere_bounds ? s->d_sphere_bounds + s->sphere_bounds_doubles : nullptr;
    std::memcpy(w.group, s->group, sizeof w.group);
    w.shade = s->d_shade;
    w.n = (int32_t)s->n;
    w.bvh = s->d_bvh;
    w.bvh_root = s->bvh_root;
    w.bvh_origin_limit = s->bvh_origin_limit;
    for (int i = 0; i < 3; ++i) {
        w.light_pos[i] = s->light_pos[i];
        w.light_int[i] = s->light_int[i];
    }
    w.light_reach = ((std::fabs(s->light_pos[0]) + std::fabs(s->light_pos[1])) + std::fabs(s->light_pos[2])) + 1.0;
    return w;
}

void tile_counts(uint64_t hsize, uint64_t vsize, uint32_t rank, uint32_t ranks, uint32_t* tiles_x,
                 uint32_t* tiles_total, uint64_t* local_tiles) {
    const uint64_t tx = (hsize + RTC_TILE_W - 1) / RTC_TILE_W, ty = (vsize + RTC_TILE_H - 1) / RTC_TILE_H;
    const uint64_t total = tx * ty;
    *tiles_x = (uint32_t)tx;
    *tiles_total = (uint32_t)total;
    *local_tiles = total > rank ? (total - rank + ranks - 1) / ranks : 0;
}

int check_camera(const RtcCameraDesc* c) {
    if (!c) return fail(RTC_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (c->hsize == 0 || c->vsize == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "camera hsize/vsize must be non-zero");
    const uint64_t tx = (c->hsize + RTC_TILE_W - 1) / RTC_TILE_W, ty = (c->vsize + RTC_TILE_H - 1) / RTC_TILE_H;
    if (c->hsize > (1u << 20) || c->vsize > (1u << 20) || tx * ty * RTC_TILE_W * RTC_TILE_H > (1ull << 31))
        return fail(RTC_ERR_TOO_LARGE, "canvas too large (at most 2^31 padded pixels)");
    return RTC_OK;
}

// Fill stats from the device counters (after the stream has been synchronised).
int read_stats(RtcScene* s, const uint64_t* d_stats, RtcStats* out, uint64_t pixels, float ms) {
    uint64_t h[9];
    HIP_TRY(hipMemcpy(h, d_stats, sizeof h, hipMemcpyDeviceToHost));
    out->radiance_rays = h[0];
    out->shadow_rays = h[1];
    out->sphere_tests = (h[0] + h[1]) * s->n_spheres;
    out->plane_tests = (h[0] + h[1]) * s->n_planes;
    out->sphere_hits = h[2];
    out->plane_hits = h[3];
    out->shaded_hits = h[4];
    out->sphere_bound_skips = h[5];
    out->plane_side_skips = h[6];
    out->cube_tests = (h[0] + h[1]) * s->n_cubes;
    out->cube_hits = h[7];
    out->sphere_prefilter_skips = h[8];
    out->pixels = pixels;
    out->kernel_ms = ms;
    return RTC_OK;
}

// Upper bound of the largest singular value of the 3x3 part of a row-major matrix with row
// stride `stride`: exact for a diagonal matrix, else min(Frobenius, sqrt(max column sum * max
// row sum)).
double stretch_bound(const double* m, int stride) {
    const double* r0 = m;
    const double* r1 = m + stride;
    const double* r2 = m + 2 * stride;
    if (r0[1] == 0 && r0[2] == 0 && r1[0] == 0 && r1[2] == 0 && r2[0] == 0 && r2[1] == 0)
        return std::fmax(std::fabs(r0[0]), std::fmax(std::fabs(r1[1]), std::fabs(r2[2])));
    double fro = 0, n1 = 0, ninf = 0;
    for (int a = 0; a < 3; ++a) {
        const double* row = m + a * stride;
        fro += row[0] * row[0] + row[1] * row[1] + row[2] * row[2];
        ninf = std::fmax(ninf, std::fabs(row[0]) + std::fabs(row[1]) + std::fabs(row[2]));
        n1 = std::fmax(n1, std::fabs(r0[a]) + std::fabs(r1[a]) + std::fabs(r2[a]));
    }
    return std::fmin(std::sqrt(fro), std::sqrt(n1 * ninf)) * (1.0 + 0x1p-40);
}

// World-space bounding sphere of a sphere shape for the sweep's early-out: out[0..2] centre,
// out[3] squared radius plus the error margin, divided by 1 - 2^-27.  The sweep skips a (ray, sphere) pair when
//     (|c - o|^2 (1 - 2^-27) - bound) |d|^2 > ((c - o).d)^2 ,   evaluated as (|c - o|^2 - out[3]) (|d|^2 (1 - 2^-27)),
// i.e. when the line passes the bounding sphere by more than 2^-28 (|o|^2 + |c|^2 + r^2) in squared
// distance (|o|^2 <= 2|c - o|^2 + 2|c|^2).  In exact arithmetic on the stored inverse the
// reference's discriminant is 4|ld|^2 (1 - (distance/r')^2) < 0 there (r' <= r the extent of the
// ellipsoid along the line's normal).  Its floating-point evaluation, the test's own rounding
// and the difference between the forward transform's centre and the stored inverse's move that
// squared distance by at most ~50 cond^2 2^-53 (|o|^2 + |c|^2 + r^2) with cond = stretch(transform)
// * stretch(inverse); the margin covers cond^2 up to 2^12 with a factor 2^8 to spare, and
// shapes beyond that (or with non-finite numbers) get no bound.
void sphere_bound(const RtcShapeDesc& s, const double* inv12, double* out) {
    out[0] = out[1] = out[2] = 0.0;
    out[3] = std::numeric_limits<double>::infinity();
    if (s.geo != RTC_GEO_SPHERE) return;
    const double* t = s.transform;
    const double r = stretch_bound(t, 4), rinv = stretch_bound(inv12, 4);
    const double cond2 = (r * rinv) * (r * rinv);
    const double cc = t[3] * t[3] + t[7] * t[7] + t[11] * t[11];
    const double r2 = r * r * (1.0 + 0x1p-30);
    const double bound = r2 + 0x1p-28 * (3.0 * cc + r2) * (1.0 + 0x1p-30);
    if (!(cond2 < 4096.0) || !std::isfinite(bound) || !std::isfinite(cc)) return;
    out[0] = t[3];
    out[1] = t[7];
    out[2] = t[11];
    // stored divided by the test's shrink factor 1 - 2^-27 (the device multiplies |d|^2 by it instead of
    // |c - o|^2), rounded up so that the stored bound is never the smaller one
    out[3] = std::nextafter(bound / (1.0 - 0x1p-27), std::numeric_limits<double>::infinity());
}

// Box around the enlarged bounding sphere of sphere_bound (centre b[0..2], squared radius b[3]):
// a line that misses the box misses that sphere, and the sweep would dismiss the pair anyway.
// The box is grown once more, by 2^-20 of its size and position and by 2^-30 of the extent of
// the whole scene; the slab test that is run against it rounds coordinates of at most 2^10
// extents (rays from farther out are not pruned) to 2^-53, i.e. 2^-43 extents.  An unbounded
// sphere gets an infinite box.
RtcScene::SphereBox sphere_box(const double* b, int32_t pos, double scene_extent) {
    RtcScene::SphereBox box;
    box.pos = pos;
    const double r = std::sqrt(b[3]);
    for (int a = 0; a < 3; ++a) {
        const double pad = std::isfinite(r) ? r + 0x1p-20 * (r + std::fabs(b[a])) + 0x1p-30 * scene_extent : r;
        box.lo[a] = b[a] - pad;
        box.hi[a] = b[a] + pad;
    }
    return box;
}

// Hierarchy over items[begin, end); returns the child code of the subtree.  Split by the surface
// area heuristic over 16 bins of the centres along each axis; where that finds no split (all
// centres in one bin) or the tree is already kMaxSahDepth deep, by the median along the widest
// axis, which halves the range, so the depth stays below kMaxSahDepth + 21 < the traversal stack.
constexpr int kMaxSahDepth = 40;
inline double half_area(const double lo[3], const double hi[3]) {
    const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + y * z + z * x;
}
int32_t build_bvh(std::vector<RtcScene::SphereBox>& items, size_t begin, size_t end, std::vector<rtc::BvhNode>& nodes,
                  double lo[3], double hi[3], int depth = 0) {
    if (end - begin == 1) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = items[begin].lo[a];
            hi[a] = items[begin].hi[a];
        }
        return ~items[begin].pos;
    }
    // an unbounded sphere has an infinite box; its centre is taken as 0 so that the keys below
    // stay totally ordered (-inf + inf would be NaN)
    auto centre = [](const RtcScene::SphereBox& x, int a) {
        const double c = 0.5 * x.lo[a] + 0.5 * x.hi[a];
        return c == c ? c : 0.0;
    };
    double cmin[3] = {1e300, 1e300, 1e300}, cmax[3] = {-1e300, -1e300, -1e300};
    bool finite = true;
    for (size_t k = begin; k < end; ++k)
        for (int a = 0; a < 3; ++a) {
            const double c = centre(items[k], a);
            finite = finite && std::isfinite(items[k].lo[a]) && std::isfinite(items[k].hi[a]);
            if (c < cmin[a]) cmin[a] = c;
            if (c > cmax[a]) cmax[a] = c;
        }
    size_t mid = 0;
    bool split = false;
    if (finite && depth < kMaxSahDepth && end - begin > 2) {
        constexpr int kBins = 16;
        double best = 1e300;
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
            const double width = cmax[a] - cmin[a];
            if (!(width > 0)) continue;
            size_t count[kBins] = {0};
            double blo[kBins][3], bhi[kBins][3];
            for (int b = 0; b < kBins; ++b)
                for (int c = 0; c < 3; ++c) blo[b][c] = 1e300, bhi[b][c] = -1e300;
            for (size_t k = begin; k < end; ++k) {
                int b = (int)((centre(items[k], a) - cmin[a]) / width * kBins);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                ++count[b];
                for (int c = 0; c < 3; ++c) {
                    blo[b][c] = std::fmin(blo[b][c], items[k].lo[c]);
                    bhi[b][c] = std::fmax(bhi[b][c], items[k].hi[c]);
                }
            }
            // cost of splitting after bin b: area(left) * n(left) + area(right) * n(right)
            double rlo[kBins][3], rhi[kBins][3];
            size_t rcount[kBins];
            double accl[3] = {1e300, 1e300, 1e300}, acch[3] = {-1e300, -1e300, -1e300};
            size_t n = 0;
            for (int b = kBins - 1; b >= 0; --b) {
                for (int c = 0; c < 3; ++c) {
                    accl[c] = std::fmin(accl[c], blo[b][c]);
                    acch[c] = std::fmax(acch[c], bhi[b][c]);
                    rlo[b][c] = accl[c];
                    rhi[b][c] = acch[c];
                }
                n += count[b];
                rcount[b] = n;
            }
            double ll[3] = {1e300, 1e300, 1e300}, lh[3] = {-1e300, -1e300, -1e300};
            size_t ln = 0;
            for (int b = 0; b + 1 < kBins; ++b) {
                for (int c = 0; c < 3; ++c) {
                    ll[c] = std::fmin(ll[c], blo[b][c]);
                    lh[c] = std::fmax(lh[c], bhi[b][c]);
                }
                ln += count[b];
                if (ln == 0 || rcount[b + 1] == 0) continue;
                const double cost = half_area(ll, lh) * (double)ln + half_area(rlo[b + 1], rhi[b + 1]) * (double)rcount[b + 1];
                if (cost < best) {
                    best = cost;
                    best_axis = a;
                    best_bin = b;
                }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const double width = cmax[a] - cmin[a];
            auto it = std::partition(items.begin() + begin, items.begin() + end, [&](const RtcScene::SphereBox& x) {
                int b = (int)((centre(x, a) - cmin[a]) / width * kBins);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            mid = (size_t)(it - items.begin());
            split = mid > begin && mid < end;
        }
    }
    if (!split) {
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
        mid = begin + (end - begin) / 2;
        std::nth_element(items.begin() + begin, items.begin() + mid, items.begin() + end,
                         [&](const RtcScene::SphereBox& x, const RtcScene::SphereBox& y) {
                             const double cx = centre(x, axis), cy = centre(y, axis);
                             return cx < cy || (cx == cy && x.pos < y.pos);
                         });
    }
    const size_t index = nodes.size();
    nodes.emplace_back();
    double l[2][3], h[2][3];
    const int32_t c0 = build_bvh(items, begin, mid, nodes, l[0], h[0], depth + 1);
    const int32_t c1 = build_bvh(items, mid, end, nodes, l[1], h[1], depth + 1);
    rtc::BvhNode& n = nodes[index];
    std::memset(&n, 0, sizeof n);
    n.child[0] = c0;
    n.child[1] = c1;
    for (int k = 0; k < 2; ++k)
        for (int a = 0; a < 3; ++a) {
            n.lo[k][a] = l[k][a];
            n.hi[k][a] = h[k][a];
        }
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::fmin(l[0][a], l[1][a]);
        hi[a] = std::fmax(h[0][a], h[1][a]);
    }
    return (int32_t)index;
}

// Largest factor by which the 3x3 part of a row-major 4x4 stretches a vector (its largest
// singular value), by power iteration on A^T A; scheduling heuristics only.
double largest_stretch(const double* m) {
    double v[3] = {0.6, 0.55, 0.58}, norm = 0;
    for (int it = 0; it < 24; ++it) {
        double av[3], w[3];
        for (int r = 0; r < 3; ++r) av[r] = m[r * 4] * v[0] + m[r * 4 + 1] * v[1] + m[r * 4 + 2] * v[2];
        for (int c = 0; c < 3; ++c) w[c] = m[c] * av[0] + m[4 + c] * av[1] + m[8 + c] * av[2];
        norm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (!(norm > 0) || !std::isfinite(norm)) return 0;
        for (int c = 0; c < 3; ++c) v[c] = w[c] / norm;
    }
    return std::sqrt(norm);
}

// The persistent kernel hands out tiles from a queue; whichever tiles come last decide how
// long the GPU runs partly empty at the end, because one pixel is a serial chain of up to 2^depth
// nodes.  Pixels that show a transparent shape cost several times the average (each such hit has two
// subtrees), so the rows of tiles that can show one are queued first, the fullest first, and
// the remaining rows follow by their distance from those, nearest first.  Rows stay whole and
// are scanned left to right: measured on the 4K reference frame, a queue that keeps a row's
// tiles together is faster than one that scatters them, and queueing the costly rows first is
// worth about 7 % of the frame time over a top-to-bottom scan.
// This only orders the work: every pixel is computed by the same arithmetic whatever its
// place in the queue.  This function is the host arithmetic (queue slot -> local tile index);
// `order` comes back empty when the natural order is kept.
struct PixelBox {
    double x0, x1, y0, y1;  // extent on the canvas, in pixels, of where a transparent shape can show
};
int compute_tile_order(const RtcScene* s, const RtcCameraDesc* cam, const rtc::DevJob* job, uint64_t local_tiles,
                       std::vector<uint32_t>& order, std::vector<PixelBox>* pixel_boxes = nullptr) {
    order.clear();
    if (pixel_boxes) pixel_boxes->clear();
    if (local_tiles < 2 || s->costly_everywhere || s->costly.empty()) return RTC_OK;
    const uint32_t tiles_x = job->tiles_x, ranks = job->tile_ranks, rank = job->tile_rank;
    const uint32_t tiles_y = job->tiles_total / tiles_x;
    std::vector<uint64_t> row_score(tiles_y, 0);
    struct TileBox { uint32_t tx0, tx1, ty0, ty1; };
    std::vector<TileBox> boxes;
    const double* T = cam->transform;  // world -> camera; the camera looks down -z (camera.zig:48-72)
    const double stretch = largest_stretch(T);
    bool any = false;
    for (const RtcScene::Costly& k : s->costly) {
        const double xc = T[0] * k.c[0] + T[1] * k.c[1] + T[2] * k.c[2] + T[3];
        const double yc = T[4] * k.c[0] + T[5] * k.c[1] + T[6] * k.c[2] + T[7];
        const double dist = -(T[8] * k.c[0] + T[9] * k.c[1] + T[10] * k.c[2] + T[11]);
        const double r = k.r * stretch * 1.05;
        if (!(dist + r > 0)) continue;  // entirely behind the eye
        double x0 = -1e300, x1 = 1e300, y0 = -1e300, y1 = 1e300;  // extent on the plane z = -1
        if (dist - r > 1e-9) {
            const double near = dist - r, far = dist + r;
            x0 = std::fmin((xc - r) / near, (xc - r) / far);
            x1 = std::fmax((xc + r) / near, (xc + r) / far);
            y0 = std::fmin((yc - r) / near, (yc - r) / far);
            y1 = std::fmax((yc + r) / near, (yc + r) / far);
        }
        // pixel = (half - offset) / pixel_size, larger coordinate = smaller pixel index
        const double px0 = (cam->half_width - x1) / cam->pixel_size, px1 = (cam->half_width - x0) / cam->pixel_size;
        const double py0 = (cam->half_height - y1) / cam->pixel_size, py1 = (cam->half_height - y0) / cam->pixel_size;
        if (!(px1 >= 0) || !(py1 >= 0) || !(px0 < (double)cam->hsize) || !(py0 < (double)cam->vsize)) continue;
        const uint32_t tx0 = px0 > 0 ? (uint32_t)(px0 / RTC_TILE_W) : 0;
        const uint32_t ty0 = py0 > 0 ? (uint32_t)(py0 / RTC_TILE_H) : 0;
        const uint32_t tx1 = px1 < (double)cam->hsize ? (uint32_t)(px1 / RTC_TILE_W) : tiles_x - 1;
        const uint32_t ty1 = py1 < (double)cam->vsize ? (uint32_t)(py1 / RTC_TILE_H) : tiles_y - 1;
        for (uint32_t ty = ty0; ty <= ty1 && ty < tiles_y; ++ty) row_score[ty] += (uint64_t)(tx1 - tx0 + 1);
        boxes.push_back({tx0, tx1, ty0, ty1});
        if (pixel_boxes) pixel_boxes->push_back({px0, px1, py0, py1});
        any = true;
    }
    if (!any) return RTC_OK;
    // rank of every row of tiles in the queue
    std::vector<uint32_t> dist_to_costly(tiles_y, tiles_y);
    {
        uint32_t d = tiles_y;
        for (uint32_t y = 0; y < tiles_y; ++y) {
            d = row_score[y] ? 0 : (d < tiles_y ? d + 1 : d);
            dist_to_costly[y] = d;
        }
        d = tiles_y;
        for (uint32_t y = tiles_y; y-- > 0;) {
            d = row_score[y] ? 0 : (d < tiles_y ? d + 1 : d);
            dist_to_costly[y] = std::min(dist_to_costly[y], d);
        }
    }
    std::vector<uint32_t> rows(tiles_y);
    for (uint32_t y = 0; y < tiles_y; ++y) rows[y] = y;
    std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) {
        if ((row_score[a] != 0) != (row_score[b] != 0)) return row_score[a] != 0;
        if (row_score[a] != row_score[b]) return row_score[a] > row_score[b];
        return dist_to_costly[a] < dist_to_costly[b];
    });
    std::vector<uint32_t> rank_of_row(tiles_y);
    for (uint32_t i = 0; i < tiles_y; ++i) rank_of_row[rows[i]] = i;
    // local tiles bucketed by the rank of their row, left to right inside a row
    std::vector<uint64_t> start(tiles_y + 1, 0);
    for (uint64_t k = 0; k < local_tiles; ++k) ++start[rank_of_row[(k * ranks + rank) / tiles_x] + 1];
    for (uint32_t i = 0; i < tiles_y; ++i) start[i + 1] += start[i];
    order.resize(local_tiles);
    for (uint64_t k = 0; k < local_tiles; ++k) order[start[rank_of_row[(k * ranks + rank) / tiles_x]]++] = (uint32_t)k;
    // A block of 64 such pixels keeps its wave for some 60 node iterations, about 0.4 ms: every one of them should be
    // some wave's first block, so the tiles a transparent shape can show go to the front of the queue (in the row order
    // above): the 4K reference frame 1.89 -> 1.84 ms on the first frame of a camera (the strip table of a camera that
    // repeats, which does the same at the granularity of blocks, gives 1.83 ms).  On large frames the rows stay whole
    // (7680x4320: no difference now, 10.08 against 10.31 ms when the pixels of a wave were dealt lane by lane).
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const uint64_t resident_waves = (uint64_t)cus * 16;
    if (local_tiles * (RTC_TILE_W * RTC_TILE_H / 64) <= 48 * resident_waves) {
        auto marked = [&](uint32_t k) {
            const uint32_t tx = (uint32_t)(((uint64_t)k * ranks + rank) % tiles_x), ty = (uint32_t)(((uint64_t)k * ranks + rank) / tiles_x);
            for (const TileBox& b : boxes)
                if (tx >= b.tx0 && tx <= b.tx1 && ty >= b.ty0 && ty <= b.ty1) return true;
            return false;
        };
        std::stable_partition(order.begin(), order.end(), marked);
    }
    return RTC_OK;
}

// The queue at a finer grain: strips of 8 x 2 pixels, four to a chunk of 64 queue positions (rtc_refill.inc).
// A block of 64 pixels that all show a transparent shape keeps its wave for some 45 node iterations, with every lane
// busy on a long chain of its own: on a share of a frame that the GPU finishes in little more than that, those waves
// end long after the others.  Here such blocks are dealt a strip at a time, each together with three strips of an
// ordinary block: a wave then holds 16 costly pixels and 48 cheap ones, the costly work is spread over four times as
// many waves from the start, and once the queue is empty the lanes the cheap pixels leave idle take over the parked
// subtrees of the costly ones.  Only the schedule changes: a pixel's arithmetic does not depend on its wave.
// An entry: low word = (first pixel's x / 8) | (first pixel's y / 2) << 16, or 0xffffffff for a strip outside the
// canvas; high word = index of the strip's first pixel in the output (canvas, or the rank's dense tile buffer).
constexpr uint32_t kStripsPerTile = RTC_TILE_W * RTC_TILE_H / 16;
bool strips_usable(const RtcCameraDesc* cam) { return cam->hsize < (1u << 19) && cam->vsize < (1u << 17); }
void build_strip_order(const RtcCameraDesc* cam, const rtc::DevJob* job, uint64_t local_tiles, const std::vector<uint32_t>& tile_order,
                       const std::vector<PixelBox>& boxes, int variant, std::vector<uint64_t>& strips) {
    const uint32_t tiles_x = job->tiles_x, ranks = job->tile_ranks, rank = job->tile_rank;
    const uint32_t hsize = (uint32_t)cam->hsize, vsize = (uint32_t)cam->vsize;
    std::vector<uint64_t> costly, cheap, outside;
    costly.reserve(local_tiles * 4);
    cheap.reserve(local_tiles * kStripsPerTile);
    for (uint64_t slot = 0; slot < local_tiles; ++slot) {
        const uint64_t lt = tile_order.empty() ? slot : tile_order[slot];
        const uint64_t gt = lt * ranks + rank;
        const uint32_t tx0 = (uint32_t)(gt % tiles_x) * RTC_TILE_W, ty0 = (uint32_t)(gt / tiles_x) * RTC_TILE_H;
        for (uint32_t blk = 0; blk < RTC_TILE_W / 8; ++blk) {
            const uint32_t bx0 = tx0 + blk * 8;
            // does the 8 x 8 block overlap the extent of a transparent shape?
            bool marked = false;
            for (const PixelBox& b : boxes)
                marked = marked || (b.x1 >= (double)bx0 && b.x0 < (double)(bx0 + 8) && b.y1 >= (double)ty0 && b.y0 < (double)(ty0 + 8));
            for (uint32_t rp = 0; rp < RTC_TILE_H / 2; ++rp) {
                const uint32_t py0 = ty0 + rp * 2;
                const uint64_t out0 = job->out_row_major ? (uint64_t)py0 * hsize + bx0
                                                         : lt * (RTC_TILE_W * RTC_TILE_H) + (uint64_t)rp * 2 * RTC_TILE_W + blk * 8;
                const bool inside = gt < job->tiles_total && bx0 < hsize && py0 < vsize;
                const uint64_t entry = inside ? ((out0 << 32) | (uint64_t)((bx0 / 8) | ((py0 / 2) << 16))) : 0xffffffffull;
                (inside ? (marked ? costly : cheap) : outside).push_back(entry);
            }
        }
    }
    strips.clear();
    strips.reserve(local_tiles * kStripsPerTile);
    size_t c = 0, p = 0;
    // variant 1: one costly strip with three cheap ones while both last ...
    while (variant == 1 && c < costly.size() && p + 3 <= cheap.size()) {
        strips.push_back(costly[c++]);
        strips.push_back(cheap[p++]);
        strips.push_back(cheap[p++]);
        strips.push_back(cheap[p++]);
    }
    // ... then what is left of either (whole blocks again, in the order above), and the strips outside the canvas last
    while (c < costly.size()) strips.push_back(costly[c++]);
    while (p < cheap.size()) strips.push_back(cheap[p++]);
    for (uint64_t e : outside) strips.push_back(e);
}

// How the queue of this launch is dealt (build_strip_order): 0 = whole tiles in the order of compute_tile_order,
// 1 = in strips, one strip of a costly block with three of an ordinary one, 2 = in strips, whole costly blocks first.
// Measured on the Ch. 11 scene (scripts/partition_timing.py, blocks of the share per wave the GPU holds):
//   up to 2 blocks per wave (1080p over 4 or 8 ranks)   1: 0.407 -> 0.387 ms, 0.357 -> 0.279 ms (2: 0.394 / 0.355)
//   4 to 32 (4K over 1 to 8 ranks, 1080p)               2: 2.23 -> 2.17, 1.14 -> 1.10, 0.73 -> 0.68, 0.475 -> 0.449 ms
//                                                          (1 mixes two kinds of pixel in every wave: 2.13 -> 2.69 ms)
//   126 (7680x4320)                                      0: 7.42 ms against 7.62 ms with 2
// RTC_STRIP_QUEUE=0/1/2 in the environment forces a variant (a developer switch for timing).
int strip_queue_variant(const rtc::DevJob* job, uint64_t local_tiles) {
    static const int forced = [] {
        const char* e = std::getenv("RTC_STRIP_QUEUE");
        return e && *e ? std::atoi(e) : -1;
    }();
    if (forced >= 0) return forced;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return 0;
    (void)job;
    const uint64_t resident_waves = (uint64_t)cus * 16, blocks = local_tiles * (RTC_TILE_W * RTC_TILE_H / 64);
    if (blocks <= 2 * resident_waves) return 1;
    if (blocks <= 48 * resident_waves) return 2;
    return 0;
}

// The draw threshold (DevJob::draw_threshold) of a queue that mixes costly and ordinary strips: half a wave - the idle
// lanes of the ordinary strips are meant to take further work while the costly strip's lanes are busy.  Measured on
// the 1080p reference frame over 4 / 8 ranks: 0.384 / 0.283 ms without a threshold, 0.374 / 0.276 at 16, 0.376 / 0.256 at
// 32, 0.487 / 0.389 at 64.  RTC_MIXED_THRESHOLD in the environment sets another (a developer switch for timing).
uint32_t mixed_strips_threshold() {
    static const uint32_t t = [] {
        const char* e = std::getenv("RTC_MIXED_THRESHOLD");
        return (uint32_t)(e && *e ? std::atoi(e) : 32);
    }();
    return t;
}

// The plan in the scene's own buffer (grown with hipMalloc when needed), cached per camera and partition.
int plan_tile_order(RtcScene* s, const RtcCameraDesc* cam, rtc::DevJob* job, uint64_t local_tiles,
                    hipStream_t stream) {
    job->tile_order = nullptr;
    job->strip_order = nullptr;
    if (local_tiles < 2 || s->costly_everywhere || s->costly.empty()) return RTC_OK;
    // (block stores need every pixel of a block in one wave: no strips then)
    const int strips = (strips_usable(cam) && !(s->canvas_store == RTC_CANVAS_STORE_BLOCKS && job->out_row_major))
                           ? strip_queue_variant(job, local_tiles)
                           : 0;
    // A new camera gets the order of the tiles (little host arithmetic, 65 KB at 4K); the strip table (2 ms of host
    // work and 4 MB at 4K) is made when the same camera comes a second time, so that a moving camera never pays for it
    // and a still one pays once.
    // (a cached strip table is of no use to a launch that must not have one: block stores selected meanwhile)
    const bool same = s->plan_valid && s->plan_rank == job->tile_rank && s->plan_ranks == job->tile_ranks &&
                      s->plan_row_major == (job->out_row_major != 0) && (strips != 0 || !s->plan_has_strips) &&
                      std::memcmp(&s->plan_camera, cam, sizeof *cam) == 0;
    const bool upgrade = same && strips != 0 && s->plan_strips != strips;
    if (same && !upgrade) {
        job->tile_order = s->plan_identity ? nullptr : s->d_tile_order;
        job->strip_order = s->plan_has_strips ? s->d_strip_order : nullptr;
        if (job->strip_order) job->tile_order = nullptr;
        if (job->strip_order && s->plan_strips == 1) job->draw_threshold = mixed_strips_threshold();
        return RTC_OK;
    }
    std::vector<uint32_t> order;
    std::vector<PixelBox> boxes;
    if (int rc = compute_tile_order(s, cam, job, local_tiles, order, &boxes)) return rc;
    std::vector<uint64_t> strip_table;
    if (upgrade && !boxes.empty()) build_strip_order(cam, job, local_tiles, order, boxes, strips, strip_table);
    // the cache describes a plan only once that plan is on the device (or needs no upload)
    s->plan_valid = false;
    s->plan_camera = *cam;
    s->plan_rank = job->tile_rank;
    s->plan_ranks = job->tile_ranks;
    s->plan_strips = upgrade ? strips : 0;  // (the variant the plan was made for once the camera had repeated)
    s->plan_row_major = job->out_row_major != 0;
    s->plan_identity = order.empty();
    s->plan_has_strips = !strip_table.empty();
    if (order.empty() && strip_table.empty()) {
        s->plan_valid = true;
        return RTC_OK;
    }
    // an earlier launch on this stream may still be reading the previous plan
    HIP_TRY(hipStreamSynchronize(stream));
    if (!strip_table.empty()) {
        if (s->strip_order_capacity < strip_table.size()) {
            if (s->d_strip_order) (void)hipFree(s->d_strip_order);
            s->d_strip_order = nullptr;
            s->strip_order_capacity = 0;
            HIP_TRY(hipMalloc((void**)&s->d_strip_order, strip_table.size() * sizeof(uint64_t)));
            s->strip_order_capacity = strip_table.size();
        }
        HIP_TRY(hipMemcpy(s->d_strip_order, strip_table.data(), strip_table.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        s->plan_valid = true;
        job->strip_order = s->d_strip_order;
        if (s->plan_strips == 1) job->draw_threshold = mixed_strips_threshold();
        return RTC_OK;
    }
    if (s->tile_order_capacity < local_tiles) {
        if (s->d_tile_order) (void)hipFree(s->d_tile_order);
        s->d_tile_order = nullptr;
        s->tile_order_capacity = 0;
        HIP_TRY(hipMalloc((void**)&s->d_tile_order, local_tiles * sizeof(uint32_t)));
        s->tile_order_capacity = local_tiles;
    }
    HIP_TRY(hipMemcpy(s->d_tile_order, order.data(), local_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
    s->plan_valid = true;
    job->tile_order = s->d_tile_order;
    return RTC_OK;
}

// What the shadow sweep's same-side test needs of a plane (see sweep_shape): the sign and size of
// the light's local y, and the guard factor; {0, 0, inf} switches the test off for the plane.
void plane_light_side(const double* inv12, const double* light_pos, double out[3]) {
    const double m4 = inv12[4], m5 = inv12[5], m6 = inv12[6], m7 = inv12[7];
    const double h = ((m4 * light_pos[0] + m5 * light_pos[1]) + m6 * light_pos[2]) + m7;
    const double guard = 0x1p-32 * std::fmax((std::fabs(m4) + std::fabs(m5)) + std::fabs(m6), std::fabs(m7));
    out[0] = h > 0 ? 1.0 : (h < 0 ? -1.0 : 0.0);
    out[1] = std::fabs(h);
    out[2] = guard;
    if (!std::isfinite(h) || !std::isfinite(guard)) {
        out[0] = 0.0;
        out[1] = 0.0;
        out[2] = std::numeric_limits<double>::infinity();
    }
}

// The bounding-sphere test of sphere_bound for a shadow ray, anchored at the light instead of the
// ray's origin.  The ray o + t d, d = normalize(light - o), dist = |light - o|, passes the light
// within 2^-51 dist, so the distance between its line and the centre c is the distance of c from
// the line through the light with direction d, up to that much.  With lc = c - light,
//     |c - o|^2 <= 2 |lc|^2 + 2 dist^2   and   |d|^2 = 1 +- 2^-50 ,
// the condition of sphere_bound follows from
//     |lc|^2 (1 - 2^-25) - bound[3] (1 + 2^-40) - 2^-25 dist^2 > (lc . d)^2 ,
// whose left side is out[3] minus a per-ray term: three multiply-adds per sphere instead of the
// origin's nine.  out[0..2] = lc.
// (`eps` is the shrink of the test the bound belongs to: 2^-27 for a sphere's own, more for a block's; the
// per-ray term is then 4 eps dist^2.)
void sphere_from_light(const double* bound, const double* light_pos, double out[4], double eps = 0x1p-27) {
    double l2 = 0;
    for (int a = 0; a < 3; ++a) {
        out[a] = bound[a] - light_pos[a];
        l2 += out[a] * out[a];
    }
    out[3] = l2 * (1.0 - 4.0 * eps) - bound[3] * (1.0 + 0x1p-40);
    if (!std::isfinite(out[0]) || !std::isfinite(out[1]) || !std::isfinite(out[2]) || !(out[3] == out[3]))
        out[3] = -std::numeric_limits<double>::infinity();
}

// A sphere around the bounding spheres of a block of spheres, for the streamed sweep's block test: a ray that
// passes it by the test's margin passes every member by the member's own margin, i.e. the block test may stand
// in for the members' tests.  A member's test (sphere_bound) dismisses when
//     D_i^2 > b_i + eps |c_i - o|^2 ,   eps = 2^-27 ,
// D_i the distance of the ray's line from the member's centre c_i, b_i its enlarged squared radius.  The block's
// test has the same form with centre C, squared radius R^2 and a shrink eps_B of its own.  With delta_i = |c_i - C|:
// D_i >= D - delta_i and |c_i - o|^2 <= 2 |C - o|^2 + 2 delta_i^2, so it is enough that
//     (D - delta_i)^2 > b_i + 2 eps delta_i^2 + 2 eps |C - o|^2 .
// For D^2 > R^2 + z, z >= 0 and R >= delta:  (D - delta)^2 >= (R - delta)^2 + z (R - delta) / R
// (expand both sides: the difference is delta (z / R - 2 (sqrt(R^2 + z) - R)) >= 0).  So with
//     rho_i = sqrt(b_i + 2 eps delta_i^2) ,   R >= delta_i + rho_i for every member ,   eps_B >= 2 eps R / rho_i ,
// the block's test D^2 > R^2 + eps_B |C - o|^2 gives (D - delta_i)^2 > rho_i^2 + 2 eps |C - o|^2, which is the
// line above.  Roundings of this function are covered by the factors 1 + 2^-40; the centre starts as the mean of
// the members' centres and is moved towards the centre of their smallest enclosing ball (any point would do).  A member without a bound leaves the block without one.
struct BlockBound {
    bool bounded = false;
    double c[3] = {0, 0, 0};
    double r2 = 0;   // R^2
    double eps = 0;  // eps_B
};
BlockBound sphere_block_bound(const RtcScene::SphereBound* const* members, size_t n) {
    BlockBound out;
    if (n == 0) return out;
    double c[3] = {0, 0, 0};
    for (size_t k = 0; k < n; ++k) {
        if (!members[k] || !std::isfinite(members[k]->b[3])) return out;
        for (int a = 0; a < 3; ++a) c[a] += members[k]->b[a] / (double)n;
    }
    const double eps = 0x1p-27, up = 1.0 + 0x1p-40;
    {
        // a better centre than the mean: a few steps towards the member that reaches farthest, each shorter than
        // the one before (the iteration of Badoiu and Clarkson for the smallest enclosing ball); the centre that
        // gave the smallest radius is kept.  Only the quality of the bound depends on it.
        auto reach = [&](const double* at, size_t* farthest) {
            double worst = -1.0;
            for (size_t k = 0; k < n; ++k) {
                double d2 = 0;
                for (int a = 0; a < 3; ++a) d2 += (members[k]->b[a] - at[a]) * (members[k]->b[a] - at[a]);
                const double r = std::sqrt(d2) + std::sqrt(members[k]->b[3]);
                if (r > worst) worst = r, *farthest = k;
            }
            return worst;
        };
        double at[3] = {c[0], c[1], c[2]};
        size_t far = 0;
        double best = reach(at, &far);
        for (int step = 1; step <= 48 && std::isfinite(best); ++step) {
            for (int a = 0; a < 3; ++a) at[a] += (members[far]->b[a] - at[a]) / (double)(step + 1);
            const double r = reach(at, &far);
            if (r < best) {
                best = r;
                for (int a = 0; a < 3; ++a) c[a] = at[a];
            }
        }
    }
    double R = 0, rho_min = std::numeric_limits<double>::infinity();
    for (size_t k = 0; k < n; ++k) {
        double d2 = 0;
        for (int a = 0; a < 3; ++a) d2 += (members[k]->b[a] - c[a]) * (members[k]->b[a] - c[a]);
        // (b[3] is the member's bound divided by 1 - 2^-27: not smaller than b_i)
        const double rho = std::sqrt((members[k]->b[3] + 2.0 * eps * d2 * up) * up) * up;
        R = std::fmax(R, (std::sqrt(d2 * up) * up + rho) * up);
        rho_min = std::fmin(rho_min, rho);
    }
    if (!(rho_min > 0) || !std::isfinite(R)) return out;
    out.bounded = true;
    for (int a = 0; a < 3; ++a) out.c[a] = c[a];
    out.r2 = R * R * up;
    out.eps = 2.0 * eps * (R / rho_min) * up * up;
    if (!std::isfinite(out.r2) || !std::isfinite(out.eps)) out.bounded = false;
    return out;
}

// Single-precision operands of the packed pre-test that runs ahead of the bounding tests in the streamed sweep
// of large worlds (pair_clear, rtc_kernels.hip).  The pre-test may only dismiss a (ray, sphere) pair when the
// double-precision test's inequality holds in exact arithmetic on the double operands,
//     radiance:  (A - bound) D k > B^2        A = |c - o|^2, B = (c - o).d, D = |d|^2, k = 1 - 2^-27
//     shadow:    lk - 2^-25 dist^2 > (lc.d)^2
// so every rounding of the float evaluation has to be covered by margins (u = 2^-24):
//  * c, o, d are rounded to float and c - o is rounded again: the computed difference vector is off by at most
//    2.01 u s, s^2 = sum (|c_j| + |o_j|)^2 <= 2 (|c|^2 + |o|^2); its squared length, three more rounded
//    operations, is off by at most 7.2 u s^2 < 2^-20 (|c|^2 + |o|^2); its product with d by beta = 6.1 u s |d|.
//  * B^2 <= (1 + 2^-19) B~^2 + (1 + 2^19) beta^2 (twice the product of two numbers is at most q times the square
//    of one plus 1/q times the square of the other), and the second term, 2^19 37.2 u^2 s^2 D, is below
//    0.15 2^-20 (|c|^2 + |o|^2) D k.  The split matters: the factor on B~^2 widens the test by that fraction of
//    the squared DISTANCE to the sphere, not of its radius - with 2^-10 a sphere of radius 0.6 at distance 60 was
//    tested as one of radius 2.7.
//  * The two subtractions that form the left side add 4.1 u (|c|^2 + |o|^2) = 0.26 2^-20 (...).  In all at most
//    (0.9 + 0.15 + 0.26) 2^-20 < 2^-19.5 times (|c|^2 + |o|^2) is lost on the left: the sphere's share, 2^-19.5
//    |c|^2, is added to its radius term here (rounded up), the ray's share is subtracted on the device (rounded up).
//  * The factor 1 + 2^-19 and the rounding of the two final products (a factor (1 + u)/(1 - u) < 1 + 2^-22.9)
//    are covered by scaling D k by 1 - 2^-18 (rounded down) on the device: (1 - 2^-22.9)/(1 - 2^-18) > 1 + 2^-19.
// Shadow rays: lc~.d~ is off by at most 5.2 u |lc| (|d| = 1 to 2^-50), so (lc.d)^2 <= (1 + 2^-21) (lc~.d~)^2 +
// (1 + 2^21) 27.1 u^2 |lc|^2, the last term below 2^-22 |lc|^2; the stored term is (lk - 2^-22 |lc|^2)(1 - 2^-19)
// rounded down, which also covers the rounding of the device's subtraction and product ((1 - 2^-23)/(1 - 2^-19) >
// 1 + 2^-21; the per-ray term is subtracted unscaled, which only makes the left side smaller); a term that is
// not positive can dismiss nothing (-inf).
// Terms below 2^-100 are raised to it, so that flushed subnormals (at most 2^-126 per operation) stay covered;
// an overflow gives an infinity or a NaN on the device, and a comparison with either dismisses nothing.
float float_above(double v) {
    float f = (float)v;
    if (!((double)f >= v)) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}
float float_below(double v) {
    float f = (float)v;
    if (!((double)f <= v)) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
void sphere_prefilter_tables(const double bound[4], const double from_light[4], float radiance[4], float shadow[4]) {
    const float inf = std::numeric_limits<float>::infinity();
    const double cc = bound[0] * bound[0] + bound[1] * bound[1] + bound[2] * bound[2];
    for (int k = 0; k < 3; ++k) radiance[k] = (float)bound[k];
    const double r = bound[3] + 0x1.6a09e667f3bcdp-20 * cc;  // 2^-19.5
    radiance[3] = (std::isfinite(r) && std::isfinite(cc)) ? std::fmax(float_above(r), 0x1p-100f) : inf;  // +inf: never dismissed
    const double ll = from_light[0] * from_light[0] + from_light[1] * from_light[1] + from_light[2] * from_light[2];
    for (int k = 0; k < 3; ++k) shadow[k] = (float)from_light[k];
    const double t = (from_light[3] - 0x1p-22 * ll) * (1.0 - 0x1p-19);
    shadow[3] = (std::isfinite(t) && std::isfinite(ll) && t > 0x1p-100) ? float_below(t) : -inf;  // -inf: never dismissed
}

// Device canvas -> the caller's (pageable, usually freshly allocated) buffer.  A plain hipMemcpy
// into such memory runs at 10-17 GB/s on this platform (staging inside the runtime plus the first
// touch of every page) against 57 GB/s into pinned memory.  Large canvases therefore go through a
// pinned staging buffer kept with the scene: the device copies it there slice by slice, and a few
// host threads move finished slices on into the caller's buffer while later slices are in flight.
struct HostSlice {
    size_t begin, len;
};
// the pinned staging buffer, grown when needed; false when no pinned memory is to be had
bool have_staging(RtcScene* s, size_t bytes) {
    if (s->staging_bytes >= bytes) return true;
    if (s->h_staging) (void)hipHostFree(s->h_staging);
    s->h_staging = nullptr;
    s->staging_bytes = 0;
    if (hipHostMalloc(&s->h_staging, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        s->h_staging = nullptr;
        return false;
    }
    s->staging_bytes = bytes;
    return true;
}
int have_slice_events(RtcScene* s, size_t n) {
    while (s->slice_events.size() < n) {
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        s->slice_events.push_back(ev);
    }
    return RTC_OK;
}
// Staging buffer -> the caller's buffer, slice k as soon as slice_events[k] has happened, by this thread and a few
// helpers; returns when every slice is in place (or a wait failed).
int move_slices_to_caller(RtcScene* s, void* dst, const std::vector<HostSlice>& slices) {
    const size_t n_slices = slices.size();
    std::atomic<size_t> next{0};
    std::atomic<int> first_error{(int)hipSuccess};
    const int device = s->device;
    auto mover = [&]() {
        if (hipSetDevice(device) != hipSuccess) return;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n_slices) return;
            const size_t begin = slices[k].begin, len = slices[k].len;
            {
                // While the slice is still on its way: touch its pages in the caller's buffer.  A freshly allocated
                // buffer gets its pages at the first write, which costs more than the move itself (199 MB by one
                // thread: 16.5 ms into fresh memory, 6.8 ms into touched memory); here that happens while the thread
                // would be waiting.  (The buffer is output only: every byte touched is overwritten below.)
                volatile char* d = (volatile char*)dst + begin;
                for (size_t at = 0; at < len; at += 4096) d[at] = 0;
                d[len - 1] = 0;
            }
            const hipError_t e = hipEventSynchronize(s->slice_events[k]);
            if (e != hipSuccess) {
                int expected = (int)hipSuccess;
                first_error.compare_exchange_strong(expected, (int)e);
                return;
            }
            std::memcpy((char*)dst + begin, (const char*)s->h_staging + begin, len);
        }
    };
    static const size_t max_helpers = [] {
        const char* e = std::getenv("RTC_COPY_HELPERS");  // developer switch for timing
        if (e && *e) return (size_t)std::atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        return (size_t)(hw >= 8 ? 7 : (hw >= 4 ? 3 : 1));
    }();
    std::vector<std::thread> helpers;
    const size_t n_helpers = std::min<size_t>(max_helpers, n_slices - 1);
    for (size_t t = 0; t < n_helpers; ++t) helpers.emplace_back(mover);
    mover();
    for (std::thread& t : helpers) t.join();
    if (first_error.load() != (int)hipSuccess) return fail_hip("copy of the canvas to the host", (hipError_t)first_error.load());
    return RTC_OK;
}

int copy_canvas_to_host(RtcScene* s, const void* d_src, void* dst, size_t bytes) {
    // 8 MiB slices for the large canvases (what is left to do once the last slice has arrived is one slice's
    // move); an 8-bit 4K canvas (25 MB) still gets six
    const size_t kSlice = bytes >= (64u << 20) ? (8u << 20) : (4u << 20);
    if (bytes < 4 * kSlice || !have_staging(s, bytes)) {  // (no pinned memory to be had: the plain copy still works)
        HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
        return RTC_OK;
    }
    std::vector<HostSlice> slices;
    for (size_t begin = 0; begin < bytes; begin += kSlice) slices.push_back({begin, std::min(kSlice, bytes - begin)});
    if (int rc = have_slice_events(s, slices.size())) return rc;
    for (size_t k = 0; k < slices.size(); ++k) {
        HIP_TRY(hipMemcpyAsync((char*)s->h_staging + slices[k].begin, (const char*)d_src + slices[k].begin, slices[k].len,
                               hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipEventRecord(s->slice_events[k], nullptr));
    }
    const int rc = move_slices_to_caller(s, dst, slices);
    HIP_TRY(hipStreamSynchronize(nullptr));
    return rc;
}

// Common launch path of rtc_render_device / rtc_render / rtc_color_at.
// `control` is the launch's control block: the scene's own (nullptr) or the caller's.
int run_job(RtcScene* s, rtc::DevJob& job, hipStream_t stream, RtcStats* stats, uint64_t pixels,
            uint32_t* control = nullptr, bool zero_with_kernel = false) {
    if (job.depth > rtc::kMaxDepthLarge) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "depth %d exceeds the supported maximum %d", job.depth, rtc::kMaxDepthLarge);
        return fail(RTC_ERR_TOO_LARGE, buf);
    }
    // calc.zig:829,946 take an i32 `remaining` and spawn no children for remaining <= 0: a negative
    // depth renders like depth 0 (and must not reach the kernel, whose task word holds it unsigned)
    if (job.depth < 0) job.depth = 0;
    // every launch of a scene shares its control block, tile plan and events: the scene's device
    // must be the current one (the caller's stream and output buffer live there too)
    int current_device = -1;
    HIP_TRY(hipGetDevice(&current_device));
    if (current_device != s->device) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "scene was created on device %d but device %d is current", s->device, current_device);
        return fail(RTC_ERR_INVALID_ARGUMENT, buf);
    }
    job.weighted = s->integrator == RTC_INTEGRATOR_WEIGHTED ? 1 : 0;
    // block stores: canvases in row-major layout, post-order integrator, frame kernels (see launch_tables)
    const bool block_stores = s->canvas_store == RTC_CANVAS_STORE_BLOCKS && job.out_row_major && !job.weighted && !stats &&
                              job.rays == nullptr && s->d_flush_stage != nullptr;
    job.flush_stage = block_stores ? s->d_flush_stage : nullptr;
    if (block_stores) job.strip_order = nullptr;  // (a block must stay with one wave; plan_tile_order does not hand one out then)
    job.flush_waves = block_stores ? s->flush_waves : 0u;
    if (job.weighted && job.out_row_major && job.tile_ranks > 1)
        return fail(RTC_ERR_INVALID_ARGUMENT,
                    "the weighted integrator adds to its output with atomics and renders a partition into its own tile "
                    "buffer only (rtc_render_device), not in place into a shared canvas");
    if (!control) control = s->d_counter;
    uint64_t* control_stats = (uint64_t*)((char*)control + rtc::kStatsOffsetBytes);
    job.counter = control;
    job.stats = stats ? control_stats : nullptr;
    if (job.total == 0) {
        if (stats) {
            std::memset(stats, 0, sizeof *stats);
        }
        return RTC_OK;
    }
    // zero the queue heads and the stats words (a launch captured into a graph does it with a kernel node: a
    // captured memset node of this block was seen not to run when the graph was replayed)
    if (zero_with_kernel)
        HIP_TRY(rtc::launch_zero_words(control, rtc::kProbeOffsetBytes / 4, stream));
    else
        HIP_TRY(hipMemsetAsync(control, 0, rtc::kProbeOffsetBytes, stream));
    if (stats) HIP_TRY(hipEventRecord(s->ev_start, stream));
    HIP_TRY(rtc::launch_render(dev_world(s), job, stats != nullptr, stream));
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev_stop, stream));
        HIP_TRY(hipEventSynchronize(s->ev_stop));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
        return read_stats(s, control_stats, stats, pixels, ms);
    }
    return RTC_OK;
}

int fill_camera_job(const RtcCameraDesc* camera, int32_t depth, uint32_t tile_rank, uint32_t tile_ranks,
                    rtc::DevJob* job, uint64_t* local_pixels) {
    if (int rc = check_camera(camera)) return rc;
    if (tile_ranks == 0 || tile_rank >= tile_ranks)
        return fail(RTC_ERR_INVALID_ARGUMENT, "tile_rank must be < tile_ranks and tile_ranks >= 1");
    double cam_inv[16];
    if (int rc = invert_affine(camera->transform, cam_inv, "camera", 0)) return rc;
    std::memcpy(job->cam_inv, cam_inv, 12 * sizeof(double));
    for (int r = 0; r < 3; ++r) {
        const double* m = cam_inv + 4 * r;  // ((m0*x + m1*y) + m2*z) + m3 at (0,0,0), not folded
        const volatile double zero = 0.0;
        job->cam_origin[r] = ((m[0] * zero + m[1] * zero) + m[2] * zero) + m[3];
    }
    job->half_width = camera->half_width;
    job->half_height = camera->half_height;
    job->pixel_size = camera->pixel_size;
    job->hsize = (uint32_t)camera->hsize;
    job->vsize = (uint32_t)camera->vsize;
    uint64_t local_tiles = 0;
    tile_counts(camera->hsize, camera->vsize, tile_rank, tile_ranks, &job->tiles_x, &job->tiles_total, &local_tiles);
    job->tile_rank = tile_rank;
    job->tile_ranks = tile_ranks;
    job->out_row_major = tile_ranks == 1 ? 1u : 0u;
    job->rays = nullptr;
    job->draw_threshold = RTC_DRAW_THRESHOLD;
    job->total = (uint32_t)(local_tiles * RTC_TILE_W * RTC_TILE_H);
    job->depth = depth;
    *local_pixels = local_tiles * RTC_TILE_W * RTC_TILE_H;
    return RTC_OK;
}

// ---- the entry point that allocates nothing (rtc_render_device_noalloc) -------------------
// What a launch needs beyond its inputs and outputs is the control block (work-queue heads and
// statistics counters) and the queue plan (one uint32 per local tile).  rtc_render_device keeps
// both in buffers the scene allocated; this variant keeps them in a file-scope device array, so
// that a caller that cannot call hipMalloc (graph capture, a serving loop with a fixed memory
// budget) can still launch.  Every GPU the library runs on holds its own copy of the array, at
// its own address, until the process ends.
constexpr uint32_t kNoAllocMaxTiles = RTC_NOALLOC_MAX_TILES;
constexpr size_t kScratchControlBytes = 4096;  // >= kProbeOffsetBytes, keeps the plan 256-byte aligned
static_assert(rtc::kProbeOffsetBytes <= (int)kScratchControlBytes, "control block fits its slot");
constexpr size_t kScratchBytes = kScratchControlBytes + (size_t)kNoAllocMaxTiles * sizeof(uint32_t);
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// Host code never uses the array's name as an address (it is a host-side stand-in): the device
// address is asked for once per device, with that device current, and kept here.
constexpr int kMaxDevices = 64;
struct ScratchSlot {
    unsigned char* ptr = nullptr;  // device address of this device's copy of `scratch`
    // the plan the copy holds: scene (by serial number), camera and partition it was computed for
    bool plan_valid = false, plan_identity = true;
    uint64_t plan_scene_serial = 0;
    RtcCameraDesc plan_camera{};
    uint32_t plan_rank = 0, plan_ranks = 0;
    // host copy of that plan: the source of the asynchronous upload, which must outlive the call; the event
    // (created with the slot, at set-up) marks the end of the last upload, after which the copy may change
    std::vector<uint32_t> host_plan;
    hipEvent_t uploaded = nullptr;
    bool upload_pending = false;
};
ScratchSlot g_scratch_slots[kMaxDevices];
std::mutex g_scratch_mutex;  // held by rtc_render_device_noalloc from the look at the slot to the launch

// (the caller holds g_scratch_mutex)
int scratch_of_current_device(ScratchSlot** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(RTC_ERR_TOO_LARGE, "device index beyond the scratch table");
    ScratchSlot& slot = g_scratch_slots[dev];
    if (!slot.ptr) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RTC_ERR_DEVICE, "hipGetSymbolAddress(scratch) returned NULL");
        HIP_TRY(hipEventCreateWithFlags(&slot.uploaded, hipEventDisableTiming));
        slot.ptr = (unsigned char*)p;
    }
    *out = &slot;
    return RTC_OK;
}

}  // namespace

namespace rtc {
int set_error(int code, const char* message) { return fail(code, message); }
int set_hip_error(const char* what, hipError_t e) { return fail_hip(what, e); }
int copy_device_to_host(RtcScene* scene, const void* d_src, void* dst, size_t bytes) {
    return copy_canvas_to_host(scene, d_src, dst, bytes);
}
}  // namespace rtc

extern "C" {

const char* rtc_last_error(void) { return g_last_error.c_str(); }

int rtc_device_count(int32_t* count) {
    if (!count) return fail(RTC_ERR_INVALID_ARGUMENT, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip("hipGetDeviceCount", e);
    }
    *count = n;
    return RTC_OK;
}

int rtc_set_device(int32_t device) {
    HIP_TRY(hipSetDevice(device));
    return RTC_OK;
}

int rtc_scene_create(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light, RtcScene** out_scene) {
    if (!out_scene) return fail(RTC_ERR_INVALID_ARGUMENT, "out_scene is NULL");
    *out_scene = nullptr;
    if (!light) return fail(RTC_ERR_INVALID_ARGUMENT, "light is NULL");
    if (n_shapes > 0 && !shapes) return fail(RTC_ERR_INVALID_ARGUMENT, "shapes is NULL");
    if (n_shapes > (1u << 21)) return fail(RTC_ERR_TOO_LARGE, "more than 2^21 shapes");
    if (!(light->position[3] == 1.0))
        return fail(RTC_ERR_INVALID_ARGUMENT, "light.position must be a point (w == 1)");

    constexpr size_t kRec = rtc::kRecordDoubles;  // doubles per sweep-table record, see DevWorld
    std::vector<double> inv12(n_shapes * 12);
    std::vector<int> group_of(n_shapes);
    std::vector<rtc::DevShade> shade(n_shapes);
    uint64_t n_spheres = 0, n_planes = 0, n_cubes = 0;
    std::unordered_map<uint64_t, std::vector<uint64_t>> buckets;
    for (uint64_t i = 0; i < n_shapes; ++i) {
        const RtcShapeDesc& s = shapes[i];
        if (s.geo != RTC_GEO_SPHERE && s.geo != RTC_GEO_PLANE && s.geo != RTC_GEO_CUBE) {
            char buf[96];
            std::snprintf(buf, sizeof buf, "shape %llu: unknown geo %d", (unsigned long long)i, s.geo);
            return fail(RTC_ERR_INVALID_ARGUMENT, buf);
        }
        (s.geo == RTC_GEO_SPHERE ? n_spheres : s.geo == RTC_GEO_PLANE ? n_planes : n_cubes)++;
        double full[16];
        if (int rc = invert_affine(s.transform, full, "shape", i)) return rc;
        std::memcpy(&inv12[i * 12], full, 12 * sizeof(double));
        // Group by what the sweep can skip: where the entries of the inverse that a shape's
        // test multiplies in are exact zeros (sphere: the 3x3 is diagonal; plane: row 1 is
        // (0,m,0)), the products are zeros that change no value and are not evaluated.
        const double* f = full;
        const bool row1_axis = f[4] == 0.0 && f[6] == 0.0;
        const bool diagonal = row1_axis && f[1] == 0.0 && f[2] == 0.0 && f[8] == 0.0 && f[9] == 0.0;
        // the group of the table (rtc::kSweepGroupTable) that serves this geo, the zero-skipping variant if it applies
        const bool zeros = s.geo == RTC_GEO_SPHERE ? diagonal : row1_axis;
        group_of[i] = -1;
        for (int g = 0; g < rtc::kSweepGroups; ++g) {
            const rtc::SweepGroupInfo& info = rtc::kSweepGroupTable[g];
            if (info.geo != s.geo || (info.axis_aligned && !zeros)) continue;
            group_of[i] = g;
            break;  // the specialised variant of a kind precedes its general one
        }
        if (group_of[i] < 0) return fail(RTC_ERR_INVALID_ARGUMENT, "no sweep group serves this shape kind");
        rtc::DevShade& d = shade[i];
        std::memset(&d, 0, sizeof d);
        const RtcMaterialDesc& m = s.material;
        std::memcpy(d.color, m.color, sizeof d.color);
        d.ambient = m.ambient;
        d.diffuse = m.diffuse;
        d.specular = m.specular;
        d.shininess = m.shininess;
        d.reflective = m.reflective;
        d.transparency = m.transparency;
        d.refractive_index = m.refractive_index;
        d.geo = s.geo;
        d.pattern = -1;
        {
            // inverse-transpose times the local normal (0,1,0), w forced to 0, normalised by
            // division (shape.zig:91-96, vector.zig:88-97), products with the zeros included
            const double* M = full;
            const double nx = (M[0] * 0.0 + M[4] * 1.0) + M[8] * 0.0;
            const double ny = (M[1] * 0.0 + M[5] * 1.0) + M[9] * 0.0;
            const double nz = (M[2] * 0.0 + M[6] * 1.0) + M[10] * 0.0;
            const double len = std::sqrt((nx * nx + ny * ny) + nz * nz);
            d.plane_normal[0] = nx / len;
            d.plane_normal[1] = ny / len;
            d.plane_normal[2] = nz / len;
        }
        // (with a light intensity that is not finite a specular term scaled by zero is NaN, not
        // nothing: such scenes take the general routine, which leaves out no term)
        const bool finite_light = std::isfinite(light->intensity[0]) && std::isfinite(light->intensity[1]) &&
                                  std::isfinite(light->intensity[2]);
        d.shininess_n = (finite_light && m.shininess >= 1.0 && m.shininess < 500.0 && m.shininess == std::floor(m.shininess))
                            ? (int32_t)m.shininess
                            : 0;
        if (m.has_pattern) {
            const int k = m.pattern.kind;
            if (k < RTC_PATTERN_POINT || k > RTC_PATTERN_CHECKERS) {
                char buf[96];
                std::snprintf(buf, sizeof buf, "shape %llu: unknown pattern kind %d", (unsigned long long)i, k);
                return fail(RTC_ERR_INVALID_ARGUMENT, buf);
            }
            double pfull[16];
            if (int rc = invert_affine(m.pattern.transform, pfull, "pattern of shape", i)) return rc;
            std::memcpy(d.pat_inv, pfull, 12 * sizeof(double));
            std::memcpy(d.pat_a, m.pattern.a, sizeof d.pat_a);
            std::memcpy(d.pat_b, m.pattern.b, sizeof d.pat_b);
            d.pattern = k;
        }
        // The reference identifies "the same object" by value (calc.zig:274,280); the kernels
        // identify it by list index.  The two agree unless two shapes are value-identical.
        bool has_nan = false;
        const uint64_t h = hash_shape(s, &has_nan);
        if (!has_nan) {
            auto& bucket = buckets[h];
            for (uint64_t j : bucket) {
                if (shapes_equal(shapes[j], s)) {
                    char buf[128];
                    std::snprintf(buf, sizeof buf, "shapes %llu and %llu are value-identical", (unsigned long long)j,
                                  (unsigned long long)i);
                    return fail(RTC_ERR_DUPLICATE_SHAPE, buf);
                }
            }
            bucket.push_back(i);
        }
    }

    std::vector<RtcScene::SphereBound> bounds;
    // table order: the groups one after the other
    std::vector<double> tab((n_shapes + 2) * kRec, 0.0);
    std::vector<double> inv_tab((n_shapes ? n_shapes : 1) * 12, 0.0);  // rows 0..2 of every inverse, table order (gathers)
    std::vector<int32_t> order(n_shapes ? n_shapes : 1, 0);
    std::vector<rtc::DevShade> shade_tab(n_shapes);
    int32_t group[rtc::kSweepGroups + 1] = {};
    uint32_t axis_mask[3] = {0, 0, 0};  // see DevWorld::sweep_axis_mask
    {
        size_t pos = 0;
        // Members of every group in list order; in a world too large for the on-chip tables the spheres of a
        // group follow a spatial order of their centres instead, so that spheres that follow each other in the
        // table lie close together and the streamed sweep can dismiss them a block at a time (sphere_block_bound).
        // Nothing depends on the order inside a group: equal t are resolved by the list index kept with each record.
        std::vector<std::vector<uint64_t>> members(rtc::kSweepGroups);
        for (uint64_t i = 0; i < n_shapes; ++i)
            if (group_of[i] >= 0) members[group_of[i]].push_back(i);
        if (n_shapes > (uint64_t)rtc::kLdsShapeLimit) {
            auto centre_ok = [&](uint64_t i) {
                const double* t = shapes[i].transform;
                return std::isfinite(t[3]) && std::isfinite(t[7]) && std::isfinite(t[11]);
            };
            // Spatial order in which the BITS of a table position mean something: a range is cut at the table
            // position inside it that is a multiple of the highest power of two, 2^b, and the spheres with the
            // smaller coordinate along axis(b) go below the cut.  Inside every aligned run of 2^(b+1) positions
            // bit b then says on which side of a plane across axis(b) a sphere lies - the same axis for every run
            // - so every aligned block of 8, 64, 512 ... positions is a cell of the subdivision (compact clusters,
            // whose bounding spheres few rays cross), and a ray that visits the entries of such a run in the order
            // of `position XOR mask`, mask bit b set where its direction along axis(b) is negative, visits them front
            // to back (sweep_spheres_lanes; sweep_axis_mask).  axis(b) follows the extents of the scene: from the
            // highest bit down, the axis along which a cell of that level is longest, the cell then halved.
            // (Shapes without a finite centre go last.)
            auto centre = [&](uint64_t i, int a) -> double { return shapes[i].transform[3 + 4 * a]; };
            int axis_of_bit[32];
            {
                double lo_c[3] = {0, 0, 0}, hi_c[3] = {0, 0, 0};
                bool any = false;
                for (uint64_t i = 0; i < n_shapes; ++i) {
                    if (shapes[i].geo != RTC_GEO_SPHERE || !centre_ok(i)) continue;
                    for (int a = 0; a < 3; ++a) {
                        lo_c[a] = any ? std::fmin(lo_c[a], centre(i, a)) : centre(i, a);
                        hi_c[a] = any ? std::fmax(hi_c[a], centre(i, a)) : centre(i, a);
                    }
                    any = true;
                }
                double ext[3] = {hi_c[0] - lo_c[0], hi_c[1] - lo_c[1], hi_c[2] - lo_c[2]};
                int top = 0;
                while ((1ull << (top + 1)) < n_shapes) ++top;
                for (int bit = 31; bit >= 0; --bit) {
                    int axis = 0;
                    for (int a = 1; a < 3; ++a)
                        if (ext[a] > ext[axis]) axis = a;
                    axis_of_bit[bit] = axis;
                    if (bit <= top) ext[axis] *= 0.5;
                }
                for (int bit = 3; bit < 32; ++bit) axis_mask[axis_of_bit[bit]] |= 1u << bit;
            }
            size_t table_pos = 0;  // where the group starts in the table: positions are counted from 0
            std::function<void(std::vector<uint64_t>&, size_t, size_t)> split = [&](std::vector<uint64_t>& v, size_t lo_i, size_t hi_i) {
                const size_t abs_lo = table_pos + lo_i, abs_hi = table_pos + hi_i;
                if (abs_hi - abs_lo < 2) return;
                // the multiple of the highest power of two strictly inside (abs_lo, abs_hi)
                int bit = 31;
                size_t cut = 0;
                for (; bit >= 3; --bit) {
                    cut = ((abs_lo >> bit) + 1) << bit;
                    if (cut < abs_hi) break;
                }
                if (bit < 3) return;  // the range lies inside one block
                const int axis = axis_of_bit[bit];
                const size_t mid = cut - table_pos;
                std::nth_element(v.begin() + lo_i, v.begin() + mid, v.begin() + hi_i, [&](uint64_t x, uint64_t y) {
                    const double cx = centre(x, axis), cy = centre(y, axis);
                    return cx < cy || (cx == cy && x < y);  // (ties by list index: deterministic)
                });
                split(v, lo_i, mid);
                split(v, mid, hi_i);
            };
            for (int g = 0; g < rtc::kSweepGroups; ++g) {
                if (rtc::kSweepGroupTable[g].geo == RTC_GEO_SPHERE) {
                    std::vector<uint64_t>& v = members[g];
                    const size_t n_finite = std::stable_partition(v.begin(), v.end(), centre_ok) - v.begin();
                    split(v, 0, n_finite);
                }
                table_pos += members[g].size();
            }
        }
        for (int g = 0; g < rtc::kSweepGroups; ++g) {
            group[g] = (int32_t)pos;
            for (uint64_t i : members[g]) {
                // the record of the shape in its group's layout (rtc_internal.h: kRec*), what each kind of sweep
                // reads side by side so that it arrives with one or two scalar loads
                const double* m = &inv12[i * 12];
                double* rec = &tab[pos * kRec];
                const int64_t order_word = (int64_t)i;
                double order_as_double;
                std::memcpy(&order_as_double, &order_word, sizeof order_word);
                double bound[4] = {0, 0, 0, std::numeric_limits<double>::infinity()}, from_light[4] = {0, 0, 0, 0}, side[3] = {0, 0, 0};
                sphere_bound(shapes[i], m, bound);
                if (shapes[i].geo == RTC_GEO_PLANE) plane_light_side(m, light->position, side);
                if (shapes[i].geo == RTC_GEO_SPHERE) sphere_from_light(bound, light->position, from_light);
                switch (g) {
                    case 0:  // sphere, diagonal inverse
                        for (int k = 0; k < 6; ++k) rec[rtc::kRecAxisSphereM + k] = rec[rtc::kRecAxisSphereShadowM + k] = m[rtc::kAxisSphereEntries[k]];
                        rec[rtc::kRecAxisSphereOrder] = order_as_double;
                        rec[rtc::kRecAxisSphereBr2] = bound[3];
                        for (int k = 0; k < 3; ++k) rec[rtc::kRecAxisSphereBc + k] = bound[k];
                        for (int k = 0; k < 4; ++k) rec[rtc::kRecAxisSphereLight + k] = from_light[k];
                        break;
                    case 2:  // plane, inverse row 1 = (0, m5, 0, m7)
                        rec[rtc::kRecAxisPlaneM] = m[5];
                        rec[rtc::kRecAxisPlaneM + 1] = m[7];
                        rec[rtc::kRecAxisPlaneOrder] = order_as_double;
                        for (int k = 0; k < 3; ++k) rec[rtc::kRecAxisPlaneLight + k] = side[k];
                        break;
                    case 3:  // plane
                        for (int k = 0; k < 4; ++k) rec[rtc::kRecPlaneM + k] = m[4 + k];
                        rec[rtc::kRecPlaneOrder] = order_as_double;
                        for (int k = 0; k < 3; ++k) rec[rtc::kRecPlaneLight + k] = side[k];
                        break;
                    default:  // sphere (1) and cube (4): the whole inverse first
                        std::memcpy(rec, m, 12 * sizeof(double));
                        rec[rtc::kRecFullOrder] = order_as_double;
                        for (int k = 0; k < 4; ++k) rec[rtc::kRecFullBound + k] = bound[k];
                        for (int k = 0; k < 4; ++k) rec[rtc::kRecFullLight + k] = from_light[k];
                        break;
                }
                if (shapes[i].geo == RTC_GEO_SPHERE) {
                    RtcScene::SphereBound sb;
                    std::memcpy(sb.b, bound, sizeof sb.b);
                    std::memcpy(sb.from_light, from_light, sizeof sb.from_light);
                    sb.pos = (int32_t)pos;
                    bounds.push_back(sb);
                }
                std::memcpy(&inv_tab[pos * 12], m, 12 * sizeof(double));
                order[pos] = (int32_t)i;
                shade_tab[pos] = shade[i];
                ++pos;
            }
        }
        group[rtc::kSweepGroups] = (int32_t)pos;
    }

    RtcScene* sc = new (std::nothrow) RtcScene();
    if (!sc) return fail(RTC_ERR_OUT_OF_MEMORY, "host allocation failed");
    {
        static std::atomic<uint64_t> next_serial{1};
        sc->serial = next_serial.fetch_add(1);
    }
    hipError_t e = hipGetDevice(&sc->device);
    if (e != hipSuccess) {
        free_scene(sc);
        return fail_hip("hipGetDevice", e);
    }
    for (uint64_t i = 0; i < n_shapes; ++i) {
        const RtcShapeDesc& s = shapes[i];
        if (!(std::fabs(s.material.transparency) > kF64Eps)) continue;
        if (s.geo == RTC_GEO_PLANE) {
            sc->costly_everywhere = true;
            continue;
        }
        // unit sphere (or the cube's corners, sqrt(3) from its centre) under the shape's transform:
        // centre = translation column
        const double* t = s.transform;
        sc->costly.push_back({{t[3], t[7], t[11]}, largest_stretch(t) * (s.geo == RTC_GEO_CUBE ? 1.7320508075688774 : 1.0)});
    }
    sc->sphere_bounds = std::move(bounds);
    sc->n = n_shapes;
    sc->n_spheres = n_spheres;
    sc->n_planes = n_planes;
    sc->n_cubes = n_cubes;
    for (int i = 0; i < 3; ++i) {
        sc->light_pos[i] = light->position[i];
        sc->light_int[i] = light->intensity[i];
    }
    const size_t n_alloc = n_shapes ? n_shapes : 1;
#define SCENE_TRY(expr)                 \
    do {                                \
        hipError_t _e = (expr);         \
        if (_e != hipSuccess) {         \
            free_scene(sc);             \
            return fail_hip(#expr, _e); \
        }                               \
    } while (0)
    SCENE_TRY(hipMalloc((void**)&sc->d_tab, tab.size() * sizeof(double)));
    SCENE_TRY(hipMalloc((void**)&sc->d_shade, n_alloc * sizeof(rtc::DevShade)));
    SCENE_TRY(hipMalloc((void**)&sc->d_counter, rtc::kProbeOffsetBytes + 19 * 8 * 8192));  // + developer probe area
    sc->d_stats = (uint64_t*)((char*)sc->d_counter + rtc::kStatsOffsetBytes);
    SCENE_TRY(hipEventCreate(&sc->ev_start));
    SCENE_TRY(hipEventCreate(&sc->ev_stop));
    {
        // side tables of the streamed sphere sweep: the spheres are the first two groups of the table
        const size_t n_sph = (size_t)group[2];
        sc->sphere_bounds_doubles = (n_sph + rtc::kSphereBoundsPadding) * 4;
        std::vector<double> side(2 * sc->sphere_bounds_doubles, 0.0);
        for (const RtcScene::SphereBound& sb : sc->sphere_bounds)  // (the spheres' table positions are 0 .. n_sph - 1)
            for (int k = 0; k < 4; ++k) {
                side[(size_t)sb.pos * 4 + k] = sb.b[k];
                side[sc->sphere_bounds_doubles + (size_t)sb.pos * 4 + k] = sb.from_light[k];
            }
        // the packed pre-test's tables: 8 floats per pair of spheres; padding pairs dismiss (radius term -inf for
        // radiance rays, +inf for shadow rays), real spheres without a bound never do
        const size_t n_pairs = (n_sph + 1) / 2 + rtc::kSphereBoundsPadding;
        sc->sphere_pairs_floats = n_pairs * 8;
        std::vector<float> pairs(2 * sc->sphere_pairs_floats, 0.0f);
        for (size_t q = 0; q < n_pairs; ++q) {
            pairs[q * 8 + 6] = pairs[q * 8 + 7] = -std::numeric_limits<float>::infinity();
            pairs[sc->sphere_pairs_floats + q * 8 + 6] = pairs[sc->sphere_pairs_floats + q * 8 + 7] = std::numeric_limits<float>::infinity();
        }
        for (const RtcScene::SphereBound& sb : sc->sphere_bounds) {
            float rad[4], sha[4];
            sphere_prefilter_tables(sb.b, sb.from_light, rad, sha);
            const size_t q = (size_t)sb.pos / 2, half = (size_t)sb.pos % 2;
            for (int k = 0; k < 4; ++k) {
                pairs[q * 8 + 2 * k + half] = rad[k];
                pairs[sc->sphere_pairs_floats + q * 8 + 2 * k + half] = sha[k];
            }
        }
        // the block tests' tables behind them: per block of kSphereBlock consecutive spheres a bounding sphere of
        // the members' bounding spheres, two blocks per record of 8 floats, padded like the tables above; and behind
        // those the same per super-block (eight blocks = 64 consecutive spheres) and per group (eight super-blocks)
        {
            std::vector<const RtcScene::SphereBound*> by_pos(n_sph, nullptr);
            for (const RtcScene::SphereBound& sb : sc->sphere_bounds) by_pos[(size_t)sb.pos] = &sb;
            double eps_blocks = 0.0;  // the largest shrink any block or super-block asks for
            auto bounds_of = [&](size_t unit) {
                std::vector<BlockBound> v((n_sph + unit - 1) / unit);
                for (size_t b = 0; b < v.size(); ++b) {
                    const size_t first = b * unit, last = std::min(n_sph, first + unit);
                    v[b] = sphere_block_bound(&by_pos[first], last - first);
                    if (v[b].bounded) eps_blocks = std::fmax(eps_blocks, v[b].eps);
                }
                return v;
            };
            const std::vector<BlockBound> blocks = bounds_of(rtc::kSphereBlock), supers = bounds_of(8 * rtc::kSphereBlock),
                                          groups = bounds_of(64 * rtc::kSphereBlock);
            // one shrink for the whole scene (a larger one only makes a test more cautious); beyond 2^-12 the
            // blocks of this scene are too loose to be worth a test
            const bool usable = eps_blocks > 0 && eps_blocks <= 0x1p-12;
            sc->block_shrink = usable ? (1.0 - eps_blocks) / (1.0 - 0x1p-27) : 0.0;
            sc->block_margin_scale = usable ? eps_blocks / 0x1p-27 : 1.0;
            // appends the radiance and the shadow table of one level; returns where the radiance table starts
            auto append_level = [&](const std::vector<BlockBound>& level, uint32_t* offset_radiance, uint32_t* offset_shadow) {
                const size_t n_records = (level.size() + 1) / 2 + rtc::kSphereBoundsPadding / 16 + 8;
                const size_t table_floats = n_records * 8;
                const size_t base = pairs.size();
                *offset_radiance = (uint32_t)(base - 0);
                *offset_shadow = (uint32_t)(base + table_floats - sc->sphere_pairs_floats);
                pairs.resize(base + 2 * table_floats, 0.0f);
                for (size_t q = 0; q < n_records; ++q) {
                    pairs[base + q * 8 + 6] = pairs[base + q * 8 + 7] = -std::numeric_limits<float>::infinity();
                    pairs[base + table_floats + q * 8 + 6] = pairs[base + table_floats + q * 8 + 7] = std::numeric_limits<float>::infinity();
                }
                for (size_t b = 0; b < level.size(); ++b) {
                    float rad[4] = {0, 0, 0, std::numeric_limits<float>::infinity()}, sha[4] = {0, 0, 0, -std::numeric_limits<float>::infinity()};
                    if (usable && level[b].bounded) {
                        // the block's sphere as sphere_bound would store it, with the scene's shrink
                        double bound[4] = {level[b].c[0], level[b].c[1], level[b].c[2],
                                           std::nextafter(level[b].r2 / (1.0 - eps_blocks), std::numeric_limits<double>::infinity())};
                        double from_light[4];
                        sphere_from_light(bound, light->position, from_light, eps_blocks);
                        sphere_prefilter_tables(bound, from_light, rad, sha);
                    }
                    const size_t q = b / 2, half = b % 2;
                    for (int k = 0; k < 4; ++k) {
                        pairs[base + q * 8 + 2 * k + half] = rad[k];
                        pairs[base + table_floats + q * 8 + 2 * k + half] = sha[k];
                    }
                }
            };
            append_level(blocks, &sc->block_pairs_offset_radiance, &sc->block_pairs_offset_shadow);
            append_level(supers, &sc->super_pairs_offset_radiance, &sc->super_pairs_offset_shadow);
            append_level(groups, &sc->group_pairs_offset_radiance, &sc->group_pairs_offset_shadow);
            // which level the sweep tests wave-wide: groups once there are enough of them to save trips
            sc->sweep_top_log2 = n_sph > 2048 ? 9 : 6;
            for (int a = 0; a < 3; ++a) sc->sweep_axis_mask[a] = axis_mask[a];
        }
        SCENE_TRY(hipMalloc((void**)&sc->d_sphere_pairs, pairs.size() * sizeof(float)));
        SCENE_TRY(hipMemcpy(sc->d_sphere_pairs, pairs.data(), pairs.size() * sizeof(float), hipMemcpyHostToDevice));
        SCENE_TRY(hipMalloc((void**)&sc->d_sphere_bounds, side.size() * sizeof(double)));
        SCENE_TRY(hipMemcpy(sc->d_sphere_bounds, side.data(), side.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    SCENE_TRY(hipMalloc((void**)&sc->d_inv, inv_tab.size() * sizeof(double)));
    SCENE_TRY(hipMemcpy(sc->d_inv, inv_tab.data(), inv_tab.size() * sizeof(double), hipMemcpyHostToDevice));
    SCENE_TRY(hipMalloc((void**)&sc->d_order, order.size() * sizeof(int32_t)));
    SCENE_TRY(hipMemcpy(sc->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    SCENE_TRY(hipMemcpy(sc->d_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (n_shapes)
        SCENE_TRY(hipMemcpy(sc->d_shade, shade_tab.data(), n_shapes * sizeof(rtc::DevShade), hipMemcpyHostToDevice));
    std::memcpy(sc->group, group, sizeof group);
#undef SCENE_TRY
    {
        // set-up time is where the library learns the device address of its fixed scratch array on this GPU
        // (rtc_render_device_noalloc then finds it resolved, also when its stream is being captured)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        ScratchSlot* slot = nullptr;
        if (int rc = scratch_of_current_device(&slot)) {
            free_scene(sc);
            return rc;
        }
    }
    *out_scene = sc;
    return RTC_OK;
}

int rtc_scene_destroy(RtcScene* scene) {
    free_scene(scene);
    return RTC_OK;
}

int rtc_scene_set_integrator(RtcScene* scene, int32_t integrator) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (integrator != RTC_INTEGRATOR_POSTORDER && integrator != RTC_INTEGRATOR_WEIGHTED)
        return fail(RTC_ERR_INVALID_ARGUMENT, "unknown integrator");
    scene->integrator = integrator;
    return RTC_OK;
}

int rtc_scene_set_canvas_store(RtcScene* scene, int32_t mode) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (mode != RTC_CANVAS_STORE_PIXELS && mode != RTC_CANVAS_STORE_BLOCKS)
        return fail(RTC_ERR_INVALID_ARGUMENT, "unknown canvas store mode");
    if (mode == RTC_CANVAS_STORE_BLOCKS && !scene->d_flush_stage) {
        // the staging area, allocated here (set-up time) for as many waves as the GPU can hold at once: four slots of
        // one 8x8 block of f64 RGB per wave
        int dev = 0, cus = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != scene->device) return fail(RTC_ERR_INVALID_ARGUMENT, "the scene's device must be the current one");
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const uint32_t waves = (uint32_t)cus * 32u;  // 2048 threads per CU at most
        const size_t bytes = (size_t)waves * rtc::kFlushSlots * 64 * 3 * sizeof(double);
        HIP_TRY(hipMalloc((void**)&scene->d_flush_stage, bytes));
        scene->flush_waves = waves;
    }
    scene->canvas_store = mode;
    return RTC_OK;
}

int rtc_scene_set_acceleration(RtcScene* scene, int32_t mode) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (mode != RTC_ACCEL_NONE && mode != RTC_ACCEL_BVH) return fail(RTC_ERR_INVALID_ARGUMENT, "unknown acceleration mode");
    HIP_TRY(hipDeviceSynchronize());  // a launch may still be reading the hierarchy
    if (scene->d_bvh) (void)hipFree(scene->d_bvh);
    scene->d_bvh = nullptr;
    if (mode == RTC_ACCEL_NONE || scene->sphere_bounds.size() < 2) return RTC_OK;
    // extent of the scene: the largest coordinate any bounded sphere reaches
    double extent = 1.0;
    for (const RtcScene::SphereBound& sb : scene->sphere_bounds) {
        const double r = std::sqrt(sb.b[3]);
        if (!std::isfinite(r)) continue;
        for (int a = 0; a < 3; ++a) extent = std::fmax(extent, std::fabs(sb.b[a]) + r);
    }
    std::vector<RtcScene::SphereBox> items;
    items.reserve(scene->sphere_bounds.size());
    for (const RtcScene::SphereBound& sb : scene->sphere_bounds) items.push_back(sphere_box(sb.b, sb.pos, extent));
    std::vector<rtc::BvhNode> nodes;
    nodes.reserve(items.size());
    double lo[3], hi[3];
    scene->bvh_root = build_bvh(items, 0, items.size(), nodes, lo, hi);
    scene->bvh_origin_limit = extent * 1024.0;  // see sphere_box
    HIP_TRY(hipMalloc((void**)&scene->d_bvh, nodes.size() * sizeof(rtc::BvhNode)));
    hipError_t e = hipMemcpy(scene->d_bvh, nodes.data(), nodes.size() * sizeof(rtc::BvhNode), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(scene->d_bvh);
        scene->d_bvh = nullptr;
        return fail_hip("hipMemcpy(bvh)", e);
    }
    return RTC_OK;
}

int rtc_scene_group_counts(const RtcScene* scene, uint64_t out_counts[RTC_SWEEP_GROUPS]) {
    if (!scene || !out_counts) return fail(RTC_ERR_INVALID_ARGUMENT, "NULL pointer");
    for (int g = 0; g < rtc::kSweepGroups; ++g) out_counts[g] = (uint64_t)(scene->group[g + 1] - scene->group[g]);
    return RTC_OK;
}

int rtc_sweep_group_info(uint32_t group, RtcSweepGroupInfo* out_info) {
    if (!out_info) return fail(RTC_ERR_INVALID_ARGUMENT, "out_info is NULL");
    if (group >= (uint32_t)rtc::kSweepGroups) return fail(RTC_ERR_INVALID_ARGUMENT, "group must be < RTC_SWEEP_GROUPS");
    const rtc::SweepGroupInfo& info = rtc::kSweepGroupTable[group];
    std::memset(out_info, 0, sizeof *out_info);
    out_info->geo = info.geo;
    out_info->axis_aligned = info.axis_aligned;
    out_info->two_roots = info.two_roots;
    out_info->flops_reference = info.flops_reference;
    out_info->flops_executed = info.flops_executed;
    out_info->flops_roots = info.flops_roots;
    std::snprintf(out_info->name, sizeof out_info->name, "%s", info.name);
    return RTC_OK;
}

int rtc_tile_buffer_doubles(const RtcCameraDesc* camera, uint32_t tile_rank, uint32_t tile_ranks,
                            uint64_t* out_doubles) {
    if (!out_doubles) return fail(RTC_ERR_INVALID_ARGUMENT, "out_doubles is NULL");
    if (int rc = check_camera(camera)) return rc;
    if (tile_ranks == 0 || tile_rank >= tile_ranks)
        return fail(RTC_ERR_INVALID_ARGUMENT, "tile_rank must be < tile_ranks and tile_ranks >= 1");
    if (tile_ranks == 1) {
        *out_doubles = camera->hsize * camera->vsize * 3;
        return RTC_OK;
    }
    uint32_t tx, tt;
    uint64_t local;
    tile_counts(camera->hsize, camera->vsize, tile_rank, tile_ranks, &tx, &tt, &local);
    *out_doubles = local * RTC_TILE_W * RTC_TILE_H * 3;
    return RTC_OK;
}

int rtc_render_device(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, uint32_t tile_rank,
                      uint32_t tile_ranks, void* d_out_rgb, void* hip_stream, RtcStats* stats) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!valid_stream_and_ptr(d_out_rgb)) return fail(RTC_ERR_INVALID_ARGUMENT, "d_out_rgb is NULL");
    rtc::DevJob job{};
    uint64_t local_pixels = 0;
    if (int rc = fill_camera_job(camera, depth, tile_rank, tile_ranks, &job, &local_pixels)) return rc;
    job.out = (double*)d_out_rgb;
    if (int rc = plan_tile_order(scene, camera, &job, local_pixels / (RTC_TILE_W * RTC_TILE_H), (hipStream_t)hip_stream))
        return rc;
    const uint64_t pixels = tile_ranks == 1 ? camera->hsize * camera->vsize : local_pixels;
    return run_job(scene, job, (hipStream_t)hip_stream, stats, pixels);
}

int rtc_render_device_to_canvas(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, uint32_t tile_rank,
                                uint32_t tile_ranks, void* d_canvas_rgb, void* hip_stream, RtcStats* stats) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!valid_stream_and_ptr(d_canvas_rgb)) return fail(RTC_ERR_INVALID_ARGUMENT, "d_canvas_rgb is NULL");
    rtc::DevJob job{};
    uint64_t local_pixels = 0;
    if (int rc = fill_camera_job(camera, depth, tile_rank, tile_ranks, &job, &local_pixels)) return rc;
    job.out = (double*)d_canvas_rgb;
    job.out_row_major = 1;
    if (int rc = plan_tile_order(scene, camera, &job, local_pixels / (RTC_TILE_W * RTC_TILE_H), (hipStream_t)hip_stream))
        return rc;
    return run_job(scene, job, (hipStream_t)hip_stream, stats, local_pixels);
}

int rtc_scatter_tiles_device(const RtcCameraDesc* camera, uint32_t tile_rank, uint32_t tile_ranks, const void* d_tiles,
                             void* d_canvas_rgb, void* hip_stream) {
    if (int rc = check_camera(camera)) return rc;
    if (!d_tiles || !d_canvas_rgb) return fail(RTC_ERR_INVALID_ARGUMENT, "device pointer is NULL");
    if (tile_ranks == 0 || tile_rank >= tile_ranks)
        return fail(RTC_ERR_INVALID_ARGUMENT, "tile_rank must be < tile_ranks and tile_ranks >= 1");
    HIP_TRY(rtc::launch_scatter_tiles((const double*)d_tiles, tile_rank, tile_ranks, (uint32_t)camera->hsize,
                                      (uint32_t)camera->vsize, (double*)d_canvas_rgb, (hipStream_t)hip_stream));
    return RTC_OK;
}

int rtc_scatter_tiles_u8_device(const RtcCameraDesc* camera, uint32_t tile_rank, uint32_t tile_ranks,
                                const void* d_tiles_u8, void* d_canvas_u8, void* hip_stream) {
    if (int rc = check_camera(camera)) return rc;
    if (!d_tiles_u8 || !d_canvas_u8) return fail(RTC_ERR_INVALID_ARGUMENT, "device pointer is NULL");
    if (tile_ranks == 0 || tile_rank >= tile_ranks)
        return fail(RTC_ERR_INVALID_ARGUMENT, "tile_rank must be < tile_ranks and tile_ranks >= 1");
    HIP_TRY(rtc::launch_scatter_tiles_u8((const uint8_t*)d_tiles_u8, tile_rank, tile_ranks, (uint32_t)camera->hsize,
                                         (uint32_t)camera->vsize, (uint8_t*)d_canvas_u8, (hipStream_t)hip_stream));
    return RTC_OK;
}

int rtc_render_device_noalloc(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, uint32_t tile_rank,
                              uint32_t tile_ranks, void* d_out_rgb, void* hip_stream, RtcStats* stats) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!valid_stream_and_ptr(d_out_rgb)) return fail(RTC_ERR_INVALID_ARGUMENT, "d_out_rgb is NULL");
#ifdef RTC_PROBE
    return fail(RTC_ERR_INVALID_ARGUMENT, "the developer probe build writes beyond the control block");
#endif
    rtc::DevJob job{};
    uint64_t local_pixels = 0;
    if (int rc = fill_camera_job(camera, depth, tile_rank, tile_ranks, &job, &local_pixels)) return rc;
    const uint64_t local_tiles = local_pixels / (RTC_TILE_W * RTC_TILE_H);
    if (local_tiles > kNoAllocMaxTiles) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "rtc_render_device_noalloc: %llu local tiles exceed RTC_NOALLOC_MAX_TILES (%u)",
                      (unsigned long long)local_tiles, kNoAllocMaxTiles);
        return fail(RTC_ERR_TOO_LARGE, buf);
    }
    job.out = (double*)d_out_rgb;
    hipStream_t stream = (hipStream_t)hip_stream;
    // While the stream is being captured into a graph nothing here may wait for the device or copy from the
    // host, and whatever the launch reads must still be there when the graph runs, any number of times, after
    // other calls have used the array: the captured launch takes the natural queue order (it reads no plan;
    // every pixel gets the same value, only the schedule differs), and statistics, which need a host
    // synchronisation, are refused.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(stream, &capture));
    const bool capturing = capture != hipStreamCaptureStatusNone;
    if (capturing && stats)
        return fail(RTC_ERR_INVALID_ARGUMENT,
                    "rtc_render_device_noalloc: statistics need a host synchronisation, which a capturing stream cannot give");
    // one host thread at a time from here to the launch: the slot's bookkeeping and the order of the
    // operations enqueued on the array (memset, plan upload, kernel) stay consistent
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ScratchSlot* slot = nullptr;
    if (int rc = scratch_of_current_device(&slot)) return rc;
    uint32_t* control = (uint32_t*)slot->ptr;
    uint32_t* d_plan = (uint32_t*)(slot->ptr + kScratchControlBytes);
    // queue plan: same arithmetic as rtc_render_device, kept in the scratch array instead of a buffer of the
    // scene; the slot remembers which scene, camera and partition the array's plan serves
    job.tile_order = nullptr;
    if (!capturing && !(local_tiles < 2 || scene->costly_everywhere || scene->costly.empty())) {
        const bool cached = slot->plan_valid && slot->plan_scene_serial == scene->serial && slot->plan_rank == tile_rank &&
                            slot->plan_ranks == tile_ranks && std::memcmp(&slot->plan_camera, camera, sizeof *camera) == 0;
        if (!cached) {
            slot->plan_valid = false;
            if (slot->upload_pending) {  // the last upload may still be reading the host copy
                HIP_TRY(hipEventSynchronize(slot->uploaded));
                slot->upload_pending = false;
            }
            if (int rc = compute_tile_order(scene, camera, &job, local_tiles, slot->host_plan)) return rc;
            // Stream-ordered upload: it runs after whatever this stream still has to do with the previous plan, and
            // the kernel below runs after it.  (Launches on OTHER streams must not overlap this call: see the header.)
            if (!slot->host_plan.empty()) {
                HIP_TRY(hipMemcpyAsync(d_plan, slot->host_plan.data(), local_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                HIP_TRY(hipEventRecord(slot->uploaded, stream));
                slot->upload_pending = true;
            }
            slot->plan_scene_serial = scene->serial;
            slot->plan_camera = *camera;
            slot->plan_rank = tile_rank;
            slot->plan_ranks = tile_ranks;
            slot->plan_identity = slot->host_plan.empty();
            slot->plan_valid = true;
        }
        job.tile_order = slot->plan_identity ? nullptr : d_plan;
    }
    const uint64_t pixels = tile_ranks == 1 ? camera->hsize * camera->vsize : local_pixels;
    return run_job(scene, job, stream, stats, pixels, control, capturing);
}

// rtc_render of a large canvas.  One launch followed by the copy leaves the link to the host idle while the kernel
// runs and the GPU idle while the canvas crosses (4K: 2.1 ms + 3.6 ms, and the moves into the caller's buffer behind
// that).  Here the frame is rendered in kBands bands of whole tile rows, each band a launch of its own on one stream
// (the queue of a launch is a list of tiles: plan_tile_order's order, cut by band); a second stream copies a band to
// the pinned staging buffer as soon as its launch has finished, slice by slice, and the host threads move arrived
// slices on.  Every pixel is computed by the same arithmetic whichever launch it belongs to.
const uint32_t kBands = [] {
    const char* e = std::getenv("RTC_BANDS");  // developer switch for timing
    const int n = e && *e ? std::atoi(e) : 6;
    return (uint32_t)(n >= 1 && n <= 64 ? n : 6);
}();
// (Not for rtc_render_u8: its 25 MB cross in 0.5 ms, less than six launches instead of one cost - 3.4 against 3.1 ms.)
int render_to_host_in_bands(RtcScene* s, const RtcCameraDesc* cam, int32_t depth, double* d_out, double* out_host) {
    rtc::DevJob job{};
    uint64_t local_pixels = 0;
    if (int rc = fill_camera_job(cam, depth, 0, 1, &job, &local_pixels)) return rc;
    job.out = d_out;
    const uint64_t local_tiles = local_pixels / (RTC_TILE_W * RTC_TILE_H);
    const uint32_t tiles_x = job.tiles_x, tiles_y = job.tiles_total / job.tiles_x;
    if (!s->band_render_stream) HIP_TRY(hipStreamCreateWithFlags(&s->band_render_stream, hipStreamNonBlocking));
    if (!s->band_copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->band_copy_stream, hipStreamNonBlocking));
    while (s->band_events.size() < kBands) {
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        s->band_events.push_back(ev);
    }
    if (!(s->band_plan_valid && std::memcmp(&s->band_camera, cam, sizeof *cam) == 0)) {
        s->band_plan_valid = false;
        // the bands: tile rows dealt evenly; inside a band the tiles keep the order of the whole-frame plan
        std::vector<uint32_t> order;
        if (int rc = compute_tile_order(s, cam, &job, local_tiles, order)) return rc;
        std::vector<uint32_t> band_of_row(tiles_y);
        s->band_row.assign(kBands + 1, (uint32_t)cam->vsize);
        for (uint32_t b = 0; b < kBands; ++b) {
            const uint32_t r0 = (uint32_t)((uint64_t)tiles_y * b / kBands), r1 = (uint32_t)((uint64_t)tiles_y * (b + 1) / kBands);
            for (uint32_t r = r0; r < r1; ++r) band_of_row[r] = b;
            s->band_row[b] = std::min<uint32_t>(r0 * RTC_TILE_H, (uint32_t)cam->vsize);
        }
        std::vector<uint32_t> count(kBands + 1, 0);
        for (uint64_t t = 0; t < local_tiles; ++t) ++count[band_of_row[t / tiles_x] + 1];
        s->band_first.assign(kBands + 1, 0);
        for (uint32_t b = 0; b < kBands; ++b) s->band_first[b + 1] = s->band_first[b] + count[b + 1];
        std::vector<uint32_t> at(s->band_first.begin(), s->band_first.end() - 1), list(local_tiles);
        for (uint64_t k = 0; k < local_tiles; ++k) {
            const uint32_t t = order.empty() ? (uint32_t)k : order[k];
            list[at[band_of_row[t / tiles_x]]++] = t;
        }
        if (s->band_order_capacity < local_tiles) {
            if (s->d_band_order) (void)hipFree(s->d_band_order);
            s->d_band_order = nullptr;
            s->band_order_capacity = 0;
            HIP_TRY(hipMalloc((void**)&s->d_band_order, local_tiles * sizeof(uint32_t)));
            s->band_order_capacity = local_tiles;
        }
        // (blocking, from a vector that goes away: launches of an earlier call have been waited for at its end)
        HIP_TRY(hipMemcpy(s->d_band_order, list.data(), local_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
        s->band_camera = *cam;
        s->band_plan_valid = true;
    }
    const size_t kSlice = 8u << 20, row_bytes = (size_t)cam->hsize * 3 * sizeof(double);
    const char* d_src = (const char*)d_out;
    std::vector<HostSlice> slices;
    std::vector<uint32_t> slices_end(kBands, 0);  // per band: one past its last slice
    for (uint32_t b = 0; b < kBands; ++b) {
        const size_t begin = (size_t)s->band_row[b] * row_bytes, end = (size_t)s->band_row[b + 1] * row_bytes;
        for (size_t at = begin; at < end; at += kSlice) slices.push_back({at, std::min(kSlice, end - at)});
        slices_end[b] = (uint32_t)slices.size();
    }
    if (int rc = have_slice_events(s, slices.size())) return rc;
    size_t k = 0;
    for (uint32_t b = 0; b < kBands; ++b) {
        rtc::DevJob band = job;
        band.tile_order = s->d_band_order + s->band_first[b];
        band.strip_order = nullptr;
        band.total = (s->band_first[b + 1] - s->band_first[b]) * (uint32_t)(RTC_TILE_W * RTC_TILE_H);
        if (int rc = run_job(s, band, s->band_render_stream, nullptr, 0)) return rc;
        HIP_TRY(hipEventRecord(s->band_events[b], s->band_render_stream));
        HIP_TRY(hipStreamWaitEvent(s->band_copy_stream, s->band_events[b], 0));
        for (; k < slices_end[b]; ++k) {
            HIP_TRY(hipMemcpyAsync((char*)s->h_staging + slices[k].begin, d_src + slices[k].begin, slices[k].len,
                                   hipMemcpyDeviceToHost, s->band_copy_stream));
            HIP_TRY(hipEventRecord(s->slice_events[k], s->band_copy_stream));
        }
    }
    const int rc = move_slices_to_caller(s, out_host, slices);
    HIP_TRY(hipStreamSynchronize(s->band_render_stream));
    HIP_TRY(hipStreamSynchronize(s->band_copy_stream));
    return rc;
}

int rtc_render(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, double* out_rgb, RtcStats* stats) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!out_rgb) return fail(RTC_ERR_INVALID_ARGUMENT, "out_rgb is NULL");
    if (int rc = check_camera(camera)) return rc;
    const size_t bytes = (size_t)camera->hsize * camera->vsize * 3 * sizeof(double);
    // the device-side canvas is kept with the scene and reused by later calls of the same size
    if (scene->canvas_bytes < bytes) {
        if (scene->d_canvas) (void)hipFree(scene->d_canvas);
        scene->d_canvas = nullptr;
        scene->canvas_bytes = 0;
        HIP_TRY(hipMalloc((void**)&scene->d_canvas, bytes));
        scene->canvas_bytes = bytes;
    }
    double* d_out = scene->d_canvas;
    // large canvases, when nobody asks for the counters of ONE launch: bands, rendering and copying at the same time
    static const bool bands_off = std::getenv("RTC_NO_BANDS") != nullptr;  // developer switch for timing
    const uint32_t tile_rows = ((uint32_t)camera->vsize + RTC_TILE_H - 1) / RTC_TILE_H;
    if (!stats && !bands_off && bytes >= (64u << 20) && tile_rows >= 4 * kBands && have_staging(scene, bytes))
        return render_to_host_in_bands(scene, camera, depth, d_out, out_rgb);
    int rc = rtc_render_device(scene, camera, depth, 0, 1, d_out, nullptr, stats);
    if (rc == RTC_OK) rc = copy_canvas_to_host(scene, d_out, out_rgb, bytes);
    return rc;
}

int rtc_render_u8(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, uint8_t* out_rgb8, RtcStats* stats) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!out_rgb8) return fail(RTC_ERR_INVALID_ARGUMENT, "out_rgb8 is NULL");
    if (int rc = check_camera(camera)) return rc;
    const size_t values = (size_t)camera->hsize * camera->vsize * 3;
    // the f64 canvas and, behind it, its 8-bit image: both stay with the scene
    const size_t bytes = values * sizeof(double) + values;
    if (scene->canvas_bytes < bytes) {
        if (scene->d_canvas) (void)hipFree(scene->d_canvas);
        scene->d_canvas = nullptr;
        scene->canvas_bytes = 0;
        HIP_TRY(hipMalloc((void**)&scene->d_canvas, bytes));
        scene->canvas_bytes = bytes;
    }
    double* d_rgb = scene->d_canvas;
    uint8_t* d_u8 = reinterpret_cast<uint8_t*>(d_rgb + values);
    int rc = rtc_render_device(scene, camera, depth, 0, 1, d_rgb, nullptr, stats);
    if (rc != RTC_OK) return rc;
    HIP_TRY(rtc::launch_quantize(d_rgb, values, d_u8, nullptr));
    return copy_canvas_to_host(scene, d_u8, out_rgb8, values);
}

int rtc_color_at(RtcScene* scene, const double* rays, uint64_t n_rays, int32_t depth, double* out_rgb,
                 RtcStats* stats) {
    if (!scene) return fail(RTC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (n_rays == 0) {
        if (stats) std::memset(stats, 0, sizeof *stats);
        return RTC_OK;
    }
    if (!rays || !out_rgb) return fail(RTC_ERR_INVALID_ARGUMENT, "rays/out_rgb is NULL");
    if (n_rays > (1ull << 31)) return fail(RTC_ERR_TOO_LARGE, "more than 2^31 rays");
    double *d_rays = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rays, n_rays * 6 * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_out, n_rays * 3 * sizeof(double));
    if (e != hipSuccess) {
        (void)hipFree(d_rays);
        return fail_hip("hipMalloc(d_out)", e);
    }
    int rc = RTC_OK;
    e = hipMemcpy(d_rays, rays, n_rays * 6 * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail_hip("hipMemcpy(rays)", e);
    if (rc == RTC_OK) {
        rtc::DevJob job{};
        job.rays = d_rays;
        job.draw_threshold = RTC_DRAW_THRESHOLD;
        job.total = (uint32_t)n_rays;
        job.depth = depth;
        job.out = d_out;
        job.tile_ranks = 1;
        job.out_row_major = 1;
        rc = run_job(scene, job, nullptr, stats, n_rays);
    }
    if (rc == RTC_OK) {
        e = hipMemcpy(out_rgb, d_out, n_rays * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail_hip("hipMemcpy(out_rgb)", e);
    }
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    return rc;
}

int rtc_deinterleave_device(const RtcCameraDesc* camera, uint32_t tile_ranks, uint64_t per_rank_doubles,
                            const void* d_gathered, void* d_out_rgb, void* hip_stream) {
    if (int rc = check_camera(camera)) return rc;
    if (!d_gathered || !d_out_rgb) return fail(RTC_ERR_INVALID_ARGUMENT, "device pointer is NULL");
    if (tile_ranks == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "tile_ranks must be >= 1");
    uint64_t need = 0;
    if (int rc = rtc_tile_buffer_doubles(camera, 0, tile_ranks, &need)) return rc;
    if (tile_ranks > 1 && per_rank_doubles < need)
        return fail(RTC_ERR_INVALID_ARGUMENT, "per_rank_doubles smaller than rank 0's tile buffer");
    HIP_TRY(rtc::launch_deinterleave((const double*)d_gathered, per_rank_doubles, tile_ranks, (uint32_t)camera->hsize,
                                     (uint32_t)camera->vsize, (double*)d_out_rgb, (hipStream_t)hip_stream));
    return RTC_OK;
}

int rtc_deinterleave_u8_device(const RtcCameraDesc* camera, uint32_t tile_ranks, uint64_t per_rank_bytes,
                               const void* d_gathered_u8, void* d_out_u8, void* hip_stream) {
    if (int rc = check_camera(camera)) return rc;
    if (!d_gathered_u8 || !d_out_u8) return fail(RTC_ERR_INVALID_ARGUMENT, "device pointer is NULL");
    if (tile_ranks == 0) return fail(RTC_ERR_INVALID_ARGUMENT, "tile_ranks must be >= 1");
    uint64_t need = 0;
    if (int rc = rtc_tile_buffer_doubles(camera, 0, tile_ranks, &need)) return rc;
    if (tile_ranks > 1 && per_rank_bytes < need)
        return fail(RTC_ERR_INVALID_ARGUMENT, "per_rank_bytes smaller than rank 0's quantised tile buffer");
    HIP_TRY(rtc::launch_deinterleave_u8((const uint8_t*)d_gathered_u8, per_rank_bytes, tile_ranks,
                                        (uint32_t)camera->hsize, (uint32_t)camera->vsize, (uint8_t*)d_out_u8,
                                        (hipStream_t)hip_stream));
    return RTC_OK;
}

int rtc_quantize_u8_device(const void* d_rgb, uint64_t n_values, void* d_out_u8, void* hip_stream) {
    if (n_values == 0) return RTC_OK;
    if (!d_rgb || !d_out_u8) return fail(RTC_ERR_INVALID_ARGUMENT, "device pointer is NULL");
    HIP_TRY(rtc::launch_quantize((const double*)d_rgb, n_values, (uint8_t*)d_out_u8, (hipStream_t)hip_stream));
    return RTC_OK;
}

int rtc_arith_probe(const double* a, const double* b, uint64_t n, double* out_div, double* out_sqrt,
                    double* out_pow) {
    if (n == 0) return RTC_OK;
    if (!a || !b || !out_div || !out_sqrt || !out_pow) return fail(RTC_ERR_INVALID_ARGUMENT, "NULL pointer");
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, n * 5 * sizeof(double)));
    int rc = RTC_OK;
    hipError_t e = hipMemcpy(d, a, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rtc::launch_arith_probe(d, d + n, n, d + 2 * n, d + 3 * n, d + 4 * n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_div, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_sqrt, d + 3 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_pow, d + 4 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail_hip("rtc_arith_probe", e);
    (void)hipFree(d);
    return rc;
}

#ifdef RTC_PROBE
int rtc_debug_probe(RtcScene* scene, uint64_t* out, uint64_t n_waves) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, (char*)scene->d_counter + rtc::kProbeOffsetBytes, 19 * 8 * 8192, hipMemcpyDeviceToHost));
    return RTC_OK;
}
#endif

int rtc_mat4_inverse(const double* m, double* out_inv, double* out_det) {
    if (!m || !out_inv) return fail(RTC_ERR_INVALID_ARGUMENT, "NULL pointer");
    const double det = det4(m);
    if (out_det) *out_det = det;
    if (!(std::fabs(det - 0.0) > kF64Eps)) return fail(RTC_ERR_SINGULAR_TRANSFORM, "matrix is not invertible");
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) out_inv[row * 4 + col] = cofactor4(m, col, row) / det;
    return RTC_OK;
}

}  // extern "C"
