// Internal declarations shared by the host side (rtc_api.hip) and the device side
// (rtc_kernels.hip) of librtc_hip.so.  Nothing here is part of the C-ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtc_hip.h"

namespace rtc {

// Deepest recursion the kernels are instantiated for.  A node at `remaining == 0` never
// spawns children (calc.zig:829,946), so `depth` frames suffice for depth `depth`.
constexpr int kMaxDepthSmall = 5;   // the reference's hard-coded depth (camera.zig:83)
constexpr int kMaxDepthLarge = 16;

// Per-shape record read by one lane after it knows which shape it hit (a gather).
// Everything a hit needs that the sweep table does not hold.
// Per-scene control block in device memory: the heads of the work queues (one 128-byte line
// each), then 128 bytes of statistics counters, then the developer probe area.
#ifndef RTC_QUEUES
#define RTC_QUEUES 16
#endif
constexpr int kQueues = RTC_QUEUES;
#ifndef RTC_DRAW_THRESHOLD
#define RTC_DRAW_THRESHOLD 64  // idle lanes a wave waits for before it draws its next chunk (rtc_refill.inc)
#endif
constexpr int kQueueStrideWords = 32;  // uint32 words between queue heads
constexpr int kStatsOffsetBytes = kQueues * kQueueStrideWords * 4;
constexpr int kProbeOffsetBytes = kStatsOffsetBytes + 128;

// Layout of a sweep-table record (kRecordDoubles doubles) per group: every sweep reads a contiguous run.
//   sphere with a diagonal inverse: [0..5] m0 m3 m5 m7 m10 m11, [6] list index, [7] bound radius term, [8..10] bound
//       centre — the radiance sweep's run [0..10]; [12..17] the six entries again, [18..21] light-anchored bound —
//       the shadow sweep's run [12..21]
//   plane with row 1 = (0, m5, 0, m7): [0] m5, [1] m7, [2] list index, [4..6] light side, light height, guard
//   plane: [0..3] m4..m7, [4] list index, [5..7] light side, light height, guard
//   sphere, cube: [0..11] rows 0..2 of the inverse, [12] list index, [13..16] bound, [20..23] light-anchored bound
constexpr int kAxisSphereEntries[6] = {0, 3, 5, 7, 10, 11};
constexpr int kRecAxisSphereM = 0, kRecAxisSphereOrder = 6, kRecAxisSphereBr2 = 7, kRecAxisSphereBc = 8,
              kRecAxisSphereShadowM = 12, kRecAxisSphereLight = 18;
constexpr int kRecAxisPlaneM = 0, kRecAxisPlaneOrder = 2, kRecAxisPlaneLight = 4;
constexpr int kRecPlaneM = 0, kRecPlaneOrder = 4, kRecPlaneLight = 5;
constexpr int kRecFullOrder = 12, kRecFullBound = 13, kRecFullLight = 20;
constexpr int kSphereBoundsPadding = 80;  // entries the streamed sweep may read past the last sphere (a block of 64 and the read-ahead)
#ifndef RTC_LDS_SHAPES
#define RTC_LDS_SHAPES 32
#endif
constexpr int kLdsShapeLimit = RTC_LDS_SHAPES;  // worlds of at most this many shapes keep their per-hit tables in LDS
constexpr int kSphereBlock = 8;           // spheres per block of the streamed sweep's block test
constexpr int kRecordDoubles = 24;  // doubles per sweep-table record (see DevWorld)
constexpr int kFlushSlots = 4;      // blocks per wave whose finished pixels can wait in the staging area (block stores)

// The sweep table is grouped by kind of test: one row per group, in table order.  A shape kind
// (Shape.geo, shape.zig:76-79) owns one or more groups; a group is swept by a loop specialised for
// it (rtc_kernels.hip: sweep_shape<KIND>), and this table is what the host side consults to place a
// shape, to report per-group counts and to price a test.  Adding a primitive = a geo value, its
// rows here, its branch in sweep_shape (local intersect) and in the hit's normal (local normal).
//   flops_reference  the reference's own arithmetic for one (ray, shape) test: Ray.transform of
//                    origin and direction plus the local intersect up to the hit/miss decision
//   flops_executed   what the specialised loop evaluates (products with exact zeros left out)
//   flops_roots      extra arithmetic when the test produces t values
constexpr int kSweepGroups = RTC_SWEEP_GROUPS;
struct SweepGroupInfo {
    int32_t geo;
    int32_t axis_aligned;  // the group's inverses have exact zeros where the test multiplies (see rtc_scene_create)
    int32_t two_roots;     // 1: closed convex solid, two t per hit, can be an open container while the origin is inside
    int32_t flops_reference, flops_executed, flops_roots;
    const char* name;
};
constexpr SweepGroupInfo kSweepGroupTable[kSweepGroups] = {
    {RTC_GEO_SPHERE, 1, 1, 54, 30, 6, "sphere, diagonal inverse"},
    {RTC_GEO_SPHERE, 0, 1, 54, 54, 6, "sphere"},
    {RTC_GEO_PLANE, 1, 0, 11, 3, 2, "plane, inverse row 1 = (0, m, 0, t)"},
    {RTC_GEO_PLANE, 0, 0, 11, 11, 2, "plane"},
    // extension beyond the reference: axis-aligned cube; 33 for the two transforms, 6 numerators,
    // 6 divisions (the comparisons are not counted, as everywhere); its t values come with the test
    {RTC_GEO_CUBE, 0, 1, 45, 45, 0, "cube"},
};

struct DevShade {
    double color[3];
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double pat_a[3];
    double pat_b[3];
    double pat_inv[12];  // rows 0..2 of inverse(pattern.transform), row-major 3x4
    // Planes: Shape.normalAt (shape.zig:84-97) before the flip towards the eye.  It is the same
    // vector at every point of a plane, evaluated once on the host with the same operations.
    double plane_normal[3];
    int32_t geo;         // RTC_GEO_*
    int32_t pattern;     // RTC_PATTERN_* or -1 when the material has no pattern
    int32_t shininess_n; // shininess when it is an integer in [1, 500), else 0 (see pow_small_integer)
    int32_t pad;
};

// Device-side view of an uploaded world.
//
// Layout in HBM (all allocated once per scene, read-only while rendering):
// The shapes are stored in "table order": five groups (spheres whose inverse is diagonal,
// other spheres, planes whose inverse row 1 is (0,m,0,t), other planes, cubes), each group in the
// reference's list order.  group[g]..group[g+1] are the table positions of group g.
//   tab   : (n+2) x 24 doubles, one 192-byte record per shape, laid out per group (kRec* above) so that what
//           a sweep reads of it is one contiguous run: the entries of rows 0..2 of inverse(shape.transform)
//           the group's test multiplies in, the shape's index in the reference's list (an integer in the
//           low bits of a word), centre and enlarged squared radius (divided by the test's shrink factor)
//           of a world-space bounding sphere (spheres only; +inf = none); planes: side of the plane the
//           light is on (+1, -1, 0 = unknown), |local y of the light|, guard factor
//           2^-32 max(|m4|+|m5|+|m6|, |m7|); spheres, for shadow rays (which all end at the light):
//           bounding-sphere centre minus light position, and |that|^2 (1 - 2^-25) - bound (1 + 2^-40).
//           The sweep reads it with wave-uniform addresses, up to two records ahead, hence two padding records.
//   shade : n x DevShade, table order — gathered per lane at a hit.
//   order : n x int32, table position -> list index (tie-breaks between equal t).
// Node of the optional bounding-volume hierarchy over the spheres (128 B): the boxes of both
// children, so that one fetch decides both.  A child code >= 0 is a node index, < 0 is a leaf
// holding the sphere at table position ~code.
struct BvhNode {
    double lo[2][3];
    double hi[2][3];
    int32_t child[2];
    int32_t pad[6];
};
static_assert(sizeof(BvhNode) == 128, "one node per 128-byte line");

struct DevWorld {
    const double* tab;
    const DevShade* shade;
    const int32_t* order;
    const double* inv;  // n x 12 doubles: rows 0..2 of inverse(shape.transform), table order — gathered per lane at a hit
    // per sphere (table order, starting at table position group[0]) the four numbers of the bounding test: for
    // radiance rays [13..16] of the record, for shadow rays [20..23]; padded by kSphereBoundsPadding entries
    const double* sphere_bounds_radiance;
    const double* sphere_bounds_shadow;
    // the same in single precision with widened margins, two spheres per record of 8 floats (x0 x1 y0 y1 z0 z1
    // k0 k1), for the packed pre-test of the streamed sweep (sphere_prefilter_tables, rtc_api.hip)
    const float* sphere_pairs_radiance;
    const float* sphere_pairs_shadow;
    // Behind each of the two: the same test for BLOCKS of kSphereBlock spheres that follow each other in the table
    // (large worlds keep their spheres in Morton order of their centres, so a block is a cluster): one record of 8
    // floats per two blocks, holding a sphere around the block's bounding spheres (sphere_block_bound, rtc_api.hip).
    // A ray that clears it clears every sphere of the block.  Offsets in floats from the two pointers above;
    // block_shrink scales the ray's |d|^2 term for these tests, block_margin_scale the shadow ray's margin
    // (block_shrink == 0: the block tests dismiss nothing).
    uint32_t block_pairs_offset_radiance, block_pairs_offset_shadow;
    // and behind those, in the same format, one entry per SUPER-BLOCK: eight blocks that follow each other in the
    // table (64 spheres; the table order is a spatial one, so they are a cluster too), with the same shrink
    uint32_t super_pairs_offset_radiance, super_pairs_offset_shadow;
    // and one entry per GROUP of eight super-blocks (512 spheres); sweep_top_log2 says which level the sweep tests
    // wave-wide: 6 = super-blocks, 9 = groups (worlds of more than 2048 spheres)
    uint32_t group_pairs_offset_radiance, group_pairs_offset_shadow;
    int32_t sweep_top_log2, sweep_pad;
    // The spheres' table order is a spatial one in which bit b of a table position (b >= 3) says on which side of a
    // plane across axis(b) the sphere lies within its aligned run of 2^(b+1) positions (rtc_scene_create):
    // sweep_axis_mask[a] has the bits b with axis(b) == a, and sweep_pad2 keeps the struct's size a multiple of 8
    uint32_t sweep_axis_mask[3], sweep_pad2;
    double block_shrink, block_margin_scale;
    int32_t n;
    int32_t group[kSweepGroups + 1];
    // acceleration (rtc_scene_set_acceleration): nullptr = every sphere is tested (the reference's list walk)
    const BvhNode* bvh;
    int32_t bvh_root;        // child code of the root
    double bvh_origin_limit; // rays starting farther out than this (any coordinate) are not pruned
    double light_pos[3];
    double light_int[3];
    double light_reach;  // 1 + |x| + |y| + |z| of the light position (see the same-side test of the plane sweep)
};

// Work description of one launch.
struct DevJob {
    // camera (camera.zig:20-27) with the inverse of its transform precomputed on the host
    double cam_inv[12];
    double cam_origin[3];  // cam_inv * point(0,0,0), evaluated as matrix.zig:215-224 does
    double half_width, half_height, pixel_size;
    uint32_t hsize, vsize;
    // tile partition (see rtc_render_device in the public header)
    uint32_t tiles_x, tiles_total, tile_rank, tile_ranks;
    // 1: `out` is the full row-major canvas (always so with one rank; with several ranks every rank
    // then stores its pixels at their final place, possibly in a peer GPU's memory); 0: the rank's
    // dense tile buffer
    uint32_t out_row_major;
    // explicit rays instead of camera rays (rtc_color_at): 6 doubles per ray, or nullptr
    const double* rays;
    uint32_t total;  // number of work items: local tile pixels (padding included) or rays
    // Order in which the queue hands out the local tiles (queue slot -> local tile index),
    // costliest first, or nullptr for the natural order.  Only the schedule depends on it.
    const uint32_t* tile_order;
    // The queue dealt in strips of 8 x 2 pixels instead (four entries per chunk of 64 queue positions; see
    // build_strip_order in rtc_api.hip for the entry format), or nullptr.  Takes precedence over tile_order.
    const unsigned long long* strip_order;
    int32_t depth;
    int32_t weighted;  // 1: the attenuation-weighted integrator (render_weighted_kernel) instead of the post-order one
    double* out;
    // block stores (rtc_scene_set_canvas_store): the waves' staging area, kFlushSlots x 64 pixels x 3 doubles for each
    // of the first flush_waves waves of the grid, or nullptr = every pixel is stored when it is finished
    double* flush_stage;
    uint32_t flush_waves;
    // idle lanes a wave waits for before it draws its next chunk (rtc_refill.inc; kernels of small worlds only)
    uint32_t draw_threshold;
    uint32_t* counter;  // control block: the work-queue heads first, zeroed before the launch
    uint64_t* stats;    // 9 counters (radiance, shadow, sphere hits, plane hits, shaded, sphere skips, plane skips, cube hits, prefilter skips) or nullptr
};

// Launchers implemented in rtc_kernels.hip.  grid_blocks == 0 lets the launcher size the
// grid from the device's CU count and the kernel's occupancy.
hipError_t launch_render(const DevWorld& world, const DevJob& job, bool with_stats, hipStream_t stream);
hipError_t launch_deinterleave(const double* gathered, uint64_t per_rank_doubles, uint32_t tile_ranks,
                               uint32_t hsize, uint32_t vsize, double* out, hipStream_t stream);
hipError_t launch_deinterleave_u8(const uint8_t* gathered, uint64_t per_rank_bytes, uint32_t tile_ranks,
                                  uint32_t hsize, uint32_t vsize, uint8_t* out, hipStream_t stream);
// One rank's dense tile buffer (f64 or 8-bit values, three per pixel) into the row-major canvas.
hipError_t launch_scatter_tiles(const double* tiles, uint32_t tile_rank, uint32_t tile_ranks, uint32_t hsize,
                                uint32_t vsize, double* canvas, hipStream_t stream);
hipError_t launch_scatter_tiles_u8(const uint8_t* tiles, uint32_t tile_rank, uint32_t tile_ranks, uint32_t hsize,
                                   uint32_t vsize, uint8_t* canvas, hipStream_t stream);
hipError_t launch_quantize(const double* rgb, uint64_t n, uint8_t* out, hipStream_t stream);
hipError_t launch_zero_words(uint32_t* p, uint32_t words, hipStream_t stream);
hipError_t launch_arith_probe(const double* a, const double* b, uint64_t n, double* out_div, double* out_sqrt,
                              double* out_pow, hipStream_t stream);


// Host-side helpers of rtc_api.hip used by the multi-GPU orchestration (rtc_multi.hip).
int set_error(int code, const char* message);            // records the thread's last error, returns code
int set_hip_error(const char* what, hipError_t e);       // same for a failed HIP call
int copy_device_to_host(RtcScene* scene, const void* d_src, void* dst, size_t bytes);  // staged, see copy_canvas_to_host

}  // namespace rtc
