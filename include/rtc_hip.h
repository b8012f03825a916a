/*
 * rtc_hip.h — C-ABI of the MI355X (gfx950) render path for the Zig ray tracer
 * stevschmid/raytracer-challenge.
 *
 * The library replaces the body of Camera.render (camera.zig:74-89) and everything
 * it calls: calc.worldColorAt (calc.zig:641-652), calc.intersectWorld (calc.zig:200-214),
 * calc.prepareComputations (calc.zig:247-306), calc.shadeHit (calc.zig:448-470),
 * calc.isShadowed (calc.zig:777-790), calc.lighting (calc.zig:26-73),
 * calc.reflectedColor (calc.zig:945-959), calc.refractedColor (calc.zig:828-853),
 * calc.schlick (calc.zig:1004-1026), Shape.intersect / Shape.normalAt (shape.zig:84-107)
 * and Pattern.patternAt (pattern.zig:77-91).
 *
 * Everything here is plain C: fixed-width integers, doubles, raw pointers.  Every
 * function returns an int status (RTC_OK == 0); the message of the last failure on
 * the calling thread is available from rtc_last_error().  No exceptions cross the ABI.
 *
 * All arithmetic on the path is IEEE-754 binary64 with no fused multiply-add and the
 * reference's evaluation order, so the colours written are value-identical to what the
 * Zig code computes from the same descriptors.
 */
#ifndef RTC_HIP_H
#define RTC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------- */
#define RTC_OK 0
#define RTC_ERR_INVALID_ARGUMENT 1   /* null pointer, zero size, bad enum value          */
#define RTC_ERR_SINGULAR_TRANSFORM 2 /* |det| <= f64 epsilon: matrix.zig:288 would panic */
#define RTC_ERR_NOT_AFFINE 3         /* inverse's last row is not (0,0,0,1)              */
#define RTC_ERR_DUPLICATE_SHAPE 4    /* two value-identical shapes (calc.zig:274,280)    */
#define RTC_ERR_OUT_OF_MEMORY 5      /* maps to Zig's error.OutOfMemory                  */
#define RTC_ERR_DEVICE 6             /* a HIP call failed; see rtc_last_error()          */
#define RTC_ERR_TOO_LARGE 7          /* size beyond what an entry point supports         */

/* ---- enums (declaration order of the reference's unions) ----------------- */
/* Shape.geo, shape.zig:76-79 */
#define RTC_GEO_SPHERE 0
#define RTC_GEO_PLANE 1
/* Extension beyond the reference, which stops at sphere and plane: the axis-aligned cube
 * [-1, 1]^3 of the book's next chapter, added through the same hook (a new member of the union, a
 * localIntersect / localNormalAt pair).  A Zig caller only passes it after adding the member to
 * its own Shape.geo. */
#define RTC_GEO_CUBE 2
/* Pattern.pattern, pattern.zig:67-73 */
#define RTC_PATTERN_POINT 0
#define RTC_PATTERN_STRIPE 1
#define RTC_PATTERN_GRADIENT 2
#define RTC_PATTERN_RING 3
#define RTC_PATTERN_CHECKERS 4

/* ---- plain-old-data descriptors ------------------------------------------ */
/* Pattern (pattern.zig:64-75).  a/b are unused for RTC_PATTERN_POINT.
 * transform is Mat4.mat row-major (matrix.zig:103). */
typedef struct RtcPatternDesc {
    int32_t kind;
    int32_t _pad0;
    double a[3];
    double b[3];
    double transform[16];
} RtcPatternDesc;

/* Material (material.zig:6-18).  has_pattern == 0 is Zig's `pattern: ?Pattern = null`. */
typedef struct RtcMaterialDesc {
    double color[3];
    double ambient;
    double diffuse;
    double specular;
    double shininess;
    double reflective;
    double transparency;
    double refractive_index;
    int32_t has_pattern;
    int32_t _pad0;
    RtcPatternDesc pattern;
} RtcMaterialDesc;

/* Shape (shape.zig:73-82). */
typedef struct RtcShapeDesc {
    int32_t geo;
    int32_t _pad0;
    double transform[16];
    RtcMaterialDesc material;
} RtcShapeDesc;

/* PointLight (light.zig:9-14).  position is a Vec4 (x,y,z,w) and must have w == 1. */
typedef struct RtcLightDesc {
    double position[4];
    double intensity[3];
} RtcLightDesc;

/* Camera (camera.zig:20-27), as produced by the unchanged Camera.init (camera.zig:29-46);
 * the library never evaluates tan(). */
typedef struct RtcCameraDesc {
    uint64_t hsize;
    uint64_t vsize;
    double field_of_view;
    double half_width;
    double half_height;
    double pixel_size;
    double transform[16];
} RtcCameraDesc;

/* Work counters of one call.  radiance_rays counts calc.worldColorAt invocations
 * (primary + reflected + refracted), shadow_rays counts calc.isShadowed invocations;
 * each is one full sweep over every shape.  sphere_hits / plane_hits count the
 * (ray, shape) pairs whose local intersect produced t values (shape.zig:35,62).
 * shaded_hits counts calc.shadeHit invocations.  kernel_ms is the device time of the
 * render kernel measured with HIP events on the stream it ran on.  sphere_bound_skips counts
 * the (ray, sphere) pairs (part of sphere_tests) that the library dismissed with its bounding
 * sphere test without evaluating the reference's 54-flop intersection: their discriminant
 * would have been negative.  plane_side_skips counts the (shadow ray, plane) pairs (part of
 * plane_tests) of which only the origin's row is evaluated, because origin and light lie
 * clearly on the same side of the plane. */
typedef struct RtcStats {
    uint64_t radiance_rays;
    uint64_t shadow_rays;
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t sphere_hits;
    uint64_t plane_hits;
    uint64_t shaded_hits;
    uint64_t pixels;
    double kernel_ms;
    uint64_t sphere_bound_skips;
    uint64_t plane_side_skips;
    uint64_t cube_tests; /* (ray, cube) pairs and those that produced t values (extension, RTC_GEO_CUBE) */
    uint64_t cube_hits;
    /* (ray, sphere) pairs (part of sphere_tests) dismissed by the single-precision pre-test that runs ahead of
     * the bounding test in worlds too large for the on-chip tables; they do not appear in sphere_bound_skips */
    uint64_t sphere_prefilter_skips;
} RtcStats;

/* Opaque handle of an uploaded world (World, world.zig:15-28). */
typedef struct RtcScene RtcScene;

/* ---- entry points -------------------------------------------------------- */

/* Text of the last error raised on the calling thread ("" if none). */
const char* rtc_last_error(void);

/* Number of visible GPUs / select the GPU used by later calls of this thread. */
int rtc_device_count(int32_t* count);
int rtc_set_device(int32_t device);

/* Flatten a world (world.zig:15-28: objects in list order, one light) into device
 * buffers on the current GPU.  For every shape and pattern the inverse transform is
 * computed once on the host with the reference's cofactor algorithm (matrix.zig:286-300),
 * which yields the same bits as the reference's per-call recomputation.  Rejects
 * singular and non-affine transforms and value-identical duplicate shapes.
 * n_shapes may be 0 (every ray misses, calc.zig:649-651) and at most 2^21. */
int rtc_scene_create(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light,
                     RtcScene** out_scene);
int rtc_scene_destroy(RtcScene* scene);

/* Optional acceleration of the sweeps.  RTC_ACCEL_NONE (the default) tests every shape against
 * every ray, as calc.intersectWorld does (calc.zig:204-209).  RTC_ACCEL_BVH builds a bounding
 * volume hierarchy over the spheres (planes are always tested) and visits only the spheres whose
 * enlarged bounding boxes the ray crosses between its origin and the best hit so far (for a
 * shadow ray: the light); the pairs left out are ones whose intersection the reference would
 * compute and then not use (no roots, roots behind the origin, or beyond the hit), and ties
 * between equal t are still resolved by list order, so the canvas is the same.  With the
 * hierarchy on, sphere_hits and sphere_bound_skips of RtcStats count visited pairs only. */
#define RTC_ACCEL_NONE 0
#define RTC_ACCEL_BVH 1
int rtc_scene_set_acceleration(RtcScene* scene, int32_t mode);

/* How a pixel's tree of reflected and refracted rays (calc.shadeHit, calc.zig:460-469) is summed.
 * RTC_INTEGRATOR_POSTORDER (the default) evaluates the reference's expressions in the reference's
 * order, surface.add(reflected.scale(..)).add(refracted.scale(..)) from the leaves up: the canvas is
 * the reference's value for value.  RTC_INTEGRATOR_WEIGHTED traces the same rays to the same hits
 * but gives every ray the product of the factors on its path from the root and adds weight x
 * surface colour at every hit: no frames, no unwinding, and parked refracted rays that any idle
 * lane can trace.  Its colours differ from the reference's in the last bits (re-association: a few
 * 1e-16 relative, tested to 1e-9 absolute and to identical 8-bit values except <= 1 LSB on <= 0.01 %
 * of the channels) and, in pixels whose subtrees were shared between lanes, from run to run.  Ray
 * counts and every other statistic are the same.  It cannot render in place into a canvas shared
 * between ranks (rtc_render_device_to_canvas with tile_ranks > 1). */
#define RTC_INTEGRATOR_POSTORDER 0
#define RTC_INTEGRATOR_WEIGHTED 1
int rtc_scene_set_integrator(RtcScene* scene, int32_t integrator);

/* The sweep table is grouped by kind of test; each group is swept by a loop specialised for it.
 * rtc_sweep_group_info describes group `group` (0 <= group < RTC_SWEEP_GROUPS): the Shape.geo value
 * it serves, whether it is the variant for inverses with exact zeros (a sphere whose inverse's 3x3
 * part is diagonal; a plane whose inverse has row 1 = (0, m, 0, t)) whose products with those zeros
 * are not evaluated (the values are unchanged), and the flops of one (ray, shape) test: as the
 * reference evaluates it, as the group's loop executes it, and extra when the test yields t values.
 * rtc_scene_group_counts tells how many shapes of a scene fall in each group.  A roofline that
 * wants executed rather than algorithmic flops needs both. */
#define RTC_SWEEP_GROUPS 5
typedef struct RtcSweepGroupInfo {
    int32_t geo;
    int32_t axis_aligned;
    int32_t two_roots;
    int32_t flops_reference;
    int32_t flops_executed;
    int32_t flops_roots;
    char name[40];
} RtcSweepGroupInfo;
int rtc_sweep_group_info(uint32_t group, RtcSweepGroupInfo* out_info);
int rtc_scene_group_counts(const RtcScene* scene, uint64_t out_counts[RTC_SWEEP_GROUPS]);

/* Camera.render (camera.zig:74-89): fill out_rgb (host memory, hsize*vsize*3 doubles,
 * pixel (x,y) at index (y*hsize + x)*3 as canvas.zig:25, channels red,green,blue as
 * color.zig:7-9) with calc.worldColorAt(world, rayForPixel(x,y), depth).  The
 * reference hard-codes depth 5 (camera.zig:83).  depth is the reference's i32 `remaining`
 * (calc.zig:641): a value <= 0 spawns no secondary rays (calc.zig:829,946), so a negative depth
 * renders like depth 0; the largest supported depth is 16 (RTC_ERR_TOO_LARGE beyond).  This holds
 * for every entry point that takes a depth.  stats may be NULL.  The call returns when out_rgb is
 * complete.  A canvas of 64 MB and more is rendered in several launches (bands of rows, on streams
 * of the scene's own) while earlier bands cross to the host, unless stats is given (the counters are
 * those of one launch on the NULL stream, as for smaller canvases).
 *
 * Concurrency (all render entry points): the launches of one scene share the scene's control
 * block, tile plan, timing events and staging buffers.  At most one launch per scene may be in
 * flight at a time (later launches on the same stream are ordered by the stream; launches on
 * different streams or threads must be serialised by the caller, or use one scene handle each),
 * and the scene's device (the one current in rtc_scene_create) must be the current device of the
 * calling thread, else RTC_ERR_INVALID_ARGUMENT. */
int rtc_render(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, double* out_rgb,
               RtcStats* stats);

/* Same, with the framebuffer left in device memory and an optional tile partition for
 * multi-GPU rendering.  The image is cut into tiles of RTC_TILE_W x RTC_TILE_H pixels,
 * numbered row-major; this call renders tiles t with t % tile_ranks == tile_rank.
 * With tile_ranks == 1, d_out_rgb is the full row-major canvas.  Otherwise d_out_rgb is
 * the rank's dense tile buffer: local tile k (global tile k*tile_ranks + tile_rank)
 * occupies doubles [k*RTC_TILE_W*RTC_TILE_H*3, ...) row-major inside the tile; pixels
 * of edge tiles that fall outside the canvas are left untouched.
 * hip_stream is a hipStream_t (NULL = default stream).  The launch is asynchronous
 * unless stats != NULL, in which case the call synchronises the stream and fills stats. */
#define RTC_TILE_W 64
#define RTC_TILE_H 8
int rtc_render_device(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth,
                      uint32_t tile_rank, uint32_t tile_ranks, void* d_out_rgb, void* hip_stream,
                      RtcStats* stats);

/* rtc_render_device for callers that cannot allocate device memory.  This entry point has no
 * counterpart in the reference; it exists for callers that must not call hipMalloc: tight serving
 * loops, harnesses that forbid allocation, code under HIP graph capture.  It renders exactly what
 * rtc_render_device renders (same arguments, same output layout, same values) but keeps the
 * launch's scratch data - the control block with the work-queue heads and statistics counters,
 * and the queue plan of one uint32 per local tile - in a fixed array in the library's device
 * image instead of buffers of the scene, and calls no hipMalloc / hipFree or other routine that
 * allocates device memory.  Limit: at most RTC_NOALLOC_MAX_TILES local tiles (tiles of this rank's
 * share: 7680x4320 on one rank is 64 800), RTC_ERR_TOO_LARGE beyond.
 * Without statistics the call only enqueues work on hip_stream (a memset of the control block, an
 * upload of the queue plan when scene, camera or partition differ from the previous call on this
 * GPU, the kernel) and never waits for the device, except that a changed plan first waits for the
 * previous plan's UPLOAD to have left the host.  With stats != NULL it synchronises like
 * rtc_render_device.
 * While hip_stream is being captured into a graph: stats must be NULL (RTC_ERR_INVALID_ARGUMENT
 * otherwise); the launch uses the natural queue order instead of a plan, because a plan would live
 * in the shared array, which later calls may overwrite before the graph runs (same pixels, a
 * schedule a few per cent slower on frames with transparent shapes); the captured nodes are the
 * memset and the kernel, and the graph may be replayed any number of times as long as scene, output
 * buffer and stream ordering stay valid.
 * The array (about 1 MiB) exists once per GPU and is shared by every call on that GPU, whatever the
 * scene: launches that overlap on one GPU (two streams, two threads, a replayed graph beside a
 * direct call) race on it - the later one resets the queue heads the earlier one is still drawing
 * from - and must be serialised by the caller: the previous launch complete, or ordered before the
 * next one on the same stream.  The scene itself must have been created before (rtc_scene_create
 * allocates, and is where the library resolves the array's device address for the current GPU). */
#define RTC_NOALLOC_MAX_TILES 262144
int rtc_render_device_noalloc(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth,
                              uint32_t tile_rank, uint32_t tile_ranks, void* d_out_rgb,
                              void* hip_stream, RtcStats* stats);

/* How the render kernel stores finished pixels into a row-major canvas (rtc_render_device with tile_ranks == 1,
 * rtc_render_device_to_canvas, rtc_render_device_noalloc; post-order integrator, calls without statistics).  No
 * counterpart in the reference (canvas.zig:24-28 is one store per pixel on one thread).
 *   RTC_CANVAS_STORE_PIXELS  (default) every lane stores its pixel's 24 bytes when the pixel is finished
 *   RTC_CANVAS_STORE_BLOCKS  the finished pixels of an 8x8 block wait in a staging area of the wave that traces the
 *                            block and are copied out together when the block is complete: eight runs of 192
 *                            contiguous bytes per store instruction instead of scattered 24-byte stores.  Meant for a
 *                            canvas in ANOTHER GPU's memory (peer access, IPC mapping), where small stores cross the
 *                            link one by one.  Allocates the staging area (about 50 MB) when first selected.
 * The canvas values are the same in both modes. */
#define RTC_CANVAS_STORE_PIXELS 0
#define RTC_CANVAS_STORE_BLOCKS 1
int rtc_scene_set_canvas_store(RtcScene* scene, int32_t mode);

/* rtc_render_device with the rank's pixels stored at their final place: d_canvas_rgb is the full
 * row-major canvas (hsize*vsize*3 doubles) and this call fills the pixels of the tiles t with
 * t % tile_ranks == tile_rank, leaving the others untouched.  The canvas may live in another GPU's
 * memory that the current GPU can address (peer access within one process, or an IPC mapping):
 * N ranks then assemble one canvas with no tile buffers, no gather and no de-interleave pass.
 * No counterpart in the reference (camera.zig:77-86 is one serial loop). */
int rtc_render_device_to_canvas(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth,
                                uint32_t tile_rank, uint32_t tile_ranks, void* d_canvas_rgb,
                                void* hip_stream, RtcStats* stats);

/* One rank's dense tile buffer (as rtc_render_device fills it, or its 8-bit image after
 * rtc_quantize_u8_device) into the row-major canvas: the sender's half of a gather.  The canvas
 * may be peer memory.  Device pointers. */
int rtc_scatter_tiles_device(const RtcCameraDesc* camera, uint32_t tile_rank, uint32_t tile_ranks,
                             const void* d_tiles, void* d_canvas_rgb, void* hip_stream);
int rtc_scatter_tiles_u8_device(const RtcCameraDesc* camera, uint32_t tile_rank, uint32_t tile_ranks,
                                const void* d_tiles_u8, void* d_canvas_u8, void* hip_stream);

/* Number of doubles rtc_render_device writes into for (camera, tile_rank, tile_ranks). */
int rtc_tile_buffer_doubles(const RtcCameraDesc* camera, uint32_t tile_rank, uint32_t tile_ranks,
                            uint64_t* out_doubles);

/* Scatter the gathered per-rank tile buffers (rank r's buffer at d_gathered +
 * r*per_rank_doubles) back into the row-major canvas d_out_rgb (device memory). */
int rtc_deinterleave_device(const RtcCameraDesc* camera, uint32_t tile_ranks,
                            uint64_t per_rank_doubles, const void* d_gathered, void* d_out_rgb,
                            void* hip_stream);

/* calc.worldColorAt (calc.zig:641-652) for caller-supplied rays: rays holds n_rays
 * records of 6 doubles (origin x,y,z then direction x,y,z; origin.w = 1 and
 * direction.w = 0 are implied), out_rgb receives n_rays*3 doubles.  Host pointers. */
int rtc_color_at(RtcScene* scene, const double* rays, uint64_t n_rays, int32_t depth,
                 double* out_rgb, RtcStats* stats);

/* byteColorValue (ppm.zig:6-9) on the device: out[i] = (uint8_t)clamp(rgb[i]*256, 0, 255)
 * for n_values doubles.  Device pointers. */
int rtc_quantize_u8_device(const void* d_rgb, uint64_t n_values, void* d_out_u8, void* hip_stream);

/* Camera.render followed by byteColorValue (ppm.zig:6-9) of every channel, the first half of
 * canvasToPPM (ppm.zig:11-38): out_rgb8 (host memory, hsize*vsize*3 bytes, same pixel and
 * channel order as rtc_render) receives the 8-bit values the reference's writer prints, so the
 * caller only formats text.  The colours are quantised on the device and 3 instead of 24 bytes
 * per pixel cross to the host.  stats may be NULL. */
int rtc_render_u8(RtcScene* scene, const RtcCameraDesc* camera, int32_t depth, uint8_t* out_rgb8,
                  RtcStats* stats);

/* rtc_deinterleave_device for quantised tile buffers: each rank quantises its dense tile buffer
 * with rtc_quantize_u8_device (one byte per value, same order), the gather moves 3 B instead of
 * 24 B per pixel, and this call scatters the gathered bytes (rank r's buffer at d_gathered_u8 +
 * r*per_rank_bytes) into the row-major 8-bit canvas canvasToPPM (ppm.zig:11-38) formats.
 * With tile_ranks == 1 the tile buffer of rtc_render_device is already the row-major canvas;
 * this call then expects the tiled layout and is not needed. */
int rtc_deinterleave_u8_device(const RtcCameraDesc* camera, uint32_t tile_ranks, uint64_t per_rank_bytes,
                               const void* d_gathered_u8, void* d_out_u8, void* hip_stream);

/* ---- Camera.render over several GPUs from one process ------------------------------------
 * The body of Camera.render (camera.zig:74-89) for a caller that is one thread of one process
 * (the Zig program) and wants the GPUs of a node: the library holds one replica of the world per
 * rank, deals the RTC_TILE_W x RTC_TILE_H tiles of the canvas round-robin to the ranks
 * (rank = tile % n_ranks), runs every rank's render kernel concurrently on its own stream and
 * assembles the row-major canvas on rank 0's GPU.  The canvas is value-identical to rtc_render's
 * for every rank count and mode.  No counterpart in the reference (it is single-threaded).
 *
 * rtc_multi_create: devices[r] is the HIP device of rank r; devices == NULL maps rank r onto
 * visible device r % device_count (so n_ranks may exceed the number of GPUs: ranks then share a
 * GPU, which is how a one-GPU machine rehearses an eight-rank frame). */
typedef struct RtcMulti RtcMulti;
#define RTC_MULTI_MAX_RANKS 64
int rtc_multi_create(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light,
                     const int32_t* devices, uint32_t n_ranks, RtcMulti** out_multi);
int rtc_multi_destroy(RtcMulti* multi);
int rtc_multi_rank_device(const RtcMulti* multi, uint32_t rank, int32_t* out_device);
/* rtc_scene_set_integrator for every rank's replica.  With RTC_INTEGRATOR_WEIGHTED and more than one rank the
 * modes that work on tile buffers apply (scatter, RCCL gather, host-direct); RTC_MULTI_PEER_STORES renders in
 * place into rank 0's canvas, which that integrator cannot do, and fails with RTC_ERR_INVALID_ARGUMENT. */
int rtc_multi_set_integrator(RtcMulti* multi, int32_t integrator);

/* How the ranks' pixels reach rank 0's canvas. */
#define RTC_MULTI_PEER_STORES 0  /* render kernels store into rank 0's canvas through peer access */
#define RTC_MULTI_PEER_SCATTER 1 /* dense tile buffers; each rank scatters its tiles into rank 0's
                                    canvas through peer access after rendering (8-bit: after
                                    quantising, so 3 instead of 24 bytes per pixel cross the link) */
#define RTC_MULTI_RCCL_GATHER 2  /* dense tile buffers, RCCL send/recv to rank 0 (single process,
                                    ncclCommInitAll), de-interleave kernel on rank 0 */
#define RTC_MULTI_HOST_DIRECT 3  /* dense tile buffers; each rank scatters its tiles over its own PCIe link
                                    into a pinned host canvas kept by the library, which host threads then
                                    move into the caller's buffer: nothing crosses between the GPUs, rank 0
                                    assembles nothing, and the copy to the host is split over the ranks'
                                    links.  The destination must not be NULL; with one rank the frame is
                                    rendered and copied as by rtc_render.  In RtcMultiTiming the transfer to
                                    the pinned canvas counts as render_and_transfer_ms, the host threads'
                                    copy as to_host_ms */
#define RTC_MULTI_PEER_BLOCK_STORES 4 /* RTC_MULTI_PEER_STORES with the render kernels' stores switched to whole
                                         8x8 blocks (rtc_scene_set_canvas_store: RTC_CANVAS_STORE_BLOCKS): what
                                         crosses the link are runs of 192 contiguous bytes instead of single
                                         pixels of 24 */

/* Device-side phases of one call, in milliseconds: from the first launch until every rank's pixels
 * are on rank 0's GPU (rendering and the transfers, which overlap in the peer modes), then rank
 * 0's own assembly (de-interleave and/or quantisation); device_ms is the host's wall clock from
 * the call until the canvas is complete on rank 0's GPU, to_host_ms the copy to the caller. */
typedef struct RtcMultiTiming {
    double render_and_transfer_ms;
    double assemble_ms;
    double device_ms;
    double to_host_ms;
} RtcMultiTiming;

/* Camera.render on all ranks.  out_rgb: host memory, hsize*vsize*3 doubles as rtc_render; NULL
 * leaves the canvas on rank 0's GPU (timing runs).  stats (may be NULL) receives the sums over the
 * ranks and the longest kernel; asking for stats serialises the ranks.  timing may be NULL. */
int rtc_render_multi(RtcMulti* multi, const RtcCameraDesc* camera, int32_t depth, int32_t mode,
                     double* out_rgb, RtcStats* stats, RtcMultiTiming* timing);
/* The same with byteColorValue (ppm.zig:6-9) applied on the GPUs, as rtc_render_u8. */
int rtc_render_multi_u8(RtcMulti* multi, const RtcCameraDesc* camera, int32_t depth, int32_t mode,
                        uint8_t* out_rgb8, RtcStats* stats, RtcMultiTiming* timing);

/* For callers that run one PROCESS per GPU (an MPI- or torchrun-style launch) and want the
 * peer-store assembly of rtc_render_device_to_canvas: rank 0 creates the canvas and a handle
 * (hipMalloc + hipIpcGetMemHandle), sends the RTC_IPC_HANDLE_BYTES bytes to the other ranks by
 * whatever channel it has, each of which maps the canvas into its own address space with
 * rtc_ipc_canvas_open (hipIpcOpenMemHandle) and renders its tiles straight into it.  The caller
 * synchronises the ranks (a barrier after the launches) before rank 0 reads the canvas.
 * rtc_ipc_canvas_close frees (owner) or unmaps (others).  No counterpart in the reference. */
#define RTC_IPC_HANDLE_BYTES 64
int rtc_ipc_canvas_create(uint64_t bytes, void** out_d_canvas, uint8_t out_handle[RTC_IPC_HANDLE_BYTES]);
int rtc_ipc_canvas_open(const uint8_t handle[RTC_IPC_HANDLE_BYTES], void** out_d_canvas);
int rtc_ipc_canvas_close(void* d_canvas, int32_t is_owner);

/* ---- diagnostics (used by the parity tests) ------------------------------ */

/* Mat4.inverse (matrix.zig:286-300) and Mat4.determinant (matrix.zig:272-280) exactly as
 * rtc_scene_create evaluates them on the host; m and out_inv are 16 doubles row-major.
 * Returns RTC_ERR_SINGULAR_TRANSFORM where the reference's assert (matrix.zig:288) fires. */
int rtc_mat4_inverse(const double* m, double* out_inv, double* out_det);

/* Element-wise a[i]/b[i], sqrt(a[i]) and std.math.pow(a[i], b[i]) (as used at
 * calc.zig:67,1024,1025) computed ON THE DEVICE with the code paths the render kernel
 * uses, for bit-level comparison with the host's IEEE results.  Host pointers. */
int rtc_arith_probe(const double* a, const double* b, uint64_t n, double* out_div, double* out_sqrt,
                    double* out_pow);

#ifdef __cplusplus
}
#endif
#endif /* RTC_HIP_H */
