// Camera.render (camera.zig:74-89) over several GPUs of one node from ONE process: the part of
// the C-ABI a single-threaded caller (the Zig program) needs to use more than one GPU.
//
// The reference has no counterpart (it is single-threaded).  Pixels are independent
// (camera.zig:77-86), so the canvas is cut into RTC_TILE_W x RTC_TILE_H tiles dealt round-robin to
// the ranks (rank = tile % ranks); every rank holds a replica of the scene on its own GPU and runs
// the same render kernel on its tiles.  How the pixels reach rank 0's row-major canvas is the mode:
//   RTC_MULTI_PEER_STORES   the render kernels store every finished pixel at its final place in
//                           rank 0's canvas through peer access (xGMI): no tile buffers, no gather,
//                           no de-interleave pass; the transfers hide under the rendering
//   RTC_MULTI_PEER_BLOCK_STORES  the same, a finished 8x8 block at a time (192-byte runs instead of 24-byte stores)
//   RTC_MULTI_PEER_SCATTER  each rank renders into a dense tile buffer in its own HBM, (quantises it,
//                           for the 8-bit canvas), and a scatter kernel on that rank writes the
//                           tiles into rank 0's canvas through peer access in long runs
//   RTC_MULTI_RCCL_GATHER   dense tile buffers, RCCL send/recv to rank 0 (ncclCommInitAll, one
//                           communicator rank per GPU), de-interleave kernel on rank 0
//   RTC_MULTI_HOST_DIRECT   dense tile buffers; each rank's scatter kernel writes its tiles at their final
//                           place in a pinned host canvas over that GPU's own PCIe link: nothing crosses
//                           between the GPUs, rank 0 assembles nothing, and the copy to the host, the
//                           bottleneck of a one-GPU frame, is split N ways
// One host thread drives every GPU: all launches are asynchronous, each rank has its own stream,
// and rank 0's stream waits on an event per rank before the canvas is used.
#include "rtc_internal.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace {

#define MULTI_HIP(expr)                                        \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return rtc::set_hip_error(#expr, _e); \
    } while (0)
#define MULTI_RTC(expr)           \
    do {                          \
        int _rc = (expr);         \
        if (_rc != RTC_OK) return _rc; \
    } while (0)

// RCCL is loaded when the gather mode is first used: the other modes, and every single-GPU caller,
// need neither the library nor its headers.  The few declarations used, as the library's C interface
// publishes them (rccl.h: ncclResult_t 0 = success; ncclDataType_t ncclUint8 = 1, ncclFloat64 = 8):
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint8 = 1, ncclDouble = 8;
struct Rccl {
    void* handle = nullptr;
    bool loaded = false;  // every symbol below resolved
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    void unload() {
        if (handle) (void)dlclose(handle);
        handle = nullptr;
        loaded = false;
        CommInitAll = nullptr, CommDestroy = nullptr, GroupStart = nullptr, GroupEnd = nullptr, Send = nullptr, Recv = nullptr,
        GetErrorString = nullptr;
    }
    bool load(std::string* why) {
        if (loaded) return true;
        unload();
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) {
            const char* e = dlerror();
            *why = std::string("dlopen(librccl.so.1): ") + (e ? e : "not found");
            return false;
        }
#define RCCL_SYM(field, symbol)                                       \
    field = reinterpret_cast<decltype(field)>(dlsym(handle, symbol)); \
    if (!field) {                                                     \
        *why = std::string("librccl has no ") + symbol;               \
        unload(); /* a later call starts over instead of calling through a null pointer */ \
        return false;                                                 \
    }
        RCCL_SYM(CommInitAll, "ncclCommInitAll")
        RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd")
        RCCL_SYM(Send, "ncclSend")
        RCCL_SYM(Recv, "ncclRecv")
        RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
        loaded = true;
        return true;
    }
};
Rccl g_rccl;

}  // namespace

struct RtcMulti {
    uint32_t ranks = 0;
    std::vector<int> device;          // HIP device of each rank
    std::vector<RtcScene*> scene;     // the world, once per rank (each owns its control block)
    std::vector<hipStream_t> stream;  // one stream per rank, on the rank's device
    std::vector<hipEvent_t> done;     // "this rank's part of the frame is in place"
    std::vector<double*> tiles;       // dense tile buffer of each rank (scatter and gather modes)
    std::vector<uint8_t*> tiles_u8;
    std::vector<size_t> tiles_doubles, tiles_u8_bytes;
    bool peer_ok = true;              // every rank's GPU can address rank 0's memory
    std::string peer_why;
    // on rank 0's device
    double* canvas = nullptr;
    size_t canvas_doubles = 0;
    uint8_t* canvas_u8 = nullptr;
    size_t canvas_u8_bytes = 0;
    double* gathered = nullptr;       // gather mode: [ranks][per-rank doubles]
    size_t gathered_doubles = 0;
    uint8_t* gathered_u8 = nullptr;
    size_t gathered_u8_bytes = 0;
    // host-direct mode: the pinned host canvas (f64 or 8-bit, whichever the frame asks for)
    void* h_canvas = nullptr;
    size_t h_canvas_bytes = 0;
    hipEvent_t t_begin = nullptr, t_rendered = nullptr, t_end = nullptr;
    // RCCL: one communicator rank per distinct GPU, in order of first appearance among the ranks
    std::vector<int> comm_device;
    std::vector<ncclComm_t> comm;
    std::vector<int> comm_of_rank;
};

namespace {

int ensure_device_buffer(void** p, size_t* have, size_t need) {
    if (*have >= need && *p) return RTC_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    MULTI_HIP(hipMalloc(p, need ? need : 1));
    *have = need;
    return RTC_OK;
}

void destroy_multi(RtcMulti* m) {
    if (!m) return;
    for (uint32_t r = 0; r < m->ranks; ++r) {
        if (r < m->device.size()) (void)hipSetDevice(m->device[r]);
        if (r < m->stream.size() && m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
    }
    for (ncclComm_t c : m->comm)
        if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    for (uint32_t r = 0; r < m->ranks; ++r) {
        if (r < m->device.size()) (void)hipSetDevice(m->device[r]);
        if (r < m->scene.size() && m->scene[r]) (void)rtc_scene_destroy(m->scene[r]);
        if (r < m->tiles.size() && m->tiles[r]) (void)hipFree(m->tiles[r]);
        if (r < m->tiles_u8.size() && m->tiles_u8[r]) (void)hipFree(m->tiles_u8[r]);
        if (r < m->done.size() && m->done[r]) (void)hipEventDestroy(m->done[r]);
        if (r < m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    }
    if (!m->device.empty()) (void)hipSetDevice(m->device[0]);
    if (m->canvas) (void)hipFree(m->canvas);
    if (m->canvas_u8) (void)hipFree(m->canvas_u8);
    if (m->gathered) (void)hipFree(m->gathered);
    if (m->gathered_u8) (void)hipFree(m->gathered_u8);
    if (m->h_canvas) (void)hipHostFree(m->h_canvas);
    for (hipEvent_t e : {m->t_begin, m->t_rendered, m->t_end})
        if (e) (void)hipEventDestroy(e);
    delete m;
}

int ensure_rccl(RtcMulti* m) {
    if (!m->comm.empty()) return RTC_OK;
    std::string why;
    if (!g_rccl.load(&why)) return rtc::set_error(RTC_ERR_DEVICE, why.c_str());
    m->comm_of_rank.assign(m->ranks, 0);
    m->comm_device.clear();
    for (uint32_t r = 0; r < m->ranks; ++r) {
        int found = -1;
        for (size_t k = 0; k < m->comm_device.size(); ++k)
            if (m->comm_device[k] == m->device[r]) found = (int)k;
        if (found < 0) {
            found = (int)m->comm_device.size();
            m->comm_device.push_back(m->device[r]);
        }
        m->comm_of_rank[r] = found;
    }
    m->comm.assign(m->comm_device.size(), nullptr);
    const ncclResult_t rc = g_rccl.CommInitAll(m->comm.data(), (int)m->comm_device.size(), m->comm_device.data());
    if (rc != ncclSuccess) {
        m->comm.clear();
        const std::string msg = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc);
        return rtc::set_error(RTC_ERR_DEVICE, msg.c_str());
    }
    return RTC_OK;
}

uint64_t doubles_of_rank(const RtcCameraDesc* cam, uint32_t rank, uint32_t ranks) {
    uint64_t n = 0;
    (void)rtc_tile_buffer_doubles(cam, rank, ranks, &n);
    return n;
}

// The frame on rank 0's device: m->canvas (f64) and, when want_u8, m->canvas_u8.  Returns with
// every stream still running; m->t_end is recorded on rank 0's stream behind the last operation.
int render_frame(RtcMulti* m, const RtcCameraDesc* cam, int32_t depth, int32_t mode, bool want_u8, RtcStats* stats) {
    const uint32_t R = m->ranks;
    if (mode != RTC_MULTI_PEER_STORES && mode != RTC_MULTI_PEER_SCATTER && mode != RTC_MULTI_RCCL_GATHER &&
        mode != RTC_MULTI_HOST_DIRECT && mode != RTC_MULTI_PEER_BLOCK_STORES)
        return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "unknown multi-GPU mode");
    // peer stores a block at a time: the same path with the kernels' canvas stores switched to whole 8x8 blocks
    const int32_t canvas_store = mode == RTC_MULTI_PEER_BLOCK_STORES ? RTC_CANVAS_STORE_BLOCKS : RTC_CANVAS_STORE_PIXELS;
    if (mode == RTC_MULTI_PEER_BLOCK_STORES) mode = RTC_MULTI_PEER_STORES;
    // one rank: the canvas is rendered where rtc_render renders it, and copied like there
    const bool host_direct = mode == RTC_MULTI_HOST_DIRECT && R > 1;
    if (mode == RTC_MULTI_HOST_DIRECT && R == 1) mode = RTC_MULTI_PEER_STORES;
    if ((mode == RTC_MULTI_PEER_STORES || mode == RTC_MULTI_PEER_SCATTER) && !m->peer_ok) {
        const std::string msg = "peer access to rank 0's GPU is not available (" + m->peer_why + "): use RTC_MULTI_RCCL_GATHER";
        return rtc::set_error(RTC_ERR_DEVICE, msg.c_str());
    }
    const bool tiled = R > 1 && mode != RTC_MULTI_PEER_STORES;
    uint64_t n0 = 0;  // rank 0 holds the largest share
    MULTI_RTC(rtc_tile_buffer_doubles(cam, 0, R, &n0));  // validates the camera as well
    const size_t values = (size_t)cam->hsize * cam->vsize * 3;

    MULTI_HIP(hipSetDevice(m->device[0]));
    // with 8-bit output and per-rank quantisation the f64 canvas is not needed on rank 0; in
    // host-direct mode no canvas is
    const bool f64_canvas = !(want_u8 && tiled) && !host_direct;
    if (host_direct) {
        const size_t need = want_u8 ? values : values * sizeof(double);
        if (m->h_canvas_bytes < need) {
            if (m->h_canvas) (void)hipHostFree(m->h_canvas);
            m->h_canvas = nullptr;
            m->h_canvas_bytes = 0;
            // portable + mapped: every rank's GPU addresses it (through the same pointer: unified addressing)
            MULTI_HIP(hipHostMalloc(&m->h_canvas, need, hipHostMallocPortable | hipHostMallocMapped));
            m->h_canvas_bytes = need;
        }
    }
    if (f64_canvas) {
        void* p = m->canvas;
        size_t have = m->canvas_doubles * sizeof(double);
        MULTI_RTC(ensure_device_buffer(&p, &have, values * sizeof(double)));
        m->canvas = (double*)p;
        m->canvas_doubles = have / sizeof(double);
    }
    if (want_u8 && !host_direct) {
        void* p = m->canvas_u8;
        MULTI_RTC(ensure_device_buffer(&p, &m->canvas_u8_bytes, values));
        m->canvas_u8 = (uint8_t*)p;
    }
    if (tiled && mode == RTC_MULTI_RCCL_GATHER) {
        MULTI_RTC(ensure_rccl(m));
        if (want_u8) {
            void* p = m->gathered_u8;
            MULTI_RTC(ensure_device_buffer(&p, &m->gathered_u8_bytes, (size_t)n0 * R));
            m->gathered_u8 = (uint8_t*)p;
        } else {
            void* p = m->gathered;
            size_t have = m->gathered_doubles * sizeof(double);
            MULTI_RTC(ensure_device_buffer(&p, &have, (size_t)n0 * R * sizeof(double)));
            m->gathered = (double*)p;
            m->gathered_doubles = have / sizeof(double);
        }
    }
    if (tiled) {
        for (uint32_t r = 0; r < R; ++r) {
            MULTI_HIP(hipSetDevice(m->device[r]));
            void* p = m->tiles[r];
            size_t have = m->tiles_doubles[r] * sizeof(double);
            MULTI_RTC(ensure_device_buffer(&p, &have, (size_t)n0 * sizeof(double)));
            m->tiles[r] = (double*)p;
            m->tiles_doubles[r] = have / sizeof(double);
            if (want_u8) {
                void* q = m->tiles_u8[r];
                MULTI_RTC(ensure_device_buffer(&q, &m->tiles_u8_bytes[r], (size_t)n0));
                m->tiles_u8[r] = (uint8_t*)q;
            }
        }
    }

    MULTI_HIP(hipSetDevice(m->device[0]));
    MULTI_HIP(hipEventRecord(m->t_begin, m->stream[0]));
    // the other ranks start behind rank 0's begin mark, so that buffers rank 0 just (re)allocated
    // or a previous frame still being read are not overtaken
    for (uint32_t r = 1; r < R; ++r) {
        MULTI_HIP(hipSetDevice(m->device[r]));
        MULTI_HIP(hipStreamWaitEvent(m->stream[r], m->t_begin, 0));
    }

    RtcStats total{};
    auto add_stats = [&](const RtcStats& s) {
        total.radiance_rays += s.radiance_rays;
        total.shadow_rays += s.shadow_rays;
        total.sphere_tests += s.sphere_tests;
        total.plane_tests += s.plane_tests;
        total.sphere_hits += s.sphere_hits;
        total.plane_hits += s.plane_hits;
        total.shaded_hits += s.shaded_hits;
        total.sphere_bound_skips += s.sphere_bound_skips;
        total.plane_side_skips += s.plane_side_skips;
        total.cube_tests += s.cube_tests;
        total.cube_hits += s.cube_hits;
        total.sphere_prefilter_skips += s.sphere_prefilter_skips;
        total.kernel_ms = s.kernel_ms > total.kernel_ms ? s.kernel_ms : total.kernel_ms;
    };

    // ---- render: every rank's kernel is in flight before any is waited for -------------------
    // (with stats the launches are synchronous one after the other: statistics runs are not timed)
    for (uint32_t r = 0; r < R; ++r) {
        MULTI_HIP(hipSetDevice(m->device[r]));
        RtcStats s{};
        RtcStats* sp = stats ? &s : nullptr;
        MULTI_RTC(rtc_scene_set_canvas_store(m->scene[r], canvas_store));  // (allocates its staging area once)
        if (R == 1) {
            MULTI_RTC(rtc_render_device(m->scene[r], cam, depth, 0, 1, m->canvas, m->stream[r], sp));
        } else if (!tiled) {
            MULTI_RTC(rtc_render_device_to_canvas(m->scene[r], cam, depth, r, R, m->canvas, m->stream[r], sp));
        } else {
            MULTI_RTC(rtc_render_device(m->scene[r], cam, depth, r, R, m->tiles[r], m->stream[r], sp));
        }
        if (stats) add_stats(s);
    }

    // ---- bring the tiles to rank 0 (or, host-direct, to the pinned host canvas) ----------------------
    if (tiled && (mode == RTC_MULTI_PEER_SCATTER || host_direct)) {
        for (uint32_t r = 0; r < R; ++r) {
            MULTI_HIP(hipSetDevice(m->device[r]));
            void* target = host_direct ? nullptr : (want_u8 ? (void*)m->canvas_u8 : (void*)m->canvas);
            if (host_direct) MULTI_HIP(hipHostGetDevicePointer(&target, m->h_canvas, 0));
            if (want_u8) {
                MULTI_RTC(rtc_quantize_u8_device(m->tiles[r], doubles_of_rank(cam, r, R), m->tiles_u8[r], m->stream[r]));
                MULTI_RTC(rtc_scatter_tiles_u8_device(cam, r, R, m->tiles_u8[r], target, m->stream[r]));
            } else {
                MULTI_RTC(rtc_scatter_tiles_device(cam, r, R, m->tiles[r], target, m->stream[r]));
            }
        }
    }
    if (tiled && mode == RTC_MULTI_RCCL_GATHER) {
        for (uint32_t r = 0; r < R; ++r) {
            if (!want_u8) break;
            MULTI_HIP(hipSetDevice(m->device[r]));
            MULTI_RTC(rtc_quantize_u8_device(m->tiles[r], doubles_of_rank(cam, r, R), m->tiles_u8[r], m->stream[r]));
        }
        // ranks on rank 0's own GPU hand their buffer over with a device copy; the others send it
        bool any_remote = false;
        for (uint32_t r = 0; r < R; ++r) any_remote = any_remote || m->comm_of_rank[r] != 0;
        if (any_remote) {
            ncclResult_t rc = g_rccl.GroupStart();
            for (uint32_t r = 0; r < R && rc == ncclSuccess; ++r) {
                if (m->comm_of_rank[r] == 0) continue;
                const uint64_t count = doubles_of_rank(cam, r, R);
                if (count == 0) continue;
                if (want_u8) {
                    rc = g_rccl.Send(m->tiles_u8[r], count, ncclUint8, 0, m->comm[m->comm_of_rank[r]], m->stream[r]);
                    if (rc == ncclSuccess)
                        rc = g_rccl.Recv(m->gathered_u8 + (size_t)r * n0, count, ncclUint8, m->comm_of_rank[r], m->comm[0],
                                         m->stream[0]);
                } else {
                    rc = g_rccl.Send(m->tiles[r], count, ncclDouble, 0, m->comm[m->comm_of_rank[r]], m->stream[r]);
                    if (rc == ncclSuccess)
                        rc = g_rccl.Recv(m->gathered + (size_t)r * n0, count, ncclDouble, m->comm_of_rank[r], m->comm[0],
                                         m->stream[0]);
                }
            }
            const ncclResult_t end = g_rccl.GroupEnd();
            if (rc == ncclSuccess) rc = end;
            if (rc != ncclSuccess) {
                const std::string msg = std::string("RCCL send/recv: ") + g_rccl.GetErrorString(rc);
                return rtc::set_error(RTC_ERR_DEVICE, msg.c_str());
            }
        }
        for (uint32_t r = 0; r < R; ++r) {
            if (m->comm_of_rank[r] != 0) continue;
            MULTI_HIP(hipSetDevice(m->device[r]));
            const uint64_t count = doubles_of_rank(cam, r, R);
            if (count == 0) continue;
            if (want_u8)
                MULTI_HIP(hipMemcpyAsync(m->gathered_u8 + (size_t)r * n0, m->tiles_u8[r], count, hipMemcpyDeviceToDevice,
                                         m->stream[r]));
            else
                MULTI_HIP(hipMemcpyAsync(m->gathered + (size_t)r * n0, m->tiles[r], count * sizeof(double),
                                         hipMemcpyDeviceToDevice, m->stream[r]));
        }
    }

    // ---- rank 0 waits for every rank, then finishes the canvas ----------------------------------------
    for (uint32_t r = 1; r < R; ++r) {
        MULTI_HIP(hipSetDevice(m->device[r]));
        MULTI_HIP(hipEventRecord(m->done[r], m->stream[r]));
    }
    MULTI_HIP(hipSetDevice(m->device[0]));
    for (uint32_t r = 1; r < R; ++r) MULTI_HIP(hipStreamWaitEvent(m->stream[0], m->done[r], 0));
    MULTI_HIP(hipEventRecord(m->t_rendered, m->stream[0]));
    if (tiled && mode == RTC_MULTI_RCCL_GATHER) {
        if (want_u8)
            MULTI_RTC(rtc_deinterleave_u8_device(cam, R, n0, m->gathered_u8, m->canvas_u8, m->stream[0]));
        else
            MULTI_RTC(rtc_deinterleave_device(cam, R, n0, m->gathered, m->canvas, m->stream[0]));
    }
    if (want_u8 && !tiled) MULTI_RTC(rtc_quantize_u8_device(m->canvas, values, m->canvas_u8, m->stream[0]));
    MULTI_HIP(hipEventRecord(m->t_end, m->stream[0]));
    if (stats) {
        total.pixels = (uint64_t)cam->hsize * cam->vsize;
        *stats = total;
    }
    return RTC_OK;
}

}  // namespace

extern "C" {

int rtc_multi_create(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light, const int32_t* devices,
                     uint32_t n_ranks, RtcMulti** out_multi) {
    if (!out_multi) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "out_multi is NULL");
    *out_multi = nullptr;
    if (n_ranks == 0 || n_ranks > RTC_MULTI_MAX_RANKS)
        return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "n_ranks must be between 1 and RTC_MULTI_MAX_RANKS");
    int visible = 0;
    MULTI_HIP(hipGetDeviceCount(&visible));
    if (visible < 1) return rtc::set_error(RTC_ERR_DEVICE, "no GPU visible");
    int previous = 0;
    MULTI_HIP(hipGetDevice(&previous));
    RtcMulti* m = new (std::nothrow) RtcMulti();
    if (!m) return rtc::set_error(RTC_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->ranks = n_ranks;
    m->device.resize(n_ranks);
    m->scene.assign(n_ranks, nullptr);
    m->stream.assign(n_ranks, nullptr);
    m->done.assign(n_ranks, nullptr);
    m->tiles.assign(n_ranks, nullptr);
    m->tiles_u8.assign(n_ranks, nullptr);
    m->tiles_doubles.assign(n_ranks, 0);
    m->tiles_u8_bytes.assign(n_ranks, 0);
    int rc = RTC_OK;
    for (uint32_t r = 0; r < n_ranks && rc == RTC_OK; ++r) {
        const int d = devices ? devices[r] : (int)(r % (uint32_t)visible);
        if (d < 0 || d >= visible) {
            rc = rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "device index out of range");
            break;
        }
        m->device[r] = d;
    }
    for (uint32_t r = 0; r < n_ranks && rc == RTC_OK; ++r) {
        hipError_t e = hipSetDevice(m->device[r]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->done[r], hipEventDisableTiming);
        if (e != hipSuccess) {
            rc = rtc::set_hip_error("stream/event of a rank", e);
            break;
        }
        rc = rtc_scene_create(shapes, n_shapes, light, &m->scene[r]);
        if (rc != RTC_OK) break;
        // ranks on another GPU than rank 0 reach rank 0's canvas through peer access
        if (m->device[r] != m->device[0]) {
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, m->device[r], m->device[0]);
            if (e == hipSuccess && can) {
                e = hipDeviceEnablePeerAccess(m->device[0], 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    e = hipSuccess;
                }
            }
            if (e != hipSuccess || !can) {
                m->peer_ok = false;
                char buf[96];
                std::snprintf(buf, sizeof buf, "GPU %d cannot address GPU %d", m->device[r], m->device[0]);
                m->peer_why = buf;
                (void)hipGetLastError();
            }
        }
    }
    if (rc == RTC_OK) {
        hipError_t e = hipSetDevice(m->device[0]);
        if (e == hipSuccess) e = hipEventCreate(&m->t_begin);
        if (e == hipSuccess) e = hipEventCreate(&m->t_rendered);
        if (e == hipSuccess) e = hipEventCreate(&m->t_end);
        if (e != hipSuccess) rc = rtc::set_hip_error("timing events", e);
    }
    if (rc != RTC_OK) {
        const std::string keep = rtc_last_error();
        destroy_multi(m);
        (void)hipSetDevice(previous);
        return rtc::set_error(rc, keep.c_str());
    }
    (void)hipSetDevice(previous);
    *out_multi = m;
    return RTC_OK;
}

int rtc_multi_destroy(RtcMulti* multi) {
    int previous = 0;
    const bool have_previous = hipGetDevice(&previous) == hipSuccess;
    destroy_multi(multi);
    if (have_previous) (void)hipSetDevice(previous);
    return RTC_OK;
}

int rtc_multi_set_integrator(RtcMulti* multi, int32_t integrator) {
    if (!multi) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "multi is NULL");
    int previous = 0;
    MULTI_HIP(hipGetDevice(&previous));
    int rc = RTC_OK;
    for (uint32_t r = 0; r < multi->ranks && rc == RTC_OK; ++r) {
        const hipError_t e = hipSetDevice(multi->device[r]);
        rc = e != hipSuccess ? rtc::set_hip_error("hipSetDevice", e) : rtc_scene_set_integrator(multi->scene[r], integrator);
    }
    (void)hipSetDevice(previous);
    return rc;
}

int rtc_multi_rank_device(const RtcMulti* multi, uint32_t rank, int32_t* out_device) {
    if (!multi || !out_device || rank >= multi->ranks) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "bad rank or NULL pointer");
    *out_device = multi->device[rank];
    return RTC_OK;
}

static int finish_frame(RtcMulti* m, RtcMultiTiming* timing, std::chrono::steady_clock::time_point t0, void* dst,
                        const void* d_src, size_t bytes) {
    MULTI_HIP(hipSetDevice(m->device[0]));
    MULTI_HIP(hipEventSynchronize(m->t_end));
    const auto t1 = std::chrono::steady_clock::now();
    if (dst && d_src == nullptr) {
        // host-direct: the frame lies in the library's pinned canvas; a few threads move it on (the caller's
        // buffer is usually fresh pageable memory, whose first touch is what takes the time)
        const size_t kSlice = 8u << 20;
        const size_t n_slices = (bytes + kSlice - 1) / kSlice;
        std::atomic<size_t> next{0};
        auto mover = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_slices) return;
                const size_t begin = k * kSlice, len = bytes - begin < kSlice ? bytes - begin : kSlice;
                std::memcpy((char*)dst + begin, (const char*)m->h_canvas + begin, len);
            }
        };
        std::vector<std::thread> helpers;
        const size_t n_helpers = n_slices > 1 ? (n_slices - 1 < 7 ? n_slices - 1 : 7) : 0;
        for (size_t t = 0; t < n_helpers; ++t) helpers.emplace_back(mover);
        mover();
        for (std::thread& t : helpers) t.join();
    } else if (dst) {
        MULTI_RTC(rtc::copy_device_to_host(m->scene[0], d_src, dst, bytes));
    }
    const auto t2 = std::chrono::steady_clock::now();
    if (timing) {
        float a = 0, b = 0;
        MULTI_HIP(hipEventElapsedTime(&a, m->t_begin, m->t_rendered));
        MULTI_HIP(hipEventElapsedTime(&b, m->t_rendered, m->t_end));
        timing->render_and_transfer_ms = a;
        timing->assemble_ms = b;
        timing->device_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        timing->to_host_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    }
    return RTC_OK;
}

// ---- one process per GPU: rank 0's canvas mapped into the other ranks' address spaces ----------
int rtc_ipc_canvas_create(uint64_t bytes, void** out_d_canvas, uint8_t out_handle[RTC_IPC_HANDLE_BYTES]) {
    static_assert(sizeof(hipIpcMemHandle_t) <= RTC_IPC_HANDLE_BYTES, "handle fits the caller's buffer");
    if (!out_d_canvas || !out_handle || bytes == 0) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "NULL pointer or zero size");
    *out_d_canvas = nullptr;
    void* p = nullptr;
    MULTI_HIP(hipMalloc(&p, bytes));
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return rtc::set_hip_error("hipIpcGetMemHandle", e);
    }
    std::memset(out_handle, 0, RTC_IPC_HANDLE_BYTES);
    std::memcpy(out_handle, &h, sizeof h);
    *out_d_canvas = p;
    return RTC_OK;
}

int rtc_ipc_canvas_open(const uint8_t handle[RTC_IPC_HANDLE_BYTES], void** out_d_canvas) {
    if (!handle || !out_d_canvas) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "NULL pointer");
    *out_d_canvas = nullptr;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void* p = nullptr;
    MULTI_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *out_d_canvas = p;
    return RTC_OK;
}

int rtc_ipc_canvas_close(void* d_canvas, int32_t is_owner) {
    if (!d_canvas) return RTC_OK;
    if (is_owner)
        MULTI_HIP(hipFree(d_canvas));
    else
        MULTI_HIP(hipIpcCloseMemHandle(d_canvas));
    return RTC_OK;
}

int rtc_render_multi(RtcMulti* multi, const RtcCameraDesc* camera, int32_t depth, int32_t mode, double* out_rgb,
                     RtcStats* stats, RtcMultiTiming* timing) {
    if (!multi) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "multi is NULL");
    if (!camera) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "camera is NULL");
    int previous = 0;
    MULTI_HIP(hipGetDevice(&previous));
    const auto t0 = std::chrono::steady_clock::now();
    const bool host_direct = mode == RTC_MULTI_HOST_DIRECT && multi->ranks > 1;
    if (host_direct && !out_rgb)
        return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "RTC_MULTI_HOST_DIRECT assembles the canvas in host memory: out_rgb is NULL");
    int rc = render_frame(multi, camera, depth, mode, false, stats);
    if (rc == RTC_OK)
        rc = finish_frame(multi, timing, t0, out_rgb, host_direct ? nullptr : multi->canvas,
                          (size_t)camera->hsize * camera->vsize * 3 * sizeof(double));
    (void)hipSetDevice(previous);
    return rc;
}

int rtc_render_multi_u8(RtcMulti* multi, const RtcCameraDesc* camera, int32_t depth, int32_t mode, uint8_t* out_rgb8,
                        RtcStats* stats, RtcMultiTiming* timing) {
    if (!multi) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "multi is NULL");
    if (!camera) return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "camera is NULL");
    int previous = 0;
    MULTI_HIP(hipGetDevice(&previous));
    const auto t0 = std::chrono::steady_clock::now();
    const bool host_direct = mode == RTC_MULTI_HOST_DIRECT && multi->ranks > 1;
    if (host_direct && !out_rgb8)
        return rtc::set_error(RTC_ERR_INVALID_ARGUMENT, "RTC_MULTI_HOST_DIRECT assembles the canvas in host memory: out_rgb8 is NULL");
    int rc = render_frame(multi, camera, depth, mode, true, stats);
    if (rc == RTC_OK)
        rc = finish_frame(multi, timing, t0, out_rgb8, host_direct ? nullptr : multi->canvas_u8,
                          (size_t)camera->hsize * camera->vsize * 3);
    (void)hipSetDevice(previous);
    return rc;
}

}  // extern "C"
