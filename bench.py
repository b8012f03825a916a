#!/usr/bin/env python3
"""Benchmark of the render path: one step = one frame of the headline workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (default `c3`): the reference's Ch. 11 reflection/refraction scene
(draw_world.zig:29-192) at 3840x2160, depth 5 — the configuration the metric is quoted on.
With N > 1 ranks the same frame is cut into interleaved 64x8 tiles (rank = tile % N); every
rank renders its tiles, the tile buffers are gathered to rank 0 over RCCL and scattered back
into the row-major canvas there.  Total work is fixed as N grows: "scaling": "strong".

The scene and the framebuffers are resident in HBM before the timed region starts; nothing
is copied to the host inside it.  `value` is radiance rays per second (primary + reflected +
refracted = calc.worldColorAt invocations) over the whole job; shadow rays are reported
separately in `config`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic flop model (every mul/add/sub/div/sqrt counted as one flop, the reference's
# redundant per-call matrix inverses NOT counted):
#   per ray-sphere test  54  (ray.zig:31-36 two transforms 18+15, shape.zig:29-33 a 5, b 6, c 6, disc 4)
#   per sphere hit       +6  (shape.zig:38-39 sqrt, two +-, 2*a, two divisions)
#   per ray-plane test   11  (the y rows of the two transforms 6+5; shape.zig:62)
#   per plane hit        +2  (shape.zig:65 negate, divide)
#   per shaded hit      190  (prepareComputations 75, shadow ray set-up 12, lighting 60 incl. pow,
#                             child set-up / Schlick / unwinding 43; patterns add ~40 where present)
FLOPS_SPHERE_HIT, FLOPS_PLANE_HIT, FLOPS_SHADED_HIT = 6, 2, 190  # per-test prices come from rtc_sweep_group_info
PEAK_FP64_NO_FMA_TFLOPS = 39.3  # 256 CU x 64 lanes x 2.4 GHz, one flop per lane-op (FMA forbidden for parity)
PEAK_FP64_FMA_TFLOPS = 78.6

WORKLOADS = {
    # name: (scene builder, hsize, vsize, description)
    "c2": ("draw_world", 1920, 1080, "draw_world.zig Ch.11 scene 1920x1080 depth 5"),
    "c3": ("draw_world", 3840, 2160, "draw_world.zig Ch.11 scene 3840x2160 depth 5"),
    "c4": ("synthetic", 3840, 2160, "synthetic 10k spheres+planes seed 0x5EED 3840x2160 depth 5"),
    "c5": ("draw_world", 7680, 4320, "draw_world.zig Ch.11 scene 7680x4320 depth 5"),
}


def algorithmic_flops(st, groups, table):
    """The reference's own arithmetic: every (ray, shape) pair at the full price of its kind
    (rtc_sweep_group_info: flops_reference), plus the roots of the pairs that produce t values,
    plus the shading of every hit."""
    rays = st["radiance_rays"] + st["shadow_rays"]
    per_ray = sum(n * g["flops_reference"] for n, g in zip(groups, table))
    return (
        rays * per_ray
        + FLOPS_SPHERE_HIT * st["sphere_hits"]
        + FLOPS_PLANE_HIT * st["plane_hits"]
        + FLOPS_SHADED_HIT * st["shaded_hits"]
    )


def executed_flops(st, groups, table):
    """Flops the sweep actually executes, group by group (rtc_sweep_group_info: flops_executed):
    groups whose inverses have exact zeros skip the products with those zeros (values unchanged):
    30 instead of 54 per ray-sphere test, 3 instead of 11 per ray-plane test; every ray-sphere
    pair first pays 17 flops for the bounding-sphere test (5 more per sweep), and the pairs it
    dismisses pay nothing else (the dismissed pairs are taken as spread evenly over the sphere
    groups); in a world too large for the on-chip tables a single-precision pre-test runs ahead of
    that test, and the pairs it dismisses (sphere_prefilter_skips) pay no double-precision flop
    at all (its own single-precision operations are not counted: the peak is an FP64 peak); a plane that a shadow ray's origin shares a side of with the light costs its origin
    row only (2 of 3, or 6 of 11 flops; taken as spread evenly over the plane groups).  Root
    filters and bookkeeping compares are not counted."""
    rays = st["radiance_rays"] + st["shadow_rays"]
    pre = st.get("sphere_prefilter_skips", 0) / max(st["sphere_tests"], 1)
    kept = 1.0 - pre - st["sphere_bound_skips"] / max(st["sphere_tests"], 1)
    side = st["plane_side_skips"] / max(st["plane_tests"], 1)
    per_ray = 5.0
    for n, g in zip(groups, table):
        if g["geo"] == 0:  # RTC_GEO_SPHERE
            per_ray += n * (17 * (1.0 - pre) + kept * g["flops_executed"])
        elif g["geo"] == 1:  # RTC_GEO_PLANE: the direction row is 1 of 3, or 5 of 11 flops
            per_ray += n * (g["flops_executed"] - side * (1 if g["axis_aligned"] else 5))
        else:
            per_ray += n * g["flops_executed"]
    return (
        rays * per_ray
        + FLOPS_SPHERE_HIT * st["sphere_hits"]
        + FLOPS_PLANE_HIT * st["plane_hits"]
        + FLOPS_SHADED_HIT * st["shaded_hits"]
    )


def pmc_counters(workload, n_gpus):
    """Hardware counters of one render_kernel launch from the committed PMC passes of this same
    command (profiles/pmc_counters.json, written by scripts/pmc_to_json.py; hardware counters
    cannot be read inside a timed run); None when no pass covers the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as f:
            entry = json.load(f).get(workload)
    except (OSError, ValueError):
        return None
    if not entry or entry.get("n_gpus") != n_gpus:
        return None
    return entry


def _device_view(torch, pointer, n_doubles, device):
    """A torch view of device memory the library allocated (the IPC canvas), without a copy."""

    class _Raw:
        __cuda_array_interface__ = {"shape": (n_doubles,), "typestr": "<f8", "data": (int(pointer), False), "version": 2}

    return torch.as_tensor(_Raw(), device=device)


def single_process_multi(native, D, world, cam, args, hsize, vsize):
    """rtc_render_multi (one process drives every rank) in each transfer mode: wall clock per frame with
    the canvas left on rank 0's GPU, checked against rtc_render."""
    import numpy as np

    res = {"ranks": args.multi_ranks, "visible_gpus": native.device_count()}
    with native.Scene(world.shape_descs(), len(world.objects), world.light_desc()) as sc:
        want = sc.render(cam, args.depth)
    with native.Multi(world.shape_descs(), len(world.objects), world.light_desc(), args.multi_ranks) as m:
        for name, mode in (("peer_stores", D.RTC_MULTI_PEER_STORES), ("peer_block_stores", D.RTC_MULTI_PEER_BLOCK_STORES),
                           ("peer_scatter", D.RTC_MULTI_PEER_SCATTER), ("rccl_gather", D.RTC_MULTI_RCCL_GATHER)):
            try:
                same = bool(np.array_equal(m.render(cam, args.depth, mode), want))
                for _ in range(args.warmup):
                    m.render(cam, args.depth, mode, to_host=False)
                tm = D.RtcMultiTiming()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    m.render(cam, args.depth, mode, None, tm, to_host=False)
                dt = (time.perf_counter() - t0) / args.steps
                res[name] = {"frame_ms": dt * 1e3, "identical_to_rtc_render": same, "last_call": tm.as_dict()}
            except native.RtcError as e:
                res[name] = {"error": str(e)}
        # with the canvas delivered to the caller's host buffer: rank 0's GPU copying the assembled canvas out
        # (one PCIe link) against every rank scattering its tiles into pinned host memory over its own link
        for name, mode in (("peer_stores_to_host", D.RTC_MULTI_PEER_STORES), ("host_direct_to_host", D.RTC_MULTI_HOST_DIRECT)):
            try:
                same = bool(np.array_equal(m.render(cam, args.depth, mode), want))
                tm = D.RtcMultiTiming()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    m.render(cam, args.depth, mode, None, tm)
                dt = (time.perf_counter() - t0) / args.steps
                res[name] = {"frame_ms": dt * 1e3, "identical_to_rtc_render": same, "last_call": tm.as_dict()}
            except native.RtcError as e:
                res[name] = {"error": str(e)}
    return res


def weighted_integrator_leg(torch, native, D, scene, cam, args, exact, stream, radiance_rays):
    """The same frame with rtc_scene_set_integrator(RTC_INTEGRATOR_WEIGHTED): the attenuation-weighted sum the
    task description sketches (same rays, same hits; products and sums re-associated).  Reported beside the
    headline, which stays the integrator whose canvas equals the reference's value for value; the weighted
    canvas is checked here against that one at the stated bar (|delta| <= 1e-9; 8-bit values identical or
    1 LSB apart on at most 0.01 % of the channels)."""
    out = torch.empty_like(exact)
    try:
        scene.set_integrator(D.RTC_INTEGRATOR_WEIGHTED)
        for _ in range(max(args.warmup, 1)):
            scene.render_device(cam, args.depth, out.data_ptr(), 0, 1, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            scene.render_device(cam, args.depth, out.data_ptr(), 0, 1, stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
    finally:
        scene.set_integrator(D.RTC_INTEGRATOR_POSTORDER)
    delta = float((out - exact).abs().max())
    q = lambda t: (t * 256.0).clamp(0.0, 255.0).to(torch.uint8).to(torch.int16)  # byteColorValue, ppm.zig:6-9
    d8 = (q(out) - q(exact)).abs()
    differing = float((d8 != 0).double().mean())
    ok = delta <= 1e-9 and int(d8.max()) <= 1 and differing <= 1e-4
    return {"frame_ms": ms if ok else None, "rays_per_s": radiance_rays / (ms * 1e-3) if ok else None,
            "max_abs_delta_vs_exact_canvas": delta, "share_of_8bit_channels_differing": differing,
            "max_8bit_difference": int(d8.max()), "within_stated_tolerance": ok}


def two_streams_that_overlap(torch, launch):
    """Two streams whose kernels run side by side.  The runtime deals its streams over a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) and launches of streams that share a queue run one after the other, so a
    pair is picked by measurement: a few alternating launches (`launch(which, stream)`, which = 0 or 1) with one
    stream and each of three partners, the fastest pair wins."""
    first = torch.cuda.Stream()
    best = None
    for _ in range(3):
        partner = torch.cuda.Stream()
        pair = (first, partner)
        for k in range(4):
            launch(k % 2, pair[k % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(6):
            launch(k % 2, pair[k % 2])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, partner)
    return first, best[1]


def frames_in_flight_leg(torch, native, D, world, cam, args, exact, radiance_rays):
    """A stream of frames instead of one frame at a time: two scene handles of the same world (a handle owns its
    work queue, so a handle has one frame in flight), two streams, two canvases, frames launched alternately.  The
    waves of the next frame start on the compute units the tail of the current frame leaves idle.  Every frame is
    complete and both canvases are compared with the headline's; reported beside the headline, which stays the
    time of one frame on its own."""
    scenes_ = [native.Scene(world.shape_descs(), len(world.objects), world.light_desc()) for _ in range(2)]
    try:
        outs = [torch.empty_like(exact) for _ in range(2)]
        streams = two_streams_that_overlap(
            torch, lambda which, st: scenes_[which].render_device(cam, args.depth, outs[which].data_ptr(), 0, 1, st.cuda_stream))
        for k in range(2 * max(args.warmup, 3)):
            scenes_[k % 2].render_device(cam, args.depth, outs[k % 2].data_ptr(), 0, 1, streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            scenes_[k % 2].render_device(cam, args.depth, outs[k % 2].data_ptr(), 0, 1, streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)
        ok = bool(torch.equal(outs[0], exact) and torch.equal(outs[1], exact))
    finally:
        for sc in scenes_:
            sc.close()
    return {"frames_in_flight": 2, "ms_per_frame": ms if ok else None,
            "rays_per_s": radiance_rays / (ms * 1e-3) if ok else None, "canvases_identical_to_the_headline_s": ok}


def cpu_baseline(world, camera, depth, budget_s=20.0):
    """Single-thread CPU baseline on this host: the oracle (a port of the reference) on a
    bounded sample of the same frame.  Pixels are sampled with a fixed stride; the literal
    integrator mirrors the reference's cost model (per-call inverse, heap lists, stable
    sort), the streaming one is the strong CPU baseline (hoisted inverses, running min)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from raytracer_challenge_amd import descs as D

    import numpy as np

    n_pixels = camera.hsize * camera.vsize
    # one core, and always the same one: the first the process is allowed on
    pinned_to = None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {allowed[0]})
        pinned_to = allowed[0]
    except (AttributeError, OSError):
        allowed = None

    def run(mode, n_sample):
        # every k-th pixel of the frame in row-major order, k odd so that the samples do not line up in columns
        n_sample = int(min(max(n_sample, 16), n_pixels))
        k = max(n_pixels // n_sample, 1) | 1
        idx = np.arange(k // 2, n_pixels, k, dtype=np.uint64)
        st = D.RtcStats()
        t0 = time.perf_counter()
        O.render_pixels(world, camera, idx, depth, mode, stats=st)
        dt = time.perf_counter() - t0
        return st.radiance_rays / dt, st.radiance_rays, dt, idx.size, k

    # calibrate the sample sizes so that each leg takes a bounded time whatever the host and the scene
    _, _, probe_dt, probe_n, _ = run(O.STREAMING, 64)
    s_rate, s_rays, s_dt, s_px, s_k = run(O.STREAMING, probe_n * budget_s * 0.25 / max(probe_dt, 1e-6))
    _, _, probe_dt, probe_n, _ = run(O.LITERAL, 16)
    l_rate, l_rays, l_dt, l_px, l_k = run(O.LITERAL, probe_n * budget_s * 0.7 / max(probe_dt, 1e-6))
    if allowed:
        os.sched_setaffinity(0, set(allowed))
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": l_rate,
        "unit": "rays/s",
        "cores": 1,
        "kind": "port",
        "variant": "literal integrator (reference cost model: per-call inverse, heap lists, stable sort)",
        "sample": f"every {l_k}th pixel of the frame: {l_px} pixels, {l_rays} radiance rays in {l_dt:.1f} s",
        "streaming_value": s_rate,
        "streaming_sample": f"every {s_k}th pixel: {s_px} pixels, {s_rays} radiance rays in {s_dt:.1f} s",
        "cpu_model": cpu_model,
        "pinned_to_core": pinned_to,
        "host_cores_visible": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--accel", action="store_true",
                    help="sweep the spheres through the optional hierarchy (rtc_scene_set_acceleration); the default is "
                    "the reference's list walk, which the roofline accounting describes")
    ap.add_argument("--integrator", default="postorder", choices=["postorder", "weighted"],
                    help="what every timed leg renders with: the reference's order of summation (default: the canvas is "
                    "value-identical to the reference's) or the attenuation-weighted sum (same rays; colours within a few "
                    "1e-16, canvases then compared at 1e-9 instead of exactly; no in-place peer stores)")
    ap.add_argument("--canvas-store", default="pixels", choices=["pixels", "blocks"],
                    help="rtc_scene_set_canvas_store for the headline leg: every finished pixel stored at once (default) "
                    "or whole 8x8 blocks (meant for a canvas in a peer GPU's memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-leg", action="store_true",
                    help="leave out the stream-of-frames leg (its launches overlap each other, which would blur a "
                    "profiler's per-launch average of the render kernel)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    # developer switches to rehearse the N > 1 path on a box with one GPU: every rank uses GPU 0
    # and the gather goes through host memory over gloo (never used for reported numbers)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="rank 0 checks the assembled canvas against a 1-rank render")
    ap.add_argument("--multi-ranks", type=int, default=0,
                    help="(single process) also time rtc_render_multi with this many ranks mapped onto the visible GPUs, "
                    "every transfer mode; reported under config.single_process_multi, never as `value`")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracer_challenge_amd import descs as D
    from raytracer_challenge_amd import native, scenes

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    native.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    kind, hsize, vsize, desc = WORKLOADS[args.workload]
    if kind == "draw_world":
        world, camera = scenes.draw_world_scene(hsize, vsize)
    else:
        world, camera = scenes.synthetic_scene(hsize, vsize)
    cam = camera.desc()
    scene = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
    if args.canvas_store == "blocks":
        scene.set_canvas_store(D.RTC_CANVAS_STORE_BLOCKS)
    if args.accel:
        scene.set_acceleration(D.RTC_ACCEL_BVH)
    weighted = args.integrator == "weighted"
    if weighted:
        scene.set_integrator(D.RTC_INTEGRATOR_WEIGHTED)

    def same_canvas(a, b):
        # the weighted integrator's sums arrive in varying order: its canvases agree to rounding, not bit for bit
        return bool(torch.equal(a, b)) if not weighted else bool((a - b).abs().max() <= 1e-9)

    n_local = native.tile_buffer_doubles(cam, rank, world_size)
    n_rank0 = native.tile_buffer_doubles(cam, 0, world_size)  # the largest tile buffer
    local = torch.empty(n_rank0 if world_size > 1 else n_local, dtype=torch.float64, device=dev)
    canvas = gathered = None
    if world_size > 1 and rank == 0:
        gathered = torch.empty((world_size, n_rank0), dtype=torch.float64, device=dev)
        canvas = torch.empty(hsize * vsize * 3, dtype=torch.float64, device=dev)

    # second leg for N > 1: quantise on each rank (ppm.zig:6-9) and gather 3 B instead of 24 B per pixel
    local_u8 = gathered_u8 = canvas_u8 = None
    if world_size > 1:
        local_u8 = torch.empty(n_rank0, dtype=torch.uint8, device=dev)
        if rank == 0:
            gathered_u8 = torch.empty((world_size, n_rank0), dtype=torch.uint8, device=dev)
            canvas_u8 = torch.empty(hsize * vsize * 3, dtype=torch.uint8, device=dev)

    stream = torch.cuda.current_stream().cuda_stream

    # third leg for N > 1: no gather at all.  Rank 0's canvas is mapped into every rank's address space
    # (HIP IPC) and each rank's render kernel stores its pixels at their final place in it over xGMI.
    def share_a_canvas():
        """Rank 0 allocates a canvas and every other rank maps it (HIP IPC): (pointer in this process or None, why not)."""
        mapped, why = None, None
        try:
            handle = torch.zeros(D.RTC_IPC_HANDLE_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                mapped, hbytes = native.ipc_canvas_create(hsize * vsize * 3 * 8)
                handle.copy_(torch.frombuffer(bytearray(hbytes), dtype=torch.uint8))
            if args.rehearse_on_one_gpu:
                hcpu = handle.cpu()
                dist.broadcast(hcpu, src=0)
                handle = hcpu
            else:
                dist.broadcast(handle, src=0)
            if rank != 0:
                # a kernel that stores through a mapping its GPU cannot address faults the GPU: ask first
                # (torch.distributed.run leaves every GPU visible, so rank 0's GPU is device 0 here too)
                if (not args.rehearse_on_one_gpu and local_rank != 0 and torch.cuda.device_count() > 1
                        and not torch.cuda.can_device_access_peer(local_rank, 0)):
                    raise RuntimeError(f"GPU {local_rank} cannot address the memory of GPU 0")
                mapped = native.ipc_canvas_open(bytes(handle.cpu().numpy().tobytes()))
        except Exception as e:  # no peer mapping on this machine: the leg is reported as unavailable
            why = f"{type(e).__name__}: {e}"
            mapped = None
        ok = torch.tensor([1 if mapped else 0], dtype=torch.int32, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return mapped, False, why or "another rank could not map the canvas"
        return mapped, True, None

    peer_canvas = None       # device pointer of rank 0's canvas as this process addresses it
    peer_canvas2 = None      # a second one, for the leg that keeps two frames in flight
    peer_why = None
    peer_canvas_usable = peer_canvas2_usable = False
    if world_size > 1 and weighted:
        peer_why = "the weighted integrator does not render in place into a shared canvas"
    if world_size > 1 and not weighted:
        peer_canvas, peer_canvas_usable, peer_why = share_a_canvas()
        if peer_canvas_usable:
            peer_canvas2, peer_canvas2_usable, _ = share_a_canvas()
    # fourth leg for N > 1: no GPU-to-GPU traffic at all.  Every rank copies its dense tile buffer to pinned
    # host memory of its own over its GPU's own PCIe link; the tiles of the frame then lie in N host buffers
    # (in one process, rtc_render_multi's RTC_MULTI_HOST_DIRECT scatters them into one host canvas instead).
    local_host = None
    host_why = None
    if world_size > 1:
        try:
            local_host = torch.empty(n_local, dtype=torch.float64, pin_memory=True)
        except Exception as e:  # no pinned memory to be had: the leg is reported as unavailable
            host_why = f"{type(e).__name__}: {e}"
        ok = torch.tensor([1 if local_host is not None else 0], dtype=torch.int32,
                          device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            local_host = None
            host_why = host_why or "another rank could not allocate pinned host memory"

    def render_and_copy_to_own_host_buffer():
        scene.render_device(cam, args.depth, local.data_ptr(), rank, world_size, stream)
        local_host.copy_(local[:n_local], non_blocking=True)
        torch.cuda.synchronize()
        dist.barrier()

    sync_token = torch.zeros(1, dtype=torch.int32, device=dev) if world_size > 1 else None

    def render_into_peer_canvas():
        """Every rank stores its tiles into rank 0's canvas; the all-reduce that follows on the same
        stream is the barrier after which rank 0 may read it."""
        scene.render_device_to_canvas(cam, args.depth, peer_canvas, rank, world_size, stream)
        if args.rehearse_on_one_gpu:
            torch.cuda.synchronize()
            dist.barrier()
        else:
            dist.all_reduce(sync_token)

    def gather_and_assemble():
        """RCCL gather of the per-rank tile buffers to rank 0, then scatter into the canvas."""
        if args.rehearse_on_one_gpu:
            host = local.cpu()
            parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                gathered.copy_(torch.stack(parts))
        else:
            dist.gather(local, list(gathered.unbind(0)) if rank == 0 else None, dst=0)
        if rank == 0:
            native.deinterleave_device(cam, world_size, n_rank0, gathered.data_ptr(), canvas.data_ptr(), stream)

    def gather_and_assemble_u8():
        native.quantize_u8_device(local.data_ptr(), n_rank0, local_u8.data_ptr(), stream)
        if args.rehearse_on_one_gpu:
            host = local_u8.cpu()
            parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                gathered_u8.copy_(torch.stack(parts))
        else:
            dist.gather(local_u8, list(gathered_u8.unbind(0)) if rank == 0 else None, dst=0)
        if rank == 0:
            native.deinterleave_u8_device(cam, world_size, n_rank0, gathered_u8.data_ptr(), canvas_u8.data_ptr(), stream)

    def step():
        scene.render_device(cam, args.depth, local.data_ptr(), rank, world_size, stream)
        if world_size > 1:
            gather_and_assemble()

    # deterministic work counters of this rank's share (outside the timed region)
    st = D.RtcStats()
    scene.render_device(cam, args.depth, local.data_ptr(), rank, world_size, stream, st)
    counts = torch.tensor(
        [st.radiance_rays, st.shadow_rays, st.sphere_tests, st.plane_tests, st.sphere_hits, st.plane_hits,
         st.shaded_hits, st.sphere_bound_skips, st.plane_side_skips, st.cube_tests, st.cube_hits,
         st.sphere_prefilter_skips],
        dtype=torch.int64, device=dev,
    )
    if world_size > 1:
        dist.all_reduce(counts)
    names = ["radiance_rays", "shadow_rays", "sphere_tests", "plane_tests", "sphere_hits", "plane_hits", "shaded_hits",
             "sphere_bound_skips", "plane_side_skips", "cube_tests", "cube_hits", "sphere_prefilter_skips"]
    total = dict(zip(names, (int(v) for v in counts.tolist())))
    group_table = native.sweep_group_table()
    local_flops = algorithmic_flops(st.as_dict(), scene.group_counts(), group_table)
    local_executed = executed_flops(st.as_dict(), scene.group_counts(), group_table)

    for _ in range(args.warmup):
        step()

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # HIP events on the stream the render kernel is launched on, around that launch only
        ev[i][0].record()
        scene.render_device(cam, args.depth, local.data_ptr(), rank, world_size, stream)
        ev[i][1].record()
        if world_size > 1:
            gather_and_assemble()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)

    elapsed_u8 = 0.0
    if world_size > 1:
        # the same frames with the 8-bit gather (reported beside the headline, never as `value`)
        scene.render_device(cam, args.depth, local.data_ptr(), rank, world_size, stream)
        gather_and_assemble_u8()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            scene.render_device(cam, args.depth, local.data_ptr(), rank, world_size, stream)
            gather_and_assemble_u8()
        fence()
        elapsed_u8 = time.perf_counter() - t0

    elapsed_peer = elapsed_peer_blocks = 0.0
    peer_blocks_copy = None
    if world_size > 1 and peer_canvas_usable:
        # the same leg with the kernels' stores a finished 8x8 block at a time (192-byte runs over the link), first:
        # its canvas is kept for the comparison below, then the per-pixel stores overwrite it
        scene.set_canvas_store(D.RTC_CANVAS_STORE_BLOCKS)
        for _ in range(max(args.warmup, 1)):
            render_into_peer_canvas()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            render_into_peer_canvas()
        fence()
        elapsed_peer_blocks = time.perf_counter() - t0
        if rank == 0:
            peer_blocks_copy = _device_view(torch, peer_canvas, hsize * vsize * 3, dev).clone()
        scene.set_canvas_store(D.RTC_CANVAS_STORE_PIXELS)
        fence()
        for _ in range(max(args.warmup, 1)):
            render_into_peer_canvas()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            render_into_peer_canvas()
        fence()
        elapsed_peer = time.perf_counter() - t0

    # The peer-store leg as a stream of frames: two scene handles, two streams and two shared canvases per rank,
    # frames launched alternately, so that the tail of one frame's share (the serial chains of single glass pixels)
    # overlaps with the next frame's start on every GPU.  No per-frame completion token here: the ranks meet once,
    # after the last frame (a consumer on rank 0 would need one per frame, on a third stream).
    elapsed_stream = 0.0
    stream_copies = None
    if world_size > 1 and peer_canvas_usable and peer_canvas2_usable:
        scene2 = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
        both = (scene, scene2), (peer_canvas, peer_canvas2)
        pair = two_streams_that_overlap(
            torch, lambda which, st_: both[0][which].render_device_to_canvas(cam, args.depth, both[1][which], rank, world_size, st_.cuda_stream))
        lanes_ = ((scene, peer_canvas, pair[0].cuda_stream), (scene2, peer_canvas2, pair[1].cuda_stream))
        for k in range(2 * max(args.warmup, 3)):
            h, c, st_ = lanes_[k % 2]
            h.render_device_to_canvas(cam, args.depth, c, rank, world_size, st_)
        torch.cuda.synchronize()
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            h, c, st_ = lanes_[k % 2]
            h.render_device_to_canvas(cam, args.depth, c, rank, world_size, st_)
        torch.cuda.synchronize()  # this rank's stores have arrived ...
        fence()                   # ... and so have everybody else's
        elapsed_stream = time.perf_counter() - t0
        if rank == 0:
            stream_copies = [_device_view(torch, c, hsize * vsize * 3, dev).clone() for c in (peer_canvas, peer_canvas2)]
        scene2.close()
        fence()
        # (the canvas of the per-pixel peer leg is put back for the comparison below)
        render_into_peer_canvas()
        fence()

    elapsed_host = 0.0
    if world_size > 1 and local_host is not None:
        for _ in range(max(args.warmup, 1)):
            render_and_copy_to_own_host_buffer()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            render_and_copy_to_own_host_buffer()
        fence()
        elapsed_host = time.perf_counter() - t0
        # what arrived in the host buffers goes through the (already verified) gather once more, outside the
        # timed region, and is compared with the 1-rank render below
        local[:n_local].copy_(local_host)
        gather_and_assemble()
        fence()
        host_canvas_copy = canvas.clone() if rank == 0 else None
        # (and the canvas of the headline leg is put back)
        step()
        fence()

    t = torch.tensor([elapsed, kernel_ms, elapsed_u8, elapsed_peer, elapsed_host, elapsed_peer_blocks, elapsed_stream],
                     dtype=torch.float64, device=dev)
    if world_size > 1:
        if args.rehearse_on_one_gpu:
            tc = t.cpu()
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            t = tc
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    (elapsed_max, kernel_ms_max, elapsed_u8_max, elapsed_peer_max, elapsed_host_max, elapsed_peer_blocks_max,
     elapsed_stream_max) = (float(v) for v in t.tolist())
    # how even the shares are: every rank's own kernel time and ray count (an uneven partition shows here)
    mine = torch.tensor([kernel_ms, float(st.radiance_rays)], dtype=torch.float64,
                        device="cpu" if (world_size > 1 and args.rehearse_on_one_gpu) else dev)
    per_rank = [mine]
    if world_size > 1:
        per_rank = [torch.empty_like(mine) for _ in range(world_size)]
        dist.all_gather(per_rank, mine)
    rank_kernel_ms = [float(v[0]) for v in per_rank]
    rank_rays = [int(v[1]) for v in per_rank]

    # Every N > 1 run checks what it assembled: rank 0 renders the whole frame alone (outside the timed
    # regions) and compares the canvases value for value.  A leg that differs is not reported.
    gather_ok = peer_ok = host_ok = peer_blocks_ok = stream_ok = None
    if world_size > 1 and rank == 0:
        whole = torch.empty(hsize * vsize * 3, dtype=torch.float64, device=dev)
        scene.render_device(cam, args.depth, whole.data_ptr(), 0, 1, stream)
        torch.cuda.synchronize()
        gather_ok = same_canvas(whole, canvas)
        if local_host is not None:
            host_ok = same_canvas(whole, host_canvas_copy)
        if peer_canvas_usable:
            peer_t = _device_view(torch, peer_canvas, hsize * vsize * 3, dev)
            peer_ok = bool(torch.equal(whole, peer_t))
            peer_blocks_ok = bool(torch.equal(whole, peer_blocks_copy)) if peer_blocks_copy is not None else None
            stream_ok = all(bool(torch.equal(whole, c)) for c in stream_copies) if stream_copies is not None else None

    if world_size > 1 and rank == 0:
        q = torch.empty(hsize * vsize * 3, dtype=torch.uint8, device=dev)
        native.quantize_u8_device(whole.data_ptr(), hsize * vsize * 3, q.data_ptr(), stream)
        torch.cuda.synchronize()
        if weighted:
            d8 = (q.to(torch.int16) - canvas_u8.to(torch.int16)).abs()
            u8_ok = bool(d8.max() <= 1 and (d8 != 0).double().mean() <= 1e-4)
        else:
            u8_ok = bool(torch.equal(q, canvas_u8))
        if args.verify:
            if not (gather_ok and u8_ok and peer_ok is not False and host_ok is not False and peer_blocks_ok is not False
                    and stream_ok is not False):
                raise SystemExit(f"verify FAILED: gather {gather_ok}, 8-bit gather {u8_ok}, peer stores {peer_ok}, "
                                 f"own D2H {host_ok}, block stores {peer_blocks_ok}, stream of frames {stream_ok}")
            print(f"verify ok: assembled canvases identical to the 1-rank render (f64 gather, 8-bit gather, "
                  f"peer stores: {peer_ok}, own D2H: {host_ok})", file=sys.stderr)

    if rank == 0:
        ms_gather = elapsed_max * 1e3 / args.steps
        ms_per_step = ms_gather
        headline_leg = "single GPU" if world_size == 1 else "RCCL gather of f64 tile buffers + de-interleave"
        if world_size > 1:
            ms_peer = elapsed_peer_max * 1e3 / args.steps if (peer_canvas_usable and peer_ok) else None
            if not gather_ok:
                raise SystemExit("the gathered canvas differs from the 1-rank render: no number is reported")
            # the headline is the faster of the two verified ways to the same f64 canvas on rank 0
            ms_peer_blocks = elapsed_peer_blocks_max * 1e3 / args.steps if (peer_canvas_usable and peer_blocks_ok) else None
            if ms_peer is not None and ms_peer < ms_per_step:
                ms_per_step = ms_peer
                headline_leg = "peer stores into rank 0's canvas (HIP IPC mapping, no gather)"
            if ms_peer_blocks is not None and ms_peer_blocks < ms_per_step:
                ms_per_step = ms_peer_blocks
                headline_leg = "peer stores into rank 0's canvas, a finished 8x8 block at a time (HIP IPC mapping, no gather)"
        rays_per_s = total["radiance_rays"] / (ms_per_step * 1e-3)
        # roofline of the dominant kernel (render_kernel) on THIS rank: algorithmic flops of
        # one launch / average launch duration from the HIP events above
        # `achieved` counts the flops the kernel EXECUTES for the reference's arithmetic (the model of
        # executed_flops above); the reference's own count (54 per ray-sphere, 11 per ray-plane test,
        # zeros and dismissed pairs included) is reported beside it as *_algorithmic
        sec = kernel_ms * 1e-3
        achieved = local_executed / sec / 1e12
        achieved_alg = local_flops / sec / 1e12
        frac_alg = achieved_alg / PEAK_FP64_NO_FMA_TFLOPS
        # (the committed counter passes are of the default command: depth 5, list walk)
        pmc = pmc_counters(args.workload, world_size) if args.depth == 5 and not args.accel else None
        roofline = {
            "bound": "fp64-valu",
            "achieved": achieved,
            "peak": PEAK_FP64_NO_FMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_NO_FMA_TFLOPS,
            "traffic": (pmc["read_bytes"] + pmc["write_bytes"]) if pmc else None,
            "kernel": "render_kernel",
            "kernel_ms": kernel_ms,
            # (with several ranks the headline may be the peer-store leg; the launches timed here are always these)
            "kernel_ms_measured_on": "rank 0's launches of rtc_render_device into its " +
            ("row-major canvas" if world_size == 1 else "dense tile buffer (the gather leg)"),
            "executed_flops_per_launch": local_executed,
            "algorithmic_flops_per_launch": local_flops,
            "achieved_algorithmic": achieved_alg,
            # the reference's count can exceed what any kernel could execute at peak when most of it
            # is products with exact zeros and pairs a bound dismisses: then it is no fraction of a peak
            "frac_algorithmic": frac_alg if frac_alg <= 1.0 else None,
            "peak_with_fma": PEAK_FP64_FMA_TFLOPS,
            "frac_of_fma_peak": achieved / PEAK_FP64_FMA_TFLOPS,
            "note": "compute-bound FP64 vector path (no dense contraction, so not MFMA; scene fits in cache, so "
            "not HBM); FMA contraction is forbidden by the parity contract, hence the 39.3 TFLOP/s peak (one "
            "flop per lane per issue). `achieved`/`frac` count executed flops: the reference's arithmetic "
            "minus products with exact zeros of axis-aligned inverses and minus ray-sphere pairs the "
            "bounding test dismisses, plus that test; `*_algorithmic` count the reference's 54 flops per "
            "ray-sphere and 11 per ray-plane test whether executed or not",
        }
        if pmc:
            # the flop model against the hardware's own count of f64 instructions (same workload, PMC pass)
            lanes = pmc["active_lanes_per_valu_inst"]
            wave = pmc["f64_add_wave_insts"] + pmc["f64_mul_wave_insts"] + pmc["f64_fma_wave_insts"]
            lane_ops = wave * lanes
            lane_flops = (wave + pmc["f64_fma_wave_insts"]) * lanes
            roofline["counter_check"] = {
                "f64_lane_ops_per_launch": lane_ops,
                "f64_lane_flops_per_launch_fma_as_2": lane_flops,
                "model_over_lane_flops": local_executed / lane_flops if lane_flops else None,
                "issue_frac_from_counters": lane_ops / sec / 1e12 / PEAK_FP64_NO_FMA_TFLOPS,
                "active_lanes_per_valu_inst": lanes,
                "valu_busy": pmc["valu_busy"],
                "f64_share_of_valu_insts": wave / pmc["valu_wave_insts"],
                "source": pmc["source"],
                "note": "f64 add+mul+fma wave-instructions x average active lanes (fused operations occur "
                "only in the bounding tests and inside division/sqrt expansions); counted once per "
                "instruction for the issue fraction, FMA twice for the flop comparison. The counters also "
                "see division, sqrt and pow expansions, which the model prices at one flop each",
            }
        if st.sphere_prefilter_skips:
            # a world too large for the on-chip tables: nearly every ray-sphere pair ends in the packed
            # single-precision pre-tests (per group of 512 spheres - per super-block of 64 in worlds of at most
            # 2048 spheres - for every ray, then per super-block, block of eight and sphere only where a ray
            # cannot dismiss the level above), which the FP64 figures above do not see.  A test costs 20
            # single-precision flops for a radiance ray (difference 3, two dot products 5 + 5, margin 2, the
            # clamped product 4, one product 1) and 10 for a shadow ray (dot product 5, margin 1, clamped
            # product 4).  Counted here: the top-level tests only, which every ray runs (how many tests follow
            # depends on the ray).
            groups_now = scene.group_counts()
            n_sph = sum(n for n, g in zip(groups_now, group_table) if g["geo"] == 0)
            top = 512 if n_sph > 2048 else 64
            n_blocks = (n_sph + top - 1) // top
            f32 = (st.radiance_rays * 20 + st.shadow_rays * 10) * n_blocks
            roofline["f32_top_level_test_flops_per_launch"] = f32
            roofline["pairs_dismissed_by_f32_pre_tests"] = st.sphere_prefilter_skips / max(st.sphere_tests, 1)
            # What bounds this kernel is the issue rate of the vector units, whatever the type of the instruction:
            # `achieved` = vector wave-instructions x average active lanes per launch (hardware counters of the same
            # command, profiles/pmc_counters.json) / the launch duration measured in THIS run; `peak` = 256 CUs x 64
            # lanes x 2.4 GHz lane-operations per second.  The FP64 flop figures stay, as secondary ones.
            fp64 = {k: roofline[k] for k in ("achieved", "frac", "executed_flops_per_launch", "algorithmic_flops_per_launch",
                                              "achieved_algorithmic", "frac_algorithmic", "peak_with_fma", "frac_of_fma_peak")}
            fp64["unit"] = "TFLOP/s"
            fp64["peak"] = PEAK_FP64_NO_FMA_TFLOPS
            for k in ("executed_flops_per_launch", "algorithmic_flops_per_launch", "achieved_algorithmic", "frac_algorithmic",
                      "peak_with_fma", "frac_of_fma_peak"):
                del roofline[k]
            lane_ops = pmc["valu_wave_insts"] * pmc["active_lanes_per_valu_inst"] if pmc else None
            roofline["bound"] = "valu-issue"
            roofline["unit"] = "T lane-ops/s"
            roofline["peak"] = PEAK_FP64_NO_FMA_TFLOPS  # the same 256 x 64 x 2.4e9 per second, counted in lane-operations
            roofline["achieved"] = lane_ops / sec / 1e12 if lane_ops else None
            roofline["frac"] = roofline["achieved"] / roofline["peak"] if lane_ops else None
            roofline["valu_lane_ops_per_launch"] = lane_ops
            roofline["fp64_flops_only"] = fp64
            roofline["note"] = ("vector-issue bound: in a world too large for the on-chip tables nearly every ray-sphere pair "
                                "ends in packed single-precision pre-tests (group of 512, super-block of 64, block of eight, "
                                "sphere), integer bookkeeping of the per-lane work lists and per-lane loads, so FP64 flops "
                                "describe a small part of the work (`fp64_flops_only`, 17 % of the vector instructions). "
                                "`achieved` counts every vector instruction of a launch times its active lanes (hardware "
                                "counters of this command, `counter_check.source`) over this run's launch duration; `frac` = "
                                "share of the chip's lane-cycles that issue a vector operation = VALU busy x active lanes / 64")
            if "counter_check" in roofline:
                # the few lanes that go on to the double test run it with most of the wave masked off, so
                # wave-instructions x average lanes over-states the FP64 lane operations
                roofline["counter_check"]["model_over_lane_flops"] = None
        out = {
            "metric": "rays/sec (primary+secondary) at 3840x2160 depth 5" if args.workload == "c3"
            else f"rays/sec (primary+secondary), workload {args.workload}",
            "value": rays_per_s,
            "unit": "rays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "shapes": len(world.objects),
                "pixels": hsize * vsize,
                "integrator": "attenuation-weighted sum (rtc_scene_set_integrator; colours within 1e-9 of the reference's order)"
                if weighted else "post-order, the reference's order of summation (value-identical canvas)",
                "partition": f"interleaved {D.RTC_TILE_W}x{D.RTC_TILE_H} tiles, rank = tile % {world_size}",
                "radiance_rays_per_frame": total["radiance_rays"],
                "shadow_rays_per_frame": total["shadow_rays"],
                "all_rays_per_s": (total["radiance_rays"] + total["shadow_rays"]) / (ms_per_step * 1e-3),
                "frame_ms": ms_per_step,
                "sweep": "hierarchy over the spheres (optional mode; flop figures below assume the list walk)"
                if args.accel else "every shape against every ray (calc.zig:204-209)",
                "canvas_store": args.canvas_store,
                "render_kernel_ms_max_over_ranks": kernel_ms_max,
                "render_kernel_ms_per_rank": {"min": min(rank_kernel_ms), "mean": sum(rank_kernel_ms) / len(rank_kernel_ms),
                                              "max": max(rank_kernel_ms)},
                "radiance_rays_per_rank": {"min": min(rank_rays), "max": max(rank_rays)},
            },
            "roofline": roofline,
        }
        if args.accel:
            # the flop model describes the list walk, not the hierarchy's traversal
            for k in ("achieved", "frac", "achieved_algorithmic", "frac_algorithmic", "frac_of_fma_peak"):
                out["roofline"][k] = None
            out["roofline"]["note"] = "--accel: the hierarchy skips most of the reference's ray-shape tests; no roofline is claimed for it"
        if world_size > 1:
            ms_u8 = elapsed_u8_max * 1e3 / args.steps
            out["config"]["headline_leg"] = headline_leg
            out["config"]["frame_ms_with_f64_gather"] = ms_gather
            out["config"]["frame_ms_with_u8_gather"] = ms_u8 if u8_ok else None
            out["config"]["rays_per_s_with_u8_gather"] = total["radiance_rays"] / (ms_u8 * 1e-3) if u8_ok else None
            out["config"]["frame_ms_with_peer_stores"] = ms_peer
            out["config"]["frame_ms_with_peer_block_stores"] = ms_peer_blocks
            out["config"]["peer_stores_unavailable_because"] = peer_why
            # never the headline: a stream of frames (two in flight per GPU), not one frame on its own
            ms_stream = elapsed_stream_max * 1e3 / args.steps if stream_ok else None
            out["config"]["stream_of_frames"] = {
                "frames_in_flight": 2, "ms_per_frame": ms_stream,
                "rays_per_s": total["radiance_rays"] / (ms_stream * 1e-3) if ms_stream else None,
                "how": "peer stores into two shared canvases on rank 0, two scene handles and two streams per rank, "
                       "the ranks meet after the last frame"}
            # every GPU copying its own tiles to pinned host memory, no gather (never the headline: the other legs
            # leave the canvas on rank 0's GPU, this one ends in N host buffers and includes the trip over PCIe)
            out["config"]["frame_ms_with_own_d2h_no_gather"] = (
                elapsed_host_max * 1e3 / args.steps if (local_host is not None and host_ok) else None)
            out["config"]["own_d2h_unavailable_because"] = host_why
            out["config"]["canvases_verified_against_1_rank_render"] = {"f64_gather": gather_ok, "u8_gather": u8_ok,
                                                                        "peer_stores": peer_ok, "own_d2h": host_ok,
                                                                        "peer_block_stores": peer_blocks_ok,
                                                                        "stream_of_frames": stream_ok}
        if world_size == 1 and not args.accel and not weighted:
            out["config"]["weighted_integrator"] = weighted_integrator_leg(torch, native, D, scene, cam, args, local, stream,
                                                                           total["radiance_rays"])
        if world_size == 1 and not args.accel and not weighted and not args.no_stream_leg:
            out["config"]["stream_of_frames"] = frames_in_flight_leg(torch, native, D, world, cam, args, local,
                                                                     total["radiance_rays"])
        if world_size == 1 and args.multi_ranks > 0:
            out["config"]["single_process_multi"] = single_process_multi(native, D, world, cam, args, hsize, vsize)
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, camera, args.depth, args.cpu_budget)
            out["config"]["speedup_vs_cpu_literal"] = rays_per_s / out["cpu_baseline"]["value"]
            out["config"]["speedup_vs_cpu_streaming"] = rays_per_s / out["cpu_baseline"]["streaming_value"]
        print(json.dumps(out), flush=True)

    scene.close()
    if world_size > 1:
        torch.cuda.synchronize()
        dist.barrier()
        for c in (peer_canvas, peer_canvas2):
            if c and rank != 0:
                native.ipc_canvas_close(c, False)
        dist.barrier()
        for c in (peer_canvas, peer_canvas2):
            if c and rank == 0:
                native.ipc_canvas_close(c, True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
