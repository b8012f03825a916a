"""Developer tool: kernel time of one rank's share of the 4K Ch. 11 frame for 1, 2, 4, 8 ranks
(interleaved tiles), i.e. what the render leg of an N-GPU run costs per GPU.  With --stream also the time per
share when the shares of consecutive frames overlap (two scene handles on two streams, launched alternately)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_challenge_amd import native, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
hs, vs = {"c3": (3840, 2160), "c5": (7680, 4320), "c2": (1920, 1080)}[wl]
world, cam = scenes.draw_world_scene(hs, vs)
cd = cam.desc()
sc = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
full = None
for ranks in (1, 2, 4, 8):
    n = native.tile_buffer_doubles(cd, 0, ranks)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    worst = 0.0
    per_rank = []
    for r in range(ranks if (ranks <= 2 or "--all-ranks" in sys.argv) else 2):
        for _ in range(3):
            sc.render_device(cd, 5, out.data_ptr(), r, ranks)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); sc.render_device(cd, 5, out.data_ptr(), r, ranks); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        worst = max(worst, sorted(ts)[len(ts) // 2])
        per_rank.append(sorted(ts)[len(ts) // 2])
    if ranks == 1:
        full = worst
    print(f"{wl} ranks {ranks}: render {worst:.3f} ms per rank; render-only efficiency {full / (ranks * worst):.2f}"
          + (f"; every rank's share: {' '.join(f'{t:.3f}' for t in per_rank)} ms" if len(per_rank) > 2 else ""))
    if "--stream" in sys.argv:
        import time
        handles = [native.Scene(world.shape_descs(), len(world.objects), world.light_desc()) for _ in range(2)]
        streams = globals().setdefault('_two_streams', [torch.cuda.Stream() for _ in range(2)])  # (the same two for every share size)
        outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(2)]
        for k in range(8):
            handles[k % 2].render_device(cd, 5, outs[k % 2].data_ptr(), 0, ranks, streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(40):
            handles[k % 2].render_device(cd, 5, outs[k % 2].data_ptr(), 0, ranks, streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 40
        if ranks == 1:
            full_stream = ms
        print(f"{wl} ranks {ranks}: as a stream of frames {ms:.3f} ms per share; render-only efficiency {full_stream / (ranks * ms):.2f} "
              f"(against one frame at a time on one GPU: {full / (ranks * ms):.2f})")
        for h in handles:
            h.close()
