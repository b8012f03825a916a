import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from raytracer_challenge_amd import native, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
world, cam = {"c3": lambda: scenes.draw_world_scene(3840, 2160), "c2": lambda: scenes.draw_world_scene(1920, 1080), "tiny": lambda: scenes.draw_world_scene(128, 64), "c5": lambda: scenes.draw_world_scene(7680, 4320),
              "c4m": lambda: scenes.synthetic_scene(1920, 1080)}[wl]()
cd = cam.desc()
sc = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # probe rank 0's share of an interleaved partition
out = torch.empty(native.tile_buffer_doubles(cd, 0, ranks), dtype=torch.float64, device="cuda")
for _ in range(2):
    sc.render_device(cd, 5, out.data_ptr(), 0, ranks)
torch.cuda.synchronize()
L = native.lib()
n = 4096
raw = np.zeros(19 * 8192, dtype=np.uint64)
buf = raw[: 3 * 8192].reshape(8192, 3)[:n]
L.rtc_debug_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert L.rtc_debug_probe(sc._h, raw.ctypes.data, n) == 0
t0 = buf[:, 0].astype(np.int64); tx = buf[:, 1].astype(np.int64); te = buf[:, 2].astype(np.int64)
ok = te > 0
base = t0[ok].min()
f = 1e-5  # 100 MHz ticks -> ms
print("waves", ok.sum(), "start spread ms", (t0[ok].max() - base) * f)
print("exhausted at ms: min %.3f med %.3f max %.3f" % tuple(np.percentile((tx[ok] - base) * f, [0, 50, 100])))
print("end at ms: min %.3f p10 %.3f med %.3f p90 %.3f p99 %.3f max %.3f" % tuple(np.percentile((te[ok] - base) * f, [0, 10, 50, 90, 99, 100])))
tail = (te[ok] - tx[ok]) * f
print("per-wave (end - exhausted) ms: med %.3f p90 %.3f max %.3f" % tuple(np.percentile(tail, [50, 90, 100])))
end = np.sort((te[ok] - base) * f)
print("waves still running at ms:", " ".join("%.2f:%d" % (t, (end > t).sum()) for t in np.arange(0.1, end[-1], 0.05)))
it = raw[3 * 8192:].reshape(8192, 16)[:n][ok][:, 15].astype(np.int64)
print("iterations per wave: min %d med %d p90 %d max %d" % tuple(np.percentile(it, [0, 50, 90, 100])))

pc = raw[3 * 8192:].reshape(8192, 16)[:n][ok].astype(np.float64)
names = ["refill+steal", "sweep1", "prep", "sweep2", "gather", "pattern", "light", "children", "push", "unwind"]
tot = pc[:, :10].sum()
iters = pc[:, 15].sum()
print("wave-iterations", int(iters), " cycles per wave-iteration %.0f" % (tot / iters))
for k, nme in enumerate(names):
    print("  %-13s %5.1f%%  %8.0f cycles/iter" % (nme, 100 * pc[:, k].sum() / tot, pc[:, k].sum() / iters))
