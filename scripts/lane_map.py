"""Developer tool: lane utilisation of the regions of the frame kernel's loop, from a library built with
-DRTC_PROBE -DRTC_LANE_MAP (RTC_HIP_LIB=...): per region, the lanes that enter it per wave iteration and per visit
of a wave.  usage: lane_map.py [c3|c2|c5] [ranks]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from raytracer_challenge_amd import native, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
world, cam = {"c3": lambda: scenes.draw_world_scene(3840, 2160), "c2": lambda: scenes.draw_world_scene(1920, 1080),
              "c5": lambda: scenes.draw_world_scene(7680, 4320)}[wl]()
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cd = cam.desc()
sc = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
out = torch.empty(native.tile_buffer_doubles(cd, 0, ranks), dtype=torch.float64, device="cuda")
for _ in range(3):
    sc.render_device(cd, 5, out.data_ptr(), 0, ranks)
torch.cuda.synchronize()
L = native.lib()
raw = np.zeros(19 * 8192, dtype=np.uint64)
L.rtc_debug_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert L.rtc_debug_probe(sc._h, raw.ctypes.data, 4096) == 0
pc = raw[3 * 8192:].reshape(8192, 16)[:4096]
lanes = (pc & np.uint64(0xffffffff)).sum(axis=0).astype(np.float64)
waves = (pc >> np.uint64(32)).sum(axis=0).astype(np.float64)
names = ["live lanes (not idle)", "radiance sweep", "hit: prepare, shadow sweep, lighting", "pattern", "diffuse term",
         "specular term", "refraction direction", "Schlick", "push a frame", "unwinding (have a colour)",
         "unwind level 0", "unwind level 1", "unwind level 2", "unwind level 3", "unwind level 4", "pixel stored"]
iters = waves[0]
print(f"{wl}, rank 0 of {ranks}: {int(iters)} wave iterations")
print(f"{'region':40s} {'visits per iteration':>22s} {'lanes per visit':>16s} {'lanes per iteration':>20s}")
for k, nme in enumerate(names):
    if waves[k] > 0:
        print(f"{nme:40s} {waves[k] / iters:22.3f} {lanes[k] / waves[k]:16.1f} {lanes[k] / iters:20.1f}")
