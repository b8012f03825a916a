"""Developer tool: wall time of the host-buffer entry point rtc_render (device allocation, kernel,
D2H copy into pageable memory) next to the device-resident rtc_render_device, 4K Ch. 11 frame."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from raytracer_challenge_amd import native, scenes
world, cam = scenes.draw_world_scene(3840, 2160)
cd = cam.desc()
sc = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
out = torch.empty(cd.hsize * cd.vsize * 3, dtype=torch.float64, device="cuda")
for _ in range(3):
    sc.render_device(cd, 5, out.data_ptr())
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    sc.render_device(cd, 5, out.data_ptr())
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t) * 100
sc.render(cd, 5)
ts = []
for _ in range(5):
    t = time.perf_counter(); sc.render(cd, 5); ts.append((time.perf_counter() - t) * 1e3)
t = time.perf_counter(); sc2 = native.Scene(world.shape_descs(), len(world.objects), world.light_desc()); create_ms = (time.perf_counter() - t) * 1e3
sc.render_u8(cd, 5)
t8 = []
for _ in range(5):
    t = time.perf_counter(); sc.render_u8(cd, 5); t8.append((time.perf_counter() - t) * 1e3)
print(f"rtc_render_u8 (kernel + quantisation + 25 MB to the caller) min {min(t8):.1f} median {sorted(t8)[2]:.1f} ms")
print(f"rtc_render_device {dev_ms:.2f} ms/frame; rtc_render (kernel + 199 MB to the caller's fresh pageable buffer) min {min(ts):.1f} median {sorted(ts)[2]:.1f} ms; rtc_scene_create {create_ms:.2f} ms")
