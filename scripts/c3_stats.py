"""Developer tool: statistics of one Ch. 11 frame (where the ray-sphere pairs end)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_challenge_amd import native, scenes, descs as D
h, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
world, cam = scenes.draw_world_scene(h, v)
sc = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
st = D.RtcStats()
out = torch.empty(h * v * 3, dtype=torch.float64, device="cuda")
sc.render_device(cam.desc(), 5, out.data_ptr(), 0, 1, 0, st)
torch.cuda.synchronize()
d = st.as_dict()
rays = d["radiance_rays"] + d["shadow_rays"]
print(d)
print(f"per ray: sphere pairs {d['sphere_tests'] / rays:.2f}, past the bounding test {(d['sphere_tests'] - d['sphere_bound_skips']) / rays:.2f}, "
      f"with roots {d['sphere_hits'] / rays:.2f}; plane pairs {d['plane_tests'] / rays:.2f}, with t {d['plane_hits'] / rays:.2f}, "
      f"same-side skips {d['plane_side_skips'] / rays:.2f}; shaded hits per radiance ray {d['shaded_hits'] / d['radiance_rays']:.2f}")
