#!/usr/bin/env python3
"""A/B kernel timing of several builds of librtc_hip.so, interleaved rounds, one subprocess per
measurement (developer tool).  usage: python scripts/ab.py [--rounds N] [--workloads c3,c4s] lib1.so lib2.so ..."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, os
sys.path.insert(0, %r)
import torch
from raytracer_challenge_amd import native, scenes, descs as D
wl = sys.argv[1]
weighted = wl.endswith("w")  # e.g. c3w: the same workload with the attenuation-weighted integrator
if weighted: wl = wl[:-1]
if wl == "c3": world, cam = scenes.draw_world_scene(3840, 2160)
elif wl == "c2": world, cam = scenes.draw_world_scene(1920, 1080)
elif wl == "c5": world, cam = scenes.draw_world_scene(7680, 4320)
elif wl == "c1k": world, cam = scenes.draw_world_scene(960, 540)
elif wl == "c4m": world, cam = scenes.synthetic_scene(1920, 1080)
elif wl == "c4s": world, cam = scenes.synthetic_scene(960, 540)
elif wl == "c4": world, cam = scenes.synthetic_scene(3840, 2160)
cd = cam.desc()
sc = native.Scene(world.shape_descs(), len(world.objects), world.light_desc())
if weighted: sc.set_integrator(D.RTC_INTEGRATOR_WEIGHTED)
out = torch.empty(cd.hsize * cd.vsize * 3, dtype=torch.float64, device="cuda")
DEPTH = int(os.environ.get('RTC_AB_DEPTH', '5'))
for _ in range(2): sc.render_device(cd, DEPTH, out.data_ptr())
torch.cuda.synchronize()
ts = []
for _ in range(int(sys.argv[2])):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sc.render_device(cd, DEPTH, out.data_ptr()); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
import hashlib
print(json.dumps({"ms": ts, "sha": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]}))
''' % ROOT

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--workloads", default="c3,c4s")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    res = {}
    for wl in a.workloads.split(","):
        for r in range(a.rounds):
            for lib in a.libs:
                # a "library" may carry environment settings: lib.so:NAME=value[:NAME=value]
                parts = lib.split(":")
                env = dict(os.environ, RTC_HIP_LIB=os.path.abspath(parts[0]))
                for kv in parts[1:]:
                    k, v = kv.split("=", 1)
                    env[k] = v  # RTC_AB_ENV
                p = subprocess.run([sys.executable, "-c", CHILD, wl, str(a.iters)], env=env, capture_output=True, text=True, timeout=600)
                if p.returncode != 0:
                    print("FAILED", lib, wl, p.stderr[-2000:]); continue
                j = json.loads(p.stdout.strip().splitlines()[-1])
                res.setdefault((wl, lib), {"ms": [], "sha": set()})
                res[(wl, lib)]["ms"] += j["ms"]; res[(wl, lib)]["sha"].add(j["sha"])
    for (wl, lib), v in res.items():
        ms = sorted(v["ms"])
        print(f"{wl:5s} {os.path.basename(lib):32s} min {ms[0]:10.3f} ms  median {ms[len(ms)//2]:10.3f} ms  n={len(ms)} sha={','.join(sorted(v['sha']))}", flush=True)

if __name__ == "__main__":
    main()
