"""Executed-walk counts per ray of one workload with the tree the benchmark uses (the device builder's):
  python tools/walk_counts.py cornell|mesh1000|textured|frames"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import workloads  # noqa: E402

name = sys.argv[1]
scene, cam, label, sqrt_spp, extra = workloads.load(name)
W, H = extra.get("width", 1920), extra.get("height", 1080)
kw = {}
if extra.get("max_path_segments"):
    kw["max_path_segments"] = extra["max_path_segments"]
t0, t1 = extra["frames"][0] if extra.get("frames") else (0.0, 0.0)
with pt.DeviceScene(scene) as dev:
    dev.build_bvh_device(t0, t1)
    p = pt.make_params(W // 2, H // 2, 4, seed=1, t0=t0, t1=t1, **kw)
    dev.render(cam, p)
    dev.count_walk(True)
    dev.render(cam, p)
    c = dev.walk_counters()
    dev.count_walk(False)
    rays = max(c.get("path_rays", 0) + c.get("shadow_rays", 0), 1)
    samples = (W // 2) * (H // 2) * 16
    print(name, "rays/sample %.2f (path %.2f, shadow %.2f); per ray:" % (
        rays / samples, c.get("path_rays", 0) / samples, c.get("shadow_rays", 0) / samples),
        {k: round(v / rays, 5) for k, v in c.items() if v and k not in ("path_rays", "shadow_rays")})
