"""Per-launch time of the renderer's kernels in a render that is mostly the draining of the pool (default: 1920x1080
at 9 spp, one filling of a 16 M pool and then the long paths): what an iteration costs regardless of the number of
paths, which is what the last iterations of every batch pay.
  python tools/fixed_costs.py cornell|frames|mesh1000|textured [width height sqrt_spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import workloads  # noqa: E402

name = sys.argv[1]
scene, cam, label, sqrt_spp, extra = workloads.load(name)
kw = {}
if extra.get("max_path_segments"):
    kw["max_path_segments"] = extra["max_path_segments"]
t0, t1 = extra["frames"][0] if extra.get("frames") else (0.0, 0.0)
with pt.DeviceScene(scene) as dev:
    dev.build_bvh_device(t0, t1)
    W, H, Q = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (1920, 1080, 3)
    p = pt.make_params(W, H, Q, seed=1, t0=t0, t1=t1, **kw)
    dev.render(cam, p)
    for profiled in (True, False):
        dev.profile(profiled)
        t = time.time()
        for _ in range(5):
            dev.render(cam, p)
        wall = (time.time() - t) / 5
        if profiled:
            k = dev.kernel_times()
            its = max(v[0] for v in k.values()) / 5
            print(name, "iterations %.0f, wall %.2f ms per render (timed kernels: %.2f ms)" % (
                its, wall * 1e3, sum(v[1] for v in k.values()) / 5),
                {n: "%.1f us x %d" % (v[1] / max(v[0], 1) * 1e3, v[0] / 5) for n, v in k.items()})
        else:
            print(name, "untimed: wall %.2f ms per render" % (wall * 1e3))
