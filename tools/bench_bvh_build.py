#!/usr/bin/env python3
"""Time of building the acceleration structure on the device: pt_scene_build_bvh_device with both of its builders,
on the displaced-grid mesh.  usage: tools/bench_bvh_build.py [N ...]   (N x N cells, 2 N^2 + 2 triangles)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import workloads  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [150, 500, 1000, 2000]:
    s, cam = workloads.big_mesh_scene(n)
    best = {}
    with pt.DeviceScene(s) as dev:
        for which in ("sah", "morton"):
            os.environ["PT_BVH_BUILDER"] = which
            dev.build_bvh_device()
            times = []
            for _ in range(5):
                t = time.perf_counter()
                dev.build_bvh_device()  # returns after the stream has drained
                times.append(time.perf_counter() - t)
            best[which] = min(times)
    del os.environ["PT_BVH_BUILDER"]
    print("%8d surfaces: surface-area builder %.2f ms, Morton radix tree %.2f ms (best of 5)"
          % (len(s.surfaces), best["sah"] * 1e3, best["morton"] * 1e3), flush=True)
