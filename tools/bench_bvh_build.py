#!/usr/bin/env python3
"""Time of building the acceleration structure: the host builder that restates the reference's
(pt_bvh_build, one thread) against pt_scene_build_bvh_device with both of its builders, on the displaced-grid mesh.
usage: tools/bench_bvh_build.py [N ...]   (N x N cells, 2 N^2 + 2 triangles)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracing_amd as pt  # noqa: E402
import scenes  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [150, 500, 1000]:
    t = time.perf_counter()
    s, cam = scenes.big_mesh_scene(n)  # builds the host tree
    t_scene = time.perf_counter() - t
    t = time.perf_counter()
    s.build_bvh()
    t_host = time.perf_counter() - t
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
    print("%8d surfaces: host builder %.3f s, device: surface-area builder %.2f ms (%.0fx), Morton radix tree %.2f ms (best of 5)"
          % (len(s.surfaces), t_host, best["sah"] * 1e3, t_host / best["sah"], best["morton"] * 1e3), flush=True)
