#!/usr/bin/env python3
"""pt_render_multi_device rehearsed with several handles on one GPU: what the combine costs at full image size.
   python tools/multi_combine.py [handles=3] [width=1920] [height=1080]   (on the GPU box)
Prints, per sharding mode, the device time of the combine (peer copies + kernels on the first handle's GPU), the
number of kernels and copies, and the whole call's wall time at a small sample count."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracing_amd as pt  # noqa: E402
from hiprt import Hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
golden = os.path.join(ROOT, "tests", "golden")
scene = pt.LoadedScene(os.path.join(golden, "cornell_glossy.ptscene"))
cam = pt.load_camera(os.path.join(golden, "cornell_wide.ptcam"))
hip = Hip()
devs = [pt.DeviceScene(scene) for _ in range(n)]
out = hip.zeros(W * H * 12)
p = pt.make_params(W, H, 3, seed=1)
for mode, name in ((pt.abi.PT_SHARD_TILES, "tiles"), (pt.abi.PT_SHARD_SAMPLES, "samples")):
    for rep in range(3):
        t = time.perf_counter()
        pt.render_multi_device(devs, cam, p, out.value, mode)
        hip.sync()
        wall = (time.perf_counter() - t) * 1e3
        ms, kernels, copies, direct = pt.multi_last_combine(devs[0])
    print("%d handles on one GPU, %dx%d, %s: combine %.3f ms on the device (%d kernels, %d peer copies of %.1f MB, %d direct); "
          "whole call at 9 spp %.1f ms" % (n, W, H, name, ms, kernels, copies, W * H * 12 / 1e6, direct, wall))
for d in devs:
    d.close()
hip.free(out)
