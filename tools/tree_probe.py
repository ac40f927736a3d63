"""Experiment: rate and executed-walk counts of one workload with different trees over the same leaves: the
reference's builder, the device's Morton builder, and a surface-area-heuristic tree (tools/micro/sah_tree.cpp).
  python tools/tree_probe.py mesh1000|textured [sqrt_spp]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import workloads  # noqa: E402
from path_tracing_amd.api import NODE_DTYPE  # noqa: E402

name = sys.argv[1]
scene, cam, label, sqrt_spp, extra = workloads.load(name)
if len(sys.argv) > 2:
    sqrt_spp = int(sys.argv[2])
W, H = extra.get("width", 1920), extra.get("height", 1080)
flat = scene if isinstance(scene, pt.FlatScene) else pt.FlatScene.from_desc(scene.desc())
host = np.array(flat.nodes, copy=True)
leaves = host[(host["link"] & 1) == 0]
n = len(leaves)
boxes = np.zeros((n, 6), np.float32)
surf = (leaves["link"] // 2).astype(np.int64)
boxes[surf, :3] = leaves["lo"]
boxes[surf, 3:] = leaves["hi"]
sah = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libsah_tree.so"))


def sah_nodes(sweep_below):
    out = np.zeros(2 * n - 1, NODE_DTYPE)
    t = time.time()
    sah.sah_tree(boxes.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_int(sweep_below), out.ctypes.data_as(C.c_void_p))
    return out, time.time() - t


def measure(dev, what):
    kw = dict(seed=1, max_path_segments=extra.get("max_path_segments", 0) or 16)
    if "max_path_segments" not in extra:
        kw.pop("max_path_segments")
    p = pt.make_params(W, H, sqrt_spp, **kw)
    dev.render(cam, p)
    t = time.time()
    img = dev.render(cam, p)
    dt = time.time() - t
    dev.count_walk(True)
    dev.render(cam, pt.make_params(W // 4, H // 4, 2, **kw))
    c = dev.walk_counters()
    dev.count_walk(False)
    spp = sqrt_spp * sqrt_spp
    rays = max(c.get("path_rays", 0) + c.get("shadow_rays", 0), 1)
    print("%-28s %7.1f M samples/s  mean %.5f  counters per ray: %s" % (
        what, W * H * spp / dt / 1e6, float(img.mean()),
        {k: round(v / rays, 2) for k, v in c.items() if v and k not in ("path_rays", "shadow_rays")}), flush=True)


with pt.DeviceScene(scene) as dev:
    measure(dev, "reference builder")
    for which in ("morton", "sah"):
        os.environ["PT_BVH_BUILDER"] = which
        dev.build_bvh_device()
        t = time.time()
        dev.build_bvh_device()
        ms = (time.time() - t) * 1e3
        measure(dev, "device %s builder (%.2f ms)" % (which, ms))
    for sweep, bins, boxdom in ((0, 64, 0), (64, 32, 0), (64, 32, 1), (64, 16, 1)):
        out = np.zeros(2 * n - 1, NODE_DTYPE)
        t = time.time()
        sah.sah_tree2(boxes.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_int(sweep), C.c_int(bins), C.c_int(boxdom),
                      out.ctypes.data_as(C.c_void_p))
        secs = time.time() - t
        dev.update_bvh(out)
        measure(dev, "SAH %d bins %s domain, sweep<=%d (%.1f s)" % (bins, "box" if boxdom else "centroid", sweep, secs))
