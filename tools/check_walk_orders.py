#!/usr/bin/env python3
"""One-off check at scale: the nearer-child-first walk (four boxes per row, and two) against the storage-order walk
on the 2 M-triangle mesh, 1920x1080 at 4 spp (8.3 M paths, about 31 M rays): the images must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracing_amd as pt  # noqa: E402
import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
s, cam = scenes.big_mesh_scene(n)
p = pt.make_params(1920, 1080, int(os.environ.get("CHECK_SQRT_SPP", "2")), seed=int(sys.argv[2]) if len(sys.argv) > 2 else 7)
with pt.DeviceScene(s) as dev:
    t = time.time()
    fast = dev.render(cam, p)
    t_fast = time.time() - t
    os.environ["PT_NO_QUAD_ROWS"] = "1"
    two_wide = dev.render(cam, p)
    del os.environ["PT_NO_QUAD_ROWS"]
    os.environ["PT_TRAVERSAL"] = "reference"
    t = time.time()
    slow = dev.render(cam, p)
    t_slow = time.time() - t
bad = 0
for name, img in (("four boxes per row", fast), ("two boxes per row", two_wide)):
    diff = (img.view(np.uint32) != slow.view(np.uint32)) & ~(np.isnan(img) & np.isnan(slow))
    print("%s: pixels differing from the storage-order walk: %d of %d" % (name, diff.any(axis=2).sum(), diff.shape[0] * diff.shape[1]))
    bad += int(diff.any())
print("ordered %.2f s, storage order %.2f s" % (t_fast, t_slow))
sys.exit(1 if bad else 0)
