#!/usr/bin/env python3
"""Time of the importer's two image passes on a 4096 x 4096 texture: device kernels (HIP events, data resident)
against the compiled reference on the host (one thread, as the importer runs them)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracing_amd as pt  # noqa: E402
import refbind  # noqa: E402

W = H = 4096
rng = np.random.default_rng(1)
img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
L = pt.lib()
d_in = torch.from_numpy(img).cuda()
d_tex = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
d_nrm = torch.empty_like(d_tex)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


ms_tex = timed(lambda: L.pt_texture_from_bytes_device(0, C.c_void_p(d_in.data_ptr()), W, H, 1, 1, C.c_void_p(d_tex.data_ptr()),
                                                      C.c_void_p(stream)))
ms_nrm = timed(lambda: L.pt_bump_to_normal_map_device(0, C.c_void_p(d_tex.data_ptr()), W, H, C.c_float(2.0),
                                                      C.c_void_p(d_nrm.data_ptr()), C.c_void_p(stream)))
gb_tex = W * H * 3 * 5 / 1e9
gb_nrm = W * H * 3 * 8 / 1e9
print("texture_from_bytes %d x %d: %.3f ms on the device (%.0f GB/s of 1 B in + 4 B out per sample)" % (W, H, ms_tex, gb_tex / ms_tex * 1e3))
print("bump_to_normal_map %d x %d: %.3f ms on the device (%.0f GB/s of 12 B in + 12 B out per texel)" % (W, H, ms_nrm, gb_nrm / ms_nrm * 1e3))
if os.path.exists(refbind.REF_LIB):
    R = refbind.Reference()
    t = time.perf_counter()
    tex = R.texture_from_bytes(img, linearize=True)
    t_tex = time.perf_counter() - t
    t = time.perf_counter()
    nrm = R.bump_to_normal_map(tex, 2.0)
    t_nrm = time.perf_counter() - t
    print("reference on the host: texture %.2f s, normal map %.2f s" % (t_tex, t_nrm))
    assert np.array_equal(d_tex.cpu().numpy().view(np.uint32), tex.view(np.uint32))
    assert np.array_equal(d_nrm.cpu().numpy().view(np.uint32), nrm.view(np.uint32))
    print("device results equal the reference's bit for bit")
