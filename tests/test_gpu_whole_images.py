"""Whole images at the benchmark's resolutions, one sample per pixel (four on the two small scenes), bit for bit against the oracle: every pixel of
all four workloads goes through the comparison once (the other full-size tests compare windows), with the tree the
benchmark uses (built on the device, handed to the oracle through pt_scene_get_bvh)."""
import ctypes as C

import numpy as np
import pytest

import path_tracing_amd as pt
from path_tracing_amd import workloads

pytestmark = pytest.mark.gpu


def differing_pixels(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return int(np.count_nonzero(~same.all(axis=-1)))


@pytest.mark.parametrize("name", ["cornell", "frames1", "textured", "mesh1000"])
def test_whole_image(name, oracle):
    scene, cam, label, sqrt_spp, extra = workloads.load(name)
    W, H = extra.get("width", 1920), extra.get("height", 1080)
    t0, t1 = extra["frames"][0] if "frames" in extra else (0.0, 0.0)
    flat = scene if isinstance(scene, pt.FlatScene) else pt.FlatScene.from_desc(scene.desc())
    p = pt.make_params(W, H, 2 if name in ("cornell", "frames1") else 1, seed=11, t0=t0, t1=t1, max_path_segments=extra.get("max_path_segments", 0))
    with pt.DeviceScene(scene) as dev:
        dev.build_bvh_device(t0, t1)
        flat.nodes = dev.get_bvh()
        got = dev.render(cam, p)
    d = flat.desc()
    want = oracle.render(C.byref(d), cam, p, math_mode=1)
    assert np.isfinite(got).all() and got.mean() > 0.01
    assert differing_pixels(got, want) == 0


@pytest.mark.parametrize("name", ["cornell", "frames1", "mesh1000"])
def test_fixed_and_taken_runs_give_the_same_image(name):
    """PT_TRACE_RUNS=fixed|taken only changes which wave traces which ray: the image is the same bit for bit, on the
    LDS kernel (static and moving surfaces) and on the nearer-child-first kernel, at a size where both ways differ."""
    import os
    scene, cam, label, sqrt_spp, extra = workloads.load(name)
    W, H = extra.get("width", 1920), extra.get("height", 1080)
    t0, t1 = extra["frames"][0] if "frames" in extra else (0.0, 0.0)
    p = pt.make_params(W, H, 1, seed=3, t0=t0, t1=t1)
    images = {}
    with pt.DeviceScene(scene) as dev:
        dev.build_bvh_device(t0, t1)
        for mode in ("fixed", "taken"):
            os.environ["PT_TRACE_RUNS"] = mode
            try:
                images[mode] = dev.render(cam, p)
            finally:
                del os.environ["PT_TRACE_RUNS"]
    assert differing_pixels(images["fixed"], images["taken"]) == 0
