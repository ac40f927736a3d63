"""The benchmark's other configurations at their full sizes, through properties that need no full-size render
on the CPU: the 2 M-triangle and the 8 M-triangle mesh at 1920x1080, the textured multi-material OBJ scene at 3840x2160 with path depth 32
(also split over two scene handles), and a frame of the reference's animation scene at 1920x1080.  (The Cornell configuration's counterpart is
test_gpu_parity.py::test_full_size_properties.)"""
import ctypes as C
import os

import numpy as np
import pytest

import path_tracing_amd as pt
import scenes

pytestmark = pytest.mark.gpu

NO_HIT = 0xFFFFFFFF


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def test_two_million_triangles(oracle):
    """Both walks give the same 1080p image bit for bit; a window of primary hits equals the oracle's; a tile and a
    sample range of a render equal the same part of the whole; with the tree built on the device every primary ray
    finds the same distance (surfaces may differ on shared edges)."""
    s, cam = scenes.big_mesh_scene(1000)
    d = s.desc()
    W, H = 1920, 1080
    with pt.DeviceScene(s) as dev:
        p = pt.make_params(W, H, 2, seed=7)
        fast = dev.render(cam, p)
        os.environ["PT_TRAVERSAL"] = "reference"
        try:
            slow = dev.render(cam, p)
        finally:
            del os.environ["PT_TRAVERSAL"]
        assert same_bits(fast, slow)
        os.environ["PT_NO_QUAD_ROWS"] = "1"  # the two-wide form of the nearer-child-first walk
        try:
            two_wide = dev.render(cam, p)
        finally:
            del os.environ["PT_NO_QUAD_ROWS"]
        assert same_bits(fast, two_wide)
        assert np.isfinite(fast).all() and fast.mean() > 0.01
        tile = np.zeros_like(fast)
        dev.render(cam, pt.make_params(W, H, 2, seed=7, tile=(700, 400, 1100, 650)), out=tile)
        assert same_bits(tile[400:650, 700:1100], fast[400:650, 700:1100])
        parts = [dev.render(cam, pt.make_params(W, H, 2, seed=7, samples=r, flags=pt.abi.PT_RENDER_SUM_ONLY))
                 for r in ((0, 3), (3, 4))]
        one = pt.make_params(W, H, 1, seed=7)
        window = pt.make_params(W, H, 1, seed=7, tile=(800, 500, 1120, 620))
        hits = dev.primary_hits(cam, window)
        assert hits.tobytes() == oracle.primary_hits(C.byref(d), cam, window, math_mode=1).tobytes()
        hits_ref_tree = dev.primary_hits(cam, one)
        dev.build_bvh_device()
        hits_dev_tree = dev.primary_hits(cam, one)
        again = dev.render(cam, p)
    # sample ranges: the sums over [0, 3) and [3, 4) add up to the sum over all four (the order of additions differs)
    combined = (parts[0] + parts[1]) / 4.0
    assert np.allclose(combined, fast, rtol=1e-5, atol=1e-6)
    both = (hits_ref_tree["surface"] != NO_HIT) & (hits_dev_tree["surface"] != NO_HIT)
    assert np.mean((hits_ref_tree["surface"] != NO_HIT) != (hits_dev_tree["surface"] != NO_HIT)) <= 1e-5
    assert np.mean(hits_ref_tree["a"][both] != hits_dev_tree["a"][both]) <= 1e-5
    assert np.mean(hits_ref_tree["surface"][both] != hits_dev_tree["surface"][both]) <= 1e-3
    assert np.mean(bits(again) != bits(fast)) <= 1e-3  # paths that met a tie go another way


def test_eight_million_triangles(oracle):
    """The mesh workload at 8 M triangles -- node rows and surfaces (256 + 384 MB) no longer fit the 256 MiB last-level
    cache -- with the tree built on the device: both walks give the same 1080p image bit for bit; a window of primary
    hits and of per-sample radiance equals the oracle's on the very same tree (copied out with pt_scene_get_bvh); a
    tile and sample ranges of a render equal the same part of the whole; equal inputs give equal trees."""
    from path_tracing_amd import workloads
    s, cam = workloads.big_mesh_scene(2000)
    assert len(s.surfaces) == 8000002
    W, H = 1920, 1080
    with pt.DeviceScene(s) as dev:
        dev.build_bvh_device()
        p = pt.make_params(W, H, 1, seed=7)
        fast = dev.render(cam, p)
        os.environ["PT_TRAVERSAL"] = "reference"
        try:
            slow = dev.render(cam, p)
        finally:
            del os.environ["PT_TRAVERSAL"]
        assert same_bits(fast, slow)
        os.environ["PT_NO_QUAD_ROWS"] = "1"  # the two-wide form of the nearer-child-first walk
        try:
            two_wide = dev.render(cam, p)
        finally:
            del os.environ["PT_NO_QUAD_ROWS"]
        assert same_bits(fast, two_wide)
        assert np.isfinite(fast).all() and fast.mean() > 0.01
        tile = np.zeros_like(fast)
        dev.render(cam, pt.make_params(W, H, 1, seed=7, tile=(700, 400, 1100, 650)), out=tile)
        assert same_bits(tile[400:650, 700:1100], fast[400:650, 700:1100])
        q = pt.make_params(W, H, 2, seed=7, tile=(600, 300, 1240, 660))
        whole = dev.render(cam, q)
        parts = [dev.render(cam, pt.make_params(W, H, 2, seed=7, tile=(600, 300, 1240, 660), samples=r, flags=pt.abi.PT_RENDER_SUM_ONLY))
                 for r in ((0, 3), (3, 4))]
        assert np.allclose(((parts[0] + parts[1]) / 4.0)[300:660, 600:1240], whole[300:660, 600:1240], rtol=1e-5, atol=1e-6)
        # the oracle on the same tree
        s.nodes = dev.get_bvh()
        assert len(s.nodes) == 2 * 8000002 - 1
        d = s.desc()
        window = pt.make_params(W, H, 1, seed=7, tile=(800, 500, 1120, 620))
        hits = dev.primary_hits(cam, window)
        assert hits.tobytes() == oracle.primary_hits(C.byref(d), cam, window, math_mode=1).tobytes()
        small = pt.make_params(W, H, 2, seed=7, tile=(900, 520, 964, 552))
        assert same_bits(dev.render_samples(cam, small), oracle.render_samples(C.byref(d), cam, small, math_mode=1))
        dev.build_bvh_device()
        assert dev.get_bvh().tobytes() == s.nodes.tobytes()
        assert same_bits(dev.render(cam, p), fast)


def test_textured_scene_at_4k(golden_dir, oracle):
    """The textured multi-material OBJ scene at 3840x2160, 4 of the benchmark's 512 samples, path depth 32: a tile of
    the full render bit for bit, a window against the oracle; the same image from two scene handles splitting it by
    tiles (pt_render_multi), bit for bit; sums over sample ranges against the sum over all."""
    scene = pt.LoadedScene(os.path.join(golden_dir, "textured.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "textured.ptcam"))
    W, H = 3840, 2160
    kw = dict(seed=3, max_path_segments=32, samples=(0, 4), flags=pt.abi.PT_RENDER_SUM_ONLY)
    with pt.DeviceScene(scene) as dev, pt.DeviceScene(scene) as dev2:
        p = pt.make_params(W, H, 23, **kw)
        full = dev.render(cam, p)
        assert np.isfinite(full).all() and full.mean() > 0.1
        tile = np.zeros_like(full)
        q = pt.make_params(W, H, 23, tile=(1900, 1000, 2028, 1064), **kw)
        dev.render(cam, q, out=tile)
        assert same_bits(tile[1000:1064, 1900:2028], full[1000:1064, 1900:2028])
        assert same_bits(pt.render_multi([dev, dev2], cam, p, pt.abi.PT_SHARD_TILES), full)
        halves = [dev.render(cam, pt.make_params(W, H, 23, seed=3, max_path_segments=32, samples=r,
                                                 flags=pt.abi.PT_RENDER_SUM_ONLY)) for r in ((0, 1), (1, 4))]
        assert np.allclose(halves[0] + halves[1], full, rtol=1e-5, atol=1e-6)
    want = oracle.render(scene.ptr, cam, q, math_mode=1)
    assert same_bits(tile[1000:1064, 1900:2028], want[1000:1064, 1900:2028])


def test_animation_frame_at_1080p(golden_dir, oracle):
    """One frame (number 100 of 240) of the reference's animation scene at 1920x1080 with its own tree (the
    reference's builder, then the device's): a window of the image against the oracle."""
    from path_tracing_amd import workloads
    loaded = pt.LoadedScene(os.path.join(golden_dir, "frames.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "frames.ptcam"))
    s = pt.FlatScene.from_desc(loaded.desc())
    t0, t1 = workloads.frame_interval(100)
    W, H = 1920, 1080
    q = pt.make_params(W, H, 2, seed=142, t0=t0, t1=t1, tile=(900, 500, 1028, 564))
    with pt.DeviceScene(loaded) as dev:
        dev.update_bvh(s.build_bvh(t0, t1))
        d = s.desc()
        full = dev.render(cam, pt.make_params(W, H, 2, seed=142, t0=t0, t1=t1))
        assert np.isfinite(full).all() and full.mean() > 0.1
        want = oracle.render(C.byref(d), cam, q, math_mode=1)
        assert same_bits(full[500:564, 900:1028], want[500:564, 900:1028])
        dev.build_bvh_device(t0, t1)
        s.nodes = dev.get_bvh()
        d = s.desc()
        tile = dev.render(cam, q)
        want = oracle.render(C.byref(d), cam, q, math_mode=1)
        assert same_bits(tile[500:564, 900:1028], want[500:564, 900:1028])
