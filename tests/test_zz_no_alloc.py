"""pt_render_device_noalloc -- the render entry point that allocates nothing (its work areas are parts of a
fixed device array in the library's image) -- against pt_render_device: the same image bit for bit, on calls in a
row that reuse the area, the later ones smaller than the earlier ones; free device memory unchanged across them."""
import ctypes as C
import os

import numpy as np
import pytest

import path_tracing_amd as pt
import scenes
from hiprt import Hip

pytestmark = pytest.mark.gpu


def render_both(hip, dev, cam, p, stream=None):
    nbytes = p.width * p.height * 12
    a, b = hip.zeros(nbytes), hip.zeros(nbytes)
    sp = stream.value if stream is not None else None
    dev.render_device(cam, p, a.value, sp)
    hip.sync()
    before = hip.free_bytes()
    dev.render_device_noalloc(cam, p, b.value, sp)
    hip.sync()
    after = hip.free_bytes()
    ia, ib = hip.to_host(a, (p.height, p.width, 3)), hip.to_host(b, (p.height, p.width, 3))
    hip.free(a)
    hip.free(b)
    return ia, ib, before, after


def test_noalloc_render_equals_allocating_render(golden_dir, oracle):
    hip = Hip()
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_wide.ptcam"))
    stream = hip.stream()
    with pt.DeviceScene(scene) as dev:
        # first call: 196 608 pixels x 16 samples = 3.1 M samples, more than the area's sample buffer holds, so two
        # batches whose sums are carried over; second call much smaller, on another stream; third: a tile and a
        # sample range of a larger image
        calls = [(pt.make_params(512, 384, 4, seed=3), stream),
                 (pt.make_params(64, 48, 2, seed=4), None),
                 (pt.make_params(900, 700, 3, seed=5, tile=(100, 50, 400, 260), samples=(2, 7)), stream)]
        for p, st in calls:
            want, got, before, after = render_both(hip, dev, cam, p, st)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (p.width, p.height)
            assert before == after, "device memory was allocated or freed during the call"
            assert np.isfinite(got).all() and got.mean() > 0.0
        # the small one once more against the oracle
        p = pt.make_params(40, 30, 2, seed=6)
        _, got, _, _ = render_both(hip, dev, cam, p)
        assert np.array_equal(got.view(np.uint32), oracle.render(scene.ptr, cam, p, math_mode=1).view(np.uint32))
        # beyond the stated limit: a clear error, nothing rendered
        out = hip.zeros(1024 * 1024 * 12)
        with pytest.raises(pt.PtError) as e:
            dev.render_device_noalloc(cam, pt.make_params(1024, 1024, 1), out.value, None)
        assert e.value.code == -5 and "524288" in str(e.value)
        # ... and the same image in two tiles is fine
        p = pt.make_params(1024, 1024, 1, seed=8)
        dev.render_device_noalloc(cam, pt.make_params(1024, 1024, 1, seed=8, tile=(0, 0, 1024, 512)), out.value, None)
        dev.render_device_noalloc(cam, pt.make_params(1024, 1024, 1, seed=8, tile=(0, 512, 1024, 1024)), out.value, None)
        ref = hip.zeros(1024 * 1024 * 12)
        dev.render_device(cam, p, ref.value, None)
        hip.sync()
        assert np.array_equal(hip.to_host(out, (1024, 1024, 3)).view(np.uint32), hip.to_host(ref, (1024, 1024, 3)).view(np.uint32))
        hip.free(out)
        hip.free(ref)
    hip.destroy_stream(stream)


def test_noalloc_render_of_a_scene_in_memory_and_a_second_scene():
    """A mesh too large for LDS (nearer-child-first kernel, second pass, shadow queues in the fixed area), then a
    smaller render of another scene on the same GPU: both share the area."""
    hip = Hip()
    s, cam = scenes.big_mesh_scene(150)
    s2, cam2, p2 = scenes.build("materials")
    with pt.DeviceScene(s) as dev, pt.DeviceScene(s2) as dev2:
        p = pt.make_params(320, 180, 3, seed=11)
        want, got, before, after = render_both(hip, dev, cam, p)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert before == after
        q = pt.make_params(48, 40, 2, seed=12)
        want, got, before, after = render_both(hip, dev2, cam2, q)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert before == after


def test_noalloc_calls_of_two_scenes_on_two_streams_are_ordered_by_the_library(golden_dir):
    """Two scenes of one GPU, each on its own stream, no synchronisation between the calls: the second call's
    kernels wait on the device for the first call's (the area's event), so both images are the allocating
    renderer's, bit for bit -- several rounds, so that a call regularly finds the other's kernels still running."""
    hip = Hip()
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_wide.ptcam"))
    s2, cam2, _ = scenes.build("materials")
    sa, sb = hip.stream(), hip.stream()
    pa, pb = pt.make_params(256, 192, 3, seed=21), pt.make_params(160, 120, 2, seed=22)
    with pt.DeviceScene(scene) as dev, pt.DeviceScene(s2) as dev2:
        wa, wb = hip.zeros(pa.width * pa.height * 12), hip.zeros(pb.width * pb.height * 12)
        dev.render_device(cam, pa, wa.value, None)
        dev2.render_device(cam2, pb, wb.value, None)
        hip.sync()
        want_a, want_b = hip.to_host(wa, (pa.height, pa.width, 3)), hip.to_host(wb, (pb.height, pb.width, 3))
        ga, gb = hip.zeros(pa.width * pa.height * 12), hip.zeros(pb.width * pb.height * 12)
        for _ in range(4):
            dev.render_device_noalloc(cam, pa, ga.value, sa.value)
            dev2.render_device_noalloc(cam2, pb, gb.value, sb.value)
        hip.sync()
        got_a, got_b = hip.to_host(ga, (pa.height, pa.width, 3)), hip.to_host(gb, (pb.height, pb.width, 3))
        assert np.array_equal(got_a.view(np.uint32), want_a.view(np.uint32))
        assert np.array_equal(got_b.view(np.uint32), want_b.view(np.uint32))
        for b in (wa, wb, ga, gb):
            hip.free(b)
    hip.destroy_stream(sa)
    hip.destroy_stream(sb)
