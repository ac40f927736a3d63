"""GPU parity: the HIP path through the C ABI against the CPU oracle and the golden vectors.

Bars: ids, flags and counters bit-exact; radiance bit-exact against the oracle evaluated with the
same portable transcendentals; against vectors produced by the reference itself (C library
transcendentals) within 1e-4 relative per sample, for all but a small stated fraction of samples
whose paths branch differently after a one-ulp difference (1e-3 of the samples; none on the committed vectors).
"""
import ctypes as C
import os

import numpy as np
import pytest

import path_tracing_amd as pt
import scenes

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4           # per-sample relative tolerance against reference-generated vectors
OUTLIER_FRACTION = 1e-3  # samples allowed to exceed it (paths that branch differently after a one-ulp difference
                         # between the portable and the C library's transcendentals); measured on the committed
                         # vectors: none, largest relative error 3.6e-5


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bit-identical, except that any NaN matches any NaN."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def rel_outliers(a, ref):
    scale = np.maximum(np.abs(ref).max(axis=-1), 1e-3)
    err = np.abs(a - ref).max(axis=-1) / scale
    return float(np.mean(~(err <= REL_TOL)))


@pytest.fixture(scope="module")
def cornell(golden_dir):
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_square.ptcam"))
    dev = pt.DeviceScene(scene)
    yield scene, cam, dev
    dev.close()


def test_cornell_primary_hits_bit_exact(cornell, golden_dir, oracle):
    scene, cam, dev = cornell
    p = pt.make_params(64, 64, 1)
    got = dev.primary_hits(cam, p)
    want = oracle.primary_hits(scene.ptr, cam, p, math_mode=1)
    assert got.tobytes() == want.tobytes()
    golden = np.load(os.path.join(golden_dir, "cornell_hits_64.npy"))
    # the camera here has an identity rotation and no lens: its rays need no transcendental,
    # so even the distances agree with the reference's to the last bit
    assert got.tobytes() == golden.tobytes()


def test_cornell_trace_identical_rays(cornell, oracle, reference):
    scene, cam, dev = cornell
    p = pt.make_params(48, 48, 2)
    rays = reference.camera_rays(cam, p)
    got = dev.trace_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
    rs = reference.scene(scene.ptr)
    want = rs.trace_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
    assert got.tobytes() == want.tobytes()


def test_cornell_samples(cornell, golden_dir, oracle):
    scene, cam, dev = cornell
    p = pt.make_params(32, 32, 4)
    got = dev.render_samples(cam, p)
    want = oracle.render_samples(scene.ptr, cam, p, math_mode=1)
    assert same_bits(got, want)
    golden = np.load(os.path.join(golden_dir, "cornell_samples_32x16.npy"))
    assert rel_outliers(got, golden) <= OUTLIER_FRACTION


def test_cornell_glossy_samples(golden_dir, oracle):
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_square.ptcam"))
    p = pt.make_params(32, 32, 4)
    with pt.DeviceScene(scene) as dev:
        got = dev.render_samples(cam, p)
    assert same_bits(got, oracle.render_samples(scene.ptr, cam, p, math_mode=1))
    golden = np.load(os.path.join(golden_dir, "cornell_glossy_samples_32x16.npy"))
    assert rel_outliers(got, golden) <= OUTLIER_FRACTION


def test_cornell_image_and_counters(cornell, oracle):
    scene, cam, dev = cornell
    p = pt.make_params(40, 24, 3)
    img, cnt = dev.render_counted(cam, p)
    want, wcnt = oracle.render(scene.ptr, cam, p, math_mode=1, counters=True)
    assert same_bits(img, want)
    assert cnt.as_dict() == wcnt.as_dict()
    assert same_bits(dev.render(cam, p), want)


def test_cornell_tiles_and_sample_ranges(cornell, oracle):
    scene, cam, dev = cornell
    full = dev.render(cam, pt.make_params(32, 32, 2))
    # tiles give the pixels of the full render, bit for bit, and leave the rest untouched
    out = np.full((32, 32, 3), -1.0, np.float32)
    dev.render(cam, pt.make_params(32, 32, 2, tile=(5, 7, 21, 30)), out=out)
    assert same_bits(out[7:30, 5:21], full[7:30, 5:21])
    assert np.all(out[:7] == -1.0) and np.all(out[:, :5] == -1.0) and np.all(out[30:] == -1.0)
    # sample ranges: sums of disjoint ranges add up to the full sum (float addition order differs)
    a = dev.render(cam, pt.make_params(32, 32, 2, samples=(0, 3), flags=pt.abi.PT_RENDER_SUM_ONLY))
    b = dev.render(cam, pt.make_params(32, 32, 2, samples=(3, 4), flags=pt.abi.PT_RENDER_SUM_ONLY))
    np.testing.assert_allclose((a + b) / 4.0, full, rtol=1e-5, atol=1e-6)
    want = oracle.render(scene.ptr, cam, pt.make_params(32, 32, 2, samples=(0, 3), flags=pt.abi.PT_RENDER_SUM_ONLY),
                         math_mode=1)
    assert same_bits(a, want)


@pytest.mark.parametrize("name", scenes.FAMILIES)
def test_family_samples(name, golden_dir, oracle):
    s, cam, p = scenes.build(name)
    d = s.desc()
    with pt.DeviceScene(s) as dev:
        got = dev.render_samples(cam, p)
        hits = dev.primary_hits(cam, p)
    want = oracle.render_samples(C.byref(d), cam, p, math_mode=1)
    assert hits.tobytes() == oracle.primary_hits(C.byref(d), cam, p, math_mode=1).tobytes()
    assert same_bits(got, want)
    golden = np.load(os.path.join(golden_dir, "family_%s_samples.npy" % name))
    # (the animated family evaluates a slerp per camera ray: a fifth of its samples differ from the C-library
    # evaluation in some bit, none by more than the tolerance)
    assert rel_outliers(got, golden) <= OUTLIER_FRACTION


def test_materials_that_read_texture_coordinates_and_tangents(oracle):
    """Texture coordinates and tangent of a hit are computed only for surfaces whose material reads them (decided
    per surface at upload).  Every way a material can read them, on spheres and on a mesh: a normal map on a
    sphere, a two-sided material with a textured side, a mirror and a shininess taken from images, an opacity map;
    beside materials that read neither.  Samples bit-equal to the oracle's, scene in LDS and in memory."""
    rng = np.random.default_rng(21)
    s = pt.FlatScene()
    grey = s.material_lambertian(s.texture_constant((0.6, 0.6, 0.6)))
    light = s.material_light((9.0, 9.0, 8.0))
    scenes.room(s, grey, light)
    image = lambda w, h: s.texture_image(scenes.image_texture(rng, w, h))
    normal_map = s.texture_image((rng.random((8, 8, 3), dtype=np.float32) * 0.4 + np.array([0.3, 0.3, 0.6], np.float32)))
    const = s.texture_constant
    bumpy = s.material_phong(const((0.5, 0.3, 0.2)), const((0.4, 0.4, 0.4)), const(30.0), -1, normal_map)
    s.add_sphere((-2.0, -3.0, 0.0), 1.0, bumpy)
    s.add_sphere((0.0, -3.0, -1.5), 1.0, s.material_twosided(grey, s.material_lambertian(image(8, 4))))
    s.add_sphere((2.0, -3.0, 0.0), 1.0, s.material_mirror(image(6, 6)))
    s.add_sphere((0.0, -3.3, 1.5), 0.7, s.material_twosided(bumpy, s.material_base()))
    shin = s.texture_image(np.full((4, 4, 3), 20.0, np.float32) + rng.random((4, 4, 3), dtype=np.float32) * 40)
    opacity = s.texture_image(np.repeat(rng.random((5, 5, 1), dtype=np.float32) * 0.7 + 0.3, 3, axis=2))
    pos, nrm, tc, idx = scenes.uv_sphere_mesh(0.9, (-0.3, -1.0, 0.3))
    s.add_mesh(pos, idx, s.material_phong(const((0.2, 0.4, 0.6)), const((0.3, 0.3, 0.3)), shin, opacity, -1), nrm, tc)
    pos, nrm, tc, idx = scenes.uv_sphere_mesh(0.6, (2.2, -0.8, -1.0))
    s.add_mesh(pos, idx, s.material_phong(const((0.5, 0.5, 0.1)), const((0.2, 0.2, 0.2)), const(15.0)), nrm, tc)
    s.build_bvh()
    cam = scenes.build("materials")[1]
    p = pt.make_params(32, 24, 3, seed=13)
    d = s.desc()
    want = oracle.render_samples(C.byref(d), cam, p, math_mode=1)
    for env in (None, "1"):
        if env:
            os.environ["PT_NO_LDS_SCENE"] = env
        try:
            with pt.DeviceScene(s) as dev:
                got = dev.render_samples(cam, p)
        finally:
            if env:
                del os.environ["PT_NO_LDS_SCENE"]
        assert same_bits(got, want), env


def test_converged_image_rmse(cornell, golden_dir):
    scene, cam, dev = cornell
    ref = np.load(os.path.join(golden_dir, "cornell_converged_96_mt4096.npy"))  # reference generator, 4096 spp
    img = dev.render(cam, pt.make_params(96, 96, 64))                            # counter generator, 4096 spp
    rmse = float(np.sqrt(np.mean((img - ref) ** 2)))
    # two independent 4096-spp estimates of the same image: per-pixel noise ~ sigma * sqrt(2 / 4096)
    assert rmse < 0.02, rmse
    assert abs(float(img.mean()) - float(ref.mean())) < 2e-3


def test_mesh_primary_hits_and_samples(golden_dir, oracle):
    """A 45 002-triangle displaced grid: the tree no longer fits on chip, so the tracing kernel reads it
    from memory.  Against the oracle bit for bit, and against vectors produced by the reference."""
    s, cam = scenes.big_mesh_scene(150)
    d = s.desc()
    p_hits = pt.make_params(96, 54, 1)
    p_smp = pt.make_params(32, 18, 2)
    with pt.DeviceScene(s) as dev:
        hits = dev.primary_hits(cam, p_hits)
        smp = dev.render_samples(cam, p_smp)
        img = dev.render(cam, pt.make_params(48, 27, 3))
    assert hits.tobytes() == oracle.primary_hits(C.byref(d), cam, p_hits, math_mode=1).tobytes()
    golden = np.load(os.path.join(golden_dir, "mesh150_hits_96x54.npy"))
    # ids and sides as the reference has them; distances differ by an ulp where the camera's
    # rotation went through a C-library sine or cosine
    assert np.array_equal(hits["surface"], golden["surface"])
    assert np.array_equal(hits["material"], golden["material"])
    assert np.array_equal(hits["backside"], golden["backside"])
    np.testing.assert_allclose(hits["a"], golden["a"], rtol=1e-5)
    assert same_bits(smp, oracle.render_samples(C.byref(d), cam, p_smp, math_mode=1))
    assert rel_outliers(smp, np.load(os.path.join(golden_dir, "mesh150_samples_32x18x4.npy"))) <= OUTLIER_FRACTION
    want = oracle.render(C.byref(d), cam, pt.make_params(48, 27, 3), math_mode=1)
    assert same_bits(img, want)


def test_update_bvh_between_frames(oracle):
    """Frame loop of the reference's animation program (src/07/pathtracer-frames.cpp:205-233): the scene
    stays on the device, only the tree over the frame's shutter interval is replaced."""
    s, cam, p = scenes.build("animated")
    with pt.DeviceScene(s) as dev:
        for (t0, t1) in [(p.t0, p.t1), (6.0, 6.4), (0.5, 0.5)]:
            nodes = s.build_bvh(t0, t1)
            dev.update_bvh(nodes)
            q = pt.make_params(24, 14, 2, seed=5, t0=t0, t1=t1)
            got = dev.render_samples(cam, q)
            d = s.desc()
            assert same_bits(got, oracle.render_samples(C.byref(d), cam, q, math_mode=1)), (t0, t1)


def test_kernel_times_are_recorded(cornell):
    scene, cam, dev = cornell
    dev.profile(True)
    dev.render(cam, pt.make_params(64, 64, 2))
    times = dev.kernel_times()
    dev.profile(False)
    assert times["traceKernel"][0] > 0 and times["traceKernel"][1] > 0.0
    assert times["traceOrderedKernel"][0] == 0
    assert times["shadeHitsKernel"][0] > 0 and times["shadeHitsKernel"][1] > 0.0
    assert times["resolveKernel"][0] == 1


def test_walk_counters_of_the_executed_walk(cornell, oracle):
    """The instrumented tracing kernels.  With every shadow ray traced (PT_TRACE_NULL_SHADOW_RAYS) the walk in
    storage order (scene in LDS, or PT_TRAVERSAL=reference) is the reference's, so its sums equal the work counters
    of the reference algorithm -- except that camera rays missing the root box are finished by the generating kernel
    (one ray and one node visit each, never traced), and that the walk from LDS starts below the root.  By default the shading kernel leaves out the shadow rays that
    can only add zeros (they arrive at the light's dark side): fewer rays, rows and tests, counted, same image.  The
    nearer-child-first walk fetches fewer rows; the image is the same with and without counting."""
    scene, cam, dev = cornell
    p = pt.make_params(64, 48, 3)
    want, wcnt = oracle.render(scene.ptr, cam, p, math_mode=1, counters=True)
    ref = wcnt.as_dict()

    def counted(**env):
        os.environ.update(env)
        try:
            dev.count_walk(True)
            img = dev.render(cam, p)
            c = dev.walk_counters()
            dev.count_walk(False)
        finally:
            for k in env:
                del os.environ[k]
        assert same_bits(img, want)
        return c

    root_misses = None
    rows_of_every_node = None
    # (PT_NO_WALK_PRUNE: the walk from LDS with every node of the tree in it)
    for env in ({"PT_NO_WALK_PRUNE": "1"}, {"PT_NO_LDS_SCENE": "1", "PT_TRAVERSAL": "reference"}):
        in_lds = "PT_NO_LDS_SCENE" not in env
        c = counted(PT_TRACE_NULL_SHADOW_RAYS="1", **env)
        root_misses = ref["rays"] - (c["path_rays"] + c["shadow_rays"])
        assert 0 <= root_misses <= 64 * 48 * 9
        # (the walk from LDS does not test the root's box: a ray that misses it misses both children as well, which
        # are two visits where the reference has one -- a handful of rays that leave the open box near its rim)
        root_tests = c["path_rays"] + c["shadow_rays"] if in_lds else 0
        extra = c["box_rows"] + root_misses + root_tests - ref["node_visits"]
        assert extra % 2 == 0 and 0 <= extra <= (64 if in_lds else 0), (env, extra)
        if in_lds:
            rows_of_every_node = c["box_rows"]
        assert c["surface_tests"] == ref["leaf_tests"], env
        assert c["retry_rays"] == 0 and c["leaf_rows"] == 0 and c["null_shadow_rays"] == 0
        d = counted(**env)
        assert d["path_rays"] == c["path_rays"]
        assert d["null_shadow_rays"] > 0 and d["shadow_rays"] + d["null_shadow_rays"] == c["shadow_rays"], env
        assert d["box_rows"] < c["box_rows"] and d["surface_tests"] < c["surface_tests"]
    # The walk from LDS by default: inner nodes whose boxes most rays hit are left out of it (link table made with
    # the tree).  Fewer box tests, and exactly the reference's surface tests: a leaf's own box is still tested, and a
    # ray that misses a box that was left out misses every box inside it.
    c = counted(PT_TRACE_NULL_SHADOW_RAYS="1")
    assert c["path_rays"] + c["shadow_rays"] + root_misses == ref["rays"]
    assert c["surface_tests"] == ref["leaf_tests"]
    assert 0 < c["box_rows"] < rows_of_every_node
    print("box tests of the walk from LDS: %d with every node, %d with the link table" % (rows_of_every_node, c["box_rows"]))
    c = counted(PT_NO_LDS_SCENE="1", PT_TRACE_NULL_SHADOW_RAYS="1")
    assert c["path_rays"] + c["shadow_rays"] + root_misses == ref["rays"]
    assert 0 < c["box_rows"] < ref["node_visits"]
    assert c["retry_node_visits"] >= c["retry_rays"]
    # (four boxes per row by default; two with PT_NO_QUAD_ROWS: more rows for the same rays)
    two = counted(PT_NO_LDS_SCENE="1", PT_TRACE_NULL_SHADOW_RAYS="1", PT_NO_QUAD_ROWS="1")
    assert two["path_rays"] == c["path_rays"] and two["shadow_rays"] == c["shadow_rays"]
    assert c["box_rows"] < two["box_rows"] < ref["node_visits"]
    c = counted(PT_NO_LDS_SCENE="1")
    assert c["path_rays"] + c["shadow_rays"] + c["null_shadow_rays"] + root_misses == ref["rays"]
    # switched off: the sums stay where they are
    dev.render(cam, p)
    assert dev.walk_counters() == c


def test_tracing_kernel_variants_of_a_small_scene(golden_dir, oracle):
    """The walks a small static scene can be traced with give the oracle's samples bit for bit: node table held
    eight-fold in LDS (one copy per combination of direction signs, the default for a few dozen nodes), held once
    (PT_NO_OCTANT_ROWS), either with every node in the walk (PT_NO_WALK_PRUNE) or with the link table that leaves
    often-hit inner nodes out, left in memory in storage order and nearer child first; each with the shadow rays that can
    only add zeros dropped (default) and traced (PT_TRACE_NULL_SHADOW_RAYS), and with the last paths of the batch
    left to the iterations (PT_FINISH_PATHS=0) or all paths finished by the one-lane-per-path kernel as soon as the
    last one has been started; on the tree of the file and on the device-built one (whose root is not the file's:
    the walk from LDS starts below it)."""
    loaded = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    scene = pt.FlatScene.from_desc(loaded.desc())  # (a copy whose tree can be exchanged)
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_wide.ptcam"))
    p = pt.make_params(96, 54, 3)
    d = scene.desc()
    want = oracle.render_samples(C.byref(d), cam, p, math_mode=1)
    dev = pt.DeviceScene(scene)
    try:
        for built in (False, True):
            if built:
                dev.build_bvh_device(0.0, 0.0)
                scene.nodes = dev.get_bvh()
                d = scene.desc()
                want = oracle.render_samples(C.byref(d), cam, p, math_mode=1)
            for walk in ({}, {"PT_NO_OCTANT_ROWS": "1"}, {"PT_NO_WALK_PRUNE": "1"}, {"PT_NO_WALK_PRUNE": "1", "PT_NO_OCTANT_ROWS": "1"},
                         {"PT_NO_LDS_SCENE": "1", "PT_TRAVERSAL": "reference"}, {"PT_NO_LDS_SCENE": "1"},
                         {"PT_NO_LDS_SCENE": "1", "PT_NO_QUAD_ROWS": "1"}):
                for rays in ({}, {"PT_TRACE_NULL_SHADOW_RAYS": "1"}, {"PT_FINISH_PATHS": "0"}, {"PT_FINISH_PATHS": "100000000"}):
                    env = dict(walk, **rays)
                    os.environ.update(env)
                    try:
                        got = dev.render_samples(cam, p)
                    finally:
                        for k in env:
                            del os.environ[k]
                    assert same_bits(got, want), (built, env)
    finally:
        dev.close()


def test_render_multi_with_two_handles_on_one_gpu(cornell, oracle):
    """pt_render_multi rehearsed on one GPU: two scene handles (as if on two GPUs), one host thread each.  Tiles:
    the single-handle image bit for bit (also with a rectangle and a sample range, leaving the other pixels
    alone).  Samples: the sum of the handles' partial sums in handle order, then one division -- checked against
    that very sequence on the oracle's partial images, bit for bit, and against the sequential image to rounding."""
    scene, cam, dev = cornell
    with pt.DeviceScene(scene) as dev2, pt.DeviceScene(scene) as dev3:
        p = pt.make_params(600, 300, 3, seed=5)
        single = dev.render(cam, p)
        assert same_bits(pt.render_multi([dev, dev2], cam, p, pt.abi.PT_SHARD_TILES), single)
        assert same_bits(pt.render_multi([dev, dev2, dev3], cam, p, pt.abi.PT_SHARD_TILES), single)
        q = pt.make_params(600, 300, 3, seed=5, tile=(17, 40, 590, 299), samples=(2, 7))
        out = np.full((300, 600, 3), -2.0, np.float32)
        want = np.full((300, 600, 3), -2.0, np.float32)
        dev.render(cam, q, out=want)
        assert same_bits(pt.render_multi([dev, dev2], cam, q, pt.abi.PT_SHARD_TILES, out=out), want)
        # sample ranges
        small = pt.make_params(96, 64, 3, seed=6)
        got = pt.render_multi([dev, dev2, dev3], cam, small, pt.abi.PT_SHARD_SAMPLES)
        parts = [oracle.render(scene.ptr, cam, c, math_mode=1) for k in range(3)
                 for c in pt.multi_plan(small, pt.abi.PT_SHARD_SAMPLES, 3, k)]
        acc = parts[0]
        for part in parts[1:]:
            acc = acc + part
        assert same_bits(got, acc / np.float32(9.0))
        np.testing.assert_allclose(got, dev.render(cam, small), rtol=2e-6, atol=1e-7)
        # (the combine of three handles: one copy and one kernel per peer, nothing for the first handle's own sums)
        ms, kernels, copies, direct = pt.multi_last_combine(dev)
        assert kernels <= 3 and copies == 2 and direct == 2 and ms >= 0.0
        out = np.full((64, 96, 3), 5.0, np.float32)
        r = pt.make_params(96, 64, 3, seed=6, tile=(8, 8, 90, 60), samples=(1, 8))
        pt.render_multi([dev, dev2], cam, r, pt.abi.PT_SHARD_SAMPLES, out=out)
        assert np.all(out[:8] == 5.0) and np.all(out[:, 90:] == 5.0)
        np.testing.assert_allclose(out[8:60, 8:90], dev.render(cam, r)[8:60, 8:90], rtol=2e-6, atol=1e-7)
        # the device-output form: result in a buffer of the first handle's GPU, on a stream of the caller's; twice,
        # the second time into the buffers the first call left with the handles
        from hiprt import Hip
        hip = Hip()
        stream = hip.stream()
        d_out = hip.zeros(300 * 600 * 12)
        try:
            for mode, want_img in ((pt.abi.PT_SHARD_TILES, single), (pt.abi.PT_SHARD_SAMPLES, None), (pt.abi.PT_SHARD_TILES, single)):
                pt.render_multi_device([dev, dev2, dev3], cam, p, d_out.value, mode, stream.value)
                hip.sync()
                got = hip.to_host(d_out, (300, 600, 3))
                if want_img is not None:
                    assert same_bits(got, want_img)
                else:
                    np.testing.assert_allclose(got, single, rtol=2e-6, atol=1e-7)
            ms, kernels, copies, direct = pt.multi_last_combine(dev)
            assert kernels == 2 and copies == 2
            print("combine of three handles at 600x300: %.3f ms" % ms)
        finally:
            hip.free(d_out)
            hip.destroy_stream(stream)


def test_full_size_properties(oracle, golden_dir):
    """The benchmark's image size (1920x1080), through properties that need no full-size oracle render:
    primary hits against the oracle (2 M rays, seconds on the CPU), a tile of the full render bit for
    bit, sample ranges adding up, and the image mean against the converged reference image."""
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_wide.ptcam"))
    W, H = 1920, 1080
    with pt.DeviceScene(scene) as dev:
        p1 = pt.make_params(W, H, 1)
        hits = dev.primary_hits(cam, p1)
        assert hits.tobytes() == oracle.primary_hits(scene.ptr, cam, p1, math_mode=1).tobytes()
        full = dev.render(cam, pt.make_params(W, H, 4))
        assert np.isfinite(full).all()
        tile = np.zeros_like(full)
        dev.render(cam, pt.make_params(W, H, 4, tile=(901, 333, 1187, 700)), out=tile)
        assert same_bits(tile[333:700, 901:1187], full[333:700, 901:1187])
        a = dev.render(cam, pt.make_params(W, H, 4, samples=(0, 7), flags=pt.abi.PT_RENDER_SUM_ONLY))
        b = dev.render(cam, pt.make_params(W, H, 4, samples=(7, 16), flags=pt.abi.PT_RENDER_SUM_ONLY))
        np.testing.assert_allclose((a + b) / 16.0, full, rtol=2e-5, atol=1e-6)
        # one 64x64 block of the full image against the oracle, bit for bit
        blk = oracle.render(scene.ptr, cam, pt.make_params(W, H, 4, tile=(928, 508, 992, 572)), math_mode=1)
        assert same_bits(full[508:572, 928:992], blk[508:572, 928:992])


@pytest.fixture
def scene_in_memory():
    """Make the renderer treat every scene as too large for its on-chip copy, so that small test scenes
    go through the kernels that big scenes use (nearer-child-first walk + storage-order second pass)."""
    os.environ["PT_NO_LDS_SCENE"] = "1"
    yield
    del os.environ["PT_NO_LDS_SCENE"]


@pytest.mark.parametrize("name", ["cornell"] + scenes.FAMILIES)
def test_samples_with_the_scene_in_memory(name, golden_dir, oracle, scene_in_memory):
    if name == "cornell":
        s = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
        cam = pt.load_camera(os.path.join(golden_dir, "cornell_square.ptcam"))
        p = pt.make_params(40, 40, 3)
        d = s.ptr
    else:
        s, cam, p = scenes.build(name)
        desc = s.desc()
        d = C.byref(desc)
    with pt.DeviceScene(s) as dev:
        got = dev.render_samples(cam, p)
    assert same_bits(got, oracle.render_samples(d, cam, p, math_mode=1))


def test_walk_orders_agree_on_a_mesh(oracle):
    """The nearer-child-first walk and the storage-order walk give the same image, bit for bit, and
    both the oracle's; 230 400 paths over a 45 002-triangle mesh seen at a grazing angle."""
    s, cam = scenes.big_mesh_scene(150)
    p = pt.make_params(160, 90, 4)
    with pt.DeviceScene(s) as dev:
        fast = dev.render(cam, p)
        os.environ["PT_TRAVERSAL"] = "reference"
        try:
            slow = dev.render(cam, p)
        finally:
            del os.environ["PT_TRAVERSAL"]
    assert same_bits(fast, slow)
    d = s.desc()
    assert same_bits(fast, oracle.render(C.byref(d), cam, p, math_mode=1))


def test_output_stage(cornell, golden_dir, oracle):
    """Device to8Bit + row flip (+ uniformRationalQuantization): byte-exact against the oracle, and against the
    reference-generated bytes up to the stated rate of one-off differences from the C library's pow."""
    img = scenes.output_stage_image()
    g = np.load(os.path.join(golden_dir, "output_stage.npz"))
    for key, kw in (("plain", {}), ("urq", {"urq": (2.5, 20.0)}), ("unflipped", {"flip_rows": False})):
        got = pt.image_to_8bit(img, **kw)
        assert np.array_equal(got, oracle.image_to_8bit(img, math_mode=1, **kw)), key
        diff = np.abs(got.astype(np.int32) - g[key].astype(np.int32))
        assert diff.max() <= 1 and np.mean(diff != 0) <= 1e-3, key
    # a rendered image, as the reference's main does after the pixel loop (src/15/pathtracer.cpp:225)
    scene, cam, dev = cornell
    rgb = dev.render(cam, pt.make_params(96, 64, 4))
    assert np.array_equal(pt.image_to_8bit(rgb), oracle.image_to_8bit(rgb, math_mode=1))


def test_edge_cases(cornell, oracle):
    """Ragged and degenerate inputs: one pixel, sizes that are not multiples of the 8x8 path tiles, one
    sample, an empty tile and an empty sample range, a scene of a single surface, a scene without lights."""
    scene, cam, dev = cornell
    for (w, h, s) in [(1, 1, 1), (1, 1, 3), (33, 17, 1), (7, 9, 2), (130, 3, 1)]:
        p = pt.make_params(w, h, s)
        assert same_bits(dev.render_samples(cam, p), oracle.render_samples(scene.ptr, cam, p, math_mode=1)), (w, h, s)
        assert same_bits(dev.render(cam, p), oracle.render(scene.ptr, cam, p, math_mode=1)), (w, h, s)
    # nothing to do: the output is left alone
    out = np.full((16, 16, 3), 7.0, np.float32)
    dev.render(cam, pt.make_params(16, 16, 2, tile=(5, 5, 5, 9)), out=out)
    assert np.all(out == 7.0)
    # bad arguments are errors
    with pytest.raises(pt.PtError):
        dev.render(cam, pt.make_params(16, 16, 2, tile=(5, 5, 40, 9)))
    with pytest.raises(pt.PtError):
        dev.render(cam, pt.make_params(16, 16, 2, samples=(3, 9)))
    with pytest.raises(pt.PtError):
        dev.render(cam, pt.make_params(0, 16, 2))
    # a single triangle that is its own light, seen from both sides; and a sphere with no light at all
    one = pt.FlatScene()
    one.add_mesh(np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32), np.array([0, 1, 2], np.uint32),
                 one.material_light((2.0, 3.0, 4.0)), is_light=True)
    one.build_bvh()
    dark = pt.FlatScene()
    dark.add_sphere((0.0, 0.0, 0.0), 1.0, dark.material_lambertian(dark.texture_constant((0.5, 0.5, 0.5))))
    dark.set_envmap_equirect(dark.texture_constant((1.0, 0.9, 0.8)))
    dark.build_bvh()
    for s in (one, dark):
        for eye in ((0.0, 0.3, 3.0), (0.2, 0.0, -3.0)):
            c = pt.make_camera(0.5, 0.5, 1.0, 0.0, scenes.lookat_animation(eye, (0.0, 0.0, 0.0)))
            p = pt.make_params(20, 20, 2)
            d = s.desc()
            with pt.DeviceScene(s) as ds:
                assert same_bits(ds.render_samples(c, p), oracle.render_samples(C.byref(d), c, p, math_mode=1))


def test_small_work_areas(cornell, oracle):
    """A sample buffer that holds only a few samples per pixel (several batches, sums carried from batch to
    batch in sample order) and a path pool of 1024 slots (every slot reused hundreds of times) give the same
    bits as the default work areas."""
    scene, cam, dev = cornell
    p = pt.make_params(128, 128, 4)
    want = oracle.render(scene.ptr, cam, p, math_mode=1)
    os.environ["PT_SAMPLE_BUFFER_MB"] = "1"   # 128*128*12 B per sample -> 5 samples per batch, 4 batches
    os.environ["PT_WAVEFRONT_SLOTS"] = "1024"
    try:
        got = dev.render(cam, p)
    finally:
        del os.environ["PT_SAMPLE_BUFFER_MB"]
        del os.environ["PT_WAVEFRONT_SLOTS"]
    assert same_bits(got, want)
    assert same_bits(dev.render(cam, p), want)


def test_dropin_program(tmp_path, oracle):
    """oracle/_ref/dropin_demo is a program written against the reference's own classes whose pixel loop is
    ptamd::render and whose 8-bit conversion is ptamd::to8Bit: what it renders through the binding and the
    library equals the oracle's render of the flat scene the binding produced, bit for bit."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "dropin_demo")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_demo not present")
    prefix = str(tmp_path / "demo")
    w, h, s = 72, 48, 3
    r = subprocess.run([exe, prefix, str(w), str(h), str(s)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = np.fromfile(prefix + ".img", np.float32).reshape(h, w, 3)
    u8 = np.fromfile(prefix + ".u8", np.uint8).reshape(h, w, 3)
    scene = pt.LoadedScene(prefix + ".ptscene")
    cam = pt.load_camera(prefix + ".ptcam")
    want = oracle.render(scene.ptr, cam, pt.make_params(w, h, s, seed=42), math_mode=1)
    assert same_bits(img, want)
    assert np.array_equal(u8, oracle.image_to_8bit(img, flip_rows=False, math_mode=1))
    assert img.mean() > 0.01
    # the same program rendering through ptamd::renderMulti (two scene handles, image split by tiles): the same bits
    r = subprocess.run([exe, prefix + "_multi", "600", "300", str(s), "multi"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img_multi = np.fromfile(prefix + "_multi.img", np.float32).reshape(300, 600, 3)
    cam_multi = pt.load_camera(prefix + "_multi.ptcam")
    assert same_bits(img_multi, oracle.render(scene.ptr, cam_multi, pt.make_params(600, 300, s, seed=42), math_mode=1))
    # the same program leaving the tree to the binding (built on the GPU): the same image except where a ray
    # meets two surfaces at exactly the same distance
    r = subprocess.run([exe, prefix + "_dev", str(w), str(h), str(s), "devicebvh"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    img_dev = np.fromfile(prefix + "_dev.img", np.float32).reshape(h, w, 3)
    assert np.mean(bits(img_dev) != bits(want)) <= 2e-3


def test_importer_image_passes(golden_dir, oracle):
    """pt_texture_from_bytes and pt_bump_to_normal_map against the oracle and the reference-generated vectors,
    bit for bit; also on an image large enough for many blocks."""
    img = scenes.ingest_image()
    g = np.load(os.path.join(golden_dir, "ingest.npz"))
    for key, lin in (("linear", True), ("raw", False)):
        got = pt.texture_from_bytes(img, linearize=lin)
        assert np.array_equal(bits(got), bits(g[key]))
    assert np.array_equal(bits(pt.texture_from_bytes(img, flip_rows=False)), bits(g["linear"][::-1]))
    for key, f in (("normal_1", 1.0), ("normal_8", 8.0)):
        assert np.array_equal(bits(pt.bump_to_normal_map(g["raw"], f)), bits(g[key]))
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (700, 900, 3)).astype(np.uint8)
    tex = pt.texture_from_bytes(big, linearize=False)
    assert np.array_equal(bits(tex), bits(oracle.texture_from_bytes(big, linearize=False)))
    assert np.array_equal(bits(pt.bump_to_normal_map(tex, 3.0)), bits(oracle.bump_to_normal_map(tex, 3.0)))
