"""The acceleration structure built on the device (pt_scene_build_bvh_device) against the host builder
that restates the reference's (pt_bvh_build = BVHNode::buildBVH + flattenBVH, reference
src/15/bvh.hpp:36-108,146-176) and against the oracle walking the device-built tree.

What must hold: leaf boxes are the reference's Surface::aabb bit for bit; the tree is a valid pre-order
tree whose inner boxes are the merge of their children's; the device's walks over that tree give what
the oracle's storage-order walk gives over the same tree, bit for bit; and against the reference's own
tree every ray finds the same distance, the same surface except for exact ties."""
import ctypes as C
import os

import numpy as np
import pytest

import path_tracing_amd as pt
import scenes

pytestmark = pytest.mark.gpu

NO_HIT = 0xFFFFFFFF


@pytest.fixture(params=["sah", "morton"], autouse=True)
def builder(request):
    """Every test runs with both device builders (PT_BVH_BUILDER; the surface-area builder is the default)."""
    os.environ["PT_BVH_BUILDER"] = request.param
    yield request.param
    del os.environ["PT_BVH_BUILDER"]


def check_tree(nodes, n_surfaces):
    """Pre-order layout, every surface in exactly one leaf, inner boxes = merge of the children's."""
    n = len(nodes)
    assert n == 2 * n_surfaces - 1
    link = nodes["link"]
    inner = (link % 2) == 1
    second = (link // 2).astype(np.int64)
    sids = second[~inner]
    assert np.array_equal(np.sort(sids), np.arange(n_surfaces))
    idx = np.arange(n)
    assert np.all(second[inner] > idx[inner] + 1) and np.all(second[inner] < n)
    # subtree sizes from the back: a leaf is 1, an inner node 1 + size(first) + size(second); the second
    # child must start right after the first child's subtree
    size = np.ones(n, np.int64)
    for k in range(n - 1, -1, -1):
        if inner[k]:
            assert second[k] == k + 1 + size[k + 1], k
            size[k] = 1 + size[k + 1] + size[second[k]]
    assert size[0] == n
    k = idx[inner]
    lo = np.minimum(nodes["lo"][k + 1], nodes["lo"][second[k]])
    hi = np.maximum(nodes["hi"][k + 1], nodes["hi"][second[k]])
    assert np.array_equal(lo, nodes["lo"][k]) and np.array_equal(hi, nodes["hi"][k])


def leaf_boxes(nodes, n_surfaces):
    leaf = (nodes["link"] % 2) == 0
    sid = (nodes["link"][leaf] // 2).astype(np.int64)
    lo = np.zeros((n_surfaces, 3), np.float32)
    hi = np.zeros((n_surfaces, 3), np.float32)
    lo[sid] = nodes["lo"][leaf]
    hi[sid] = nodes["hi"][leaf]
    return lo, hi


def check_quad_rows(dev, nodes):
    """The rows of the four-wide walk as made on the device are those the host makes from the same tree: the same
    boxes in the same slots, the same rows behind inner slots; a leaf is linked by its node or, a static triangle
    with the box of its vertices, by its surface."""
    got = dev.get_quad_rows()
    if len(got) == 0:
        return
    want, _, _ = pt.host_quad_rows(nodes)
    assert got.shape == want.shape
    assert np.array_equal(got[:, :12], want[:, :12])
    g, w = got[:, 12:].astype(np.int64), want[:, 12:].astype(np.int64)
    by_node = ((w & 0xC0000000) == 0x40000000) & (w != 0x7FFFFFFF)
    assert np.array_equal(g[~by_node], w[~by_node])
    node = w[by_node] & 0x3FFFFFFF
    surface = (nodes["link"][node] // 2).astype(np.int64)
    assert np.all((g[by_node] == w[by_node]) | (g[by_node] == surface))


def rays_for(oracle, d, cam, p, seed=1):
    rays = oracle.camera_rays(d, cam, p, math_mode=1)
    o, dirs, t = rays[:, 0:3].copy(), rays[:, 3:6].copy(), rays[:, 6].copy()
    first = oracle.trace_rays(d, o, dirs, t)
    hit = first["surface"] != NO_HIT
    rng = np.random.default_rng(seed)
    P = (o[hit] + dirs[hit] * first["a"][hit][:, None]).astype(np.float32)
    D = rng.normal(size=P.shape).astype(np.float32)
    D /= np.linalg.norm(D, axis=1)[:, None]
    return (np.concatenate([o, P]), np.concatenate([dirs, D]), np.concatenate([t, t[hit]]))


@pytest.mark.parametrize("in_memory", [False, True])
@pytest.mark.parametrize("name", scenes.FAMILIES + ["mesh"])
def test_device_built_tree(name, in_memory, oracle):
    if name == "mesh":
        s, cam = scenes.big_mesh_scene(60)
        p = pt.make_params(96, 54, 1, seed=3)
    else:
        s, cam, p = scenes.build(name)
    ref_nodes = s.build_bvh(p.t0, p.t1).copy()
    d_ref = s.desc()
    o, dirs, t = rays_for(oracle, C.byref(d_ref), cam, p)
    want_ref = oracle.trace_rays(C.byref(d_ref), o, dirs, t)
    if in_memory:
        os.environ["PT_NO_LDS_SCENE"] = "1"
    try:
        with pt.DeviceScene(s) as dev:
            dev.build_bvh_device(p.t0, p.t1)
            nodes = dev.get_bvh()
            check_tree(nodes, len(s.surfaces))
            dev.build_bvh_device(p.t0, p.t1)
            assert dev.get_bvh().tobytes() == nodes.tobytes()  # equal inputs, equal trees
            # leaf boxes: the host builder's (static surfaces bit for bit; a moving surface's 16 poses go through
            # the C library's sin/cos there and the portable ones here, an ulp apart at most)
            lo, hi = leaf_boxes(nodes, len(s.surfaces))
            rlo, rhi = leaf_boxes(ref_nodes, len(s.surfaces))
            if name == "animated":
                assert np.allclose(lo, rlo, rtol=1e-5, atol=1e-6) and np.allclose(hi, rhi, rtol=1e-5, atol=1e-6)
            else:
                assert np.array_equal(lo, rlo) and np.array_equal(hi, rhi)
            check_quad_rows(dev, nodes)
            # the device's walk over its tree = the oracle's storage-order walk over the same tree
            s.nodes = nodes
            d_dev = s.desc()
            got = dev.trace_rays(o, dirs, t)
            want = oracle.trace_rays(C.byref(d_dev), o, dirs, t)
            assert got.tobytes() == want.tobytes()
            # and per-sample radiance
            q = pt.make_params(24, 16, 2, seed=9, t0=p.t0, t1=p.t1)
            assert np.array_equal(dev.render_samples(cam, q).view(np.uint32),
                                  oracle.render_samples(C.byref(d_dev), cam, q, math_mode=1).view(np.uint32))
    finally:
        if in_memory:
            del os.environ["PT_NO_LDS_SCENE"]
    # against the reference's tree: same distance everywhere, same surface except for ties
    if name != "animated":
        both = (got["surface"] != NO_HIT) & (want_ref["surface"] != NO_HIT)
        differ_hit = np.mean((got["surface"] != NO_HIT) != (want_ref["surface"] != NO_HIT))
        assert differ_hit <= 1e-4, differ_hit
        assert np.mean(got["a"][both] != want_ref["a"][both]) <= 1e-4
        assert np.mean(got["surface"][both] != want_ref["surface"][both]) <= 5e-3


def tree_cost(nodes):
    """Surface-area cost of a tree: box areas of all nodes relative to the root's."""
    d = (nodes["hi"] - nodes["lo"]).astype(np.float64)
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    return float(area.sum() / area[0])


def test_surface_area_builder_gives_the_cheaper_tree(builder):
    """On the displaced grid (7 200 triangles: passes over large nodes and one-wave subtrees) the surface-area
    builder's tree costs less than the radix tree over Morton codes and than the reference builder's."""
    if builder != "sah":
        pytest.skip("compares the two builders itself")
    s, cam = scenes.big_mesh_scene(60)
    host = tree_cost(s.nodes)
    cost = {}
    with pt.DeviceScene(s) as dev:
        for which in ("sah", "morton"):
            os.environ["PT_BVH_BUILDER"] = which
            dev.build_bvh_device()
            cost[which] = tree_cost(dev.get_bvh())
    os.environ["PT_BVH_BUILDER"] = "sah"
    assert cost["sah"] < 0.95 * cost["morton"] and cost["sah"] < host, (cost, host)


def test_unknown_builder_is_refused():
    s, cam, p = scenes.build("spheres")
    with pt.DeviceScene(s) as dev:
        os.environ["PT_BVH_BUILDER"] = "best"
        with pytest.raises(pt.PtError):
            dev.build_bvh_device()


def test_scene_created_without_a_tree(oracle):
    """n_bvh_nodes == 0 at creation: the first tree comes from the device builder; rebuilding per frame keeps
    working (frame loop of reference src/07/pathtracer-frames.cpp:205-233)."""
    s, cam, p = scenes.build("animated")
    s.nodes = np.zeros(0, pt.NODE_DTYPE)
    with pt.DeviceScene(s) as dev:
        for (t0, t1) in [(p.t0, p.t1), (6.0, 6.4)]:
            dev.build_bvh_device(t0, t1)
            nodes = dev.get_bvh()
            check_tree(nodes, len(s.surfaces))
            s.nodes = nodes
            d = s.desc()
            q = pt.make_params(24, 14, 2, seed=5, t0=t0, t1=t1)
            assert np.array_equal(dev.render_samples(cam, q).view(np.uint32),
                                  oracle.render_samples(C.byref(d), cam, q, math_mode=1).view(np.uint32))
            s.nodes = np.zeros(0, pt.NODE_DTYPE)


def test_single_surface_scene(oracle):
    """One surface: the tree is one leaf."""
    s = pt.FlatScene()
    s.add_sphere((0.0, 0.0, -3.0), 1.0, s.material_light((2.0, 3.0, 4.0)), is_light=True)
    s.build_bvh()
    cam = scenes.build("spheres")[1]
    with pt.DeviceScene(s) as dev:
        dev.build_bvh_device()
        nodes = dev.get_bvh()
        assert len(nodes) == 1 and nodes["link"][0] == 0
        assert np.array_equal(nodes["lo"], s.nodes["lo"]) and np.array_equal(nodes["hi"], s.nodes["hi"])
        q = pt.make_params(16, 12, 2, seed=2)
        d = s.desc()
        assert np.array_equal(dev.render_samples(cam, q).view(np.uint32),
                              oracle.render_samples(C.byref(d), cam, q, math_mode=1).view(np.uint32))


def test_degenerate_inputs(oracle):
    """Surfaces with identical boxes (equal Morton codes are told apart by position; no plane separates their
    centres, so the surface-area builder halves them by place -- 300 of them exercise that in its passes over large
    nodes, 37 in its one-wave subtrees), a scene flat in one axis, and two surfaces only."""
    cam = scenes.build("spheres")[1]
    q = pt.make_params(20, 14, 2, seed=4)
    for variant in ("duplicates", "many duplicates", "flat", "two"):
        s = pt.FlatScene()
        light = s.material_light((6.0, 6.0, 6.0))
        grey = s.material_lambertian(s.texture_constant((0.6, 0.6, 0.6)))
        if variant in ("duplicates", "many duplicates"):
            for _ in range(37 if variant == "duplicates" else 300):
                s.add_sphere((0.5, 0.0, 0.0), 0.8, grey)
            for _ in range(5):
                s.add_sphere((-1.5, 1.0, 0.5), 0.4, light, is_light=True)
            s.add_sphere((0.0, -50.0, 0.0), 49.0, grey)
        elif variant == "flat":
            for i in range(9):
                for j in range(9):
                    s.add_sphere((i - 4.0, 0.0, j - 4.0), 0.0, grey)  # radius 0: boxes are points in one plane
            s.add_sphere((0.0, 0.0, 0.0), 0.0, light, is_light=True)
        else:
            s.add_sphere((0.0, 0.0, 0.0), 1.0, grey)
            s.add_sphere((0.0, 4.0, 0.0), 1.0, light, is_light=True)
        s.build_bvh()
        with pt.DeviceScene(s) as dev:
            dev.build_bvh_device()
            nodes = dev.get_bvh()
            check_tree(nodes, len(s.surfaces))
            s.nodes = nodes
            d = s.desc()
            for env in (None, "1"):
                if env:
                    os.environ["PT_NO_LDS_SCENE"] = env
                try:
                    got = dev.render_samples(cam, q)
                finally:
                    if env:
                        del os.environ["PT_NO_LDS_SCENE"]
                assert np.array_equal(got.view(np.uint32),
                                      oracle.render_samples(C.byref(d), cam, q, math_mode=1).view(np.uint32)), (variant, env)


def test_surfaces_of_very_different_sizes(oracle):
    """3 000 spheres with radii over five orders of magnitude, a third of them in one tight cluster and a hundred
    identical ones inside it: bins that hold almost everything, planes that separate nothing, subtrees of one wave
    and passes over large nodes -- the tree is valid and both device walks agree with the oracle on it."""
    rng = np.random.default_rng(7)
    s = pt.FlatScene()
    grey = s.material_lambertian(s.texture_constant((0.6, 0.6, 0.6)))
    light = s.material_light((5.0, 5.0, 5.0))
    n = 3000
    centres = rng.uniform(-50.0, 50.0, (n, 3))
    centres[: n // 3] = rng.normal(0.0, 0.05, (n // 3, 3)) + np.array([3.0, 1.0, -2.0])
    radii = 10.0 ** rng.uniform(-4.0, 1.0, n)
    radii[: n // 3] *= 1e-2
    for k in range(n):
        s.add_sphere(tuple(float(c) for c in centres[k]), float(radii[k]), grey)
    for _ in range(100):
        s.add_sphere((3.0, 1.0, -2.0), 0.01, grey)
    s.add_sphere((0.0, 80.0, 0.0), 10.0, light, is_light=True)
    s.build_bvh()
    cam = scenes.build("spheres")[1]
    q = pt.make_params(48, 32, 2, seed=6)
    for env in (None, "1"):
        if env:
            os.environ["PT_NO_LDS_SCENE"] = env
        try:
            with pt.DeviceScene(s) as dev:
                dev.build_bvh_device()
                nodes = dev.get_bvh()
                check_tree(nodes, len(s.surfaces))
                flat_nodes, s.nodes = s.nodes, nodes
                d = s.desc()
                got = dev.render_samples(cam, q)
                want = oracle.render_samples(C.byref(d), cam, q, math_mode=1)
                s.nodes = flat_nodes
        finally:
            if env:
                del os.environ["PT_NO_LDS_SCENE"]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), env
