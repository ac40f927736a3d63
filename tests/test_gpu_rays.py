"""Caller-supplied rays through the tracing kernels of the renderer (pt_trace_rays with the render loop's bounds),
hit for hit against the oracle's storage-order walk (reference src/15/bvh.hpp:183-213, aabb.hpp:41-60).

The rays are the ones a slab test can get wrong: directions parallel to one or two axes (infinite inverse
components, 0 * inf = NaN slabs), negative zeros, origins lying in a box plane or on a surface, rays grazing
large and thin triangles.  Every scene is traced three ways: tables in LDS (storage-order kernel), tables in
memory (nearer-child-first kernel + second pass), and tables in memory in storage order.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import path_tracing_amd as pt
import scenes
from hiprt import Hip

pytestmark = pytest.mark.gpu

MODES = [("default", {}), ("memory", {"PT_NO_LDS_SCENE": "1"}),
         ("memory, two-wide rows", {"PT_NO_LDS_SCENE": "1", "PT_NO_QUAD_ROWS": "1"}),
         ("memory, storage order", {"PT_NO_LDS_SCENE": "1", "PT_TRAVERSAL": "reference"}),
         ("one lane per ray", {"PT_TRACE_RAYS": "simple"})]


def trace_all_modes(dev, o, d, t=None):
    out = {}
    for name, env in MODES:
        os.environ.update(env)
        try:
            out[name] = dev.trace_rays(o, d, t)
        finally:
            for k in env:
                del os.environ[k]
    return out


def edge_rays(lo, hi, rng, n_random=400):
    """Rays around the box [lo, hi]: axis-parallel from inside, outside and from the box planes, with +0 and -0 in
    the zero components, diagonal rays in coordinate planes, and rays starting on the planes."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    c, e = (lo + hi) / 2, (hi - lo)
    O, D = [], []
    axes = np.eye(3)
    zeros = [0.0, -0.0]
    for a in range(3):
        for sgn in (1.0, -1.0):
            for z1 in zeros:
                for z2 in zeros:
                    d = np.insert(np.array([z1, z2]), a, sgn)
                    # from the centre, from outside along the axis, and from points in the planes of the box
                    starts = [c, c - sgn * axes[a] * e[a], c + 0.25 * e * np.array([1, -1, 1])]
                    for b in range(3):
                        p = c.copy()
                        p[b] = lo[b]
                        starts.append(p.copy())
                        p[b] = hi[b]
                        starts.append(p.copy())
                    starts.append(lo.copy())
                    starts.append(hi.copy())
                    for s in starts:
                        O.append(s)
                        D.append(d)
    # diagonals inside coordinate planes (one zero component)
    for a in range(3):
        for k in range(24):
            ang = 2 * np.pi * k / 24
            d = np.insert(np.array([np.cos(ang), np.sin(ang)]), a, rng.choice(zeros))
            O.append(c + rng.uniform(-0.5, 0.5, 3) * e)
            D.append(d)
    # random origins on the planes with random and axis-parallel directions
    for _ in range(n_random):
        p = lo + rng.random(3) * e
        b = rng.integers(0, 3)
        p[b] = (lo, hi)[rng.integers(0, 2)][b]
        if rng.random() < 0.5:
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
        else:
            d = np.zeros(3)
            d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
            d[d == 0] = rng.choice(zeros, int((d == 0).sum()))
        O.append(p)
        D.append(d)
    return np.asarray(O, np.float32), np.asarray(D, np.float32)


def check(dev, desc, oracle, o, d, t=None, what="", in_plane=None):
    """Every way of tracing gives the oracle's storage-order result.  `in_plane` marks rays that lie in the plane
    of some triangle to within rounding: the reference's triangle test returns an arbitrary distance for those
    (its only guard is |Dpre| >= FLT_EPSILON, reference src/15/surface_triangle.hpp:110-114), possibly nearer than
    the triangle's own box.  The storage-order kernels reproduce that; the nearer-child-first kernel is only
    promised to when the triangle in question lies in a subtree it visits, so for those rays it is held to its CPU
    model (oracle.trace_rays_ordered) instead, which is checked against the reference on every other ray."""
    want = oracle.trace_rays(desc, o, d, t)
    got = trace_all_modes(dev, o, d, t)
    models = {}
    for name, hits in got.items():
        if in_plane is not None and name in ("default", "memory", "memory, two-wide rows"):
            width = 2 if name == "memory, two-wide rows" else 4
            if width not in models:
                models[width], _ = oracle.trace_rays_ordered(desc, o, d, t, width=width)
                assert models[width][~in_plane].tobytes() == want[~in_plane].tobytes(), what
            if hits.tobytes() != want.tobytes():
                assert hits.tobytes() == models[width].tobytes(), (what, name)
                assert np.all(in_plane[hits != want]), (what, name)
            continue
        assert hits.tobytes() == want.tobytes(), (what, name, int(np.sum(hits != want)))
    return want


def rays_in_a_triangle_plane(pos, dirs, threshold=1e-6):
    """True for directions within `threshold` (sine of the angle, times the sine of the angle between the two
    edges) of the plane of any of the triangles -- the rays for which Dpre of the reference's triangle test is
    rounding noise."""
    tri = pos.reshape(-1, 3, 3).astype(np.float64)
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    n = np.cross(e1, e2)
    scale = np.linalg.norm(e1, axis=1) * np.linalg.norm(e2, axis=1)
    out = np.zeros(len(dirs), bool)
    for i in range(0, len(dirs), 512):
        dd = dirs[i:i + 512].astype(np.float64)
        out[i:i + 512] = (np.abs(dd @ n.T) / scale[None, :]).min(axis=1) < threshold
    return out


def test_edge_rays_cornell(golden_dir, oracle):
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    nodes = scene.nodes()
    rng = np.random.default_rng(2)
    with pt.DeviceScene(scene) as dev:
        # around the root box and around a few inner and leaf boxes
        for k in (0, 1, 5, 17, len(nodes) - 1):
            o, d = edge_rays(nodes["lo"][k], nodes["hi"][k], rng)
            want = check(dev, scene.ptr, oracle, o, d, what="box %d" % k)
            assert (want["surface"] != 0xFFFFFFFF).mean() > 0.3
        # from points on the surfaces (hit points of camera rays), axis-parallel and random
        cam = pt.load_camera(os.path.join(golden_dir, "cornell_square.ptcam"))
        rays = oracle.camera_rays(scene.ptr, cam, pt.make_params(40, 40, 1), math_mode=1)
        h = oracle.trace_rays(scene.ptr, rays[:, 0:3], rays[:, 3:6])
        ok = h["surface"] != 0xFFFFFFFF
        P = (rays[ok, 0:3] + rays[ok, 3:6] * h["a"][ok][:, None]).astype(np.float32)
        A = np.zeros_like(P)
        A[np.arange(len(P)), rng.integers(0, 3, len(P))] = rng.choice([-1.0, 1.0], len(P))
        A[A == 0] = rng.choice([0.0, -0.0], int((A == 0).sum()))
        check(dev, scene.ptr, oracle, P, A, what="axis rays from surfaces")
        R = rng.normal(size=P.shape).astype(np.float32)
        R /= np.linalg.norm(R, axis=1)[:, None]
        check(dev, scene.ptr, oracle, P, R, what="random rays from surfaces")


def triangle_soup(rng, n, extent, thin):
    """n triangles in [-extent, extent]^3; `thin`: long needles and large axis-aligned sheets among them."""
    pos = np.zeros((3 * n, 3), np.float32)
    for i in range(n):
        c = rng.uniform(-extent, extent, 3)
        if thin and i % 3 == 0:
            # a needle: two vertices close together, the third far away
            a = rng.normal(size=3)
            a /= np.linalg.norm(a)
            tri = np.stack([c, c + a * extent * rng.uniform(0.5, 2.0), c + rng.normal(size=3) * extent * 1e-3])
        elif thin and i % 3 == 1:
            # a large sheet lying exactly in an axis plane (flat box)
            tri = c + rng.uniform(-extent, extent, (3, 3))
            tri[:, rng.integers(0, 3)] = np.float32(c[0])
        else:
            tri = c + rng.normal(size=(3, 3)) * extent * 0.15
        pos[3 * i:3 * i + 3] = tri
    return pos, np.arange(3 * n, dtype=np.uint32)


@pytest.mark.parametrize("n,thin", [(60, True), (2500, False), (2500, True)])
def test_edge_and_grazing_rays_soup(n, thin, oracle):
    """Triangle soups (small enough for LDS, and larger), with needles and flat sheets: edge rays around boxes of
    the tree, and rays grazing the triangles (aimed at points on their edges and just beside them from far away,
    and running inside the planes of the sheets)."""
    rng = np.random.default_rng(n + int(thin))
    s = pt.FlatScene()
    pos, idx = triangle_soup(rng, n, 10.0, thin)
    s.add_mesh(pos, idx, s.material_lambertian(s.texture_constant((0.5, 0.5, 0.5))))
    s.add_sphere((0.0, 0.0, 0.0), 1.5, s.material_light((1.0, 1.0, 1.0)), is_light=True)
    nodes = s.build_bvh()
    d = s.desc()
    with pt.DeviceScene(s) as dev:
        for k in (0, 1, len(nodes) // 2, len(nodes) - 1):
            o, dirs = edge_rays(nodes["lo"][k], nodes["hi"][k], rng, n_random=200)
            check(dev, C.byref(d), oracle, o, dirs, what="box %d" % k)
        # grazing: towards points on triangle edges, exactly and a hair to either side
        tri = pos.reshape(-1, 3, 3).astype(np.float64)
        m = min(len(tri), 600)
        pick = rng.choice(len(tri), m, replace=False)
        w = rng.random(m)[:, None]
        on_edge = tri[pick, 0] * w + tri[pick, 1] * (1 - w)
        normal = np.cross(tri[pick, 1] - tri[pick, 0], tri[pick, 2] - tri[pick, 0])
        normal /= np.maximum(np.linalg.norm(normal, axis=1), 1e-30)[:, None]
        O, D = [], []
        for eps in (0.0, 1e-6, -1e-6, 1e-4):
            target = on_edge + eps * (tri[pick, 2] - on_edge)
            # nearly in the plane of the triangle: incidence within a fraction of a degree
            inplane = tri[pick, 1] - tri[pick, 0]
            inplane /= np.maximum(np.linalg.norm(inplane, axis=1), 1e-30)[:, None]
            side = np.cross(normal, inplane)
            for tilt in (0.0, 1e-4, -1e-3):
                origin = target - 40.0 * (side + tilt * normal)
                dd = target - origin
                dd /= np.linalg.norm(dd, axis=1)[:, None]
                O.append(origin)
                D.append(dd)
        O = np.concatenate(O).astype(np.float32)
        D = np.concatenate(D).astype(np.float32)
        in_plane = rays_in_a_triangle_plane(pos, D)
        assert 0.2 < in_plane.mean() < 0.5  # the rays built with no tilt
        want = check(dev, C.byref(d), oracle, O, D, what="grazing", in_plane=in_plane)
        assert (want["surface"] != 0xFFFFFFFF).mean() > 0.2


def test_rays_with_other_bounds_and_times(oracle):
    """amin / amax other than the render loop's go through the one-lane-per-ray kernel; moving surfaces take the
    ray's time."""
    s, cam, p = scenes.build("animated")
    d = s.desc()
    rays = oracle.camera_rays(C.byref(d), cam, pt.make_params(48, 27, 2, seed=9, t0=p.t0, t1=p.t1), math_mode=1)
    o, dirs, t = rays[:, 0:3], rays[:, 3:6], rays[:, 6].copy()
    with pt.DeviceScene(s) as dev:
        check(dev, C.byref(d), oracle, o, dirs, t, what="animated")
        for (amin, amax) in ((0.0, 15.0), (5.0, 1e30), (1e-4, 12.5)):
            want = oracle.trace_rays(C.byref(d), o, dirs, t, amin=amin, amax=amax)
            assert dev.trace_rays(o, dirs, t, amin=amin, amax=amax).tobytes() == want.tobytes()
        assert len(dev.trace_rays(o[:0], dirs[:0])) == 0
        assert dev.trace_rays(o[:1], dirs[:1], t[:1]).tobytes() == oracle.trace_rays(C.byref(d), o[:1], dirs[:1], t[:1]).tobytes()


def test_two_streams_and_two_threads_on_one_scene(golden_dir, oracle):
    """Two renders of one scene enqueued from two threads on two streams: the library orders them, and both
    images are the oracle's bit for bit."""
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell_glossy.ptscene"))
    cam = pt.load_camera(os.path.join(golden_dir, "cornell_square.ptcam"))
    pa, pb = pt.make_params(96, 80, 3, seed=1), pt.make_params(64, 48, 2, seed=2)
    want_a = oracle.render(scene.ptr, cam, pa, math_mode=1)
    want_b = oracle.render(scene.ptr, cam, pb, math_mode=1)
    hip = Hip()
    with pt.DeviceScene(scene) as dev:
        sa, sb = hip.stream(), hip.stream()
        for rounds in range(3):
            out_a, out_b, out_c = hip.zeros(80 * 96 * 12), hip.zeros(48 * 64 * 12), hip.zeros(48 * 64 * 12)
            hip.sync()
            errors = []

            def run(p, out, stream):
                try:
                    dev.render_device(cam, p, out.value, stream.value)
                except Exception as e:  # noqa: BLE001
                    errors.append(e)

            ta = threading.Thread(target=run, args=(pa, out_a, sa))
            tb = threading.Thread(target=run, args=(pb, out_b, sb))
            ta.start()
            tb.start()
            ta.join()
            tb.join()
            assert not errors, errors
            # a third call, from this thread on the default stream, right behind them
            dev.render_device(cam, pb, out_c.value, None)
            hip.sync()
            for got, want in ((hip.to_host(out_a, (80, 96, 3)), want_a), (hip.to_host(out_b, (48, 64, 3)), want_b),
                              (hip.to_host(out_c, (48, 64, 3)), want_b)):
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            for p in (out_a, out_b, out_c):
                hip.free(p)
        hip.destroy_stream(sa)
        hip.destroy_stream(sb)


def test_scene_without_a_tree_refuses_to_render(oracle):
    s, cam, p = scenes.build("spheres")
    nodes = s.nodes
    s.nodes = np.zeros(0, pt.NODE_DTYPE)
    with pt.DeviceScene(s) as dev:
        with pytest.raises(pt.PtError):
            dev.render(cam, p)
        with pytest.raises(pt.PtError):
            dev.trace_rays(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32))
        with pytest.raises(pt.PtError):
            dev.primary_hits(cam, p)
        dev.update_bvh(nodes)
        s.nodes = nodes
        d = s.desc()
        got = dev.render_samples(cam, p)
        want = oracle.render_samples(C.byref(d), cam, p, math_mode=1)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        with pytest.raises(pt.PtError):
            dev.render(cam, pt.make_params(8, 8, 1, max_path_segments=65536))
