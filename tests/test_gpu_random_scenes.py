"""Seeded random scenes: a few dozen surfaces of both kinds with random materials, textures, poses, lights and
environment, rendered by the device (scene in LDS, scene in memory, and with the device-built tree) and by the
oracle on the same flat description: per-sample radiance must agree bit for bit."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import path_tracing_amd as pt
from path_tracing_amd import abi
import scenes

pytestmark = pytest.mark.gpu


def random_scene(seed):
    rng = np.random.default_rng(seed)
    s = pt.FlatScene()

    def tex():
        k = rng.integers(0, 4)
        if k == 0:
            return s.texture_constant(rng.random(3) * 0.9)
        if k == 1:
            return s.texture_checker(s.texture_constant(rng.random(3)), s.texture_constant(rng.random(3) * 0.3),
                                     int(rng.integers(2, 9)), int(rng.integers(2, 9)))
        if k == 2:
            return s.texture_image(scenes.image_texture(rng, int(rng.integers(2, 7)), int(rng.integers(2, 7))))
        return s.texture_transformer(s.texture_constant(rng.random(3)), (2.0, 3.0), (0.25, 0.5))

    def material():
        k = rng.integers(0, 6)
        if k == 0:
            return s.material_lambertian(tex())
        if k == 1:
            return s.material_phong(tex(), s.texture_constant(rng.random(3) * 0.5), s.texture_constant(float(1 + 200 * rng.random())))
        if k == 2:
            return s.material_mirror(s.texture_constant(0.5 + 0.5 * rng.random(3)))
        if k == 3:
            return s.material_glass(tuple(rng.random(3) * 0.2), 1.2 + 0.6 * rng.random())
        if k == 4:
            return s.material_twosided(s.material_lambertian(tex()), s.material_mirror(s.texture_constant(0.8)))
        return s.material_light(rng.random(3) * 4.0)

    def animation():
        k = rng.integers(0, 4)
        if k == 0:
            return -1
        if k == 1:
            T = pt.make_transform(tuple(rng.normal(0, 0.5, 3)), scenes.quat(float(rng.random() * 90), (0.3, 1, 0.2)),
                                  (1.0 + rng.random(),) * 3)
            return s.add_animation(pt.make_animation(abi.PT_ANIM_CONSTANT, T))
        T0 = pt.make_transform(tuple(rng.normal(0, 0.5, 3)), scenes.quat(float(rng.random() * 60), (0, 1, 0)), (1.0, 1.0, 1.0))
        T1 = pt.make_transform(tuple(rng.normal(0, 0.5, 3)), scenes.quat(float(rng.random() * 120), (1, 0.2, 0)),
                               (1.0 + rng.random(),) * 3)
        return s.add_animation(pt.make_animation(abi.PT_ANIM_TWOKEY, T0, T1, 10.0))

    # a room to keep paths alive, a guaranteed light
    scenes.room(s, s.material_lambertian(s.texture_constant((0.7, 0.7, 0.7))), s.material_light((8.0, 8.0, 8.0)), size=6.0,
                with_attrs=False)
    for _ in range(int(rng.integers(3, 12))):
        m = material()
        s.add_sphere(tuple(rng.uniform(-4, 4, 3)), float(0.2 + rng.random()), m, animation=animation(),
                     is_light=bool(rng.random() < 0.2))
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, 7))
        pos = (rng.uniform(-4, 4, (1, 3)) + rng.normal(0, 0.8, (3 * n, 3))).astype(np.float32)
        idx = np.arange(3 * n, dtype=np.uint32)
        s.add_mesh(pos, idx, material(), animation=animation(), is_light=bool(rng.random() < 0.2))
    if rng.random() < 0.5:
        s.set_envmap_equirect(s.texture_constant(rng.random(3) * 0.5))
    t0, t1 = float(rng.random() * 5), float(5 + rng.random() * 4)
    s.build_bvh(t0, t1)
    cam = pt.make_camera(math.tan(math.radians(30.0)), math.tan(math.radians(30.0)) * 1.5, 8.0, 0.05 * float(rng.random()),
                         scenes.lookat_animation((0.0, 1.0, 11.0), (0.0, 0.0, 0.0)))
    return s, cam, pt.make_params(24, 16, 2, seed=int(seed) + 100, t0=t0, t1=t1)


@pytest.mark.parametrize("seed", range(32))
def test_random_scene(seed, oracle):
    s, cam, p = random_scene(seed)
    d = s.desc()
    want = oracle.render_samples(C.byref(d), cam, p, math_mode=1).view(np.uint32)
    with pt.DeviceScene(s) as dev:
        assert np.array_equal(dev.render_samples(cam, p).view(np.uint32), want), "scene in LDS"
        os.environ["PT_NO_LDS_SCENE"] = "1"
        try:
            assert np.array_equal(dev.render_samples(cam, p).view(np.uint32), want), "scene in memory"
            dev.build_bvh_device(p.t0, p.t1)
            s.nodes = dev.get_bvh()
            d2 = s.desc()
            want2 = oracle.render_samples(C.byref(d2), cam, p, math_mode=1).view(np.uint32)
            assert np.array_equal(dev.render_samples(cam, p).view(np.uint32), want2), "device-built tree, in memory"
        finally:
            del os.environ["PT_NO_LDS_SCENE"]
        assert np.array_equal(dev.render_samples(cam, p).view(np.uint32), want2), "device-built tree, in LDS"


def soup_scene(seed, n_tri, n_sph):
    """A soup of overlapping random triangles and spheres: deep trees with heavily overlapping boxes, so the
    nearer-child-first walk meets many near-ties and stack overflows and leans on its second pass."""
    rng = np.random.default_rng(seed)
    s = pt.FlatScene()
    grey = s.material_lambertian(s.texture_constant((0.6, 0.6, 0.6)))
    mirror = s.material_mirror(s.texture_constant((0.9, 0.9, 0.9)))
    centres = rng.uniform(-5, 5, (n_tri, 1, 3))
    pos = (centres + rng.normal(0, 0.6, (n_tri, 3, 3))).reshape(-1, 3).astype(np.float32)
    # some exactly coplanar, edge-sharing pairs: ties in distance
    pos[3:6] = pos[0:3][[0, 2, 1]] + np.float32(0.0)
    s.add_mesh(pos, np.arange(3 * n_tri, dtype=np.uint32), grey)
    for i in range(n_sph):
        s.add_sphere(tuple(rng.uniform(-5, 5, 3)), float(0.1 + 0.5 * rng.random()), mirror if i % 2 else grey)
    s.add_sphere((0.0, 9.0, 0.0), 2.0, s.material_light((10.0, 10.0, 10.0)), is_light=True)
    s.set_envmap_equirect(s.texture_constant((0.3, 0.3, 0.4)))
    s.build_bvh()
    cam = pt.make_camera(math.tan(math.radians(30.0)), math.tan(math.radians(30.0)) * 1.5, 8.0, 0.0,
                         scenes.lookat_animation((0.0, 2.0, 14.0), (0.0, 0.0, 0.0)))
    return s, cam


@pytest.mark.parametrize("seed,n_tri,n_sph", [(1, 600, 40), (2, 4000, 300), (3, 20000, 0)])
def test_soup(seed, n_tri, n_sph, oracle):
    s, cam = soup_scene(seed, n_tri, n_sph)
    d = s.desc()
    p1 = pt.make_params(160, 100, 1, seed=seed)
    p = pt.make_params(64, 40, 2, seed=seed)
    want_hits = oracle.primary_hits(C.byref(d), cam, p1, math_mode=1).tobytes()
    want = oracle.render_samples(C.byref(d), cam, p, math_mode=1).view(np.uint32)
    with pt.DeviceScene(s) as dev:
        for env in (None, "1"):
            if env:
                os.environ["PT_NO_LDS_SCENE"] = env
            try:
                assert dev.primary_hits(cam, p1).tobytes() == want_hits, env
                assert np.array_equal(dev.render_samples(cam, p).view(np.uint32), want), env
            finally:
                if env:
                    del os.environ["PT_NO_LDS_SCENE"]
        os.environ["PT_NO_LDS_SCENE"] = "1"
        try:
            dev.build_bvh_device()
            s.nodes = dev.get_bvh()
            d2 = s.desc()
            assert dev.primary_hits(cam, p1).tobytes() == oracle.primary_hits(C.byref(d2), cam, p1, math_mode=1).tobytes()
            assert np.array_equal(dev.render_samples(cam, p).view(np.uint32),
                                  oracle.render_samples(C.byref(d2), cam, p, math_mode=1).view(np.uint32))
        finally:
            del os.environ["PT_NO_LDS_SCENE"]
