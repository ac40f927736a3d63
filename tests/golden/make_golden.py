#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the real reference.

Run in the build container (needs /root/reference and oracle/_ref/libpt_ref.so,
built by `make -C oracle ref`).  Every expected value written here is computed
by the reference's own code (importIntoScene, buildBVH, Camera, pathSample,
BVHTreeLinear::hit, Sampler, ...) through oracle/ref_harness.cpp; inputs are
either the reference's Cornell box asset (src/11/CornellBox-Original.obj,
public domain) run through the reference's importer, or seeded synthetic data.

    python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import ctypes as C
import math
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import abi  # noqa: E402
import refbind  # noqa: E402
import scenes  # noqa: E402

REF_SRC = "/root/reference/src"
CORNELL = os.path.join(REF_SRC, "11", "CornellBox-Original.obj")


def out(name):
    return os.path.join(HERE, name)


def cornell_camera(R, aspect):
    # the camera of the reference's Cornell main (src/11/pathtracer.cpp:135-136) with the
    # five-argument constructor of the final step (src/15/camera.hpp:19-32), aperture 0
    anim = pt.make_animation(abi.PT_ANIM_CONSTANT, R.lookat((0.0, 1.0, 3.2), (0.0, 1.0, -1.0)))
    return R.make_camera(math.radians(50.0), aspect, 1.0, 0.0, anim)


def glossy_obj(tmpdir):
    """Copy of the Cornell OBJ whose two boxes get a specular term, so that the
    reference's importer picks MaterialPhong for them (src/15/import.hpp:164-217)."""
    shutil.copy(CORNELL, os.path.join(tmpdir, "CornellBox-Original.obj"))
    lines = open(CORNELL[:-4] + ".mtl").read().split("\n")
    res, cur = [], None
    for ln in lines:
        s = ln.strip()
        if s.startswith("newmtl"):
            cur = s.split()[1]
        if cur == "shortBox" and s.startswith("Ks"):
            ln = "  Ks 0.45 0.45 0.45"
        if cur == "shortBox" and s.startswith("Ns"):
            ln = "  Ns 40.0"
        if cur == "shortBox" and s.startswith("Kd"):
            ln = "  Kd 0.35 0.34 0.33"
        if cur == "tallBox" and s.startswith("Ks"):
            ln = "  Ks 0.6 0.6 0.6"
        if cur == "tallBox" and s.startswith("Ns"):
            ln = "  Ns 300.0"
        if cur == "tallBox" and s.startswith("Kd"):
            ln = "  Kd 0.2 0.2 0.25"
        res.append(ln)
    open(os.path.join(tmpdir, "CornellBox-Original.mtl"), "w").write("\n".join(res))
    return os.path.join(tmpdir, "CornellBox-Original.obj")


def gen_cornell(R):
    d = R.import_obj(CORNELL)
    R.save_desc(d, out("cornell.ptscene"))
    pt.save_camera(cornell_camera(R, 1.0), out("cornell_square.ptcam"))
    pt.save_camera(cornell_camera(R, 16.0 / 9.0), out("cornell_wide.ptcam"))
    with tempfile.TemporaryDirectory() as tmp:
        g = R.import_obj(glossy_obj(tmp))
        R.save_desc(g, out("cornell_glossy.ptscene"))
        R.free_desc(g)
    return d


def gen_cornell_vectors(R):
    scene = pt.LoadedScene(out("cornell.ptscene"))
    cam = pt.load_camera(out("cornell_square.ptcam"))
    rs = R.scene(scene.ptr)
    assert rs.bvh_mismatch == 0
    # primary hits of a 64x64 image, 1 sample per pixel
    np.save(out("cornell_hits_64.npy"), rs.primary_hits(cam, pt.make_params(64, 64, 1)))
    # per-sample radiance, 32x32 pixels x 16 samples, counter-based generator
    np.save(out("cornell_samples_32x16.npy"), rs.render_samples(cam, pt.make_params(32, 32, 4)))
    # the same for the glossy variant (Phong boxes)
    g = pt.LoadedScene(out("cornell_glossy.ptscene"))
    gs = R.scene(g.ptr)
    np.save(out("cornell_glossy_samples_32x16.npy"), gs.render_samples(cam, pt.make_params(32, 32, 4)))
    # converged image with the reference's own generator: 96x96, 64^2 = 4096 spp
    img, secs = rs.render(cam, pt.make_params(96, 96, 64), rng_mode=1)
    np.save(out("cornell_converged_96_mt4096.npy"), img)
    print("converged reference image: %.1f s" % secs)


def gen_family_vectors(R):
    """Per-sample radiance of the synthetic scene families (tests/scenes.py): every
    material, texture, environment map, surface and animation type of the final step."""
    for name in scenes.FAMILIES:
        s, cam, params = scenes.build(name, R)
        d = s.desc()
        rs = R.scene(C.byref(d), params.t0, params.t1)
        assert rs.bvh_mismatch == 0, name
        np.save(out("family_%s_samples.npy" % name), rs.render_samples(cam, params))
        rs.close()


def gen_mesh_vectors(R):
    """Primary hits and per-sample radiance of the seeded displaced-grid mesh (tests/scenes.py
    big_mesh_scene(150): 45 002 triangles, tree too large for the on-chip copy of small scenes)."""
    s, cam = scenes.big_mesh_scene(150)
    d = s.desc()
    rs = R.scene(C.byref(d))
    assert rs.bvh_mismatch == 0
    np.save(out("mesh150_hits_96x54.npy"), rs.primary_hits(cam, pt.make_params(96, 54, 1)))
    np.save(out("mesh150_samples_32x18x4.npy"), rs.render_samples(cam, pt.make_params(32, 18, 2)))
    rs.close()


def gen_ingest(R):
    """The importer's image passes: TextureImage's constructor through stb_image (gamma 2.2 and 1), and
    bumpMapToNormalMap at two bump factors (reference src/15/texture_image.hpp:17-33, import.hpp:47-67)."""
    img = scenes.ingest_image()
    lin = R.texture_from_bytes(img, linearize=True)
    raw = R.texture_from_bytes(img, linearize=False)
    np.savez(out("ingest.npz"), linear=lin, raw=raw, normal_1=R.bump_to_normal_map(raw, 1.0),
             normal_8=R.bump_to_normal_map(raw, 8.0))


def gen_output_stage(R):
    img = scenes.output_stage_image()
    np.savez(out("output_stage.npz"), plain=R.image_to_8bit(img), urq=R.image_to_8bit(img, urq=(2.5, 20.0)),
             unflipped=R.image_to_8bit(img, flip_rows=False))


def gen_kat(R):
    rng = np.random.default_rng(7)
    n = 256
    u0 = rng.random(n, dtype=np.float32)
    u1 = rng.random(n, dtype=np.float32)
    shin = (rng.random(n, dtype=np.float32) * 200).astype(np.float32)
    cmax = rng.random(n, dtype=np.float32)
    data = {"u0": u0, "u1": u1, "shininess": shin, "cos_theta_max": cmax}
    for which, name, par in [(0, "uniformOnSphere", None), (1, "uniformOnHemisphere", None), (2, "uniformInDisk", None),
                             (3, "cosineWeightedOnHemisphere", None), (4, "phongWeightedOnHemisphere", shin),
                             (5, "uniformTowardsSphere", cmax), (6, "uniformInTriangle", None)]:
        data[name] = R.kat_warp(which, u0, u1, par)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    vec = rng.standard_normal((n, 3)).astype(np.float32)
    data["ts_normals"], data["ts_vecs"] = nrm, vec
    data["tangentspace"] = R.kat_tangentspace(nrm, vec)
    cosI = rng.random(n, dtype=np.float32)
    cosT = rng.random(n, dtype=np.float32)
    n1 = np.where(rng.random(n) < 0.5, 1.0, 1.5).astype(np.float32)
    n2 = np.where(n1 == 1.0, 1.5, 1.0).astype(np.float32)
    data["fr_cosI"], data["fr_cosT"], data["fr_n1"], data["fr_n2"] = cosI, cosT, n1, n2
    data["fresnel"] = R.kat_fresnel(cosI, cosT, n1, n2)
    inc = rng.standard_normal((n, 3)).astype(np.float32)
    inc /= np.linalg.norm(inc, axis=1, keepdims=True).astype(np.float32)
    eta = np.where(rng.random(n) < 0.5, 1.0 / 1.5, 1.5).astype(np.float32)
    data["rf_i"], data["rf_n"], data["rf_eta"] = inc, nrm, eta
    data["refract"] = R.kat_refract(inc, nrm, eta)
    np.savez(out("kat_functions.npz"), **data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    R = refbind.Reference()
    steps = {"ingest": gen_ingest, "cornell": gen_cornell, "cornell_vectors": gen_cornell_vectors, "families": gen_family_vectors,
             "mesh": gen_mesh_vectors, "output": gen_output_stage, "kat": gen_kat}
    for name, fn in steps.items():
        if args.only and args.only != name:
            continue
        print("==", name)
        fn(R)


if __name__ == "__main__":
    main()
