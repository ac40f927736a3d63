#!/usr/bin/env python3
"""From an OBJ file on disk to a scene that can be rendered, end to end, on the mesh workload's geometry:

  reference   importIntoScene + Scene::buildBVH (reference src/15/import.hpp:118-293, scene.hpp:69-72, one host thread):
              tinyobj parse, per-(material, shape) re-indexing through a std::map, one SurfaceTriangle per face, the
              sorted-sweep tree -- the compiled reference (oracle/_ref), timed as a whole and flattened
  this path   the same tinyobj parse (text parsing stays with tinyobj), then pt_mesh_reindex on the GPU, the flat
              scene tables, pt_scene_create (upload) and pt_scene_build_bvh_device

usage: python tests/bench_ingest_end_to_end.py [N=1000]   (N x N cells, 2 N^2 triangles; writes the OBJ to $TMPDIR)"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import workloads  # noqa: E402
import refbind  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pos, idx = workloads.displaced_grid(n, amplitude=0.01)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "mesh%d.obj" % n)
t = time.perf_counter()
with open(path, "w") as f:
    np.savetxt(f, pos, fmt="v %.9g %.9g %.9g")
    np.savetxt(f, idx.reshape(-1, 3).astype(np.int64) + 1, fmt="f %d %d %d")
print("wrote %s: %d vertices, %d triangles, %.0f MB (%.1f s)" % (path, len(pos), len(idx) // 3, os.path.getsize(path) / 1e6,
                                                              time.perf_counter() - t), flush=True)

R = refbind.Reference()
devnull = os.open(os.devnull, os.O_WRONLY)
saved = os.dup(2)
os.dup2(devnull, 2)  # the reference prints progress on stderr
try:
    t = time.perf_counter()
    desc = R.import_obj(path)
    t_ref = time.perf_counter() - t
finally:
    os.dup2(saved, 2)
    os.close(devnull)
    os.close(saved)
n_ref_surfaces = int(desc.contents.n_surfaces)
R.free_desc(desc)
print("reference: importIntoScene + buildBVH + flatten: %.2f s (%d surfaces)" % (t_ref, n_ref_surfaces), flush=True)

# warm the GPU paths (first use loads code objects and allocates work areas)
small_pos, small_idx = workloads.displaced_grid(8)
pt.mesh_reindex(small_idx.astype(np.int32), np.full(len(small_idx), -1, np.int32), np.full(len(small_idx), -1, np.int32),
                np.zeros(len(small_idx) // 3, np.int32), 0, small_pos, np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32))

t0 = time.perf_counter()
o = R.parse_obj(path)
t_parse = time.perf_counter() - t0
t = time.perf_counter()
mesh = pt.mesh_reindex(o["vi"], o["ni"], o["ti"], o["tri_material"], -1 if o["n_materials"] == 0 else 0, o["vertices"],
                       o["normals"], o["texcoords"])
t_reindex = time.perf_counter() - t
t = time.perf_counter()
s = pt.FlatScene()
s.add_mesh(mesh["positions"], mesh["indices"], s.material_lambertian(s.texture_constant((0.5, 0.5, 0.5))))
d = s.desc()
t_tables = time.perf_counter() - t
t = time.perf_counter()
dev = pt.DeviceScene(s)
t_upload = time.perf_counter() - t
t = time.perf_counter()
dev.build_bvh_device()
t_tree = time.perf_counter() - t
total = time.perf_counter() - t0
assert len(s.surfaces) == n_ref_surfaces
print("this path: tinyobj parse %.2f s + pt_mesh_reindex %.3f s + flat tables %.3f s + pt_scene_create %.3f s + "
      "pt_scene_build_bvh_device %.3f s = %.2f s; after the parse %.3f s" % (t_parse, t_reindex, t_tables, t_upload, t_tree, total,
                                                                              total - t_parse))
print("end to end %.1fx; everything after the text parse %.0fx (reference without its parse: %.2f s)"
      % (t_ref / total, (t_ref - t_parse) / max(total - t_parse, 1e-9), t_ref - t_parse))
dev.close()
os.remove(path)
os.rmdir(tmp)
