#!/usr/bin/env python3
"""Time the importer's re-indexing (reference src/15/import.hpp:233-289) for one large shape: the library's GPU
version (pt_mesh_reindex, host arrays in and out, so transfers are inside the time) against the CPU restatement
with the importer's std::map (oracle_mesh_reindex, one host thread as in the reference)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracing_amd as pt  # noqa: E402
from path_tracing_amd import workloads  # noqa: E402
import refbind  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pos, idx = workloads.displaced_grid(n)
vi = idx.astype(np.int32)
ni = np.full_like(vi, -1)
ti = np.full_like(vi, -1)
mat = np.zeros(len(vi) // 3, np.int32)
empty3, empty2 = np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32)
O = refbind.Oracle()
pt.mesh_reindex(vi[:300], ni[:300], ti[:300], mat[:100], 0, pos, empty3, empty2)  # first use
t = time.perf_counter()
got = pt.mesh_reindex(vi, ni, ti, mat, 0, pos, empty3, empty2)
t_gpu = time.perf_counter() - t
t = time.perf_counter()
want = O.mesh_reindex(vi, ni, ti, mat, 0, pos, empty3, empty2)
t_cpu = time.perf_counter() - t
assert np.array_equal(got["indices"], want["indices"]) and got["positions"].tobytes() == want["positions"].tobytes()
print("%d triangles, %d distinct vertices: GPU %.1f ms (with transfers), std::map on one host thread %.1f ms, x%.1f"
      % (len(mat), len(got["positions"]), t_gpu * 1e3, t_cpu * 1e3, t_cpu / t_gpu))
