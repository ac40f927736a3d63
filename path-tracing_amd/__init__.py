"""MI355X-native render loop for the marlam/path-tracing course tracer (final step, src/15).

The product is lib/libptamd.so (HIP kernels for gfx950 behind the C ABI of
include/pt_amd.h); this package is its Python face for tests and benchmarks.
"""
from . import abi  # noqa: F401
from .api import (  # noqa: F401
    DeviceScene, FlatScene, LoadedScene, PtError, device_count, lib, load_camera, make_animation, make_camera,
    make_params, make_transform, identity_transform, n_samples, save_camera, image_to_8bit, image_to_8bit_device,
    texture_from_bytes, bump_to_normal_map, multi_plan, render_multi, render_multi_device, multi_last_combine, mesh_reindex, set_host_tree_builder, host_quad_rows,
    HIT_DTYPE, NODE_DTYPE, LIB_PATH,
)
