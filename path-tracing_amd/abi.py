"""ctypes mirror of include/pt_amd.h (struct layouts and enums only)."""
import ctypes as C

PT_ABI_VERSION = 1

PT_OK = 0
PT_NO_HIT = 0xFFFFFFFF
PT_RENDER_SUM_ONLY = 1

PT_ANIM_IDENTITY, PT_ANIM_CONSTANT, PT_ANIM_TWOKEY = 0, 1, 2
(PT_TEX_CONSTANT, PT_TEX_IMAGE, PT_TEX_CHECKER, PT_TEX_TRANSFORMER, PT_TEX_VALUE_NOISE,
 PT_TEX_GRADIENT_NOISE, PT_TEX_WORLEY_NOISE) = range(7)
(PT_MAT_BASE, PT_MAT_LIGHT, PT_MAT_LAMBERTIAN, PT_MAT_PHONG, PT_MAT_MIRROR, PT_MAT_GLASS,
 PT_MAT_TWOSIDED) = range(7)
PT_SURF_TRIANGLE, PT_SURF_SPHERE = 0, 1
PT_ENV_NONE, PT_ENV_BASE, PT_ENV_EQUIRECT, PT_ENV_CUBE = range(4)


class pt_transform(C.Structure):
    _fields_ = [("translation", C.c_float * 3), ("rotation", C.c_float * 4), ("scaling", C.c_float * 3)]


class pt_animation(C.Structure):
    _fields_ = [("type", C.c_int32), ("duration", C.c_float), ("T0", pt_transform), ("T1", pt_transform)]


class pt_texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("tex0", C.c_int32),
                ("tex1", C.c_int32), ("reserved", C.c_int32), ("data_offset", C.c_uint64), ("v", C.c_float * 4)]


class pt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("tex", C.c_int32 * 5), ("front", C.c_int32), ("back", C.c_int32),
                ("v", C.c_float * 4)]


class pt_mesh(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32), ("first_index", C.c_uint32),
                ("n_triangles", C.c_uint32), ("has_normals", C.c_int32), ("has_texcoords", C.c_int32),
                ("has_tangents", C.c_int32), ("material", C.c_int32), ("animation", C.c_int32),
                ("reserved", C.c_int32)]


class pt_sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_int32),
                ("animation", C.c_int32)]


class pt_surface(C.Structure):
    _fields_ = [("type", C.c_uint32), ("owner", C.c_uint32), ("index", C.c_uint32), ("reserved", C.c_uint32)]


class pt_bvh_node(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("link", C.c_uint64)]


class pt_envmap(C.Structure):
    _fields_ = [("type", C.c_int32), ("tex", C.c_int32 * 6), ("reserved", C.c_int32)]


class pt_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("reserved", C.c_uint32),
        ("n_vertices", C.c_uint64),
        ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
        ("texcoords", C.POINTER(C.c_float)), ("tangents", C.POINTER(C.c_float)),
        ("n_indices", C.c_uint64), ("indices", C.POINTER(C.c_uint32)),
        ("n_meshes", C.c_uint64), ("meshes", C.POINTER(pt_mesh)),
        ("n_spheres", C.c_uint64), ("spheres", C.POINTER(pt_sphere)),
        ("n_surfaces", C.c_uint64), ("surfaces", C.POINTER(pt_surface)),
        ("n_lights", C.c_uint64), ("lights", C.POINTER(C.c_uint32)),
        ("n_materials", C.c_uint64), ("materials", C.POINTER(pt_material)),
        ("n_textures", C.c_uint64), ("textures", C.POINTER(pt_texture)),
        ("n_texels", C.c_uint64), ("texels", C.POINTER(C.c_float)),
        ("n_animations", C.c_uint64), ("animations", C.POINTER(pt_animation)),
        ("n_bvh_nodes", C.c_uint64), ("bvh_nodes", C.POINTER(pt_bvh_node)),
        ("envmap", pt_envmap),
    ]


class pt_camera(C.Structure):
    _fields_ = [("t", C.c_float), ("b", C.c_float), ("r", C.c_float), ("l", C.c_float),
                ("focus_distance", C.c_float), ("aperture_radius", C.c_float), ("has_animation", C.c_int32),
                ("animation", pt_animation)]


class pt_render_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sqrt_spp", C.c_int32), ("t0", C.c_float),
                ("t1", C.c_float), ("flags", C.c_uint32), ("seed", C.c_uint64),
                ("tile_x0", C.c_int32), ("tile_y0", C.c_int32), ("tile_x1", C.c_int32), ("tile_y1", C.c_int32),
                ("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("max_path_segments", C.c_int32),
                ("rr_start_segment", C.c_int32)]


class pt_hit(C.Structure):
    _fields_ = [("surface", C.c_uint32), ("material", C.c_uint32), ("a", C.c_float), ("backside", C.c_uint32)]


class pt_counters(C.Structure):
    _fields_ = [("path_samples", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("leaf_tests", C.c_uint64), ("light_tests", C.c_uint64), ("vertices", C.c_uint64),
                ("rng_draws", C.c_uint64)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


# struct sizes as the C compiler lays them out (checked by tests against the library)
class pt_kernel_time(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("launches", C.c_uint64), ("total_ms", C.c_double)]


PT_SHARD_TILES, PT_SHARD_SAMPLES = 0, 1


class pt_walk_counters(C.Structure):
    _fields_ = [("path_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("box_rows", C.c_uint64),
                ("leaf_rows", C.c_uint64), ("surface_tests", C.c_uint64), ("retry_rays", C.c_uint64),
                ("retry_node_visits", C.c_uint64), ("retry_surface_tests", C.c_uint64),
                ("null_shadow_rays", C.c_uint64)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


EXPECTED_SIZES = {
    "pt_transform": 40, "pt_animation": 88, "pt_texture": 48, "pt_material": 48, "pt_mesh": 40,
    "pt_sphere": 24, "pt_surface": 16, "pt_bvh_node": 32, "pt_envmap": 32, "pt_scene_desc": 240,
    "pt_camera": 116, "pt_render_params": 64, "pt_hit": 16, "pt_counters": 56, "pt_kernel_time": 40,
    "pt_walk_counters": 72,
}
