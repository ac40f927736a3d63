"""Python face of the C ABI in include/pt_amd.h (ctypes over lib/libptamd.so).

Everything here is a thin call into the library: scene tables are assembled as
numpy arrays, tangents / acceleration structure / rendering are done by the
native code.  The module fails loudly if the library is missing: there is no
Python or CPU fallback for any render entry point.
"""
import ctypes as C
import math
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB selects another build of the library (used to compare kernel variants)
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "lib", "libptamd.so")

_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("pt_amd error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None


def lib():
    """The loaded libptamd.so (loaded on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libptamd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C path-tracing_amd/csrc`)")
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_version.restype = C.c_char_p
        L.pt_last_kernel_ms.restype = C.c_float
        L.pt_last_kernel_ms.argtypes = [C.c_void_p]
        L.pt_scene_profile.argtypes = [C.c_void_p, C.c_int]
        L.pt_scene_kernel_times.argtypes = [C.c_void_p, C.POINTER(abi.pt_kernel_time), C.c_int, C.POINTER(C.c_int)]
        L.pt_scene_count_walk.argtypes = [C.c_void_p, C.c_int]
        L.pt_scene_walk_counters.argtypes = [C.c_void_p, C.POINTER(abi.pt_walk_counters)]
        L.pt_image_to_8bit.argtypes = [C.c_int, _fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.pt_image_to_8bit_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                              C.c_void_p, C.c_void_p]
        L.pt_scene_destroy.argtypes = [C.c_void_p]
        L.pt_scene_destroy.restype = None
        L.pt_scene_desc_free.argtypes = [C.c_void_p]
        L.pt_scene_desc_free.restype = None
        L.pt_scene_create.argtypes = [C.POINTER(abi.pt_scene_desc), C.c_int, C.POINTER(C.c_void_p)]
        L.pt_scene_update_bvh.argtypes = [C.c_void_p, C.POINTER(abi.pt_bvh_node), C.c_uint64]
        L.pt_texture_from_bytes.argtypes = [C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_float)]
        L.pt_texture_from_bytes_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_void_p]
        L.pt_bump_to_normal_map.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
        L.pt_bump_to_normal_map_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.pt_scene_build_bvh_device.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.pt_scene_get_bvh.argtypes = [C.c_void_p, C.POINTER(abi.pt_bvh_node), C.c_uint64, C.POINTER(C.c_uint64)]
        L.pt_render.argtypes = [C.c_void_p, C.POINTER(abi.pt_camera), C.POINTER(abi.pt_render_params), _fp]
        L.pt_render_device.argtypes = [C.c_void_p, C.POINTER(abi.pt_camera), C.POINTER(abi.pt_render_params),
                                       C.c_void_p, C.c_void_p]
        L.pt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(abi.pt_camera), C.POINTER(abi.pt_render_params),
                                               C.c_void_p, C.c_void_p]
        L.pt_multi_plan.argtypes = [C.POINTER(abi.pt_render_params), C.c_int, C.c_int, C.c_int,
                                    C.POINTER(abi.pt_render_params), C.c_int, C.POINTER(C.c_int)]
        L.pt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(abi.pt_camera),
                                      C.POINTER(abi.pt_render_params), C.c_int, _fp]
        L.pt_render_multi_device.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(abi.pt_camera),
                                             C.POINTER(abi.pt_render_params), C.c_int, C.c_void_p, C.c_void_p]
        L.pt_multi_last_combine.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int)]
        L.pt_render_counted.argtypes = [C.c_void_p, C.POINTER(abi.pt_camera), C.POINTER(abi.pt_render_params), _fp,
                                        C.POINTER(abi.pt_counters)]
        L.pt_render_samples.argtypes = [C.c_void_p, C.POINTER(abi.pt_camera), C.POINTER(abi.pt_render_params), _fp]
        L.pt_primary_hits.argtypes = [C.c_void_p, C.POINTER(abi.pt_camera), C.POINTER(abi.pt_render_params),
                                      C.POINTER(abi.pt_hit)]
        L.pt_trace_rays.argtypes = [C.c_void_p, C.c_uint64, _fp, _fp, _fp, C.c_float, C.c_float,
                                    C.POINTER(abi.pt_hit)]
        L.pt_mesh_compute_tangents.argtypes = [_fp, _fp, _fp, C.c_uint32, _u32p, C.c_uint32, _fp]
        L.pt_scene_desc_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(abi.pt_scene_desc))]
        L.pt_scene_desc_save.argtypes = [C.POINTER(abi.pt_scene_desc), C.c_char_p]
        L.pt_scene_desc_validate.argtypes = [C.POINTER(abi.pt_scene_desc)]
        _lib = L
    return _lib


def _check(rc):
    if rc != abi.PT_OK:
        raise PtError(rc, lib().pt_last_error().decode())


def _f(a):
    return a.ctypes.data_as(_fp)


def device_count():
    return int(lib().pt_device_count())


HIT_DTYPE = np.dtype([("surface", np.uint32), ("material", np.uint32), ("a", np.float32), ("backside", np.uint32)])
NODE_DTYPE = np.dtype([("lo", np.float32, 3), ("hi", np.float32, 3), ("link", np.uint64)])


# ------------------------------------------------------------------ poses --

def identity_transform():
    T = abi.pt_transform()
    T.translation[:] = (0.0, 0.0, 0.0)
    T.rotation[:] = (0.0, 0.0, 0.0, 1.0)
    T.scaling[:] = (1.0, 1.0, 1.0)
    return T


def make_transform(translation=(0, 0, 0), rotation=(0, 0, 0, 1), scaling=(1, 1, 1)):
    T = abi.pt_transform()
    T.translation[:] = [float(v) for v in translation]
    T.rotation[:] = [float(v) for v in rotation]
    T.scaling[:] = [float(v) for v in scaling]
    return T


def make_animation(kind=abi.PT_ANIM_CONSTANT, T0=None, T1=None, duration=1.0):
    a = abi.pt_animation()
    a.type = kind
    a.duration = duration
    a.T0 = T0 if T0 is not None else identity_transform()
    a.T1 = T1 if T1 is not None else a.T0
    return a


def make_camera(t, r, focus_distance=1.0, aperture_radius=0.0, animation=None):
    """Camera POD from the frustum half-extents (t = tan(vfov / 2), r = t * aspect)."""
    cam = abi.pt_camera()
    cam.t, cam.b, cam.r, cam.l = t, -t, r, -r
    cam.focus_distance = focus_distance
    cam.aperture_radius = aperture_radius
    cam.has_animation = 1 if animation is not None else 0
    cam.animation = animation if animation is not None else make_animation(abi.PT_ANIM_IDENTITY)
    return cam


def make_params(width, height, sqrt_spp, seed=42, t0=0.0, t1=0.0, tile=None, samples=None, flags=0,
                max_path_segments=0, rr_start_segment=0):
    p = abi.pt_render_params()
    p.width, p.height, p.sqrt_spp = width, height, sqrt_spp
    p.t0, p.t1 = t0, t1
    p.seed = seed
    p.flags = flags
    if tile is not None:
        p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1 = tile
    if samples is not None:
        p.sample_begin, p.sample_end = samples
    p.max_path_segments = max_path_segments
    p.rr_start_segment = rr_start_segment
    return p


def image_to_8bit(img, device=0, urq=None, flip_rows=True):
    """The reference's output stage on the GPU: (H, W, 3) float32 with row 0 = bottom -> (H, W, 3) uint8, rows in
    file order if flip_rows.  urq = (max_val, brightness) applies uniformRationalQuantization first."""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[0], img.shape[1]
    out = np.empty((h, w, 3), np.uint8)
    mx, br = urq if urq else (0.0, 1.0)
    _check(lib().pt_image_to_8bit(device, _f(img), w, h, mx, br, 1 if flip_rows else 0, out.ctypes.data_as(C.c_void_p)))
    return out


def image_to_8bit_device(d_rgb_ptr, width, height, d_out_ptr, device=0, urq=None, flip_rows=True, stream_ptr=None):
    """Same, on device memory (raw pointers) and asynchronous on the given stream."""
    mx, br = urq if urq else (0.0, 1.0)
    _check(lib().pt_image_to_8bit_device(device, C.c_void_p(d_rgb_ptr), width, height, mx, br, 1 if flip_rows else 0,
                                         C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr) if stream_ptr else None))


def texture_from_bytes(rgb8, device=0, linearize=True, flip_rows=True):
    """What TextureImage's constructor makes of an 8-bit RGB file (reference src/15/texture_image.hpp:17-33), on the
    GPU: (H, W, 3) uint8 in file order -> (H, W, 3) float32 texels, bottom row first if flip_rows."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w = rgb8.shape[0], rgb8.shape[1]
    out = np.empty((h, w, 3), np.float32)
    _check(lib().pt_texture_from_bytes(device, rgb8.ctypes.data_as(C.POINTER(C.c_ubyte)), w, h, int(linearize),
                                       int(flip_rows), _f(out)))
    return out


def bump_to_normal_map(texels, bump_factor, device=0):
    """bumpMapToNormalMap (reference src/15/import.hpp:47-67) on the GPU: (H, W, 3) float32 -> (H, W, 3) float32."""
    texels = np.ascontiguousarray(texels, np.float32)
    h, w = texels.shape[0], texels.shape[1]
    out = np.empty((h, w, 3), np.float32)
    _check(lib().pt_bump_to_normal_map(device, _f(texels), w, h, float(bump_factor), _f(out)))
    return out


def n_samples(p):
    x0, y0, x1, y1 = p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1
    if (x0, y0, x1, y1) == (0, 0, 0, 0):
        x1, y1 = p.width, p.height
    s0, s1 = p.sample_begin, p.sample_end
    if (s0, s1) == (0, 0):
        s1 = p.sqrt_spp * p.sqrt_spp
    return (x1 - x0) * (y1 - y0) * (s1 - s0)


# ------------------------------------------------------------ flat scenes --

_host_tree_builder = None


def set_host_tree_builder(fn):
    """fn(desc, t0, t1, nodes) fills the NODE_DTYPE array `nodes` (2 * surfaces - 1 entries) and returns the depth."""
    global _host_tree_builder
    _host_tree_builder = fn


class FlatScene:
    """A flat scene in host memory: numpy arrays behind one pt_scene_desc.

    Mirrors the way the reference's Scene is filled (reference src/15/scene.hpp:29-72):
    textures, materials, animations are registered first and referred to by index;
    add_mesh / add_sphere append surfaces in call order (optionally as lights);
    build_bvh(t0, t1) must be called last.
    """

    def __init__(self):
        self.positions = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.texcoords = np.zeros((0, 2), np.float32)
        self.tangents = np.zeros((0, 3), np.float32)
        self.indices = np.zeros(0, np.uint32)
        self.meshes = []
        self.spheres = []
        self.surfaces = np.zeros((0, 4), np.uint32)
        self.lights = []
        self.materials = []
        self.textures = []
        self.texels = np.zeros(0, np.float32)
        self.animations = []
        self.envmap = abi.pt_envmap()
        self.envmap.type = abi.PT_ENV_NONE
        self.envmap.tex[:] = [-1] * 6
        self.nodes = np.zeros(0, NODE_DTYPE)
        self.depth = 0
        self._keep = []

    @classmethod
    def from_desc(cls, d):
        """A copy of the arrays behind a pt_scene_desc (e.g. a scene file loaded by the library), open to changes."""
        def arr(ptr, n, dtype, width=None):
            if n == 0:
                return np.zeros((0, width) if width else 0, dtype)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize * (width or 1),))
            a = a.view(dtype).copy()
            return a.reshape(-1, width) if width else a

        s = cls()
        nv = int(d.n_vertices)
        s.positions, s.normals = arr(d.positions, nv, np.float32, 3), arr(d.normals, nv, np.float32, 3)
        s.texcoords, s.tangents = arr(d.texcoords, nv, np.float32, 2), arr(d.tangents, nv, np.float32, 3)
        s.indices = arr(d.indices, int(d.n_indices), np.uint32)
        s.surfaces = arr(d.surfaces, int(d.n_surfaces), np.uint32, 4)
        s.lights = [int(v) for v in arr(d.lights, int(d.n_lights), np.uint32)]
        s.texels = arr(d.texels, int(d.n_texels), np.float32)
        s.nodes = arr(d.bvh_nodes, int(d.n_bvh_nodes), NODE_DTYPE)
        for name, ctype, n in (("meshes", abi.pt_mesh, d.n_meshes), ("spheres", abi.pt_sphere, d.n_spheres),
                               ("materials", abi.pt_material, d.n_materials), ("textures", abi.pt_texture, d.n_textures),
                               ("animations", abi.pt_animation, d.n_animations)):
            src = getattr(d, name)
            setattr(s, name, [ctype.from_buffer_copy(src[i]) for i in range(int(n))])
        s.envmap = abi.pt_envmap.from_buffer_copy(d.envmap)
        return s

    # -- animations / textures / materials
    def add_animation(self, anim):
        self.animations.append(anim)
        return len(self.animations) - 1

    def _add_texture(self, t):
        self.textures.append(t)
        return len(self.textures) - 1

    def _new_texture(self, kind):
        t = abi.pt_texture()
        t.type = kind
        t.tex0 = t.tex1 = -1
        return t

    def texture_constant(self, rgb):
        t = self._new_texture(abi.PT_TEX_CONSTANT)
        rgb = (rgb, rgb, rgb) if np.isscalar(rgb) else rgb
        t.v[0], t.v[1], t.v[2] = [float(np.float32(v)) for v in rgb]
        return self._add_texture(t)

    def _push_texels(self, data):
        off = self.texels.size
        self.texels = np.concatenate([self.texels, np.ascontiguousarray(data, np.float32).ravel()])
        return off

    def texture_image(self, img):
        """img: (height, width, 3) float32, row 0 = bottom (the reference flips images on load)."""
        img = np.ascontiguousarray(img, np.float32)
        t = self._new_texture(abi.PT_TEX_IMAGE)
        t.height, t.width = img.shape[0], img.shape[1]
        t.data_offset = self._push_texels(img)
        return self._add_texture(t)

    def texture_checker(self, tex0, tex1, n, m):
        t = self._new_texture(abi.PT_TEX_CHECKER)
        t.tex0, t.tex1, t.width, t.height = tex0, tex1, n, m
        return self._add_texture(t)

    def texture_transformer(self, tex, factor, offset):
        t = self._new_texture(abi.PT_TEX_TRANSFORMER)
        t.tex0 = tex
        t.v[0], t.v[1], t.v[2], t.v[3] = factor[0], factor[1], offset[0], offset[1]
        return self._add_texture(t)

    def texture_value_noise(self, values):
        values = np.ascontiguousarray(values, np.float32)
        t = self._new_texture(abi.PT_TEX_VALUE_NOISE)
        t.height, t.width = values.shape
        t.data_offset = self._push_texels(values)
        return self._add_texture(t)

    def texture_gradient_noise(self, gradients):
        gradients = np.ascontiguousarray(gradients, np.float32)
        t = self._new_texture(abi.PT_TEX_GRADIENT_NOISE)
        t.height, t.width = gradients.shape[:2]
        t.data_offset = self._push_texels(gradients)
        return self._add_texture(t)

    def texture_worley_noise(self, points):
        points = np.ascontiguousarray(points, np.float32)
        t = self._new_texture(abi.PT_TEX_WORLEY_NOISE)
        t.width = len(points)
        t.data_offset = self._push_texels(points)
        return self._add_texture(t)

    def _new_material(self, kind):
        m = abi.pt_material()
        m.type = kind
        m.tex[:] = [-1] * 5
        m.front = m.back = -1
        self.materials.append(m)
        return m, len(self.materials) - 1

    def material_base(self):
        return self._new_material(abi.PT_MAT_BASE)[1]

    def material_light(self, radiance):
        m, i = self._new_material(abi.PT_MAT_LIGHT)
        m.v[0], m.v[1], m.v[2] = [float(v) for v in radiance]
        return i

    def material_lambertian(self, albedo_tex):
        m, i = self._new_material(abi.PT_MAT_LAMBERTIAN)
        m.tex[0] = albedo_tex
        return i

    def material_phong(self, kd, ks, s, opacity=-1, normal=-1):
        m, i = self._new_material(abi.PT_MAT_PHONG)
        m.tex[0], m.tex[1], m.tex[2], m.tex[3], m.tex[4] = kd, ks, s, opacity, normal
        return i

    def material_mirror(self, color_tex):
        m, i = self._new_material(abi.PT_MAT_MIRROR)
        m.tex[0] = color_tex
        return i

    def material_glass(self, absorption=(0.0, 0.0, 0.0), refractive_index=1.5):
        m, i = self._new_material(abi.PT_MAT_GLASS)
        m.v[0], m.v[1], m.v[2] = [float(v) for v in absorption]
        m.v[3] = refractive_index
        return i

    def material_twosided(self, front, back):
        m, i = self._new_material(abi.PT_MAT_TWOSIDED)
        m.front, m.back = front, back
        return i

    # -- surfaces
    def add_mesh(self, positions, indices, material, normals=None, texcoords=None, animation=-1, is_light=False):
        """Mesh constructor + Scene::take(Mesh*, isLight) (reference src/15/mesh.hpp:73-89, scene.hpp:55-61)."""
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, np.uint32).ravel()
        nv, nt = len(positions), indices.size // 3
        has_n = normals is not None and len(normals) > 0
        has_tc = texcoords is not None and len(texcoords) > 0
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3) if has_n else np.zeros((nv, 3), np.float32)
        tc = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2) if has_tc else np.zeros((nv, 2), np.float32)
        tng = np.zeros((nv, 3), np.float32)
        if has_n and has_tc:
            _check(lib().pt_mesh_compute_tangents(_f(positions), _f(nrm), _f(tc), nv,
                                                  indices.ctypes.data_as(_u32p), nt, _f(tng)))
        m = abi.pt_mesh()
        m.first_vertex, m.n_vertices = len(self.positions), nv
        m.first_index, m.n_triangles = self.indices.size, nt
        m.has_normals, m.has_texcoords, m.has_tangents = int(has_n), int(has_tc), int(has_n and has_tc)
        m.material, m.animation = material, animation
        mesh_id = len(self.meshes)
        self.meshes.append(m)
        self.positions = np.concatenate([self.positions, positions])
        self.normals = np.concatenate([self.normals, nrm])
        self.texcoords = np.concatenate([self.texcoords, tc])
        self.tangents = np.concatenate([self.tangents, tng])
        self.indices = np.concatenate([self.indices, indices])
        first_surface = len(self.surfaces)
        rows = np.zeros((nt, 4), np.uint32)
        rows[:, 0] = abi.PT_SURF_TRIANGLE
        rows[:, 1] = mesh_id
        rows[:, 2] = np.arange(nt, dtype=np.uint32)
        self.surfaces = np.concatenate([self.surfaces, rows])
        if is_light:
            self.lights.extend(range(first_surface, first_surface + nt))
        return mesh_id

    def add_sphere(self, center, radius, material, animation=-1, is_light=False):
        """SurfaceSphere + Scene::take(Surface*, isLight) (reference src/15/surface_sphere.hpp:45-48)."""
        q = abi.pt_sphere()
        q.center[:] = [float(v) for v in center]
        q.radius = radius
        q.material, q.animation = material, animation
        self.spheres.append(q)
        sid = len(self.surfaces)
        self.surfaces = np.concatenate(
            [self.surfaces, np.array([[abi.PT_SURF_SPHERE, len(self.spheres) - 1, 0, 0]], np.uint32)])
        if is_light:
            self.lights.append(sid)
        return sid

    def set_envmap_base(self):
        self.envmap.type = abi.PT_ENV_BASE

    def set_envmap_equirect(self, tex):
        self.envmap.type = abi.PT_ENV_EQUIRECT
        self.envmap.tex[0] = tex

    def set_envmap_cube(self, texs):
        self.envmap.type = abi.PT_ENV_CUBE
        self.envmap.tex[:] = list(texs)

    # -- desc
    @staticmethod
    def _carray(ctype, items):
        arr = (ctype * max(len(items), 1))()
        for i, it in enumerate(items):
            arr[i] = it
        return arr

    def desc(self):
        """A pt_scene_desc pointing into this object's arrays (kept alive by self)."""
        d = abi.pt_scene_desc()
        d.abi_version = abi.PT_ABI_VERSION
        keep = {}
        keep["pos"] = np.ascontiguousarray(self.positions, np.float32)
        keep["nrm"] = np.ascontiguousarray(self.normals, np.float32)
        keep["tc"] = np.ascontiguousarray(self.texcoords, np.float32)
        keep["tng"] = np.ascontiguousarray(self.tangents, np.float32)
        keep["idx"] = np.ascontiguousarray(self.indices, np.uint32)
        keep["surf"] = np.ascontiguousarray(self.surfaces, np.uint32)
        keep["lights"] = np.ascontiguousarray(np.asarray(self.lights, np.uint32))
        keep["texels"] = np.ascontiguousarray(self.texels, np.float32)
        keep["nodes"] = np.ascontiguousarray(self.nodes)
        keep["meshes"] = self._carray(abi.pt_mesh, self.meshes)
        keep["spheres"] = self._carray(abi.pt_sphere, self.spheres)
        keep["materials"] = self._carray(abi.pt_material, self.materials)
        keep["textures"] = self._carray(abi.pt_texture, self.textures)
        keep["animations"] = self._carray(abi.pt_animation, self.animations)
        d.n_vertices = len(keep["pos"])
        d.positions, d.normals = _f(keep["pos"]), _f(keep["nrm"])
        d.texcoords, d.tangents = _f(keep["tc"]), _f(keep["tng"])
        d.n_indices, d.indices = keep["idx"].size, keep["idx"].ctypes.data_as(_u32p)
        d.n_meshes, d.meshes = len(self.meshes), keep["meshes"]
        d.n_spheres, d.spheres = len(self.spheres), keep["spheres"]
        d.n_surfaces = len(keep["surf"])
        d.surfaces = keep["surf"].ctypes.data_as(C.POINTER(abi.pt_surface))
        d.n_lights, d.lights = len(self.lights), keep["lights"].ctypes.data_as(_u32p)
        d.n_materials, d.materials = len(self.materials), keep["materials"]
        d.n_textures, d.textures = len(self.textures), keep["textures"]
        d.n_texels, d.texels = keep["texels"].size, _f(keep["texels"])
        d.n_animations, d.animations = len(self.animations), keep["animations"]
        d.n_bvh_nodes = len(keep["nodes"])
        d.bvh_nodes = keep["nodes"].ctypes.data_as(C.POINTER(abi.pt_bvh_node))
        d.envmap = self.envmap
        # every descriptor handed out stays valid for the life of the scene object
        self._keep.append(keep)
        return d

    def build_bvh(self, t0=0.0, t1=0.0):
        """A host-built tree for this scene, Scene::buildBVH(t0, t1) (reference src/15/scene.hpp:69-72), through the
        builder registered with set_host_tree_builder.  The library itself builds trees on the device
        (DeviceScene.build_bvh_device) or takes them from the caller; the tests register the reference's builder
        as restated by their checker, so that scenes they put together get the reference's topology."""
        if _host_tree_builder is None:
            raise PtError(-2, "no host-side tree builder registered: build the tree on the device "
                          "(DeviceScene.build_bvh_device) or hand over the reference's nodes")
        self.nodes = np.zeros(0, NODE_DTYPE)
        d = self.desc()
        _check(lib().pt_scene_desc_validate(C.byref(d)))
        nodes = np.zeros(2 * len(self.surfaces) - 1, NODE_DTYPE)
        self.depth = int(_host_tree_builder(d, t0, t1, nodes))
        self.nodes = nodes
        return nodes

    def save(self, path):
        d = self.desc()
        _check(lib().pt_scene_desc_save(C.byref(d), path.encode()))

    def validate(self):
        d = self.desc()
        _check(lib().pt_scene_desc_validate(C.byref(d)))


class LoadedScene:
    """A scene file loaded by the library (arrays owned by the library)."""

    def __init__(self, path):
        self.ptr = C.POINTER(abi.pt_scene_desc)()
        _check(lib().pt_scene_desc_load(path.encode(), C.byref(self.ptr)))

    def desc(self):
        return self.ptr.contents

    def nodes(self):
        d = self.ptr.contents
        buf = (abi.pt_bvh_node * d.n_bvh_nodes).from_address(C.addressof(d.bvh_nodes.contents))
        return np.frombuffer(buf, dtype=NODE_DTYPE).copy()

    def close(self):
        if self.ptr:
            lib().pt_scene_desc_free(self.ptr)
            self.ptr = C.POINTER(abi.pt_scene_desc)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_camera(path):
    """Read a camera file: one raw pt_camera record."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) != C.sizeof(abi.pt_camera):
        raise PtError(-4, "camera file %s has the wrong size" % path)
    return abi.pt_camera.from_buffer_copy(data)


def save_camera(cam, path):
    with open(path, "wb") as f:
        f.write(bytes(cam))


# ---------------------------------------------------------- device scenes --

def _reindex(fn, lead, vi, ni, ti, tri_material, material, vertices, normals, texcoords):
    vi, ni, ti = (np.ascontiguousarray(a, np.int32) for a in (vi, ni, ti))
    tri_material = np.ascontiguousarray(tri_material, np.int32)
    vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    texcoords = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    nt = len(tri_material)
    idx = np.zeros(max(3 * nt, 1), np.uint32)
    pos, nrm, tc = np.zeros((max(3 * nt, 1), 3), np.float32), np.zeros((max(3 * nt, 1), 3), np.float32), np.zeros((max(3 * nt, 1), 2), np.float32)
    n_idx, n_v, hn, ht = C.c_uint64(0), C.c_uint64(0), C.c_int(0), C.c_int(0)
    ip = lambda a: a.ctypes.data_as(C.c_void_p)
    sizes = lambda a: [C.c_uint64(len(a))] if lead else []
    args = list(lead) + [ip(vi), ip(ni), ip(ti), ip(tri_material), C.c_uint64(nt), C.c_int32(material), ip(vertices)] + sizes(vertices) \
        + [ip(normals)] + sizes(normals) + [ip(texcoords)] + sizes(texcoords) \
        + [ip(idx), ip(pos), ip(nrm), ip(tc), C.byref(n_idx), C.byref(n_v), C.byref(hn), C.byref(ht)]
    rc = fn(*args)
    nv = int(n_v.value)
    return rc, {"indices": idx[:int(n_idx.value)].copy(), "positions": pos[:nv].copy(),
                "normals": nrm[:nv].copy() if hn.value else None, "texcoords": tc[:nv].copy() if ht.value else None}


def mesh_reindex(vi, ni, ti, tri_material, material, vertices, normals, texcoords, device=0):
    """pt_mesh_reindex: the importer's re-indexing of one (material, shape) pair on the GPU."""
    rc, out = _reindex(lib().pt_mesh_reindex, [C.c_int(device)], vi, ni, ti, tri_material, material, vertices, normals, texcoords)
    _check(rc)
    return out


def multi_plan(params, mode, n_shares, share):
    """The render calls (pt_render_params) of one share of an image split over n_shares GPUs (pt_multi_plan)."""
    n = C.c_int(0)
    _check(lib().pt_multi_plan(C.byref(params), mode, n_shares, share, None, 0, C.byref(n)))
    calls = (abi.pt_render_params * max(n.value, 1))()
    _check(lib().pt_multi_plan(C.byref(params), mode, n_shares, share, calls, n.value, C.byref(n)))
    return [abi.pt_render_params.from_buffer_copy(calls[i]) for i in range(n.value)]


def render_multi(device_scenes, cam, params, mode=abi.PT_SHARD_TILES, out=None):
    """pt_render_multi over a list of DeviceScene handles (one per GPU): (height, width, 3) float32."""
    if out is None:
        out = np.zeros((params.height, params.width, 3), np.float32)
    handles = (C.c_void_p * len(device_scenes))(*[d.handle for d in device_scenes])
    _check(lib().pt_render_multi(handles, len(device_scenes), C.byref(cam), C.byref(params), mode, _f(out)))
    return out


def render_multi_device(device_scenes, cam, params, d_out, mode=abi.PT_SHARD_TILES, stream=None):
    """pt_render_multi_device: the image into device memory of the first handle's GPU (address d_out)."""
    handles = (C.c_void_p * len(device_scenes))(*[d.handle for d in device_scenes])
    _check(lib().pt_render_multi_device(handles, len(device_scenes), C.byref(cam), C.byref(params), mode, C.c_void_p(d_out),
                                        C.c_void_p(stream or 0)))


def multi_last_combine(first_scene):
    """(milliseconds, kernel launches, peer copies, direct copies) of the last combine on the first handle."""
    ms, k, c, d = C.c_float(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _check(lib().pt_multi_last_combine(first_scene.handle, C.byref(ms), C.byref(k), C.byref(c), C.byref(d)))
    return ms.value, k.value, c.value, d.value


class DeviceScene:
    """A scene resident in HBM on one GPU (pt_scene)."""

    def __init__(self, scene, device=0):
        d = scene.desc()
        self.handle = C.c_void_p()
        _check(lib().pt_scene_create(C.byref(d), device, C.byref(self.handle)))
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            lib().pt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def update_bvh(self, nodes):
        nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        _check(lib().pt_scene_update_bvh(self.handle, nodes.ctypes.data_as(C.POINTER(abi.pt_bvh_node)), len(nodes)))

    def build_bvh_device(self, t0=0.0, t1=0.0, stream=None):
        """Build the acceleration structure over [t0, t1] on the device and install it."""
        _check(lib().pt_scene_build_bvh_device(self.handle, t0, t1, C.c_void_p(stream or 0)))

    def get_bvh(self):
        """The installed tree as an array of NODE_DTYPE in the layout of pt_scene_desc.bvh_nodes."""
        n = C.c_uint64(0)
        _check(lib().pt_scene_get_bvh(self.handle, None, 0, C.byref(n)))
        nodes = np.zeros(n.value, NODE_DTYPE)
        _check(lib().pt_scene_get_bvh(self.handle, nodes.ctypes.data_as(C.POINTER(abi.pt_bvh_node)), n.value, C.byref(n)))
        return nodes

    def get_quad_rows(self):
        """The rows of the four-wide walk as the device made them: (rows, 16) uint32, 12 words of boxes and 4 links
        (empty if the scene has none)."""
        L = lib()
        L.pt_scene_get_quad_rows.restype = C.c_int
        n = C.c_uint64(0)
        _check(L.pt_scene_get_quad_rows(self.handle, None, C.c_uint64(0), C.byref(n)))
        rows = np.zeros((n.value, 16), np.uint32)
        if n.value:
            _check(L.pt_scene_get_quad_rows(self.handle, rows.ctypes.data_as(C.c_void_p), C.c_uint64(n.value), C.byref(n)))
        return rows

    def render(self, cam, params, out=None):
        """Host-buffer render: (height, width, 3) float32, row 0 = bottom."""
        if out is None:
            out = np.zeros((params.height, params.width, 3), np.float32)
        _check(lib().pt_render(self.handle, C.byref(cam), C.byref(params), _f(out)))
        return out

    def render_counted(self, cam, params):
        out = np.zeros((params.height, params.width, 3), np.float32)
        cnt = abi.pt_counters()
        _check(lib().pt_render_counted(self.handle, C.byref(cam), C.byref(params), _f(out), C.byref(cnt)))
        return out, cnt

    def render_device(self, cam, params, d_out_ptr, stream_ptr=None):
        """Asynchronous render into device memory (raw pointers: e.g. torch data_ptr / cuda_stream)."""
        _check(lib().pt_render_device(self.handle, C.byref(cam), C.byref(params), C.c_void_p(d_out_ptr),
                                      C.c_void_p(stream_ptr) if stream_ptr else None))

    def render_device_noalloc(self, cam, params, d_out_ptr, stream_ptr=None):
        """pt_render_device_noalloc: the same render from the fixed work area in the library's image."""
        _check(lib().pt_render_device_noalloc(self.handle, C.byref(cam), C.byref(params), C.c_void_p(d_out_ptr),
                                              C.c_void_p(stream_ptr) if stream_ptr else None))

    def last_kernel_ms(self):
        return float(lib().pt_last_kernel_ms(self.handle))

    def profile(self, enable=True):
        """Switch the per-kernel event timing on (clearing the sums) or off."""
        _check(lib().pt_scene_profile(self.handle, 1 if enable else 0))

    def kernel_times(self):
        """{kernel name: (launches, total milliseconds)} since profile(True)."""
        recs = (abi.pt_kernel_time * 16)()
        n = C.c_int(0)
        _check(lib().pt_scene_kernel_times(self.handle, recs, 16, C.byref(n)))
        return {recs[i].name.decode(): (int(recs[i].launches), float(recs[i].total_ms)) for i in range(min(n.value, 16))}

    def count_walk(self, enable=True):
        """Switch the instrumented tracing kernels on (clearing their sums) or off."""
        _check(lib().pt_scene_count_walk(self.handle, 1 if enable else 0))

    def walk_counters(self):
        """Work of the executed walk since count_walk(True), as a dict (see pt_walk_counters)."""
        c = abi.pt_walk_counters()
        _check(lib().pt_scene_walk_counters(self.handle, C.byref(c)))
        return c.as_dict()

    def render_samples(self, cam, params):
        out = np.empty((n_samples(params), 3), np.float32)
        _check(lib().pt_render_samples(self.handle, C.byref(cam), C.byref(params), _f(out)))
        return out

    def primary_hits(self, cam, params):
        out = (abi.pt_hit * n_samples(params))()
        _check(lib().pt_primary_hits(self.handle, C.byref(cam), C.byref(params), out))
        return np.frombuffer(out, dtype=HIT_DTYPE).copy()

    def trace_rays(self, origins, directions, times=None, amin=1e-4, amax=float(np.finfo(np.float32).max)):
        origins = np.ascontiguousarray(origins, np.float32)
        directions = np.ascontiguousarray(directions, np.float32)
        n = len(origins)
        tp = None
        if times is not None:
            times = np.ascontiguousarray(times, np.float32)
            tp = _f(times)
        out = (abi.pt_hit * max(n, 1))()
        _check(lib().pt_trace_rays(self.handle, n, _f(origins), _f(directions), tp, amin, amax, out))
        return np.frombuffer(out, dtype=HIT_DTYPE)[:n].copy()


def host_quad_rows(nodes):
    """The rows of the four-wide walk made on the host from a tree in the layout of pt_scene_desc.bvh_nodes (every
    leaf linked by its node: no surface table is consulted).  Returns (rows (n, 16) uint32, base (3,), step (3,))."""
    L = lib()
    L.pt_bvh_quad_rows.restype = C.c_int
    n = len(nodes)
    rows = np.zeros(((n - 1) // 2 + 1, 16), np.uint32)
    n_rows = C.c_uint64(0)
    base, step = (C.c_float * 3)(), (C.c_float * 3)()
    _check(L.pt_bvh_quad_rows(nodes.ctypes.data_as(C.c_void_p), C.c_uint64(n), rows.ctypes.data_as(C.c_void_p),
                              C.c_uint64(len(rows)), C.byref(n_rows), base, step))
    return rows[:n_rows.value], np.array(base[:], np.float32), np.array(step[:], np.float32)
