/*
 * pt_amd.h -- C ABI of the MI355X render-loop library (libptamd.so).
 *
 * The library replaces one block of the reference program: the OpenMP pixel
 * loop of the final course step (reference src/15/pathtracer.cpp:199-221) and
 * everything that loop calls (pathSample, BVH traversal, intersection,
 * materials, textures, environment maps, camera ray generation).  Scene
 * construction, OBJ import and image output stay in the caller.
 *
 * All structs are plain data; all pointers in a pt_scene_desc are host
 * pointers that only have to stay valid for the duration of the call they
 * are passed to.  Every function returns 0 on success and a negative
 * pt_status on failure; pt_last_error() gives the message of the last
 * failure on the calling thread.  Nothing in here aborts the process.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 1u

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad size, index out of range */
    PT_ERR_UNSUPPORTED = -2,      /* a dynamic type outside the closed set below */
    PT_ERR_DEVICE = -3,           /* a HIP call failed; message has the HIP error */
    PT_ERR_IO = -4,               /* scene file could not be read or written */
    PT_ERR_LIMIT = -5             /* a stated limit was exceeded (e.g. BVH depth > 128) */
} pt_status;

/* ---------------------------------------------------------------- scene -- */

/* Pose = translation, rotation quaternion (x,y,z,w), scaling.  Same fields
 * and meaning as class Transformation (reference src/15/transformation.hpp:6-18):
 * apply(v) = translation + rotation * (v * scaling). */
typedef struct pt_transform {
    float translation[3];
    float rotation[4];
    float scaling[3];
} pt_transform;

/* Closed set of Animation::at(t) behaviours (reference src/15/animation.hpp:5-12,
 * animation_constant.hpp:5-18; two-key form = AnimationSphere::at in
 * reference src/07/pathtracer-frames.cpp:83-87). */
enum {
    PT_ANIM_IDENTITY = 0, /* base class: at(t) = Transformation() */
    PT_ANIM_CONSTANT = 1, /* at(t) = T0 */
    PT_ANIM_TWOKEY = 2    /* at(t) = mix(T0, T1, t / duration), not clamped */
};
typedef struct pt_animation {
    int32_t type;
    float duration;
    pt_transform T0;
    pt_transform T1;
} pt_animation;

/* Texture::value(tc, t) family (reference src/15/texture_*.hpp). */
enum {
    PT_TEX_CONSTANT = 0,       /* v[0..2] = value */
    PT_TEX_IMAGE = 1,          /* width x height RGB float texels at data_offset, row 0 = bottom */
    PT_TEX_CHECKER = 2,        /* tex0 / tex1 children, width = n, height = m */
    PT_TEX_TRANSFORMER = 3,    /* tex0 child, v[0..1] = factor, v[2..3] = offset */
    PT_TEX_VALUE_NOISE = 4,    /* width x height scalar lattice at data_offset */
    PT_TEX_GRADIENT_NOISE = 5, /* width x height vec2 lattice at data_offset */
    PT_TEX_WORLEY_NOISE = 6    /* width = number of vec2 feature points at data_offset */
};
typedef struct pt_texture {
    int32_t type;
    int32_t width, height;
    int32_t tex0, tex1;
    int32_t reserved;
    uint64_t data_offset; /* in floats, into pt_scene_desc.texels */
    float v[4];
} pt_texture;

/* Material::{Le, scatter, scatterToDirection} family (reference src/15/material*.hpp). */
enum {
    PT_MAT_BASE = 0,       /* base class: black, ScatterNone */
    PT_MAT_LIGHT = 1,      /* v[0..2] = radiance */
    PT_MAT_LAMBERTIAN = 2, /* tex[0] = albedo */
    PT_MAT_PHONG = 3,      /* tex[0..4] = k_d, k_s, s, opacity (-1 = none), normal (-1 = none) */
    PT_MAT_MIRROR = 4,     /* tex[0] = color */
    PT_MAT_GLASS = 5,      /* v[0..2] = absorption, v[3] = refractive index */
    PT_MAT_TWOSIDED = 6    /* front / back = material indices */
};
typedef struct pt_material {
    int32_t type;
    int32_t tex[5];
    int32_t front, back;
    float v[4];
} pt_material;

/* One Mesh (reference src/15/mesh.hpp:61-102).  Vertices of all meshes are
 * concatenated in the four vertex streams of the scene; indices are relative
 * to first_vertex, three per triangle, starting at indices[first_index]. */
typedef struct pt_mesh {
    uint32_t first_vertex, n_vertices;
    uint32_t first_index, n_triangles;
    int32_t has_normals, has_texcoords, has_tangents;
    int32_t material;
    int32_t animation; /* -1 = none */
    int32_t reserved;
} pt_mesh;

/* One SurfaceSphere (reference src/15/surface_sphere.hpp:40-48). */
typedef struct pt_sphere {
    float center[3];
    float radius;
    int32_t material;
    int32_t animation; /* -1 = none */
} pt_sphere;

/* Entry i describes Scene::surfaces[i] (reference src/15/scene.hpp:19); the
 * index i is the surface id reported in pt_hit and used in pt_bvh_node. */
enum { PT_SURF_TRIANGLE = 0, PT_SURF_SPHERE = 1 };
typedef struct pt_surface {
    uint32_t type;
    uint32_t owner; /* mesh index or sphere index */
    uint32_t index; /* triangle number inside the mesh; 0 for spheres */
    uint32_t reserved;
} pt_surface;

/* One BVHNodeLinear (reference src/15/bvh.hpp:125-133), with the surface
 * pointer of a leaf replaced by 2 * surface id.  link even: leaf, surface =
 * link / 2.  link odd: inner node, first child = this index + 1, second
 * child = link / 2.  Nodes are in the reference's pre-order. */
typedef struct pt_bvh_node {
    float lo[3];
    float hi[3];
    uint64_t link;
} pt_bvh_node;

/* Scene::envMap (reference src/15/envmap*.hpp). */
enum {
    PT_ENV_NONE = 0,     /* envMap is null: a miss adds nothing */
    PT_ENV_BASE = 1,     /* base class: value = 0 */
    PT_ENV_EQUIRECT = 2, /* tex[0] = map */
    PT_ENV_CUBE = 3      /* tex[0..5] = +x, -x, +y, -y, +z, -z */
};
typedef struct pt_envmap {
    int32_t type;
    int32_t tex[6];
    int32_t reserved;
} pt_envmap;

/* Flat image of a reference Scene after buildBVH (reference src/15/scene.hpp:13-73). */
typedef struct pt_scene_desc {
    uint32_t abi_version; /* PT_ABI_VERSION */
    uint32_t reserved;
    uint64_t n_vertices;
    const float* positions; /* 3 per vertex */
    const float* normals;   /* 3 per vertex; read only where pt_mesh.has_normals */
    const float* texcoords; /* 2 per vertex; read only where pt_mesh.has_texcoords */
    const float* tangents;  /* 3 per vertex; read only where pt_mesh.has_tangents */
    uint64_t n_indices;
    const uint32_t* indices;
    uint64_t n_meshes;
    const pt_mesh* meshes;
    uint64_t n_spheres;
    const pt_sphere* spheres;
    uint64_t n_surfaces;
    const pt_surface* surfaces;
    uint64_t n_lights;
    const uint32_t* lights; /* surface ids, in Scene::lights order */
    uint64_t n_materials;
    const pt_material* materials;
    uint64_t n_textures;
    const pt_texture* textures;
    uint64_t n_texels;
    const float* texels;
    uint64_t n_animations;
    const pt_animation* animations;
    uint64_t n_bvh_nodes; /* 2 * n_surfaces - 1, or 0 if not built yet */
    const pt_bvh_node* bvh_nodes;
    pt_envmap envmap;
} pt_scene_desc;

/* Camera (reference src/15/camera.hpp:11-32): frustum corners at distance 1,
 * thin-lens parameters and an optional pose animation. */
typedef struct pt_camera {
    float t, b, r, l;
    float focus_distance;
    float aperture_radius;
    int32_t has_animation;
    pt_animation animation;
} pt_camera;

/* ---------------------------------------------------------------- render -- */

#define PT_RENDER_SUM_ONLY 1u /* write the plain sum over the sample range, do not divide by spp */

/* Parameters of one render call.  width, height, sqrt_spp, t0, t1 are the
 * literals of the reference loop (src/15/pathtracer.cpp:136-139,215); the
 * rest are knobs the reference lacks.  Zero-initialised optional fields give
 * the reference behaviour: whole image, all sqrt_spp^2 samples, at most 128
 * path segments, Russian roulette from segment 5. */
typedef struct pt_render_params {
    int32_t width, height, sqrt_spp;
    float t0, t1;
    uint32_t flags;
    uint64_t seed;
    int32_t tile_x0, tile_y0, tile_x1, tile_y1; /* pixel rectangle [x0,x1) x [y0,y1); all 0 = whole image */
    int32_t sample_begin, sample_end;           /* sample indices [begin,end) of sqrt_spp^2; both 0 = all */
    int32_t max_path_segments;                  /* 0 = 128 */
    int32_t rr_start_segment;                   /* 0 = 5; use -1 for "from the first segment" */
} pt_render_params;

/* Closest hit of one ray: what HitRecord (reference src/15/surface.hpp:9-30)
 * carries that identifies the hit. surface = 0xFFFFFFFF on a miss. */
#define PT_NO_HIT 0xFFFFFFFFu
typedef struct pt_hit {
    uint32_t surface;
    uint32_t material;
    float a;
    uint32_t backside;
} pt_hit;

/* Work counters of the reference algorithm, summed over a render call
 * (definitions: one ray = one BVHTreeLinear::hit call). */
typedef struct pt_counters {
    uint64_t path_samples;
    uint64_t rays;         /* BVH closest-hit queries (path segments + shadow rays) */
    uint64_t node_visits;  /* AABB slab tests */
    uint64_t leaf_tests;   /* primitive tests reached through leaves */
    uint64_t light_tests;  /* primitive tests made by light pdf evaluation */
    uint64_t vertices;     /* path vertices shaded (hits) */
    uint64_t rng_draws;
} pt_counters;

typedef struct pt_scene pt_scene; /* opaque: a scene resident in HBM on one GPU */

/* Message of the last failure on the calling thread ("" if none). */
const char* pt_last_error(void);

/* Library version string; also usable as a load check. */
const char* pt_version(void);

/* Number of visible GPUs (0 without a GPU; never fails). */
int pt_device_count(void);

/* Upload a flat scene to GPU `device`.  Replaces the implicit "scene is in
 * host memory" state at reference src/15/pathtracer.cpp:198.  The desc
 * includes the BVH (n_bvh_nodes == 2 * n_surfaces - 1), or none at all
 * (n_bvh_nodes == 0): then pt_scene_build_bvh_device or pt_scene_update_bvh
 * has to give the scene a tree before the first render; until then every
 * pt_render* / pt_trace_rays / pt_primary_hits call fails with
 * PT_ERR_INVALID_ARGUMENT.  The host arrays may be freed after the call.
 *
 * Calls on one scene: the work areas, the tree and the timing state of a
 * scene serve one call at a time.  The library serialises them itself: an
 * entry point holds the scene's lock while it enqueues its work, and its
 * stream first waits (hipStreamWaitEvent) for the work the previous call on
 * that scene enqueued, whichever stream that was.  So two threads, or two
 * streams, may use one scene; they do not overlap on the GPU.  Use one scene
 * handle per stream for concurrent rendering. */
int pt_scene_create(const pt_scene_desc* desc, int device, pt_scene** out_scene);

/* Replace the acceleration structure (per-frame rebuild over a new shutter
 * interval: reference src/07/pathtracer-frames.cpp:209-211). */
int pt_scene_update_bvh(pt_scene* scene, const pt_bvh_node* nodes, uint64_t n_nodes);

void pt_scene_destroy(pt_scene* scene);

/* Render into a host buffer of 3 * width * height floats: pixel (x, y) at
 * out_rgb[3 * (y * width + x)], y = 0 is the bottom row, linear radiance,
 * mean over the samples -- the contents of `img` after the loop at reference
 * src/15/pathtracer.cpp:199-221.  Pixels outside the tile are left untouched. */
int pt_render(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
              float* out_rgb);

/* Same, into device memory on the scene's GPU, enqueued on `hip_stream`
 * (a hipStream_t passed as void*; NULL = default stream).  The call returns
 * before the image is complete -- the last tracing and the summing kernel may
 * still be running; the image is complete for work enqueued on `hip_stream`
 * after the call, or after synchronising that stream -- but it is not a
 * non-blocking call: while paths are in flight the host reads the path
 * counters back every few iterations (hipStreamSynchronize on `hip_stream`)
 * to learn when the batch has drained, so it cannot be captured into a graph.
 * max_path_segments is at most 4095 (PT_ERR_LIMIT beyond; the reference stops at 128). */
int pt_render_device(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
                     float* d_out_rgb, void* hip_stream);

/* pt_render_device for callers that cannot allocate device memory (code under
 * HIP graph capture set-up, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  Same arguments, same image
 * bit for bit; but the work areas of the renderer (path pools, queues,
 * counters, sample buffer, running sums) are not allocated: they are carved
 * from a 64 MiB device array that is part of the library's image, one copy
 * per GPU, held until the process ends.  The call makes no hipMalloc /
 * hipFree / hipHostMalloc and creates no events (per-kernel timing, see
 * pt_scene_profile, is off during it).
 * Limits: at most 524 288 pixels in the tile rectangle of one call, counted
 * in the whole 8 x 8 pixel tiles that cover it (PT_ERR_LIMIT beyond; render
 * larger images tile by tile), any number of samples (they are run in batches
 * that fit the area).  The pools hold 131 072 paths, a fraction of what
 * pt_render_device uses by default, so this entry point is slower on large
 * renders.
 * All calls on one GPU share that GPU's copy of the area, whatever their
 * scene and stream: the library orders them on the device -- every call
 * makes its stream wait for an event that the previous call on that GPU
 * recorded behind its last kernel (the event is created with the first scene
 * on the GPU, not by the call).  Only a call that arrives while another
 * thread is still inside a call on that GPU is refused, with
 * PT_ERR_INVALID_ARGUMENT, rather than made to wait. */
int pt_render_device_noalloc(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
                             float* d_out_rgb, void* hip_stream);

/* One image rendered by several GPUs of a node.  The reference shards by
 * process and file: tiles written as raw rows and stitched by a second program
 * (reference src/07/pathtracer-tiles.cpp:72-139, tile-composer.cpp:10-37).
 * Here the caller creates the scene once per GPU (pt_scene_create with
 * different device numbers; two handles on one GPU are allowed) and
 * pt_render_multi renders `params` with one host thread per handle:
 *   PT_SHARD_TILES    256 x 256 tiles of the rectangle round robin over the
 *                     handles, gathered into the first handle's GPU: the image
 *                     equals pt_render's bit for bit;
 *   PT_SHARD_SAMPLES  contiguous ranges of the sample indices, partial sums
 *                     added on the first handle's GPU in handle order: equals
 *                     pt_render's image to rounding (another addition order).
 * The first handle renders into the result; per other handle the combine is
 * one asynchronous peer copy of the rectangle's rows (over xGMI where
 * hipDeviceCanAccessPeer says so, which is then switched on; staged through
 * the host by the runtime where not) and one kernel on the first handle's GPU.
 * The work buffers are kept with the handles, so a call allocates only when
 * the image is larger than any before.  (Cross-GPU copies have only been
 * rehearsed with several handles on one GPU so far.)
 * pt_render_multi: out_rgb as for pt_render (host memory).
 * pt_render_multi_device: the result goes to d_out_rgb, a buffer of
 * width*height*3 floats on the FIRST handle's GPU; the first handle's work and
 * the combine are enqueued on hip_stream (a stream of that GPU, NULL = default
 * stream), the call returns once everything is enqueued.  Pixels outside the
 * rectangle are not touched by either.  One call at a time per first handle.
 * pt_multi_last_combine: device time between the first handle's last render
 * kernel and the end of the last combine on first_scene (waits for it), and
 * what it consisted of: kernels launched, peer copies, and how many of those
 * went between GPUs that can reach each other's memory (or within one GPU).
 * pt_multi_plan gives the render calls of share `share` of `n_shares` (pure
 * host code; call with capacity 0 for the count); a distributed caller (one
 * process per GPU) renders its share's calls with pt_render_device and
 * combines with a collective instead. */
enum { PT_SHARD_TILES = 0, PT_SHARD_SAMPLES = 1 };
int pt_multi_plan(const pt_render_params* params, int mode, int n_shares, int share, pt_render_params* out_calls,
                  int capacity, int* out_count);
int pt_render_multi(pt_scene* const* scenes, int n_scenes, const pt_camera* camera, const pt_render_params* params,
                    int mode, float* out_rgb);
int pt_render_multi_device(pt_scene* const* scenes, int n_scenes, const pt_camera* camera, const pt_render_params* params,
                           int mode, float* d_out_rgb, void* hip_stream);
int pt_multi_last_combine(pt_scene* first_scene, float* out_ms, int* out_kernel_launches, int* out_peer_copies,
                          int* out_direct_copies);

/* Closest hits of n caller-supplied rays: BVHTreeLinear::hit(ray, amin, amax)
 * (reference src/15/bvh.hpp:183-213) for each.  Host arrays: origins and
 * directions 3 floats per ray, times 1 per ray (may be NULL = 0). */
int pt_trace_rays(pt_scene* scene, uint64_t n_rays, const float* origins, const float* directions,
                  const float* times, float amin, float amax, pt_hit* out_hits);

/* Closest hit of the camera ray of every (pixel, sample) of a render call,
 * in pixel-major, sample-minor order: n = tile pixels * samples entries. */
int pt_primary_hits(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
                    pt_hit* out_hits);

/* Radiance of every single path sample of a render call (same order as
 * pt_primary_hits), 3 floats each: the return values of pathSample
 * (reference src/15/pathtracer.cpp:50-130) before accumulation. */
int pt_render_samples(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
                      float* out_rgb_per_sample);

/* Render with instrumentation: fills the work counters of the reference
 * algorithm for this call (slower than pt_render; same image). */
int pt_render_counted(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
                      float* out_rgb, pt_counters* out_counters);

/* Milliseconds the GPU spent in the kernels of the last pt_render* call on
 * this scene (HIP events on the launch stream); negative if none yet. */
float pt_last_kernel_ms(pt_scene* scene);

/* Device time per kernel of the wavefront renderer (generateKernel, traceKernel
 * = every tracing kernel of one step, classifyKernel, shadeHitsKernel,
 * resolveKernel), measured with HIP events recorded on the launch stream between the
 * kernels.  The reference has no counterpart (it has one loop and no kernels);
 * this is what a benchmark prices against the hardware's limits.
 * pt_scene_profile(scene, 1) clears the sums and switches the events on for the
 * following pt_render* calls, (scene, 0) switches them off.  pt_scene_kernel_times
 * waits for the scene's outstanding work, then writes up to `capacity` records
 * and the number of kernels to *count. */
typedef struct pt_kernel_time {
    char name[24];
    uint64_t launches;
    double total_ms;
} pt_kernel_time;

int pt_scene_profile(pt_scene* scene, int enable);
int pt_scene_kernel_times(pt_scene* scene, pt_kernel_time* out, int capacity, int* count);

/* Work of the walk the tracing kernels actually execute -- which for scenes
 * held in memory is not the reference's storage-order walk (pt_counters counts
 * that one) but a nearer-child-first walk plus a storage-order second pass for
 * the rays it leaves open.  The reference has no counterpart.  These sums
 * price a tracing kernel against the memory system: every box row is 32 bytes,
 * every surface test 48 bytes of surface rows.  pt_scene_count_walk(scene, 1)
 * clears the sums and makes the following pt_render* / pt_trace_rays calls
 * launch instrumented builds of the tracing kernels (slower; same results),
 * (scene, 0) goes back to the plain ones; pt_scene_walk_counters waits for the
 * scene's work and reads the sums. */
typedef struct pt_walk_counters {
    uint64_t path_rays;            /* rays of paths traced by the first pass */
    uint64_t shadow_rays;          /* queued shadow rays traced by the first pass */
    uint64_t box_rows;             /* 32-byte rows fetched for box tests: one node (storage-order walk) or both
                                      children of an inner node on a 16-bit grid (nearer-child-first walk) */
    uint64_t leaf_rows;            /* nearer-child-first walk: exact leaf boxes read from the node table, 32 bytes */
    uint64_t surface_tests;        /* triangle / sphere rows fetched and tested, 48 bytes */
    uint64_t retry_rays;           /* rays handed to the storage-order second pass */
    uint64_t retry_node_visits;    /* its node visits, 32 bytes each */
    uint64_t retry_surface_tests;  /* its surface tests, 48 bytes each */
    uint64_t null_shadow_rays;     /* shadow rays of the reference algorithm that were not traced: the light-pdf test
                                      of the chosen light had already shown that they could add nothing but zeros
                                      (they miss the light, or arrive at a side that does not emit) */
} pt_walk_counters;

int pt_scene_count_walk(pt_scene* scene, int enable);
int pt_scene_walk_counters(pt_scene* scene, pt_walk_counters* out);

/* ----------------------------------------------------------- output stage -- */

/* What the reference does with the finished image before it writes image.ppm
 * (reference src/15/pathtracer.cpp:225): to8Bit (src/15/color.hpp:84-98: the
 * sRGB transfer of color.hpp:36-39, clamp and round of color.hpp:66-70) and
 * the bottom-up to top-down row order of saveImageAsPPM (src/15/imgsave.hpp:
 * 29-33); optionally preceded by uniformRationalQuantization (color.hpp:71-82;
 * enabled by urq_max_val > 0).  in: width * height * 3 floats, row 0 = bottom
 * (what pt_render writes); out: width * height * 3 bytes, row 0 = top if
 * flip_rows != 0 (the payload of the PPM file), else in the order of the input.
 * The _device form reads and writes memory of GPU `device` on the given stream. */
int pt_image_to_8bit_device(int device, const float* d_rgb, int width, int height, float urq_max_val,
                            float urq_brightness, int flip_rows, unsigned char* d_out, void* hip_stream);
int pt_image_to_8bit(int device, const float* rgb, int width, int height, float urq_max_val, float urq_brightness,
                     int flip_rows, unsigned char* out);

/* The acceleration structure built on the device for the time interval
 * [t0, t1] and installed in the scene: what BVHNode::buildBVH +
 * BVHTreeLinear::flattenBVH produce on one host thread (reference
 * src/15/bvh.hpp:36-108,146-176) and src/07/pathtracer-frames.cpp:209-211
 * repeats per frame.  Surface bounds are the reference's (Surface::aabb,
 * surface_triangle.hpp:32-63, surface_sphere.hpp:50-69); the tree is built
 * top-down with the reference's surface-area criterion (bvh.hpp:60-96) over
 * 32 bins per axis, and over every split of the sorted order below 65
 * surfaces -- not the reference's one-axis sorted sweep, so rays find the
 * same surface at the same distance except where two surfaces are hit at
 * exactly equal distances (the walk order decides those).  Equal inputs give
 * equal trees.  With PT_BVH_BUILDER=morton in the environment the tree is a
 * radix tree over Morton codes instead: a fifth of the build time, slower to
 * walk.  A scene may be created with n_bvh_nodes == 0 and get its first tree
 * here.
 * pt_scene_get_bvh copies the installed tree out in the layout of
 * pt_scene_desc.bvh_nodes (call with out_nodes == NULL to get the count). */
int pt_scene_build_bvh_device(pt_scene* scene, float t0, float t1, void* hip_stream);
int pt_scene_get_bvh(pt_scene* scene, pt_bvh_node* out_nodes, uint64_t capacity, uint64_t* out_count);

/* Diagnostics of the table that the tracing kernel for scenes in memory reads
 * (no counterpart in the reference, whose walk, src/15/bvh.hpp:183-213, reads
 * the nodes themselves).  A row stands for an inner node and holds up to four
 * of its descendants -- beginning with its two children, the inner node with
 * the largest box surface gives way to its own children until there are four
 * or only leaves -- as 16 words: per slot three words of box corners on a
 * 16-bit grid over the root box (lo.x | lo.y << 16, lo.z | hi.x << 16,
 * hi.y | hi.z << 16; coordinate q of axis a stands for fmaf(q, step[a],
 * base[a]), boxes rounded outwards), then four links: 0x80000000 | row for an
 * inner node, 0x40000000 | node for a leaf, a surface id for a static triangle
 * with the box of its vertices, 0x7FFFFFFF for a slot not used.  Rows are in
 * the order of the nodes they stand for.
 * pt_scene_get_quad_rows copies out the rows that the library made on the GPU
 * for the installed tree (out_rows == NULL: the count only).
 * pt_bvh_quad_rows makes them on the host from a node array (every leaf linked
 * by its node: no surface table is consulted); it needs no GPU. */
int pt_scene_get_quad_rows(pt_scene* scene, uint32_t* out_rows, uint64_t capacity_rows, uint64_t* out_n_rows);
int pt_bvh_quad_rows(const pt_bvh_node* nodes, uint64_t n_nodes, uint32_t* out_rows, uint64_t capacity_rows,
                     uint64_t* out_n_rows, float out_base[3], float out_step[3]);

/* The two per-texel image passes of the reference's importer, on the GPU.
 * pt_texture_from_bytes: what TextureImage's constructor gets from stb_image
 * for an 8-bit RGB file (reference src/15/texture_image.hpp:17-33): texel =
 * (float) pow(sample / 255.0f, linearize ? 2.2f : 1.0f), rows bottom-up if
 * flip_rows (the reference flips on load).  rgb8 = width * height * 3 bytes in
 * file order, out = width * height * 3 floats (the texels of a PT_TEX_IMAGE).
 * pt_bump_to_normal_map: bumpMapToNormalMap (reference src/15/import.hpp:47-67)
 * on the texels of a bump map (first component = height).
 * The _device variants take device pointers on `device` and a HIP stream. */
int pt_texture_from_bytes(int device, const unsigned char* rgb8, int width, int height, int linearize, int flip_rows,
                          float* out_texels);
int pt_texture_from_bytes_device(int device, const unsigned char* d_rgb8, int width, int height, int linearize,
                                 int flip_rows, float* d_out_texels, void* hip_stream);
int pt_bump_to_normal_map(int device, const float* bump_texels, int width, int height, float bump_factor,
                          float* out_texels);
int pt_bump_to_normal_map_device(int device, const float* d_bump_texels, int width, int height, float bump_factor,
                                 float* d_out_texels, void* hip_stream);

/* The importer's vertex re-indexing for one (material, shape) pair, on the GPU:
 * what the loop at reference src/15/import.hpp:240-284 computes through a
 * std::map per pair.  In: the shape as the OBJ parser delivers it -- per face
 * corner (3 per triangle, file order) the position / normal / texcoord index
 * (-1 = none), per triangle its material id -- the file's attribute arrays, and
 * the pair's material id.  Out: the index list of the faces with that material
 * (a new vertex number per distinct index triple, numbered in order of first
 * appearance) and the vertices in that numbering: positions, normals
 * (normalised, import.hpp:265-267) and texcoords.  Normals (texcoords) are
 * delivered only if every distinct triple has one (out_have_* = 0 otherwise:
 * the importer then builds the Mesh without them, import.hpp:281-284).  The out
 * arrays hold up to 3 * n_triangles indices / vertices.  Streams are identical
 * to the importer's bit for bit. */
int pt_mesh_reindex(int device, const int32_t* corner_vi, const int32_t* corner_ni, const int32_t* corner_ti,
                    const int32_t* tri_material, uint64_t n_triangles, int32_t material, const float* vertices,
                    uint64_t n_vertices, const float* normals, uint64_t n_normals, const float* texcoords,
                    uint64_t n_texcoords, uint32_t* out_indices, float* out_positions, float* out_normals,
                    float* out_texcoords, uint64_t* out_n_indices, uint64_t* out_n_vertices, int* out_have_normals,
                    int* out_have_texcoords);

/* ------------------------------------------------------- host-side helpers -- */

/* Per-vertex tangents of one mesh as Mesh's constructor computes them
 * (reference src/15/mesh.hpp:10-55). */
int pt_mesh_compute_tangents(const float* positions, const float* normals, const float* texcoords,
                             uint32_t n_vertices, const uint32_t* indices, uint32_t n_triangles,
                             float* out_tangents);

/* Scene files (".ptscene"): a flat scene serialised verbatim.  The loaded
 * desc owns its arrays; release it with pt_scene_desc_free. */
int pt_scene_desc_load(const char* path, pt_scene_desc** out_desc);
int pt_scene_desc_save(const pt_scene_desc* desc, const char* path);
void pt_scene_desc_free(pt_scene_desc* desc);

/* Check indices and sizes of a desc (every index in range, node count,
 * acyclic texture / material references).  Called by pt_scene_create. */
int pt_scene_desc_validate(const pt_scene_desc* desc);

#ifdef __cplusplus
}
#endif

#endif /* PT_AMD_H */
