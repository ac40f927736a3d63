// reference_binding.hpp -- the caller-side half of the drop-in boundary.
//
// Include this header in a reference-style main, AFTER the reference headers
// of the final course step (the include block of reference
// src/15/pathtracer.cpp:4-34).  It uses the reference's own classes (Scene,
// Camera, Mesh, SurfaceTriangle, ...) and contains none of their code: it
// walks an existing object graph, writes it into the flat tables of
// pt_amd.h, and calls the C ABI.  The call
//
//     std::vector<vec3> img = ptamd::render(scene, camera, width, height, sqrtSpp, t0, t1);
//
// replaces the pixel loop at reference src/15/pathtracer.cpp:199-221 (run it
// after scene.buildBVH, reference src/15/pathtracer.cpp:198).
//
// Dynamic types are recognised by exact typeid against the closed set shipped
// in src/15.  A user-defined Animation subclass (the reference has one in
// src/07/pathtracer-frames.cpp:63-88) can be supplied through
// FlattenOptions::translateAnimation; any other unknown subclass makes
// flattenScene fail with a message instead of guessing.
#pragma once

#include <functional>
#include <map>
#include <string>
#include <typeinfo>
#include <vector>

#include "pt_amd.h"
#include "pt_scene_io.hpp"

namespace ptamd {

struct FlattenOptions {
    // Return true and fill `out` if `anim` is a type the caller knows how to express.
    std::function<bool(const Animation* anim, pt_animation& out)> translateAnimation;
};

inline pt_transform toPod(const Transformation& T)
{
    pt_transform p;
    for (int i = 0; i < 3; i++) {
        p.translation[i] = T.translation[i];
        p.scaling[i] = T.scaling[i];
    }
    p.rotation[0] = T.rotation.x;
    p.rotation[1] = T.rotation.y;
    p.rotation[2] = T.rotation.z;
    p.rotation[3] = T.rotation.w;
    return p;
}

inline bool animationToPod(const Animation* anim, const FlattenOptions& opt, pt_animation& out, std::string& err)
{
    std::memset(&out, 0, sizeof(out));
    out.T0 = toPod(Transformation());
    out.T1 = out.T0;
    out.duration = 1.0f;
    if (typeid(*anim) == typeid(Animation)) {
        out.type = PT_ANIM_IDENTITY;
        return true;
    }
    if (typeid(*anim) == typeid(AnimationConstant)) {
        out.type = PT_ANIM_CONSTANT;
        out.T0 = toPod(static_cast<const AnimationConstant*>(anim)->T);
        out.T1 = out.T0;
        return true;
    }
    if (opt.translateAnimation && opt.translateAnimation(anim, out))
        return true;
    err = std::string("unsupported Animation subclass ") + typeid(*anim).name();
    return false;
}

class SceneFlattener
{
public:
    ptio::SceneData& out;
    const FlattenOptions& opt;
    std::string err;

    std::map<const Animation*, int> animIds;
    std::map<const Texture*, int> texIds;
    std::map<const Material*, int> matIds;
    std::map<const Mesh*, int> meshIds;
    std::map<const Surface*, uint32_t> surfIds;

    SceneFlattener(ptio::SceneData& out, const FlattenOptions& opt) : out(out), opt(opt) {}

    int animId(const Animation* a)
    {
        if (!a)
            return -1;
        auto it = animIds.find(a);
        if (it != animIds.end())
            return it->second;
        pt_animation pa;
        if (!animationToPod(a, opt, pa, err))
            return -2;
        int id = int(out.animations.size());
        out.animations.push_back(pa);
        animIds[a] = id;
        return id;
    }

    int texId(const Texture* t)
    {
        if (!t)
            return -1;
        auto it = texIds.find(t);
        if (it != texIds.end())
            return it->second;
        pt_texture pt;
        std::memset(&pt, 0, sizeof(pt));
        pt.tex0 = pt.tex1 = -1;
        const std::type_info& ti = typeid(*t);
        if (ti == typeid(TextureConstant)) {
            const TextureConstant* c = static_cast<const TextureConstant*>(t);
            pt.type = PT_TEX_CONSTANT;
            for (int i = 0; i < 3; i++)
                pt.v[i] = c->val[i];
        } else if (ti == typeid(TextureImage)) {
            const TextureImage* c = static_cast<const TextureImage*>(t);
            pt.type = PT_TEX_IMAGE;
            pt.width = c->width;
            pt.height = c->height;
            pt.data_offset = out.texels.size();
            for (size_t i = 0; i < c->img.size(); i++)
                for (int k = 0; k < 3; k++)
                    out.texels.push_back(c->img[i][k]);
        } else if (ti == typeid(TextureChecker)) {
            const TextureChecker* c = static_cast<const TextureChecker*>(t);
            pt.type = PT_TEX_CHECKER;
            pt.width = c->n;
            pt.height = c->m;
            pt.tex0 = texId(c->t0);
            pt.tex1 = texId(c->t1);
            if (pt.tex0 < 0 || pt.tex1 < 0) {
                if (err.empty())
                    err = "TextureChecker with a null child";
                return -2;
            }
        } else if (ti == typeid(TextureTransformer)) {
            const TextureTransformer* c = static_cast<const TextureTransformer*>(t);
            pt.type = PT_TEX_TRANSFORMER;
            pt.tex0 = texId(c->tex);
            if (pt.tex0 < 0) {
                if (err.empty())
                    err = "TextureTransformer with a null child";
                return -2;
            }
            pt.v[0] = c->factor.x();
            pt.v[1] = c->factor.y();
            pt.v[2] = c->offset.x();
            pt.v[3] = c->offset.y();
        } else if (ti == typeid(TextureValueNoise)) {
            const TextureValueNoise* c = static_cast<const TextureValueNoise*>(t);
            pt.type = PT_TEX_VALUE_NOISE;
            pt.width = c->width;
            pt.height = c->height;
            pt.data_offset = out.texels.size();
            out.texels.insert(out.texels.end(), c->values.begin(), c->values.end());
        } else if (ti == typeid(TextureGradientNoise)) {
            const TextureGradientNoise* c = static_cast<const TextureGradientNoise*>(t);
            pt.type = PT_TEX_GRADIENT_NOISE;
            pt.width = c->width;
            pt.height = c->height;
            pt.data_offset = out.texels.size();
            for (size_t i = 0; i < c->values.size(); i++) {
                out.texels.push_back(c->values[i].x());
                out.texels.push_back(c->values[i].y());
            }
        } else if (ti == typeid(TextureWorleyNoise)) {
            const TextureWorleyNoise* c = static_cast<const TextureWorleyNoise*>(t);
            pt.type = PT_TEX_WORLEY_NOISE;
            pt.width = int(c->points.size());
            pt.data_offset = out.texels.size();
            for (size_t i = 0; i < c->points.size(); i++) {
                out.texels.push_back(c->points[i].x());
                out.texels.push_back(c->points[i].y());
            }
        } else {
            err = std::string("unsupported Texture subclass ") + ti.name();
            return -2;
        }
        int id = int(out.textures.size());
        out.textures.push_back(pt);
        texIds[t] = id;
        return id;
    }

    int matId(const Material* m)
    {
        if (!m)
            return -1;
        auto it = matIds.find(m);
        if (it != matIds.end())
            return it->second;
        pt_material pm;
        std::memset(&pm, 0, sizeof(pm));
        for (int i = 0; i < 5; i++)
            pm.tex[i] = -1;
        pm.front = pm.back = -1;
        const std::type_info& ti = typeid(*m);
        bool texOk = true;
        if (ti == typeid(Material)) {
            pm.type = PT_MAT_BASE;
        } else if (ti == typeid(MaterialLight)) {
            pm.type = PT_MAT_LIGHT;
            for (int i = 0; i < 3; i++)
                pm.v[i] = static_cast<const MaterialLight*>(m)->radiance[i];
        } else if (ti == typeid(MaterialLambertian)) {
            pm.type = PT_MAT_LAMBERTIAN;
            pm.tex[0] = texId(static_cast<const MaterialLambertian*>(m)->albedo);
            texOk = pm.tex[0] >= 0;
        } else if (ti == typeid(MaterialPhong)) {
            const MaterialPhong* c = static_cast<const MaterialPhong*>(m);
            pm.type = PT_MAT_PHONG;
            pm.tex[0] = texId(c->k_d);
            pm.tex[1] = texId(c->k_s);
            pm.tex[2] = texId(c->s);
            pm.tex[3] = texId(c->opacity);
            pm.tex[4] = texId(c->normal);
            texOk = pm.tex[0] >= 0 && pm.tex[1] >= 0 && pm.tex[2] >= 0 && pm.tex[3] >= -1 && pm.tex[4] >= -1;
        } else if (ti == typeid(MaterialMirror)) {
            pm.type = PT_MAT_MIRROR;
            pm.tex[0] = texId(static_cast<const MaterialMirror*>(m)->color);
            texOk = pm.tex[0] >= 0;
        } else if (ti == typeid(MaterialGlass)) {
            const MaterialGlass* c = static_cast<const MaterialGlass*>(m);
            pm.type = PT_MAT_GLASS;
            for (int i = 0; i < 3; i++)
                pm.v[i] = c->absorption[i];
            pm.v[3] = c->refractiveIndex;
        } else if (ti == typeid(MaterialTwoSided)) {
            const MaterialTwoSided* c = static_cast<const MaterialTwoSided*>(m);
            pm.type = PT_MAT_TWOSIDED;
            // reserve the id first so that children get larger ids
            int id = int(out.materials.size());
            out.materials.push_back(pm);
            matIds[m] = id;
            int f = matId(c->front);
            int b = matId(c->back);
            if (f < 0 || b < 0) {
                if (err.empty())
                    err = "MaterialTwoSided with a null side";
                return -2;
            }
            out.materials[id].front = f;
            out.materials[id].back = b;
            return id;
        } else {
            err = std::string("unsupported Material subclass ") + ti.name();
            return -2;
        }
        if (!texOk) {
            if (err.empty())
                err = "material with a missing required texture";
            return -2;
        }
        int id = int(out.materials.size());
        out.materials.push_back(pm);
        matIds[m] = id;
        return id;
    }

    bool run(const Scene& scene)
    {
        // ids follow the order of the owning vectors of Scene first
        for (const auto& a : scene.animations)
            if (animId(a.get()) == -2)
                return false;
        for (const auto& t : scene.textures)
            if (texId(t.get()) == -2)
                return false;
        for (const auto& m : scene.materials)
            if (matId(m.get()) == -2)
                return false;

        for (const auto& mp : scene.meshes) {
            const Mesh* mesh = mp.get();
            pt_mesh pm;
            std::memset(&pm, 0, sizeof(pm));
            pm.first_vertex = uint32_t(out.positions.size() / 3);
            pm.n_vertices = uint32_t(mesh->positions.size());
            pm.first_index = uint32_t(out.indices.size());
            pm.n_triangles = uint32_t(mesh->indices.size() / 3);
            pm.has_normals = mesh->normals.size() > 0;
            pm.has_texcoords = mesh->texcoords.size() > 0;
            pm.has_tangents = mesh->tangents.size() > 0;
            pm.material = matId(mesh->material);
            pm.animation = animId(mesh->animation);
            if (pm.material < 0 || pm.animation == -2) {
                if (err.empty())
                    err = "mesh without material";
                return false;
            }
            for (size_t i = 0; i < mesh->positions.size(); i++) {
                for (int k = 0; k < 3; k++) {
                    out.positions.push_back(mesh->positions[i][k]);
                    out.normals.push_back(pm.has_normals ? mesh->normals[i][k] : 0.0f);
                    out.tangents.push_back(pm.has_tangents ? mesh->tangents[i][k] : 0.0f);
                }
                for (int k = 0; k < 2; k++)
                    out.texcoords.push_back(pm.has_texcoords ? mesh->texcoords[i][k] : 0.0f);
            }
            out.indices.insert(out.indices.end(), mesh->indices.begin(), mesh->indices.end());
            meshIds[mesh] = int(out.meshes.size());
            out.meshes.push_back(pm);
        }

        for (const auto& sp : scene.surfaces) {
            const Surface* s = sp.get();
            pt_surface ps;
            std::memset(&ps, 0, sizeof(ps));
            const std::type_info& ti = typeid(*s);
            if (ti == typeid(SurfaceTriangle)) {
                const SurfaceTriangle* tri = static_cast<const SurfaceTriangle*>(s);
                auto it = meshIds.find(&tri->mesh);
                if (it == meshIds.end()) {
                    err = "triangle of a mesh that the scene does not own";
                    return false;
                }
                ps.type = PT_SURF_TRIANGLE;
                ps.owner = uint32_t(it->second);
                ps.index = uint32_t((tri->indices - tri->mesh.indices.data()) / 3);
            } else if (ti == typeid(SurfaceSphere)) {
                const SurfaceSphere* sph = static_cast<const SurfaceSphere*>(s);
                pt_sphere q;
                for (int k = 0; k < 3; k++)
                    q.center[k] = sph->center[k];
                q.radius = sph->radius;
                q.material = matId(sph->material);
                q.animation = animId(sph->animation);
                if (q.material < 0 || q.animation == -2) {
                    if (err.empty())
                        err = "sphere without material";
                    return false;
                }
                ps.type = PT_SURF_SPHERE;
                ps.owner = uint32_t(out.spheres.size());
                out.spheres.push_back(q);
            } else {
                err = std::string("unsupported Surface subclass ") + ti.name();
                return false;
            }
            surfIds[s] = uint32_t(out.surfaces.size());
            out.surfaces.push_back(ps);
        }

        for (const Surface* l : scene.lights) {
            auto it = surfIds.find(l);
            if (it == surfIds.end()) {
                err = "light that is not one of the scene's surfaces";
                return false;
            }
            out.lights.push_back(it->second);
        }

        std::memset(&out.envmap, 0, sizeof(out.envmap));
        for (int i = 0; i < 6; i++)
            out.envmap.tex[i] = -1;
        if (!scene.envMap) {
            out.envmap.type = PT_ENV_NONE;
        } else {
            const EnvMap* e = scene.envMap.get();
            const std::type_info& ti = typeid(*e);
            if (ti == typeid(EnvMap)) {
                out.envmap.type = PT_ENV_BASE;
            } else if (ti == typeid(EnvMapEquiRect)) {
                out.envmap.type = PT_ENV_EQUIRECT;
                out.envmap.tex[0] = texId(static_cast<const EnvMapEquiRect*>(e)->map);
                if (out.envmap.tex[0] < 0)
                    return false;
            } else if (ti == typeid(EnvMapCube)) {
                out.envmap.type = PT_ENV_CUBE;
                for (int i = 0; i < 6; i++) {
                    out.envmap.tex[i] = texId(static_cast<const EnvMapCube*>(e)->cubesides[i]);
                    if (out.envmap.tex[i] < 0)
                        return false;
                }
            } else {
                err = std::string("unsupported EnvMap subclass ") + ti.name();
                return false;
            }
        }

        // the acceleration structure, verbatim in topology and order
        const std::vector<BVHNodeLinear>& nodes = scene.bvh.nodes;
        out.bvh_nodes.resize(nodes.size());
        for (size_t i = 0; i < nodes.size(); i++) {
            pt_bvh_node& n = out.bvh_nodes[i];
            for (int k = 0; k < 3; k++) {
                n.lo[k] = nodes[i].aabb.lo[k];
                n.hi[k] = nodes[i].aabb.hi[k];
            }
            if (nodes[i].child2IndexMulTwoPlusOne % 2 == 0) {
                auto it = surfIds.find(nodes[i].surface);
                if (it == surfIds.end()) {
                    err = "BVH leaf that points outside the scene's surfaces";
                    return false;
                }
                n.link = uint64_t(it->second) * 2;
            } else {
                n.link = nodes[i].child2IndexMulTwoPlusOne;
            }
        }
        return true;
    }
};

// Flatten a Scene (after buildBVH, or before it: then the node array is empty).
inline bool flattenScene(const Scene& scene, ptio::SceneData& out, std::string& err,
                         const FlattenOptions& opt = FlattenOptions())
{
    SceneFlattener f(out, opt);
    bool ok = f.run(scene);
    err = f.err;
    return ok;
}

inline bool flattenCamera(const Camera& camera, pt_camera& out, std::string& err,
                          const FlattenOptions& opt = FlattenOptions())
{
    std::memset(&out, 0, sizeof(out));
    out.t = camera.t;
    out.b = camera.b;
    out.r = camera.r;
    out.l = camera.l;
    out.focus_distance = camera.focusDistance;
    out.aperture_radius = camera.apertureRadius;
    out.has_animation = camera.animation != nullptr;
    if (camera.animation)
        return animationToPod(camera.animation, opt, out.animation, err);
    pt_animation ident;
    std::memset(&ident, 0, sizeof(ident));
    ident.T0 = toPod(Transformation());
    ident.T1 = ident.T0;
    ident.duration = 1.0f;
    out.animation = ident;
    return true;
}

#ifndef PTAMD_BINDING_NO_RENDER
// The drop-in: same inputs and same result convention as the loop it replaces
// (img[y * width + x], y = 0 at the bottom, mean radiance).  On failure the
// image is empty and `err` holds the message.
inline std::vector<vec3> render(const Scene& scene, const Camera& camera, int width, int height, int sqrtSpp,
                                float t0, float t1, std::string* err = nullptr, uint64_t seed = 42,
                                int device = 0, const FlattenOptions& opt = FlattenOptions())
{
    std::vector<vec3> img;
    std::string e;
    ptio::SceneData flat;
    pt_camera cam;
    if (!flattenScene(scene, flat, e, opt) || !flattenCamera(camera, cam, e, opt)) {
        if (err)
            *err = e;
        return img;
    }
    pt_scene_desc desc = flat.desc();
    pt_scene* dev = nullptr;
    if (pt_scene_create(&desc, device, &dev) != PT_OK) {
        if (err)
            *err = pt_last_error();
        return img;
    }
    // a scene handed over before scene.buildBVH(t0, t1) gets its tree built on the GPU
    // (the reference's builder is one host thread: seconds per million surfaces)
    if (desc.n_bvh_nodes == 0 && pt_scene_build_bvh_device(dev, t0, t1, nullptr) != PT_OK) {
        if (err)
            *err = pt_last_error();
        pt_scene_destroy(dev);
        return img;
    }
    pt_render_params params;
    std::memset(&params, 0, sizeof(params));
    params.width = width;
    params.height = height;
    params.sqrt_spp = sqrtSpp;
    params.t0 = t0;
    params.t1 = t1;
    params.seed = seed;
    img.resize(size_t(width) * height);
    static_assert(sizeof(vec3) == 3 * sizeof(float));
    if (pt_render(dev, &cam, &params, reinterpret_cast<float*>(img.data())) != PT_OK) {
        if (err)
            *err = pt_last_error();
        img.clear();
    }
    pt_scene_destroy(dev);
    return img;
}

// The same over several GPUs of the node: the scene is uploaded to every device of `devices` and the image is
// split by tiles (bit-identical to one GPU) -- what the reference does with one process per tile and a composer
// program (src/07/pathtracer-tiles.cpp:72-139, tile-composer.cpp:10-37).
inline std::vector<vec3> renderMulti(const Scene& scene, const Camera& camera, int width, int height, int sqrtSpp,
                                     float t0, float t1, const std::vector<int>& devices, std::string* err = nullptr,
                                     uint64_t seed = 42, int mode = PT_SHARD_TILES,
                                     const FlattenOptions& opt = FlattenOptions())
{
    std::vector<vec3> img;
    std::string e;
    ptio::SceneData flat;
    pt_camera cam;
    if (devices.empty() || !flattenScene(scene, flat, e, opt) || !flattenCamera(camera, cam, e, opt)) {
        if (err)
            *err = devices.empty() ? "no devices given" : e;
        return img;
    }
    pt_scene_desc desc = flat.desc();
    std::vector<pt_scene*> handles;
    bool ok = true;
    for (int device : devices) {
        pt_scene* dev = nullptr;
        ok = pt_scene_create(&desc, device, &dev) == PT_OK;
        if (ok)
            handles.push_back(dev);
        ok = ok && (desc.n_bvh_nodes != 0 || pt_scene_build_bvh_device(dev, t0, t1, nullptr) == PT_OK);
        if (!ok)
            break;
    }
    if (ok) {
        pt_render_params params;
        std::memset(&params, 0, sizeof(params));
        params.width = width;
        params.height = height;
        params.sqrt_spp = sqrtSpp;
        params.t0 = t0;
        params.t1 = t1;
        params.seed = seed;
        img.resize(size_t(width) * height);
        ok = pt_render_multi(handles.data(), int(handles.size()), &cam, &params, mode, reinterpret_cast<float*>(img.data()))
            == PT_OK;
    }
    if (!ok) {
        if (err)
            *err = pt_last_error();
        img.clear();
    }
    for (pt_scene* h : handles)
        pt_scene_destroy(h);
    return img;
}

// Drop-in for to8Bit(img) (reference src/15/color.hpp:84-98, used at src/15/pathtracer.cpp:225):
// same bytes in the same (image) row order, computed on the GPU; saveImageAsPPM then writes
// them as before.  Empty on failure.
inline std::vector<unsigned char> to8Bit(const std::vector<vec3>& img, int width, int height, std::string* err = nullptr,
                                         int device = 0)
{
    std::vector<unsigned char> bytes(img.size() * 3);
    static_assert(sizeof(vec3) == 3 * sizeof(float));
    if (img.size() != size_t(width) * size_t(height)
        || pt_image_to_8bit(device, reinterpret_cast<const float*>(img.data()), width, height, 0.0f, 1.0f, 0, bytes.data())
            != PT_OK) {
        if (err)
            *err = img.size() != size_t(width) * size_t(height) ? "image size does not match" : pt_last_error();
        bytes.clear();
    }
    return bytes;
}
#endif

} // namespace ptamd
