// pt_host.cpp -- host-side helpers of libptamd.so that need no GPU:
// mesh tangents, scene files, validation.  (Trees are built on the device,
// pt_scene_build_bvh_device, or handed over by the caller as the reference built them.)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pt_amd.h"
#include "pt_scene_io.hpp"
#include "pt_vec.hpp"
#include "pt_internal.hpp"

using namespace ptv;

namespace {

struct OwnedDesc {
    pt_scene_desc desc; // first member: its address is the address of the record
    ptio::SceneData data;
};

bool texAcyclic(const pt_scene_desc& d, int id, int depth)
{
    if (depth > 16)
        return false;
    const pt_texture& t = d.textures[id];
    if (t.type == PT_TEX_CHECKER)
        return texAcyclic(d, t.tex0, depth + 1) && texAcyclic(d, t.tex1, depth + 1);
    if (t.type == PT_TEX_TRANSFORMER)
        return texAcyclic(d, t.tex0, depth + 1);
    return true;
}

bool matAcyclic(const pt_scene_desc& d, int id, int depth)
{
    if (depth > 8)
        return false;
    const pt_material& m = d.materials[id];
    if (m.type == PT_MAT_TWOSIDED)
        return matAcyclic(d, m.front, depth + 1) && matAcyclic(d, m.back, depth + 1);
    return true;
}

} // namespace

namespace ptint {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}

int caught()
{
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(PT_ERR_LIMIT, "out of host memory");
    } catch (const std::exception& e) {
        return fail(PT_ERR_INVALID_ARGUMENT, std::string("unexpected failure: ") + e.what());
    } catch (...) {
        return fail(PT_ERR_INVALID_ARGUMENT, "unexpected failure");
    }
}

} // namespace ptint

using ptint::fail;

extern "C" {

const char* pt_last_error(void)
{
    return ptint::g_lastError.c_str();
}

const char* pt_version(void)
{
    return "ptamd 0.1 (gfx950)";
}

int pt_scene_desc_validate(const pt_scene_desc* d)
try {
    if (!d)
        return fail(PT_ERR_INVALID_ARGUMENT, "null scene description");
    if (d->abi_version != PT_ABI_VERSION)
        return fail(PT_ERR_INVALID_ARGUMENT, "scene description has a different ABI version");
    auto need = [](uint64_t n, const void* p) { return n == 0 || p != nullptr; };
    if (!need(d->n_vertices, d->positions) || !need(d->n_vertices, d->normals) || !need(d->n_vertices, d->texcoords)
        || !need(d->n_vertices, d->tangents) || !need(d->n_indices, d->indices) || !need(d->n_meshes, d->meshes)
        || !need(d->n_spheres, d->spheres) || !need(d->n_surfaces, d->surfaces) || !need(d->n_lights, d->lights)
        || !need(d->n_materials, d->materials) || !need(d->n_textures, d->textures) || !need(d->n_texels, d->texels)
        || !need(d->n_animations, d->animations) || !need(d->n_bvh_nodes, d->bvh_nodes))
        return fail(PT_ERR_INVALID_ARGUMENT, "scene description has a null array with a non-zero count");
    if (d->n_surfaces >= 0x3FFFFFFFull || d->n_vertices >= 0xFFFFFFFFull || d->n_indices >= 0xFFFFFFFFull)
        return fail(PT_ERR_LIMIT, "scene too large: surface ids have 30 bits, vertex ids 32");

    for (uint64_t i = 0; i < d->n_animations; i++)
        if (d->animations[i].type < PT_ANIM_IDENTITY || d->animations[i].type > PT_ANIM_TWOKEY)
            return fail(PT_ERR_UNSUPPORTED, "animation " + std::to_string(i) + " has an unknown type");
    for (uint64_t i = 0; i < d->n_textures; i++) {
        const pt_texture& t = d->textures[i];
        auto child = [&](int c) { return c >= 0 && uint64_t(c) < d->n_textures; };
        uint64_t floats = 0;
        switch (t.type) {
        case PT_TEX_CONSTANT:
            break;
        case PT_TEX_IMAGE:
            floats = 3ull * uint64_t(std::max(t.width, 0)) * uint64_t(std::max(t.height, 0));
            if (t.width <= 0 || t.height <= 0)
                return fail(PT_ERR_INVALID_ARGUMENT, "image texture " + std::to_string(i) + " has no texels");
            break;
        case PT_TEX_CHECKER:
            if (!child(t.tex0) || !child(t.tex1))
                return fail(PT_ERR_INVALID_ARGUMENT, "checker texture " + std::to_string(i) + " has a bad child");
            break;
        case PT_TEX_TRANSFORMER:
            if (!child(t.tex0))
                return fail(PT_ERR_INVALID_ARGUMENT, "transformer texture " + std::to_string(i) + " has a bad child");
            break;
        case PT_TEX_VALUE_NOISE:
        case PT_TEX_GRADIENT_NOISE:
            if (t.width <= 0 || t.height <= 0)
                return fail(PT_ERR_INVALID_ARGUMENT, "noise texture " + std::to_string(i) + " has an empty lattice");
            floats = (t.type == PT_TEX_VALUE_NOISE ? 1ull : 2ull) * uint64_t(t.width) * uint64_t(t.height);
            break;
        case PT_TEX_WORLEY_NOISE:
            if (t.width < 0)
                return fail(PT_ERR_INVALID_ARGUMENT, "worley texture " + std::to_string(i) + " has a negative count");
            floats = 2ull * uint64_t(t.width);
            break;
        default:
            return fail(PT_ERR_UNSUPPORTED, "texture " + std::to_string(i) + " has an unknown type");
        }
        if (floats > 0 && (t.data_offset > d->n_texels || floats > d->n_texels - t.data_offset))
            return fail(PT_ERR_INVALID_ARGUMENT, "texture " + std::to_string(i) + " reads past the texel heap");
    }
    for (uint64_t i = 0; i < d->n_textures; i++)
        if (!texAcyclic(*d, int(i), 0))
            return fail(PT_ERR_LIMIT, "texture " + std::to_string(i) + " nests deeper than 16 levels (or is cyclic)");
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const pt_material& m = d->materials[i];
        auto tex = [&](int c, bool optional) { return (optional && c == -1) || (c >= 0 && uint64_t(c) < d->n_textures); };
        bool ok = true;
        switch (m.type) {
        case PT_MAT_BASE:
        case PT_MAT_LIGHT:
        case PT_MAT_GLASS:
            break;
        case PT_MAT_LAMBERTIAN:
        case PT_MAT_MIRROR:
            ok = tex(m.tex[0], false);
            break;
        case PT_MAT_PHONG:
            ok = tex(m.tex[0], false) && tex(m.tex[1], false) && tex(m.tex[2], false) && tex(m.tex[3], true)
                && tex(m.tex[4], true);
            break;
        case PT_MAT_TWOSIDED:
            ok = m.front >= 0 && uint64_t(m.front) < d->n_materials && m.back >= 0 && uint64_t(m.back) < d->n_materials;
            break;
        default:
            return fail(PT_ERR_UNSUPPORTED, "material " + std::to_string(i) + " has an unknown type");
        }
        if (!ok)
            return fail(PT_ERR_INVALID_ARGUMENT, "material " + std::to_string(i) + " refers to a missing texture or material");
    }
    for (uint64_t i = 0; i < d->n_materials; i++)
        if (!matAcyclic(*d, int(i), 0))
            return fail(PT_ERR_LIMIT, "material " + std::to_string(i) + " nests deeper than 8 levels (or is cyclic)");
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        const pt_mesh& m = d->meshes[i];
        if (uint64_t(m.first_vertex) + m.n_vertices > d->n_vertices
            || uint64_t(m.first_index) + 3ull * m.n_triangles > d->n_indices)
            return fail(PT_ERR_INVALID_ARGUMENT, "mesh " + std::to_string(i) + " reads past the vertex or index arrays");
        if (m.material < 0 || uint64_t(m.material) >= d->n_materials)
            return fail(PT_ERR_INVALID_ARGUMENT, "mesh " + std::to_string(i) + " has a bad material");
        if (m.animation < -1 || (m.animation >= 0 && uint64_t(m.animation) >= d->n_animations))
            return fail(PT_ERR_INVALID_ARGUMENT, "mesh " + std::to_string(i) + " has a bad animation");
        const uint32_t* idx = d->indices + m.first_index;
        for (uint64_t k = 0; k < 3ull * m.n_triangles; k++)
            if (idx[k] >= m.n_vertices)
                return fail(PT_ERR_INVALID_ARGUMENT, "mesh " + std::to_string(i) + " has a vertex index out of range");
    }
    for (uint64_t i = 0; i < d->n_spheres; i++) {
        const pt_sphere& q = d->spheres[i];
        if (q.material < 0 || uint64_t(q.material) >= d->n_materials)
            return fail(PT_ERR_INVALID_ARGUMENT, "sphere " + std::to_string(i) + " has a bad material");
        if (q.animation < -1 || (q.animation >= 0 && uint64_t(q.animation) >= d->n_animations))
            return fail(PT_ERR_INVALID_ARGUMENT, "sphere " + std::to_string(i) + " has a bad animation");
    }
    for (uint64_t i = 0; i < d->n_surfaces; i++) {
        const pt_surface& s = d->surfaces[i];
        if (s.type == PT_SURF_TRIANGLE) {
            if (s.owner >= d->n_meshes || s.index >= d->meshes[s.owner].n_triangles)
                return fail(PT_ERR_INVALID_ARGUMENT, "surface " + std::to_string(i) + " names a missing triangle");
        } else if (s.type == PT_SURF_SPHERE) {
            if (s.owner >= d->n_spheres)
                return fail(PT_ERR_INVALID_ARGUMENT, "surface " + std::to_string(i) + " names a missing sphere");
        } else {
            return fail(PT_ERR_UNSUPPORTED, "surface " + std::to_string(i) + " has an unknown type");
        }
    }
    for (uint64_t i = 0; i < d->n_lights; i++)
        if (d->lights[i] >= d->n_surfaces)
            return fail(PT_ERR_INVALID_ARGUMENT, "light " + std::to_string(i) + " is not a surface");
    {
        const pt_envmap& e = d->envmap;
        int ntex = e.type == PT_ENV_EQUIRECT ? 1 : (e.type == PT_ENV_CUBE ? 6 : 0);
        if (e.type < PT_ENV_NONE || e.type > PT_ENV_CUBE)
            return fail(PT_ERR_UNSUPPORTED, "environment map has an unknown type");
        for (int k = 0; k < ntex; k++)
            if (e.tex[k] < 0 || uint64_t(e.tex[k]) >= d->n_textures)
                return fail(PT_ERR_INVALID_ARGUMENT, "environment map refers to a missing texture");
    }
    if (d->n_bvh_nodes > 0) {
        if (d->n_surfaces == 0 || d->n_bvh_nodes != 2 * d->n_surfaces - 1)
            return fail(PT_ERR_INVALID_ARGUMENT, "node count is not 2 * surfaces - 1");
        // every inner node's children lie after it and inside the array;
        // every surface appears in exactly one leaf
        std::vector<uint8_t> seen(d->n_surfaces, 0);
        for (uint64_t i = 0; i < d->n_bvh_nodes; i++) {
            uint64_t link = d->bvh_nodes[i].link;
            if (link % 2 == 0) {
                if (link / 2 >= d->n_surfaces || seen[link / 2])
                    return fail(PT_ERR_INVALID_ARGUMENT, "node " + std::to_string(i) + " is a bad or duplicate leaf");
                seen[link / 2] = 1;
            } else {
                uint64_t c2 = link / 2;
                if (c2 <= i + 1 || c2 >= d->n_bvh_nodes)
                    return fail(PT_ERR_INVALID_ARGUMENT, "node " + std::to_string(i) + " has a bad second child");
            }
        }
    }
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_mesh_compute_tangents(const float* positions, const float* normals, const float* texcoords,
                             uint32_t n_vertices, const uint32_t* indices, uint32_t n_triangles,
                             float* out_tangents)
try {
    // mesh.hpp:10-55
    if (!positions || !normals || !texcoords || !indices || !out_tangents)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<V3> tangents(n_vertices, v3(0.0f));
    for (uint32_t t = 0; t < n_triangles; t++) {
        uint32_t i0 = indices[3 * t + 0], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
        if (i0 >= n_vertices || i1 >= n_vertices || i2 >= n_vertices)
            return fail(PT_ERR_INVALID_ARGUMENT, "vertex index out of range");
        V3 P0 = v3(positions + 3 * i0), P1 = v3(positions + 3 * i1), P2 = v3(positions + 3 * i2);
        const float* tc0 = texcoords + 2 * i0;
        const float* tc1 = texcoords + 2 * i1;
        const float* tc2 = texcoords + 2 * i2;
        V3 e1 = P1 - P0;
        V3 e2 = P2 - P0;
        float s1 = tc1[0] - tc0[0];
        float t1 = tc1[1] - tc0[1];
        float s2 = tc2[0] - tc0[0];
        float t2 = tc2[1] - tc0[1];
        float det = (s1 * t2 - s2 * t1);
        if (std::abs(det) > kEps) {
            V3 tp = 1.0f / det * (t2 * e1 - t1 * e2);
            tangents[i0] = tangents[i0] + tp;
            tangents[i1] = tangents[i1] + tp;
            tangents[i2] = tangents[i2] + tp;
        }
    }
    for (uint32_t i = 0; i < n_vertices; i++) {
        V3 n = v3(normals + 3 * i);
        V3 tp = tangents[i];
        V3 t = v3(1.0f, 0.0f, 0.0f);
        if (dot(tp, tp) > 0.0f)
            t = normalize(tp - dot(n, tp) * n);
        out_tangents[3 * i + 0] = t.x;
        out_tangents[3 * i + 1] = t.y;
        out_tangents[3 * i + 2] = t.z;
    }
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_scene_desc_load(const char* path, pt_scene_desc** out_desc)
try {
    if (!path || !out_desc)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    OwnedDesc* od = new OwnedDesc;
    if (!ptio::load(path, od->data)) {
        delete od;
        return fail(PT_ERR_IO, std::string("cannot read scene file ") + path);
    }
    od->desc = od->data.desc();
    int rc = pt_scene_desc_validate(&od->desc);
    if (rc != PT_OK) {
        delete od;
        return rc;
    }
    *out_desc = &od->desc;
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_scene_desc_save(const pt_scene_desc* desc, const char* path)
try {
    if (!desc || !path)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    int rc = pt_scene_desc_validate(desc);
    if (rc != PT_OK)
        return rc;
    if (!ptio::save(*desc, path))
        return fail(PT_ERR_IO, std::string("cannot write scene file ") + path);
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

void pt_scene_desc_free(pt_scene_desc* desc)
{
    if (desc)
        delete reinterpret_cast<OwnedDesc*>(desc);
}

} // extern "C"
