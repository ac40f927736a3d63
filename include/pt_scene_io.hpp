// pt_scene_io.hpp -- header-only reader/writer for ".ptscene" files and an
// owning container for a flat scene.  Pure data plumbing: no rendering logic.
//
// File layout (little endian): 8-byte magic "PTSCENE1", then for each array of
// pt_scene_desc, in declaration order, a uint64 element count followed by the
// raw elements; finally the pt_envmap record.
#pragma once

#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "pt_amd.h"

namespace ptio {

// Owns the arrays a pt_scene_desc points to.
struct SceneData {
    std::vector<float> positions, normals, texcoords, tangents;
    std::vector<uint32_t> indices;
    std::vector<pt_mesh> meshes;
    std::vector<pt_sphere> spheres;
    std::vector<pt_surface> surfaces;
    std::vector<uint32_t> lights;
    std::vector<pt_material> materials;
    std::vector<pt_texture> textures;
    std::vector<float> texels;
    std::vector<pt_animation> animations;
    std::vector<pt_bvh_node> bvh_nodes;
    pt_envmap envmap;

    SceneData() { std::memset(&envmap, 0, sizeof(envmap)); for (int i = 0; i < 6; i++) envmap.tex[i] = -1; }

    size_t n_vertices() const { return positions.size() / 3; }

    // Keep the vertex streams the same length so one first_vertex indexes all.
    void pad_vertex_streams()
    {
        size_t n = n_vertices();
        normals.resize(3 * n, 0.0f);
        texcoords.resize(2 * n, 0.0f);
        tangents.resize(3 * n, 0.0f);
    }

    pt_scene_desc desc() const
    {
        pt_scene_desc d;
        std::memset(&d, 0, sizeof(d));
        d.abi_version = PT_ABI_VERSION;
        d.n_vertices = n_vertices();
        d.positions = positions.data();
        d.normals = normals.data();
        d.texcoords = texcoords.data();
        d.tangents = tangents.data();
        d.n_indices = indices.size();
        d.indices = indices.data();
        d.n_meshes = meshes.size();
        d.meshes = meshes.data();
        d.n_spheres = spheres.size();
        d.spheres = spheres.data();
        d.n_surfaces = surfaces.size();
        d.surfaces = surfaces.data();
        d.n_lights = lights.size();
        d.lights = lights.data();
        d.n_materials = materials.size();
        d.materials = materials.data();
        d.n_textures = textures.size();
        d.textures = textures.data();
        d.n_texels = texels.size();
        d.texels = texels.data();
        d.n_animations = animations.size();
        d.animations = animations.data();
        d.n_bvh_nodes = bvh_nodes.size();
        d.bvh_nodes = bvh_nodes.data();
        d.envmap = envmap;
        return d;
    }

    void assign(const pt_scene_desc& d)
    {
        positions.assign(d.positions, d.positions + 3 * d.n_vertices);
        normals.assign(d.normals, d.normals + 3 * d.n_vertices);
        texcoords.assign(d.texcoords, d.texcoords + 2 * d.n_vertices);
        tangents.assign(d.tangents, d.tangents + 3 * d.n_vertices);
        indices.assign(d.indices, d.indices + d.n_indices);
        meshes.assign(d.meshes, d.meshes + d.n_meshes);
        spheres.assign(d.spheres, d.spheres + d.n_spheres);
        surfaces.assign(d.surfaces, d.surfaces + d.n_surfaces);
        lights.assign(d.lights, d.lights + d.n_lights);
        materials.assign(d.materials, d.materials + d.n_materials);
        textures.assign(d.textures, d.textures + d.n_textures);
        texels.assign(d.texels, d.texels + d.n_texels);
        animations.assign(d.animations, d.animations + d.n_animations);
        bvh_nodes.assign(d.bvh_nodes, d.bvh_nodes + d.n_bvh_nodes);
        envmap = d.envmap;
    }
};

namespace detail {

template <typename T>
inline bool put(FILE* f, const T* data, uint64_t count)
{
    if (std::fwrite(&count, sizeof(count), 1, f) != 1)
        return false;
    return count == 0 || std::fwrite(data, sizeof(T), count, f) == count;
}

template <typename T>
inline bool get(FILE* f, std::vector<T>& v)
{
    uint64_t count = 0;
    if (std::fread(&count, sizeof(count), 1, f) != 1)
        return false;
    // the elements must be in the file: a truncated or hostile count is refused before any memory is taken
    const long here = std::ftell(f);
    if (here < 0 || std::fseek(f, 0, SEEK_END) != 0)
        return false;
    const long end = std::ftell(f);
    if (end < here || std::fseek(f, here, SEEK_SET) != 0)
        return false;
    if (count > uint64_t(end - here) / sizeof(T))
        return false;
    try {
        v.resize(count);
    } catch (const std::exception&) {
        return false;
    }
    return count == 0 || std::fread(v.data(), sizeof(T), count, f) == count;
}

} // namespace detail

static const char kMagic[8] = { 'P', 'T', 'S', 'C', 'E', 'N', 'E', '1' };

inline bool save(const pt_scene_desc& d, const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f)
        return false;
    bool ok = std::fwrite(kMagic, 1, 8, f) == 8;
    ok = ok && detail::put(f, d.positions, 3 * d.n_vertices);
    ok = ok && detail::put(f, d.normals, 3 * d.n_vertices);
    ok = ok && detail::put(f, d.texcoords, 2 * d.n_vertices);
    ok = ok && detail::put(f, d.tangents, 3 * d.n_vertices);
    ok = ok && detail::put(f, d.indices, d.n_indices);
    ok = ok && detail::put(f, d.meshes, d.n_meshes);
    ok = ok && detail::put(f, d.spheres, d.n_spheres);
    ok = ok && detail::put(f, d.surfaces, d.n_surfaces);
    ok = ok && detail::put(f, d.lights, d.n_lights);
    ok = ok && detail::put(f, d.materials, d.n_materials);
    ok = ok && detail::put(f, d.textures, d.n_textures);
    ok = ok && detail::put(f, d.texels, d.n_texels);
    ok = ok && detail::put(f, d.animations, d.n_animations);
    ok = ok && detail::put(f, d.bvh_nodes, d.n_bvh_nodes);
    ok = ok && std::fwrite(&d.envmap, sizeof(d.envmap), 1, f) == 1;
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

inline bool load(const std::string& path, SceneData& s)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
        return false;
    char magic[8];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kMagic, 8) == 0;
    ok = ok && detail::get(f, s.positions);
    ok = ok && detail::get(f, s.normals);
    ok = ok && detail::get(f, s.texcoords);
    ok = ok && detail::get(f, s.tangents);
    ok = ok && detail::get(f, s.indices);
    ok = ok && detail::get(f, s.meshes);
    ok = ok && detail::get(f, s.spheres);
    ok = ok && detail::get(f, s.surfaces);
    ok = ok && detail::get(f, s.lights);
    ok = ok && detail::get(f, s.materials);
    ok = ok && detail::get(f, s.textures);
    ok = ok && detail::get(f, s.texels);
    ok = ok && detail::get(f, s.animations);
    ok = ok && detail::get(f, s.bvh_nodes);
    ok = ok && std::fread(&s.envmap, sizeof(s.envmap), 1, f) == 1;
    std::fclose(f);
    ok = ok && s.normals.size() == s.positions.size() && s.tangents.size() == s.positions.size()
        && s.texcoords.size() / 2 == s.positions.size() / 3;
    return ok;
}

} // namespace ptio
