// pt_device.hpp -- device functions of the render loop: traversal,
// intersection, hit-record reconstruction, textures, materials, environment
// maps, light sampling and the per-path integrator.
//
// Arithmetic follows the reference operation for operation (citations are
// paths in the reference repository), so that with contraction off and the
// portable transcendentals of pt_math.hpp every path computes the same bits
// as a CPU evaluation of the same sample.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_amd.h"
#include "pt_rng.h"
#include "pt_vec.hpp"

namespace ptd {

using namespace ptv;

#define PT_DEV __device__ __forceinline__

constexpr uint32_t kInner = 0xFFFFFFFFu; // "no surface": a node step that did not end on a hit leaf
constexpr uint32_t kInnerBit = 0x80000000u; // DevNode.leaf of an inner node: kInnerBit | index of its second child
constexpr uint32_t kLeafRowBit = 0x40000000u; // in a link of the `wide` table, see DevScene
constexpr uint32_t kQuadNoLink = 0x7FFFFFFFu; // link of an unused slot of a `quad` row (as a leaf link this would be
                                              // node 2^30 - 1: the builders stop below)
constexpr uint32_t kMeshMask = 0x00FFFFFFu;  // surface row 1, word 3: mesh number | material type << 24

// surface kinds (SurfRec word 3 of row 0)
enum : uint32_t {
    kTriStatic = 0,  // vertices stored in world space (posed on the host if the mesh has a constant pose)
    kTriDynamic = 1, // vertices stored unposed; pose depends on ray time
    kSphStatic = 2,  // centre / radius stored in world space
    kSphDynamic = 3
};

struct DevMesh {
    uint32_t first_vertex;
    uint32_t first_index;
    int32_t material;
    int32_t animation; // -1 = none
    uint32_t flags;    // 1 normals, 2 texcoords, 4 tangents
    uint32_t pad[3];
};

// Scene tables in HBM.  nodes: 2 x float4 per node = (lo.xyz, hi.x), (hi.y, hi.z,
// bits escape, bits leaf): `escape` is the node to continue with when this
// node's subtree is skipped or finished (n_nodes = done); `leaf` is the
// surface id, or kInnerBit | second child for an inner node.  Inner nodes have
// their first child at index + 1.
// surf: 3 x float4 per surface; triangles (A.xyz, bits kind), (B.xyz, bits
// mesh | material type << 24), (C.xyz, bits triangle-in-mesh); spheres
// (centre.xyz, bits kind), (radius, bits sphere, bits animation number, bits material type << 24), unused.  surfAttr and matConst repeat, per
// surface and per material, what shading would otherwise reach through two or
// three dependent table reads (mesh -> index list -> vertices; material ->
// texture -> value).
struct DevScene {
    const float4* nodes;
    const uint4* wide;       // per inner node (numbered in storage order, inner nodes only) 2 x uint4: the boxes of
                             // both children on a 16-bit grid over the root box, rounded outwards (6 words:
                             // lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16 for the first child, the
                             // same for the second), then the links of the two children: kInnerBit | inner
                             // number, or a static triangle's surface id (its box is that of its vertices), or
                             // kLeafRowBit | node number (box and surface are in that row of `nodes`).  Null
                             // if the boxes do not fit the grid (non-finite coordinates).
    const uint4* quad;       // the four-wide form of `wide`, or null: per row 4 x uint4 = the boxes of up to four
                             // descendants of an inner node (3 words each, as above), then their four links;
                             // kInnerBit | row number for an inner one (rows come after the rows that lead to
                             // them, row 0 stands for the root), kQuadNoLink in a slot not used
    float qbase[3], qstep[3]; // grid coordinate q of axis a stands for fmaf(q, qstep[a], qbase[a])
    const float4* surf;
    const uint4* surfAttr;   // per surface: triangles (vertex 0, 1, 2 in the vertex streams, material | flags << 24);
                             // spheres (sphere index, 0, 0, material | flags << 24).  flags: 1 normals,
                             // 2 texcoords, 4 tangents, 8 mesh has a pose, 16 the material reads texture
                             // coordinates, 32 the material reads the tangent
    const float4* matConst;  // per material 2 rows: (value of tex[0] if constant, bits mask), (value of tex[1]
                             // if constant, x of tex[2] if constant); mask bit k = tex[k] is a constant texture
    const float* positions;
    const float* normals;
    const float* texcoords;
    const float* tangents;
    const uint32_t* indices;
    const DevMesh* meshes;
    const pt_sphere* spheres;
    const uint32_t* lights;
    // Per surface, for a triangle with a fixed pose, no vertex normals and, if its mesh is posed, no tangents (flag
    // 64 in surfAttr): the unit normal of its plane, normalize(cross(B - A, C - A)) as the hit record and the light
    // pdf compute it, and half the length of that cross product (its area) -- worked out once at upload with the same
    // operations.  lightGeo repeats the row of light i (w < 0: not such a triangle).
    const float4* faceN;
    const float4* lightGeo;
    // Scenes whose tables are traced from LDS: per node the two link words of that walk (see packNodeLinks), made
    // once per tree, and in entry n_nodes the node the walk starts at.  The table may leave inner nodes out of the
    // walk: an inner node's box only decides whether its subtree is looked at, and a ray that misses it misses every
    // box inside it, so the walk tests the same surfaces in the same order without it.  Null: links as packNodeLinks
    // makes them, start below the root.
    const uint2* walkLinks;
    const pt_material* materials;
    const pt_texture* textures;
    const float* texels;
    const pt_animation* animations;
    uint32_t n_nodes;
    uint32_t n_surfaces;
    uint32_t n_lights;
    uint32_t n_animations;
    pt_envmap envmap;
};

struct Counters {
    uint32_t rays, nodeVisits, leafTests, lightTests, vertices, draws;
};

template <bool COUNT>
struct Rng {
    uint64_t key;
    uint32_t n;
    Counters* cnt;
    PT_DEV float in01()
    {
        float r = pt_rng_in01(key, n);
        n++;
        if (COUNT)
            cnt->draws++;
        return r;
    }
};

PT_DEV uint32_t fbits(float f) { return __float_as_uint(f); }

struct PM {
    PT_DEV static float sin(float x) { return ptm::sinf(x); }
    PT_DEV static float cos(float x) { return ptm::cosf(x); }
    PT_DEV static float acos(float x) { return ptm::acosf(x); }
};

// ---------------------------------------------------------------- sampler --
// reference src/15/sampler.hpp:12-80
PT_DEV V3 uniformOnSphere(float u0, float u1)
{
    float z = 1.0f - 2.0f * u0;
    float r = ::sqrtf(max_(0.0f, 1.0f - z * z));
    float phi = 2.0f * kPi * u1;
    float s, c;
    ptm::sincosf(phi, s, c);
    return v3(r * c, r * s, z);
}
PT_DEV V2 uniformInDisk(float u0, float u1)
{
    float ox = 2.0f * u0 - 1.0f;
    float oy = 2.0f * u1 - 1.0f;
    float theta, r;
    if (fabs_(ox) > fabs_(oy)) {
        r = ox;
        theta = kPi / 4.0f * (oy / ox);
    } else {
        r = oy;
        theta = 0.5f * kPi - kPi / 4.0f * (ox / oy);
    }
    float s, c;
    ptm::sincosf(theta, s, c);
    return V2{ c * r, s * r };
}
// uniformInDisk(u0, u1) * 0: what is left of the lens sample of a pinhole camera (aperture radius 0) is the sign
// of each zero, or NaN where the disk sample is 0/0 -- decided without evaluating sine and cosine: in the first
// branch |theta| <= pi/4, so the cosine is positive and the sine has theta's sign (sin(-0) = -0); in the second
// theta lies in [pi/4, 3 pi/4], the sine is positive and the cosine is positive exactly for floats below
// float(pi/2), which lies above pi/2 while its predecessor lies below.
PT_DEV V2 uniformInDiskTimesZero(float u0, float u1, float zero)
{
    float ox = 2.0f * u0 - 1.0f;
    float oy = 2.0f * u1 - 1.0f;
    float theta, r, c, s;
    if (fabs_(ox) > fabs_(oy)) {
        r = ox;
        theta = kPi / 4.0f * (oy / ox);
        c = 1.0f;
        s = __builtin_copysignf(1.0f, theta);
    } else {
        r = oy;
        theta = 0.5f * kPi - kPi / 4.0f * (ox / oy);
        c = theta < 0.5f * kPi ? 1.0f : -1.0f;
        s = 1.0f;
    }
    if (theta != theta)
        c = s = theta;
    r = __builtin_copysignf(1.0f, r);
    return V2{ c * r * zero, s * r * zero };
}
PT_DEV V3 cosineWeightedOnHemisphere(float u0, float u1)
{
    V2 d = uniformInDisk(u0, u1);
    float disc = 1.0f - dot(d, d);
    float z = ::sqrtf(max_(0.0f, disc));
    return v3(d.x, d.y, z);
}
PT_DEV V3 phongWeightedOnHemisphere(float shininess, float u0, float u1)
{
    float cosT = ptm::powf(1.0f - u0, 1.0f / (1.0f + shininess));
    float disc = 1.0f - cosT * cosT;
    float sinT = ::sqrtf(max_(0.0f, disc));
    float phi = 2.0f * kPi * u1;
    float s, c;
    ptm::sincosf(phi, s, c);
    return v3(c * sinT, s * sinT, cosT);
}
PT_DEV V3 uniformTowardsSphere(float cosThetaMax, float u0, float u1)
{
    float cosT = (1.0f - u0) + u0 * cosThetaMax;
    float sinT = ::sqrtf(max_(0.0f, 1.0f - cosT * cosT));
    float phi = 2.0f * kPi * u1;
    float s, c;
    ptm::sincosf(phi, s, c);
    return v3(c * sinT, s * sinT, cosT);
}
PT_DEV V3 uniformInTriangle(float u0, float u1)
{
    float su0 = ::sqrtf(u0);
    float b0 = 1.0f - su0;
    float b1 = u1 * su0;
    return v3(b0, b1, 1.0f - b0 - b1);
}

// reference src/15/tangentspace.hpp:6-47
struct TangentSpace {
    V3 normal, tangent, bitangent;
};
PT_DEV TangentSpace tangentSpace(V3 n, V3 t) { return TangentSpace{ n, t, cross(n, t) }; }
PT_DEV TangentSpace tangentSpace(V3 n)
{
    V3 w;
    if (fabs_(n.x) > fabs_(n.z) && fabs_(n.y) > fabs_(n.z))
        w = v3(-n.y, n.x, 0.0f);
    else if (fabs_(n.y) > fabs_(n.x))
        w = v3(0.0f, -n.z, n.y);
    else
        w = v3(-n.z, 0.0f, n.x);
    TangentSpace ts;
    ts.normal = n;
    ts.tangent = normalize(w);
    ts.bitangent = cross(ts.normal, ts.tangent);
    return ts;
}
PT_DEV V3 toWorldSpace(const TangentSpace& ts, V3 v)
{
    return v3(v.x * ts.tangent.x + v.y * ts.bitangent.x + v.z * ts.normal.x,
              v.x * ts.tangent.y + v.y * ts.bitangent.y + v.z * ts.normal.y,
              v.x * ts.tangent.z + v.y * ts.bitangent.z + v.z * ts.normal.z);
}

// reference src/15/fresnel.hpp:6-18
PT_DEV float fresnelUnpolarized(float cosI, float cosT, float n1, float n2)
{
    float Fs = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
    Fs *= Fs;
    float Fp = (n1 * cosT - n2 * cosI) / (n1 * cosT + n2 * cosI);
    Fp *= Fp;
    return 0.5f * (Fs + Fp);
}

// --------------------------------------------------------------- textures --
// reference src/15/texture_*.hpp; the Checker / Transformer combinators are
// unrolled into a loop (nesting depth is bounded at upload).
PT_DEV V3 texel3(const DevScene& sc, const pt_texture& t, int x, int y)
{
    const float* p = sc.texels + t.data_offset + 3 * (size_t(y) * t.width + x);
    return v3(p[0], p[1], p[2]);
}
PT_DEV int wrapIndex(int i, int n)
{
    int r = i % n;
    return r < 0 ? r + n : r;
}
__device__ inline V3 textureValue(const DevScene& sc, int id, V2 tc)
{
    for (int depth = 0; depth < 20; depth++) {
        const pt_texture& t = sc.textures[id];
        switch (t.type) {
        case PT_TEX_CONSTANT:
            return v3(t.v[0], t.v[1], t.v[2]);
        case PT_TEX_IMAGE: { // texture_image.hpp:45-71
            float u = fractf(tc.x);
            float v = fractf(tc.y);
            float s = u * t.width - 0.5f;
            if (s < 0.0f)
                s = 0.0f;
            float tt = v * t.height - 0.5f;
            if (tt < 0.0f)
                tt = 0.0f;
            int x0 = int(s);
            int y0 = int(tt);
            int x1 = x0 + 1;
            if (x1 >= t.width)
                x1 = t.width - 1;
            int y1 = y0 + 1;
            if (y1 >= t.height)
                y1 = t.height - 1;
            // a NaN coordinate converts to an arbitrary integer: keep every read inside the image
            if (x0 >= t.width)
                x0 = t.width - 1;
            if (y0 >= t.height)
                y0 = t.height - 1;
            if (x0 < 0)
                x0 = 0;
            if (y0 < 0)
                y0 = 0;
            float alpha = s - x0;
            float beta = tt - y0;
            V3 a = mix3(texel3(sc, t, x0, y0), texel3(sc, t, x1, y0), alpha);
            V3 b = mix3(texel3(sc, t, x0, y1), texel3(sc, t, x1, y1), alpha);
            return mix3(a, b, beta);
        }
        case PT_TEX_CHECKER: { // texture_checker.hpp:16-26
            int col = int(tc.x * t.width);
            int row = int(tc.y * t.height);
            id = (row % 2 == col % 2) ? t.tex0 : t.tex1;
            break;
        }
        case PT_TEX_TRANSFORMER: // texture_transformer.hpp:16-19
            tc = V2{ t.v[0] * tc.x + t.v[2], t.v[1] * tc.y + t.v[3] };
            id = t.tex0;
            break;
        case PT_TEX_VALUE_NOISE: { // texture_value_noise.hpp:20-37
            const float* vals = sc.texels + t.data_offset;
            float ix = ::floorf(t.width * tc.x);
            float iy = ::floorf(t.height * tc.y);
            float fx = fractf(t.width * tc.x);
            float fy = fractf(t.height * tc.y);
            int x0 = wrapIndex(int(ix + 0), t.width), x1 = wrapIndex(int(ix + 1), t.width);
            int y0 = wrapIndex(int(iy + 0), t.height), y1 = wrapIndex(int(iy + 1), t.height);
            float a = mixf(vals[size_t(y0) * t.width + x0], vals[size_t(y0) * t.width + x1], fx);
            float b = mixf(vals[size_t(y1) * t.width + x0], vals[size_t(y1) * t.width + x1], fx);
            return v3(mixf(a, b, fy));
        }
        case PT_TEX_GRADIENT_NOISE: { // texture_gradient_noise.hpp:21-42
            const float* vals = sc.texels + t.data_offset;
            float ix = ::floorf(t.width * tc.x);
            float iy = ::floorf(t.height * tc.y);
            float fx = fractf(t.width * tc.x);
            float fy = fractf(t.height * tc.y);
            float sx = fx * fx * (3.0f - 2.0f * fx);
            float sy = fy * fy * (3.0f - 2.0f * fy);
            int x0 = wrapIndex(int(ix + 0), t.width), x1 = wrapIndex(int(ix + 1), t.width);
            int y0 = wrapIndex(int(iy + 0), t.height), y1 = wrapIndex(int(iy + 1), t.height);
            const float* g00 = vals + 2 * (size_t(y0) * t.width + x0);
            const float* g10 = vals + 2 * (size_t(y0) * t.width + x1);
            const float* g01 = vals + 2 * (size_t(y1) * t.width + x0);
            const float* g11 = vals + 2 * (size_t(y1) * t.width + x1);
            float a = mixf(dot(V2{ g00[0], g00[1] }, V2{ fx - 0.0f, fy - 0.0f }),
                           dot(V2{ g10[0], g10[1] }, V2{ fx - 1.0f, fy - 0.0f }), sx);
            float b = mixf(dot(V2{ g01[0], g01[1] }, V2{ fx - 0.0f, fy - 1.0f }),
                           dot(V2{ g11[0], g11[1] }, V2{ fx - 1.0f, fy - 1.0f }), sx);
            return v3(mixf(a, b, sy));
        }
        case PT_TEX_WORLEY_NOISE: { // texture_worley_noise.hpp:19-48
            const float* pts = sc.texels + t.data_offset;
            V2 uv{ fractf(tc.x), fractf(tc.y) };
            float d1 = kFltMax, d2 = kFltMax, d3 = kFltMax;
            for (int i = 0; i < t.width; i++) {
                float dist = kFltMax;
                for (int r = -1; r <= +1; r++)
                    for (int c = -1; c <= +1; c++) {
                        V2 q{ pts[2 * i] + float(c), pts[2 * i + 1] + float(r) };
                        dist = min_(dist, length(V2{ uv.x - q.x, uv.y - q.y }));
                    }
                if (dist < d1) {
                    d3 = d2;
                    d2 = d1;
                    d1 = dist;
                } else if (dist < d2) {
                    d3 = d2;
                    d2 = dist;
                } else if (dist < d3) {
                    d3 = dist;
                }
            }
            return v3(d1, d2, d3);
        }
        default:
            return v3(0.0f);
        }
    }
    return v3(0.0f);
}

// reference src/15/envmap_equirect.hpp:15-22, envmap_cube.hpp:19-40, envmap.hpp:8-11
__device__ inline V3 envmapValue(const DevScene& sc, V3 dir)
{
    if (sc.envmap.type == PT_ENV_EQUIRECT) {
        float theta = ptm::asinf(min_(1.0f, max_(-1.0f, dir.y)));
        float phi = ptm::atan2f(-dir.x, dir.z);
        float u = phi / (2.0f * kPi);
        float v = theta / kPi + 0.5f;
        return textureValue(sc, sc.envmap.tex[0], V2{ u, v });
    }
    if (sc.envmap.type == PT_ENV_CUBE) {
        float ax = fabs_(dir.x), ay = fabs_(dir.y), az = fabs_(dir.z);
        int side;
        float u, v;
        if (ax > ay && ax > az) {
            u = 0.5f * (dir.z / -dir.x + 1.0f);
            v = 0.5f * (dir.y / ax + 1.0f);
            side = 0 + (fbits(dir.x) >> 31);
        } else if (ay > az) {
            u = 0.5f * (dir.x / ay + 1.0f);
            v = 0.5f * (dir.z / -dir.y + 1.0f);
            side = 2 + (fbits(dir.y) >> 31);
        } else {
            u = 0.5f * (dir.x / dir.z + 1.0f);
            v = 0.5f * (dir.y / az + 1.0f);
            side = 4 + (fbits(dir.z) >> 31);
        }
        return textureValue(sc, sc.envmap.tex[side], V2{ u, v });
    }
    return v3(0.0f);
}

// ------------------------------------------------------------ intersection --

// What identifies a hit; enough to rebuild the full record afterwards.
struct HitLite {
    uint32_t sid; // PT_NO_HIT = miss
    float a;
    float u, v; // triangle barycentrics
    bool backside;
    uint32_t matClass; // type of the surface's material (from the surface row), for ordering the shading work
};

// AABB::hit, reference src/15/aabb.hpp:41-60, also giving the distance at which the box is
// entered.  The reference takes the axes in turn: per axis the near and the far plane by the
// sign of the inverse direction, reject if the entry lies beyond the smallest exit so far (amax
// to begin with) or the exit before the largest entry so far (amin to begin with), else tighten
// both; a NaN slab (0 * inf) compares false, so it neither rejects nor tightens.  Every entry
// gets compared with every exit of an earlier axis and the other way round, and an axis's own
// entry never exceeds its own exit (lo <= hi and rounding is monotone), so the verdict over the
// three axes is: largest non-NaN entry > smallest non-NaN exit.  fmaxf/fminf return the other
// operand when one is NaN, which is the reference's "not tightened".
PT_DEV bool aabbEntry(V3 lo, V3 hi, const Ray& ray, float amin, float amax, float& entry)
{
    const V3 inv = ray.invDirection, org = ray.origin;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const float inX = ((nx ? hi.x : lo.x) - org.x) * inv.x, outX = ((nx ? lo.x : hi.x) - org.x) * inv.x;
    const float inY = ((ny ? hi.y : lo.y) - org.y) * inv.y, outY = ((ny ? lo.y : hi.y) - org.y) * inv.y;
    const float inZ = ((nz ? hi.z : lo.z) - org.z) * inv.z, outZ = ((nz ? lo.z : hi.z) - org.z) * inv.z;
    entry = fmaxf(fmaxf(fmaxf(inX, inY), inZ), amin);
    const float exit = fminf(fminf(fminf(outX, outY), outZ), amax);
    return !(entry > exit);
}

PT_DEV bool aabbHit(V3 lo, V3 hi, const Ray& ray, float amin, float amax)
{
    float entry;
    return aabbEntry(lo, hi, ray, amin, amax, entry);
}

// keep a 16-byte record whole: stops the compiler from splitting the load into
// pieces fetched one early-out at a time
PT_DEV void keepWhole(float4& v)
{
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// Moeller-Trumbore, reference src/15/surface_triangle.hpp:101-140
PT_DEV bool triTest(V3 A, V3 B, V3 C, const Ray& ray, float amin, float amax, float& alpha, float& u, float& v,
                    bool& backside)
{
    V3 d = ray.direction;
    V3 e1 = B - A;
    V3 e2 = C - A;
    V3 c2 = cross(d, e2);
    float Dpre = dot(c2, e1);
    if (fabs_(Dpre) < kEps)
        return false;
    backside = Dpre < 0.0f;
    float invD = 1.0f / Dpre;
    V3 t = ray.origin - A;
    V3 c1 = cross(t, e1);
    float D2 = dot(c2, t);
    u = D2 * invD;
    if (u < 0.0f || u > 1.0f)
        return false;
    float D3 = dot(c1, d);
    v = D3 * invD;
    if (v < 0.0f || u + v > 1.0f)
        return false;
    float D1 = dot(c1, e2);
    alpha = D1 * invD;
    if (alpha < amin || alpha > amax)
        return false;
    return true;
}

// the quadratic of reference src/15/surface_sphere.hpp:71-95, without the record
PT_DEV bool sphereTest(V3 c, float r, const Ray& ray, float amin, float amax, float& a)
{
    V3 oc = ray.origin - c;
    float aq = -dot(oc, ray.direction);
    V3 tmp = oc - dot(oc, ray.direction) * ray.direction;
    float disc = r * r - dot(tmp, tmp);
    if (disc > 0.0f) {
        float a1, a2;
        if (aq < 0.0f) {
            a2 = aq - ::sqrtf(disc);
            a1 = 2.0f * aq - a2;
        } else {
            a1 = aq + ::sqrtf(disc);
            a2 = 2.0f * aq - a1;
        }
        if (a2 > amin && a2 < amax) {
            a = a2;
            return true;
        } else if (a1 > amin && a1 < amax) {
            a = a1;
            return true;
        }
    }
    return false;
}

// vertices of triangle `sid` at `time`, as the hit test sees them (getVertices,
// reference src/15/surface_triangle.hpp:79-92)
PT_DEV void triVertices(const DevScene& sc, uint32_t sid, float time, V3& A, V3& B, V3& C)
{
    float4 r0 = sc.surf[3 * size_t(sid) + 0];
    float4 r1 = sc.surf[3 * size_t(sid) + 1];
    float4 r2 = sc.surf[3 * size_t(sid) + 2];
    A = v3(r0.x, r0.y, r0.z);
    B = v3(r1.x, r1.y, r1.z);
    C = v3(r2.x, r2.y, r2.z);
    if (fbits(r0.w) == kTriDynamic) {
        const DevMesh& m = sc.meshes[fbits(r1.w) & kMeshMask];
        Xform T = animationAt<PM>(sc.animations[m.animation], time);
        A = apply(T, A);
        B = apply(T, B);
        C = apply(T, C);
    }
}

// centre and radius of sphere surface `sid` at `time` (getCR, reference
// src/15/surface_sphere.hpp:97-106); T is the pose (identity when unanimated)
PT_DEV void sphereCR(const DevScene& sc, uint32_t sid, float time, V3& c, float& r, Xform& T, uint32_t& sphereId)
{
    float4 r0 = sc.surf[3 * size_t(sid) + 0];
    float4 r1 = sc.surf[3 * size_t(sid) + 1];
    sphereId = fbits(r1.y);
    c = v3(r0.x, r0.y, r0.z);
    r = r1.x;
    T = xformIdentity();
    // (the sphere's animation number travels in its row: no look into the sphere table)
    const int animation = int(fbits(r1.z));
    if (animation >= 0) {
        T = animationAt<PM>(sc.animations[animation], time);
        if (fbits(r0.w) == kSphDynamic) {
            c = apply(T, c);
            r *= T.scaling.x;
        }
    }
}

// Surface::hit without the record: triangle or sphere.  DYN = false promises that the scene
// has no surface whose pose depends on the ray time (no two-key animation), and drops that code.
// `anims` = the animation table (sc.animations, or a copy of it that a kernel keeps in LDS).
template <bool DYN>
PT_DEV bool surfaceTest(const DevScene& sc, const float4* __restrict__ surf, const pt_animation* __restrict__ anims, uint32_t sid,
                        const Ray& ray, float amin, float amax, HitLite& out)
{
    float4 r0 = surf[3 * sid + 0];
    float4 r1 = surf[3 * sid + 1];
    uint32_t kind = fbits(r0.w);
    if (kind <= kTriDynamic) {
        float4 r2 = surf[3 * sid + 2];
        V3 A = v3(r0.x, r0.y, r0.z), B = v3(r1.x, r1.y, r1.z), C = v3(r2.x, r2.y, r2.z);
        if (DYN && kind == kTriDynamic) {
            const DevMesh& m = sc.meshes[fbits(r1.w) & kMeshMask];
            Xform T = animationAt<PM>(anims[m.animation], ray.time);
            A = apply(T, A);
            B = apply(T, B);
            C = apply(T, C);
        }
        float a, u, v;
        bool bs;
        if (!triTest(A, B, C, ray, amin, amax, a, u, v, bs))
            return false;
        out.sid = sid;
        out.a = a;
        out.u = u;
        out.v = v;
        out.backside = bs;
        out.matClass = fbits(r1.w) >> 24;
        return true;
    }
    V3 c = v3(r0.x, r0.y, r0.z);
    float r = r1.x;
    if (DYN && kind == kSphDynamic) {
        // (the sphere's animation number travels in its row: no look into the sphere table)
        Xform T = animationAt<PM>(anims[fbits(r1.z)], ray.time);
        c = apply(T, c);
        r *= T.scaling.x;
    }
    float a;
    if (!sphereTest(c, r, ray, amin, amax, a))
        return false;
    out.sid = sid;
    out.a = a;
    out.u = 0.0f;
    out.v = 0.0f;
    out.backside = false;
    out.matClass = fbits(r1.w) >> 24;
    return true;
}

template <bool DYN>
PT_DEV bool surfaceTest(const DevScene& sc, const float4* __restrict__ surf, uint32_t sid, const Ray& ray, float amin,
                        float amax, HitLite& out)
{
    return surfaceTest<DYN>(sc, surf, sc.animations, sid, ray, amin, amax, out);
}

// BVHTreeLinear::hit, reference src/15/bvh.hpp:183-213.  The reference walks
// the pre-order array depth-first with an explicit stack, always first child
// first; the stack top is always the node's `escape` link, so following the
// links visits exactly the same nodes in the same order with no stack.
//
// One step of that walk: test node `cur`, test its surface if it is a leaf whose
// box is hit, and move `cur` on.  The walk is over when cur >= sc.n_nodes.
// `nodes` and `surf` are the node and surface tables (in HBM, or a copy in LDS).
// `boxAmin` (>= amin): boxes that the ray leaves before this distance are skipped as well -- for a caller who knows
// that no surface is hit before it (see traceRetryKernel); the reference's walk has boxAmin = amin.
template <bool COUNT, bool DYN>
PT_DEV void traverseStep(const DevScene& sc, const float4* __restrict__ nodes, const float4* __restrict__ surf,
                         const Ray& ray, float amin, float boxAmin, float& amax, uint32_t& cur, HitLite& best, Counters* cnt)
{
    float4 n0 = nodes[2 * cur + 0];
    float4 n1 = nodes[2 * cur + 1];
    uint32_t escape = fbits(n1.z);
    uint32_t leaf = fbits(n1.w);
    if (COUNT)
        cnt->nodeVisits++;
    if (aabbHit(v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), ray, boxAmin, amax)) {
        if (!(leaf & kInnerBit)) {
            if (COUNT)
                cnt->leafTests++;
            if (surfaceTest<DYN>(sc, surf, leaf, ray, amin, amax, best))
                amax = best.a;
            cur = escape;
        } else {
            cur = cur + 1;
        }
    } else {
        cur = escape;
    }
}

// The node half of a step: test node `cur` and move on; if it is a leaf whose box is hit,
// report its surface in `pendingLeaf` (the caller must run the surface test -- which may
// lower amax -- before the next node step, as the reference does).
// (PIN: the table is in memory; the two loads are then kept whole, which the compiler does by itself for LDS)
template <bool PIN>
PT_DEV void nodeStep(const float4* __restrict__ nodes, const Ray& ray, float amin, float amax, uint32_t& cur,
                     uint32_t& pendingLeaf)
{
    float4 n0 = nodes[2 * cur + 0];
    float4 n1 = nodes[2 * cur + 1];
    if (PIN) {
        keepWhole(n0);
        keepWhole(n1);
    }
    const uint32_t escape = fbits(n1.z);
    const uint32_t leaf = fbits(n1.w);
    const bool hit = aabbHit(v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), ray, amin, amax);
    const bool inner = (leaf & kInnerBit) != 0;
    pendingLeaf = (hit && !inner) ? leaf : kInner;
    cur = (hit && inner) ? cur + 1 : escape;
}

// The same step on a copy of the node table whose two link words are packed for this walk (see packNodeLinks):
// `cur` alone then says what a lane does next -- below the node count: walk; kLeafDue set: the surface in `leafWord`
// is due for its test (cur & ~kLeafDue = where the walk goes on afterwards); anything else: the walk is over.
constexpr uint32_t kLeafDue = 0x80000000u;
constexpr uint32_t kNoWalk = 0x7FFFFFFFu; // a lane without a ray

// word 6: escape link (low 16 bits) | surface id + 1 of a leaf (high 16 bits); word 7: where a lane goes when the
// box is hit: the first child of an inner node, kLeafDue | escape link of a leaf.  Needs node and surface counts
// below 2^16, which holds for every table that fits LDS.
PT_DEV void packNodeLinks(uint32_t node, float4& n1)
{
    const uint32_t escape = fbits(n1.z), leaf = fbits(n1.w);
    const bool inner = (leaf & kInnerBit) != 0;
    n1.z = __uint_as_float(escape | (inner ? 0u : (leaf + 1u) << 16));
    n1.w = __uint_as_float(inner ? node + 1u : (kLeafDue | escape));
}

PT_DEV void nodeStepPacked(const float4* __restrict__ nodes, const Ray& ray, float amin, float amax, uint32_t& cur,
                           uint32_t& leafWord)
{
    const float4 n0 = nodes[2 * cur + 0];
    const float4 n1 = nodes[2 * cur + 1];
    const bool hit = aabbHit(v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), ray, amin, amax);
    leafWord = fbits(n1.z);
    cur = hit ? fbits(n1.w) : (fbits(n1.z) & 0xFFFFu);
}

// The same once more for the smallest scenes: the LDS copy holds every node eight times, once per combination of
// direction signs, each copy with its planes already in the order in which a ray of those signs meets them
// (near x, far x, near y, far y / near z, far z, and the two packed link words).  A ray picks its copy once
// (octantOf), and a step has no plane to select: 19 vector instructions instead of 24.  The arithmetic is aabbEntry's.
constexpr uint32_t kOctantRow4 = 16; // float4 per node: 8 copies of 2, copy after copy (the rows of one copy
                                     // lie 32 bytes apart, as in the plain table: a stride of 256 bytes would
                                     // put them all into the same few LDS banks)

PT_DEV uint32_t octantOf(const Ray& ray)
{
    return (ray.invDirection.x < 0.0f ? 1u : 0u) | (ray.invDirection.y < 0.0f ? 2u : 0u) | (ray.invDirection.z < 0.0f ? 4u : 0u);
}

// (n1 comes with its link words packed)
PT_DEV void octantRow(uint32_t octant, float4 n0, float4 n1, float4& r0, float4& r1)
{
    const bool nx = octant & 1u, ny = octant & 2u, nz = octant & 4u;
    r0 = make_float4(nx ? n0.w : n0.x, nx ? n0.x : n0.w, ny ? n1.x : n0.y, ny ? n0.y : n1.x);
    r1 = make_float4(nz ? n1.y : n0.z, nz ? n0.z : n1.y, n1.z, n1.w);
}

// `rows` = the ray's copy of the node table
PT_DEV void nodeStepOctant(const float4* __restrict__ rows, const Ray& ray, float amin, float amax, uint32_t& cur, uint32_t& leafWord)
{
    const float4 n0 = rows[2 * cur + 0];
    const float4 n1 = rows[2 * cur + 1];
    const V3 inv = ray.invDirection, org = ray.origin;
    const float inX = (n0.x - org.x) * inv.x, outX = (n0.y - org.x) * inv.x;
    const float inY = (n0.z - org.y) * inv.y, outY = (n0.w - org.y) * inv.y;
    const float inZ = (n1.x - org.z) * inv.z, outZ = (n1.y - org.z) * inv.z;
    const float entry = fmaxf(fmaxf(fmaxf(inX, inY), inZ), amin);
    const float exit = fminf(fminf(fminf(outX, outY), outZ), amax);
    const bool hit = !(entry > exit);
    leafWord = fbits(n1.z);
    cur = hit ? fbits(n1.w) : (fbits(n1.z) & 0xFFFFu);
}

PT_DEV HitLite noHit()
{
    HitLite h;
    h.sid = PT_NO_HIT;
    h.a = 0.0f;
    h.u = h.v = 0.0f;
    h.backside = false;
    h.matClass = 0;
    return h;
}

template <bool COUNT, bool DYN = true>
PT_DEV HitLite closestHitFrom(const DevScene& sc, const Ray& ray, float amin, float boxAmin, float amax, Counters* cnt)
{
    HitLite best = noHit();
    if (COUNT)
        cnt->rays++;
    uint32_t cur = 0;
    const uint32_t end = sc.n_nodes;
    while (cur < end)
        traverseStep<COUNT, DYN>(sc, sc.nodes, sc.surf, ray, amin, boxAmin, amax, cur, best, cnt);
    return best;
}

template <bool COUNT, bool DYN = true>
PT_DEV HitLite closestHit(const DevScene& sc, const Ray& ray, float amin, float amax, Counters* cnt)
{
    return closestHitFrom<COUNT, DYN>(sc, ray, amin, amin, amax, cnt);
}

// The full HitRecord of reference src/15/surface.hpp:9-30
struct Hit {
    float a;
    V3 position, normal;
    V2 texcoord;
    V3 tangent;
    bool backside;
    uint32_t surface;
    int32_t material;
};

// Rebuild the record of an accepted hit: reference
// src/15/surface_triangle.hpp:142-175 and surface_sphere.hpp:11-37.
__device__ inline Hit makeHit(const DevScene& sc, const Ray& ray, const HitLite& h)
{
    Hit hr;
    hr.a = h.a;
    hr.surface = h.sid;
    hr.position = rayAt(ray, h.a);
    const uint4 attr = sc.surfAttr[h.sid];
    if ((attr.w >> 24) & 64u) {
        // a triangle with a fixed pose and no vertex normals (and no tangents for a pose to turn): the normal of
        // its plane was worked out at upload, the vertices are not needed
        const uint32_t flags = attr.w >> 24;
        const float4 fn = sc.faceN[h.sid];
        V3 nrm = v3(fn.x, fn.y, fn.z);
        if (h.backside)
            nrm = -nrm;
        const size_t i0 = attr.x, i1 = attr.y, i2 = attr.z;
        const float u = h.u, v = h.v;
        const float w = 1.0f - u - v;
        V2 tc = V2{ 0.0f, 0.0f };
        if ((flags & 2u) && (flags & 16u)) {
            const float* a = sc.texcoords + 2 * i0;
            const float* b = sc.texcoords + 2 * i1;
            const float* c = sc.texcoords + 2 * i2;
            tc = V2{ w * a[0] + u * b[0] + v * c[0], w * a[1] + u * b[1] + v * c[1] };
        }
        V3 tng = v3(0.0f);
        if (flags & 32u) {
            if (flags & 4u) {
                tng = w * v3(sc.tangents + 3 * i0) + u * v3(sc.tangents + 3 * i1) + v * v3(sc.tangents + 3 * i2);
                tng = tng - dot(nrm, tng) * nrm;
                if (dot(tng, tng) > kEps)
                    tng = normalize(tng);
            }
            tng = normalize(tng);
        }
        hr.normal = nrm;
        hr.texcoord = tc;
        hr.tangent = tng;
        hr.backside = h.backside;
        hr.material = int32_t(attr.w & 0xFFFFFFu);
        return hr;
    }
    float4 r0 = sc.surf[3 * size_t(h.sid) + 0];
    float4 r1 = sc.surf[3 * size_t(h.sid) + 1];
    uint32_t kind = fbits(r0.w);
    if (kind <= kTriDynamic) {
        float4 r2 = sc.surf[3 * size_t(h.sid) + 2];
        const uint32_t flags = attr.w >> 24;
        const bool animated = (flags & 8u) != 0;
        Quat rot = Quat{ 0.0f, 0.0f, 0.0f, 1.0f };
        V3 A = v3(r0.x, r0.y, r0.z), B = v3(r1.x, r1.y, r1.z), C = v3(r2.x, r2.y, r2.z);
        if (animated) {
            const DevMesh& m = sc.meshes[fbits(r1.w) & kMeshMask];
            Xform T = animationAt<PM>(sc.animations[m.animation], ray.time);
            rot = T.rotation;
            if (kind == kTriDynamic) {
                A = apply(T, A);
                B = apply(T, B);
                C = apply(T, C);
            }
        }
        const size_t i0 = attr.x, i1 = attr.y, i2 = attr.z;
        float u = h.u, v = h.v;
        float w = 1.0f - u - v;
        V3 nrm;
        if (flags & 1u) {
            nrm = w * v3(sc.normals + 3 * i0) + u * v3(sc.normals + 3 * i1) + v * v3(sc.normals + 3 * i2);
            if (animated)
                nrm = rotate(rot, nrm);
        } else {
            nrm = cross(B - A, C - A);
        }
        nrm = normalize(nrm);
        if (h.backside)
            nrm = -nrm;
        V2 tc;
        // (texture coordinates and tangent only for a material that reads them)
        if ((flags & 2u) && (flags & 16u)) {
            const float* a = sc.texcoords + 2 * i0;
            const float* b = sc.texcoords + 2 * i1;
            const float* c = sc.texcoords + 2 * i2;
            tc = V2{ w * a[0] + u * b[0] + v * c[0], w * a[1] + u * b[1] + v * c[1] };
        } else {
            tc = V2{ 0.0f, 0.0f };
        }
        V3 tng = v3(0.0f);
        if (flags & 32u) {
            if (flags & 4u) {
                tng = w * v3(sc.tangents + 3 * i0) + u * v3(sc.tangents + 3 * i1) + v * v3(sc.tangents + 3 * i2);
                if (animated)
                    tng = rotate(rot, tng);
                tng = tng - dot(nrm, tng) * nrm;
                if (dot(tng, tng) > kEps)
                    tng = normalize(tng);
            }
            tng = normalize(tng);
        }
        hr.normal = nrm;
        hr.texcoord = tc;
        hr.tangent = tng;
        hr.backside = h.backside;
        hr.material = int32_t(attr.w & 0xFFFFFFu);
        return hr;
    }
    V3 c;
    float r;
    Xform T;
    uint32_t sphereId;
    sphereCR(sc, h.sid, ray.time, c, r, T, sphereId);
    V3 n = normalize(hr.position - c);
    const uint32_t flags = attr.w >> 24;
    float u = 0.0f, v = 0.0f;
    V3 t = v3(0.0f);
    if (flags & 48u) {
        V3 rn = rotate(T.rotation, n);
        float alpha = ptm::atan2f(rn.x, rn.z);
        if (flags & 16u) {
            float beta = ptm::asinf(min_(1.0f, max_(-1.0f, rn.y)));
            u = (alpha + kPi) / (2.0f * kPi);
            v = (beta + 0.5f * kPi) / kPi;
        }
        if (flags & 32u) {
            float sa, ca;
            ptm::sincosf(alpha, sa, ca);
            t = v3(ca, 0.0f, -sa);
        }
    }
    bool backside = false;
    if (dot(n, -ray.direction) < 0.0f) {
        backside = true;
        n = -n;
    }
    hr.normal = n;
    hr.texcoord = V2{ u, v };
    hr.tangent = t;
    hr.backside = backside;
    hr.material = int32_t(attr.w & 0xFFFFFFu);
    return hr;
}

// ----------------------------------------------------------- light sampling --

// Surface::direction: reference src/15/surface_triangle.hpp:178-189,
// surface_sphere.hpp:117-139.  Two draws; the first one drawn is the LAST
// argument of the warp (the reference's compiler evaluates the argument list
// right to left).
template <bool COUNT>
__device__ inline V3 surfaceDirection(const DevScene& sc, uint32_t sid, V3 origin, float time, Rng<COUNT>& rng)
{
    float4 r0 = sc.surf[3 * size_t(sid) + 0];
    float4 r1 = sc.surf[3 * size_t(sid) + 1];
    uint32_t kind = fbits(r0.w);
    float second = rng.in01();
    float first = rng.in01();
    if (kind <= kTriDynamic) {
        float4 r2 = sc.surf[3 * size_t(sid) + 2];
        const DevMesh& m = sc.meshes[fbits(r1.w) & kMeshMask];
        V3 A, B, C;
        if (m.animation >= 0) {
            // sample the unposed triangle, then pose the point
            const uint4 attr = sc.surfAttr[sid];
            A = v3(sc.positions + 3 * size_t(attr.x));
            B = v3(sc.positions + 3 * size_t(attr.y));
            C = v3(sc.positions + 3 * size_t(attr.z));
        } else {
            A = v3(r0.x, r0.y, r0.z);
            B = v3(r1.x, r1.y, r1.z);
            C = v3(r2.x, r2.y, r2.z);
        }
        V3 bary = uniformInTriangle(first, second);
        V3 P = bary.x * A + bary.y * B + bary.z * C;
        if (m.animation >= 0)
            P = apply(animationAt<PM>(sc.animations[m.animation], time), P);
        return normalize(P - origin);
    }
    V3 c;
    float r;
    Xform T;
    uint32_t sphereId;
    sphereCR(sc, sid, time, c, r, T, sphereId);
    V3 cmo = c - origin;
    float distSq = dot(cmo, cmo);
    float radSq = r * r;
    if (distSq <= radSq)
        return uniformOnSphere(first, second);
    float disc = 1.0f - radSq / distSq;
    float cosThetaMax = ::sqrtf(max_(0.0f, disc));
    V3 around = uniformTowardsSphere(cosThetaMax, first, second);
    return normalize(toWorldSpace(tangentSpace(normalize(cmo)), around));
}

// Surface::p: reference src/15/surface_triangle.hpp:191-208, surface_sphere.hpp:141-168
template <bool COUNT>
__device__ inline float surfaceP(const DevScene& sc, uint32_t sid, const Ray& ray, Counters* cnt)
{
    if (COUNT)
        cnt->lightTests++;
    float4 r0 = sc.surf[3 * size_t(sid) + 0];
    uint32_t kind = fbits(r0.w);
    if (kind <= kTriDynamic) {
        V3 A, B, C;
        triVertices(sc, sid, ray.time, A, B, C);
        float a, u, v;
        bool bs;
        if (!triTest(A, B, C, ray, 0.0f, kFltMax, a, u, v, bs))
            return 0.0f;
        V3 edgeCross = cross(B - A, C - A);
        float edgeCrossLength = ::sqrtf(dot(edgeCross, edgeCross));
        V3 faceNormal = edgeCross / edgeCrossLength;
        float faceArea = 0.5f * edgeCrossLength;
        float cosine = fabs_(dot(faceNormal, -ray.direction));
        float distanceSquared = a * a;
        return distanceSquared / (cosine * faceArea);
    }
    V3 c;
    float r;
    Xform T;
    uint32_t sphereId;
    sphereCR(sc, sid, ray.time, c, r, T, sphereId);
    float val = 0.0f;
    V3 cmo = c - ray.origin;
    float distSq = dot(cmo, cmo);
    float radSq = r * r;
    if (distSq <= radSq) {
        val = 1.0f / (4.0f * kPi);
    } else {
        float a;
        if (sphereTest(c, r, ray, 0.0f, kFltMax, a)) {
            float disc = 1.0f - radSq / distSq;
            float cosThetaMax = ::sqrtf(max_(0.0f, disc));
            float solidAngle = 2.0f * kPi * (1.0f - cosThetaMax);
            val = 1.0f / solidAngle;
        }
    }
    return val;
}

// --------------------------------------------------------------- materials --
// reference src/15/material.hpp:7-61 and material_*.hpp

enum { ScatterNone = 0, ScatterExplicit = 1, ScatterRandom = 2 };
struct Scatter {
    int type;
    V3 direction;
    float p;
    V3 attenuation;
};
PT_DEV Scatter scatterNone() { return Scatter{ ScatterNone, v3(0.0f), 0.0f, v3(0.0f) }; }
PT_DEV Scatter scatterExplicit(V3 dir, V3 att) { return Scatter{ ScatterExplicit, dir, 1.0f, att }; }
PT_DEV Scatter scatterRandom(V3 dir, float p, V3 att) { return Scatter{ ScatterRandom, dir, p, att }; }

// MaterialTwoSided dispatch (material_twosided.hpp:8-37): descend to the
// side's material, presenting the hit as a front-side hit
PT_DEV int resolveTwoSided(const DevScene& sc, int mid, bool& backside)
{
    for (int depth = 0; depth < 10; depth++) {
        const pt_material& m = sc.materials[mid];
        if (m.type != PT_MAT_TWOSIDED)
            break;
        if (backside) {
            mid = m.back;
            backside = false;
        } else {
            mid = m.front;
        }
    }
    return mid;
}

// Material::Le (material_light.hpp:14-17, material_twosided.hpp:23-29, material.hpp:47-50)
PT_DEV V3 materialLe(const DevScene& sc, int mid, bool backside)
{
    mid = resolveTwoSided(sc, mid, backside);
    const pt_material& m = sc.materials[mid];
    if (m.type == PT_MAT_LIGHT)
        return backside ? v3(0.0f) : v3(m.v[0], m.v[1], m.v[2]);
    return v3(0.0f);
}

// material_phong.hpp:24-31; the reference calls the double-precision pow here
PT_DEV V3 phongBrdf(V3 n, V3 l, V3 v, V3 kd, V3 ks, float shininess)
{
    V3 diffuse = kd / kPi;
    float cosRV = max_(dot(reflect(-l, n), v), 0.0f);
    V3 specular = ks * (shininess + 2.0f) / (2.0f * kPi) * ptm::powf(cosRV, shininess);
    return diffuse + specular;
}

// material_phong.hpp:33-46
__device__ inline V3 phongNormal(const DevScene& sc, const pt_material& m, const Hit& hr)
{
    V3 n = hr.normal;
    if (m.tex[4] >= 0) {
        n = textureValue(sc, m.tex[4], hr.texcoord);
        n = 2.0f * n - v3(1.0f);
        if (dot(n, n) > kEps && dot(hr.tangent, hr.tangent) > kEps) {
            TangentSpace ts = tangentSpace(hr.normal, hr.tangent);
            n = normalize(toWorldSpace(ts, n));
        }
    }
    return n;
}

// value of texture slot k (0, 1, 2) of material `mid`, skipping the texture tables when the
// texture is a constant one (resolved at upload)
PT_DEV V3 materialTexture(const DevScene& sc, int mid, const pt_material& m, int k, V2 tc)
{
    const float4 c0 = sc.matConst[2 * mid + 0];
    if (fbits(c0.w) & (1u << k)) {
        if (k == 0)
            return v3(c0.x, c0.y, c0.z);
        const float4 c1 = sc.matConst[2 * mid + 1];
        if (k == 1)
            return v3(c1.x, c1.y, c1.z);
        return v3(c1.w, c1.w, c1.w); // only the x component of slot 2 is ever used
    }
    return textureValue(sc, m.tex[k], tc);
}

// What Material::scatter read from the material's textures at the hit, kept for scatterToDirection at the same
// hit (textures are functions of the texture coordinate alone, so the values are the same ones).
struct MaterialValues {
    V3 kd, ks, n;
    float shininess;
};

// Material::scatter
template <bool COUNT>
__device__ inline Scatter materialScatter(const DevScene& sc, int mid, const Ray& ray, const Hit& hrIn, Rng<COUNT>& rng,
                                          MaterialValues* keep = nullptr)
{
    bool backside = hrIn.backside;
    mid = resolveTwoSided(sc, mid, backside);
    const pt_material& m = sc.materials[mid];
    const Hit& hr = hrIn;
    switch (m.type) {
    case PT_MAT_LAMBERTIAN: { // material_lambertian.hpp:24-38
        if (backside)
            return scatterNone();
        TangentSpace ts = tangentSpace(hr.normal);
        float second = rng.in01();
        float first = rng.in01();
        V3 dirTS = cosineWeightedOnHemisphere(first, second);
        V3 dir = toWorldSpace(ts, dirTS);
        float cosTheta = dot(hr.normal, dir);
        if (cosTheta <= 0.0f)
            return scatterNone();
        float p = cosTheta / kPi;
        V3 albedo = materialTexture(sc, mid, m, 0, hr.texcoord);
        if (keep)
            keep->kd = albedo;
        V3 att = (albedo / kPi) * cosTheta;
        return scatterRandom(dir, p, att);
    }
    case PT_MAT_PHONG: { // material_phong.hpp:48-102
        if (m.tex[3] >= 0) {
            float alpha = textureValue(sc, m.tex[3], hr.texcoord).x;
            bool transparent = alpha < rng.in01();
            if (transparent)
                return scatterExplicit(ray.direction, v3(1.0f));
        }
        if (backside)
            return scatterNone();
        V3 kd = materialTexture(sc, mid, m, 0, hr.texcoord);
        V3 ks = materialTexture(sc, mid, m, 1, hr.texcoord);
        float shininess = materialTexture(sc, mid, m, 2, hr.texcoord).x;
        V3 n = phongNormal(sc, m, hr);
        if (keep) {
            keep->kd = kd;
            keep->ks = ks;
            keep->n = n;
            keep->shininess = shininess;
        }
        V3 v = -ray.direction;
        V3 r = reflect(ray.direction, n);
        float sumKd = kd.x + kd.y + kd.z;
        float sumKs = ks.x + ks.y + ks.z;
        float sum = sumKd + sumKs + 1e-4f;
        float specProb = sumKs / sum;
        if (specProb < 0.1f)
            specProb = 0.1f;
        else if (specProb > 0.9f)
            specProb = 0.9f;
        V3 l;
        if (rng.in01() < specProb) {
            float second = rng.in01();
            float first = rng.in01();
            V3 around = phongWeightedOnHemisphere(shininess, first, second);
            l = normalize(toWorldSpace(tangentSpace(r), around));
        } else {
            TangentSpace ts = tangentSpace(n);
            float second = rng.in01();
            float first = rng.in01();
            V3 dirTS = cosineWeightedOnHemisphere(first, second);
            l = normalize(toWorldSpace(ts, dirTS));
        }
        float cosTheta = dot(l, n);
        if (cosTheta <= 0.0f)
            return scatterNone();
        float diffusePdf = cosTheta / kPi;
        float specularPdf = 0.5f / kPi * (shininess + 1.0f) * ptm::powf(max_(dot(r, l), 0.0f), shininess);
        float p = mixf(diffusePdf, specularPdf, specProb);
        V3 att = phongBrdf(n, l, v, kd, ks, shininess) * cosTheta;
        return scatterRandom(l, p, att);
    }
    case PT_MAT_MIRROR: { // material_mirror.hpp:14-21
        if (backside)
            return scatterNone();
        V3 dir = normalize(reflect(ray.direction, hr.normal));
        return scatterExplicit(dir, materialTexture(sc, mid, m, 0, hr.texcoord));
    }
    case PT_MAT_GLASS: { // material_glass.hpp:17-49
        V3 att = v3(1.0f);
        float n2 = m.v[3];
        float n1 = 1.0f;
        if (backside) {
            float tmp = n2;
            n2 = n1;
            n1 = tmp;
            float dist = hr.a;
            att = v3(ptm::expf(-m.v[0] * dist), ptm::expf(-m.v[1] * dist), ptm::expf(-m.v[2] * dist));
        }
        V3 refracted = refract(ray.direction, hr.normal, n1 / n2);
        bool doReflection = true;
        if (dot(refracted, refracted) > 0.0f) {
            float cosI = dot(-ray.direction, hr.normal);
            float cosT = -dot(refracted, hr.normal);
            float fr = fresnelUnpolarized(cosI, cosT, n1, n2);
            doReflection = rng.in01() < fr;
        }
        if (doReflection)
            return scatterExplicit(normalize(reflect(ray.direction, hr.normal)), att);
        return scatterExplicit(normalize(refracted), att);
    }
    default: // Material, MaterialLight: material.hpp:52-55
        return scatterNone();
    }
}

// Material::scatterToDirection.  MaterialTwoSided does not override it in the
// reference, so a two-sided material answers like the base class: no scatter.
// `kept`: the values materialScatter read at this hit for this very material (the caller passes them only after
// a random scatter of material `mid` itself, not of a side of a two-sided one), or null.
__device__ inline Scatter materialScatterToDirection(const DevScene& sc, int mid, const Ray& ray, const Hit& hr, V3 dir,
                                                     const MaterialValues* kept = nullptr)
{
    const pt_material& m = sc.materials[mid];
    if (m.type == PT_MAT_LAMBERTIAN) { // material_lambertian.hpp:40-49
        float cosTheta = dot(hr.normal, dir);
        if (cosTheta <= 0.0f)
            return scatterNone();
        float p = cosTheta / kPi;
        V3 albedo = kept ? kept->kd : materialTexture(sc, mid, m, 0, hr.texcoord);
        V3 att = (albedo / kPi) * cosTheta;
        return scatterRandom(dir, p, att);
    }
    if (m.type == PT_MAT_PHONG) { // material_phong.hpp:104-117
        V3 n = kept ? kept->n : phongNormal(sc, m, hr);
        float cosTheta = dot(n, dir);
        if (cosTheta <= 0.0f)
            return scatterNone();
        float p = cosTheta / kPi;
        V3 kd = kept ? kept->kd : materialTexture(sc, mid, m, 0, hr.texcoord);
        V3 ks = kept ? kept->ks : materialTexture(sc, mid, m, 1, hr.texcoord);
        float shininess = kept ? kept->shininess : materialTexture(sc, mid, m, 2, hr.texcoord).x;
        V3 att = phongBrdf(n, dir, -ray.direction, kd, ks, shininess) * cosTheta;
        return scatterRandom(dir, p, att);
    }
    return scatterNone();
}

// --------------------------------------------------------------- integrator --

PT_DEV float powerHeuristicMIS(float f, float g) // reference src/15/pathtracer.cpp:42-47
{
    f *= f;
    g *= g;
    return (f + g > 0.0f ? f / (f + g) : 0.0f);
}

struct PathParams {
    int maxSegments;
    int rrStart;
    int traceNullShadowRays = 0; // trace the shadow rays that are known to add nothing, as the reference does
};

// mean of Surface::p over all lights (reference src/15/pathtracer.cpp:78-81,88-91)
template <bool COUNT>
__device__ inline float lightsPdf(const DevScene& sc, V3 origin, V3 dir, float time, Counters* cnt)
{
    Ray r = makeRay(origin, dir, time);
    float sum = 0.0f;
    for (uint32_t i = 0; i < sc.n_lights; i++)
        sum += surfaceP<COUNT>(sc, sc.lights[i], r, cnt);
    sum /= sc.n_lights; // unsigned -> float, as size_t -> float in the reference
    return sum;
}

// The same for two directions from one point at once (the scattered direction and the direction to the
// sampled light, reference src/15/pathtracer.cpp:78-81 and 88-91): every light is read once, and what depends
// on the light alone (edges, normal, area) is computed once.  Each direction sees the arithmetic of surfaceP.
// `chosen` = the light that dirB was sampled on.  `chosenSide` tells what the test of that very surface with the ray
// along dirB says, which is the test a shadow ray along dirB makes at that surface's leaf: kSideMiss (the ray cannot
// arrive at it), kSideFront, kSideBack, or kSideUnknown for a light that is no triangle.
enum : uint32_t { kSideUnknown = 0, kSideMiss, kSideFront, kSideBack };
__device__ inline void lightsPdf2(const DevScene& sc, V3 origin, V3 dirA, V3 dirB, float time, uint32_t chosen, float& pA,
                                  float& pB, uint32_t& chosenSide)
{
    const Ray ra = makeRay(origin, dirA, time), rb = makeRay(origin, dirB, time);
    float sumA = 0.0f, sumB = 0.0f;
    chosenSide = kSideUnknown;
    for (uint32_t i = 0; i < sc.n_lights; i++) {
        const uint32_t sid = sc.lights[i];
        const float4 r0 = sc.surf[3 * size_t(sid) + 0];
        if (fbits(r0.w) <= kTriDynamic) {
            V3 A, B, C;
            triVertices(sc, sid, time, A, B, C);
            float aA, aB, u, v;
            bool bs;
            const bool hitA = triTest(A, B, C, ra, 0.0f, kFltMax, aA, u, v, bs);
            const bool hitB = triTest(A, B, C, rb, 0.0f, kFltMax, aB, u, v, bs);
            if (i == chosen)
                chosenSide = !hitB ? kSideMiss : bs ? kSideBack : kSideFront;
            float valA = 0.0f, valB = 0.0f;
            if (hitA || hitB) {
                V3 faceNormal;
                float faceArea;
                const float4 lg = sc.lightGeo[i]; // (the same for every lane)
                if (lg.w >= 0.0f) {
                    faceNormal = v3(lg.x, lg.y, lg.z);
                    faceArea = lg.w;
                } else {
                    const V3 edgeCross = cross(B - A, C - A);
                    const float edgeCrossLength = ::sqrtf(dot(edgeCross, edgeCross));
                    faceNormal = edgeCross / edgeCrossLength;
                    faceArea = 0.5f * edgeCrossLength;
                }
                if (hitA)
                    valA = (aA * aA) / (fabs_(dot(faceNormal, -ra.direction)) * faceArea);
                if (hitB)
                    valB = (aB * aB) / (fabs_(dot(faceNormal, -rb.direction)) * faceArea);
            }
            sumA += valA;
            sumB += valB;
        } else {
            sumA += surfaceP<false>(sc, sid, ra, nullptr);
            sumB += surfaceP<false>(sc, sid, rb, nullptr);
        }
    }
    pA = sumA / sc.n_lights;
    pB = sumB / sc.n_lights;
}

// pathSample, reference src/15/pathtracer.cpp:50-130
// The loop may be entered in the middle of a path: at segment `firstSegment` with the radiance and throughput
// gathered so far and, if `known`, with the closest hit of `ray` already found (`firstHit`).
template <bool COUNT>
__device__ inline V3 pathSampleFrom(const DevScene& sc, const PathParams& pp, Ray ray, Rng<COUNT>& rng, Counters* cnt,
                                    int firstSegment, V3 radiance, V3 throughput, bool known, HitLite firstHit)
{
    const float MinHit = 0.0001f;
    const float MaxHit = kFltMax;
    for (int segment = firstSegment; segment < pp.maxSegments; segment++) {
        HitLite hl = (known && segment == firstSegment) ? firstHit : closestHit<COUNT>(sc, ray, MinHit, MaxHit, cnt);
        if (hl.sid == PT_NO_HIT) {
            if (sc.envmap.type != PT_ENV_NONE)
                radiance = radiance + throughput * envmapValue(sc, ray.direction);
            break;
        }
        if (COUNT)
            cnt->vertices++;
        Hit hr = makeHit(sc, ray, hl);
        Scatter sr = materialScatter<COUNT>(sc, hr.material, ray, hr, rng);
        radiance = radiance + throughput * materialLe(sc, hr.material, hr.backside);
        if (sr.type == ScatterNone)
            break;
        V3 nextThroughput = throughput * sr.attenuation / sr.p;

        if (sr.type == ScatterRandom && sc.n_lights > 0) {
            float lightsP = lightsPdf<COUNT>(sc, hr.position, sr.direction, ray.time, cnt);
            nextThroughput = nextThroughput * powerHeuristicMIS(sr.p, lightsP);
            uint32_t lightIndex = uint32_t(rng.in01() * sc.n_lights);
            if (lightIndex >= sc.n_lights) // the product can round up to n for n >= 2^24
                lightIndex = sc.n_lights - 1;
            uint32_t lightSurf = sc.lights[lightIndex];
            V3 lightDir = surfaceDirection<COUNT>(sc, lightSurf, hr.position, ray.time, rng);
            float lightDirP = lightsPdf<COUNT>(sc, hr.position, lightDir, ray.time, cnt);
            if (lightDirP > 0.0f) {
                Scatter lightSR = materialScatterToDirection(sc, hr.material, ray, hr, lightDir);
                if (lightSR.p > 0.0f) {
                    Ray lightRay = makeRay(hr.position, lightDir, ray.time);
                    HitLite lh = closestHit<COUNT>(sc, lightRay, MinHit, MaxHit, cnt);
                    if (lh.sid != PT_NO_HIT && lh.sid == lightSurf) {
                        // Le of the light needs only its material and the backside flag
                        float4 r0 = sc.surf[3 * size_t(lh.sid) + 0];
                        int lmat;
                        bool lback;
                        if (fbits(r0.w) <= kTriDynamic) {
                            lmat = int(sc.surfAttr[lh.sid].w & 0xFFFFFFu);
                            lback = lh.backside;
                        } else {
                            Hit lhr = makeHit(sc, lightRay, lh);
                            lmat = lhr.material;
                            lback = lhr.backside;
                        }
                        float weight = powerHeuristicMIS(lightDirP, lightSR.p);
                        radiance = radiance
                            + throughput * lightSR.attenuation / lightDirP * weight * materialLe(sc, lmat, lback);
                    }
                }
            }
        }

        throughput = nextThroughput;
        ray = makeRay(hr.position, sr.direction, ray.time);

        float maxT = max_(throughput.x, max_(throughput.y, throughput.z));
        if (maxT < 1.0f && segment >= pp.rrStart) {
            float q = 1.0f - maxT;
            if (q > 0.95f)
                q = 0.95f;
            if (rng.in01() < q)
                break;
            float rrWeight = 1.0f / (1.0f - q);
            throughput = throughput * rrWeight;
        }
    }
    return radiance;
}

template <bool COUNT>
__device__ inline V3 pathSample(const DevScene& sc, const PathParams& pp, Ray ray, Rng<COUNT>& rng, Counters* cnt)
{
    return pathSampleFrom<COUNT>(sc, pp, ray, rng, cnt, 0, v3(0.0f), v3(1.0f), false, noHit());
}

// Camera::getRay, reference src/15/camera.hpp:34-55
template <bool COUNT>
__device__ inline Ray cameraRay(const pt_camera& cam, float p, float q, float t0, float t1, Rng<COUNT>& rng)
{
    V3 P = v3(mixf(cam.l, cam.r, p), mixf(cam.b, cam.t, q), -1.0f);
    P = P * cam.focus_distance;
    float second = rng.in01();
    float first = rng.in01();
    V2 lens;
    if (cam.aperture_radius == 0.0f) {
        lens = uniformInDiskTimesZero(first, second, cam.aperture_radius);
    } else {
        lens = uniformInDisk(first, second);
        lens = V2{ lens.x * cam.aperture_radius, lens.y * cam.aperture_radius };
    }
    V3 O = v3(lens.x, lens.y, 0.0f);
    V3 D = P - O;
    // (a still frame: the draw is made -- the generator moves on -- but t0 + u * (t1 - t0) with t1 - t0 = +0 is
    // t0 + 0 for every u in [0, 1), so its value need not be worked out)
    float t;
    if (t0 == t1) {
        rng.n++;
        if (COUNT)
            rng.cnt->draws++;
        t = mixf(t0, t1, 0.0f);
    } else {
        t = mixf(t0, t1, rng.in01());
    }
    if (cam.has_animation) {
        Xform T = animationAt<PM>(cam.animation, t);
        O = apply(T, O);
        D = rotate(T.rotation, D);
    }
    return makeRay(O, normalize(D), t);
}

// the ray of sample `smp` of pixel (x, y): reference src/15/pathtracer.cpp:211-215
// (i, j) = the sample's stratum: smp / sqrt_spp and smp % sqrt_spp
template <bool COUNT>
__device__ inline Ray sampleRayOfStratum(const pt_camera& cam, const pt_render_params& p, int x, int y, int i, int j, Rng<COUNT>& rng)
{
    float sp = (i + rng.in01()) / p.sqrt_spp;
    float sq = (j + rng.in01()) / p.sqrt_spp;
    float pp = (x + sp) / p.width;
    float q = (y + sq) / p.height;
    return cameraRay<COUNT>(cam, pp, q, p.t0, p.t1, rng);
}

template <bool COUNT>
__device__ inline Ray sampleRay(const pt_camera& cam, const pt_render_params& p, int x, int y, int smp, Rng<COUNT>& rng)
{
    return sampleRayOfStratum<COUNT>(cam, p, x, y, smp / p.sqrt_spp, smp % p.sqrt_spp, rng);
}

} // namespace ptd
