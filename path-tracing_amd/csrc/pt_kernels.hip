// pt_kernels.hip -- kernels and the device half of the C ABI (include/pt_amd.h).
//
// Kernels (gfx950, wave64):
//   renderPixelsKernel   one lane per pixel; the lane walks its pixel's
//                        samples in order and adds them in order, so the sum
//                        is the same float sequence as the reference's loop
//                        (reference src/15/pathtracer.cpp:207-220).  A wave
//                        covers an 8x8 pixel tile.
//   renderSamplesKernel  one lane per (pixel, sample): per-sample radiance
//   primaryHitsKernel    one lane per (pixel, sample): closest hit of the camera ray
//   traceRaysKernel      one lane per caller-supplied ray
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <functional>
#include <vector>

#include "pt_amd.h"
#include "pt_device.hpp"
#include "pt_internal.hpp"
#include "pt_scene_host.hpp"

using namespace ptd;
using ptint::fail;

namespace {

// pixel of the tile handled by this lane: 8x8 pixels per wave, 16x16 per 256-thread block
PT_DEV bool lanePixel(const RenderArgs& a, int& x, int& y)
{
    const int tilesX = (a.tw + 15) / 16;
    const int bx = blockIdx.x % tilesX;
    const int by = blockIdx.x / tilesX;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int lx = ((wave & 1) << 3) + (lane & 7);
    const int ly = ((wave >> 1) << 3) + (lane >> 3);
    x = bx * 16 + lx;
    y = by * 16 + ly;
    bool ok = x < a.tw && y < a.th;
    x += a.x0;
    y += a.y0;
    return ok;
}

template <bool COUNT>
__global__ void __launch_bounds__(256) renderPixelsKernel(DevScene sc, RenderArgs a, float* __restrict__ out,
                                                          unsigned long long* __restrict__ counters)
{
    int x, y;
    if (!lanePixel(a, x, y))
        return;
    const uint64_t pixel = uint64_t(y) * a.p.width + x;
    Counters cnt = { 0, 0, 0, 0, 0, 0 };
    V3 acc = v3(0.0f);
    for (int smp = a.s0; smp < a.s1; smp++) {
        Rng<COUNT> rng;
        rng.key = pt_rng_key(a.p.seed, pixel, uint64_t(smp));
        rng.n = 0;
        rng.cnt = &cnt;
        Ray ray = sampleRay<COUNT>(a.cam, a.p, x, y, smp, rng);
        acc = acc + pathSample<COUNT>(sc, a.pp, ray, rng, &cnt);
    }
    if (!(a.p.flags & PT_RENDER_SUM_ONLY))
        acc = acc / float(a.p.sqrt_spp * a.p.sqrt_spp);
    out[3 * pixel + 0] = acc.x;
    out[3 * pixel + 1] = acc.y;
    out[3 * pixel + 2] = acc.z;
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)cnt.rays);
        atomicAdd(&counters[1], (unsigned long long)cnt.nodeVisits);
        atomicAdd(&counters[2], (unsigned long long)cnt.leafTests);
        atomicAdd(&counters[3], (unsigned long long)cnt.lightTests);
        atomicAdd(&counters[4], (unsigned long long)cnt.vertices);
        atomicAdd(&counters[5], (unsigned long long)cnt.draws);
    }
}

// index k -> (pixel of the tile in row-major order, sample)
PT_DEV void sampleOfIndex(const RenderArgs& a, uint64_t k, int& x, int& y, int& smp)
{
    const int ns = a.s1 - a.s0;
    uint64_t tp = k / ns;
    smp = a.s0 + int(k % ns);
    x = a.x0 + int(tp % a.tw);
    y = a.y0 + int(tp / a.tw);
}

__global__ void __launch_bounds__(256) renderSamplesKernel(DevScene sc, RenderArgs a, uint64_t n, float* __restrict__ out)
{
    uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    int x, y, smp;
    sampleOfIndex(a, k, x, y, smp);
    Rng<false> rng;
    rng.key = pt_rng_key(a.p.seed, uint64_t(y) * a.p.width + x, uint64_t(smp));
    rng.n = 0;
    rng.cnt = nullptr;
    Ray ray = sampleRay<false>(a.cam, a.p, x, y, smp, rng);
    V3 v = pathSample<false>(sc, a.pp, ray, rng, nullptr);
    out[3 * k + 0] = v.x;
    out[3 * k + 1] = v.y;
    out[3 * k + 2] = v.z;
}

PT_DEV pt_hit toHit(const DevScene& sc, const Ray& ray, const HitLite& h)
{
    pt_hit r;
    if (h.sid == PT_NO_HIT) {
        r.surface = PT_NO_HIT;
        r.material = PT_NO_HIT;
        r.a = 0.0f;
        r.backside = 0;
    } else {
        Hit hr = makeHit(sc, ray, h);
        r.surface = h.sid;
        r.material = uint32_t(hr.material);
        r.a = h.a;
        r.backside = hr.backside ? 1u : 0u;
    }
    return r;
}

__global__ void __launch_bounds__(256) primaryHitsKernel(DevScene sc, RenderArgs a, uint64_t n, pt_hit* __restrict__ out)
{
    uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    int x, y, smp;
    sampleOfIndex(a, k, x, y, smp);
    Rng<false> rng;
    rng.key = pt_rng_key(a.p.seed, uint64_t(y) * a.p.width + x, uint64_t(smp));
    rng.n = 0;
    rng.cnt = nullptr;
    Ray ray = sampleRay<false>(a.cam, a.p, x, y, smp, rng);
    HitLite h = closestHit<false>(sc, ray, 0.0001f, kFltMax, nullptr);
    out[k] = toHit(sc, ray, h);
}

__global__ void __launch_bounds__(256) traceRaysKernel(DevScene sc, uint64_t n, const float* __restrict__ origins,
                                                       const float* __restrict__ dirs, const float* __restrict__ times,
                                                       float amin, float amax, pt_hit* __restrict__ out)
{
    uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    Ray ray = makeRay(v3(origins + 3 * k), v3(dirs + 3 * k), times ? times[k] : 0.0f);
    HitLite h = closestHit<false>(sc, ray, amin, amax, nullptr);
    out[k] = toHit(sc, ray, h);
}

// ------------------------------------------------------------------- host --

} // namespace

namespace {

// Convert the reference-order node array into device nodes with escape links.
int buildDeviceNodes(const pt_bvh_node* nodes, uint64_t n, std::vector<float4>& out)
{
    out.resize(2 * n);
    std::vector<uint32_t> escape(n);
    // pre-order walk with an explicit stack of (node, escape)
    std::vector<std::pair<uint32_t, uint32_t>> stack;
    stack.push_back({ 0u, uint32_t(n) });
    uint64_t visited = 0;
    while (!stack.empty()) {
        auto [k, esc] = stack.back();
        stack.pop_back();
        if (++visited > n)
            return fail(PT_ERR_INVALID_ARGUMENT, "acceleration structure is not a tree");
        escape[k] = esc;
        uint64_t link = nodes[k].link;
        if (link % 2 == 1) {
            uint32_t c2 = uint32_t(link / 2);
            stack.push_back({ c2, esc });
            stack.push_back({ k + 1, c2 });
        }
    }
    if (visited != n)
        return fail(PT_ERR_INVALID_ARGUMENT, "acceleration structure has unreachable nodes");
    for (uint64_t k = 0; k < n; k++) {
        const pt_bvh_node& nd = nodes[k];
        uint32_t leaf = (nd.link % 2 == 0) ? uint32_t(nd.link / 2) : (kInnerBit | uint32_t(nd.link / 2));
        out[2 * k + 0] = make_float4(nd.lo[0], nd.lo[1], nd.lo[2], nd.hi[0]);
        out[2 * k + 1] = make_float4(nd.hi[1], nd.hi[2], __builtin_bit_cast(float, escape[k]), __builtin_bit_cast(float, leaf));
    }
    return PT_OK;
}

// grid coordinate at or below / at or above a value, checked with the expression the device decodes it with
bool gridCoordBelow(float base, float step, float v, uint32_t& q)
{
    if (!std::isfinite(v) || v < base)
        return false;
    double est = std::floor((double(v) - double(base)) / double(step));
    int64_t k = int64_t(std::min(std::max(est, 0.0), 65535.0));
    while (k > 0 && std::fmaf(float(k), step, base) > v)
        k--;
    while (k < 65535 && std::fmaf(float(k + 1), step, base) <= v)
        k++;
    if (std::fmaf(float(k), step, base) > v)
        return false;
    q = uint32_t(k);
    return true;
}

bool gridCoordAbove(float base, float step, float v, uint32_t& q)
{
    if (!std::isfinite(v))
        return false;
    double est = std::ceil((double(v) - double(base)) / double(step));
    int64_t k = int64_t(std::min(std::max(est, 0.0), 65535.0));
    while (k < 65535 && std::fmaf(float(k), step, base) < v)
        k++;
    while (k > 0 && std::fmaf(float(k - 1), step, base) >= v)
        k--;
    if (std::fmaf(float(k), step, base) < v)
        return false;
    q = uint32_t(k);
    return true;
}

// The table of the nearer-child-first walk: per inner node (numbered in storage order among the
// inner nodes) one 32-byte row with the boxes of both children on a 16-bit grid over the root
// box and the children's links (see DevScene::wide).  A grid box always contains the exact one:
// the corner values are checked with the very expression the device decodes them with.  The
// walk uses grid boxes only to skip and to order subtrees; whether a surface is tested, and the
// entry distance that goes into the rule for when a result stands, come from the leaf's exact
// box -- that of a static triangle's vertices where the two agree bit for bit (checked here),
// else the leaf's row of `nodes`.
// Returns false if the boxes cannot be put on a grid (non-finite, or outside the root box).
bool buildWideNodes(const pt_bvh_node* nodes, uint64_t n, const std::vector<float4>& surf, std::vector<uint4>& out,
                    float base[3], float step[3])
{
    out.clear();
    for (int a = 0; a < 3; a++) {
        base[a] = 0.0f;
        step[a] = 1.0f;
    }
    if (n == 0)
        return true;
    if (n >= kLeafRowBit)
        return false;
    for (int a = 0; a < 3; a++) {
        const float lo = nodes[0].lo[a], hi = nodes[0].hi[a];
        if (!std::isfinite(lo) || !std::isfinite(hi) || !(lo <= hi))
            return false;
        base[a] = lo;
        float st = (hi - lo) / 65535.0f;
        if (!std::isfinite(st))
            return false;
        if (!(st > 1.0e-30f))
            st = 1.0e-30f;
        for (int guard = 0; guard < 64 && std::fmaf(65535.0f, st, lo) < hi; guard++)
            st = std::nextafterf(st, std::numeric_limits<float>::infinity());
        if (std::fmaf(65535.0f, st, lo) < hi)
            return false;
        step[a] = st;
    }
    auto below = [&](int a, float v, uint32_t& q) { return gridCoordBelow(base[a], step[a], v, q); };
    auto above = [&](int a, float v, uint32_t& q) { return gridCoordAbove(base[a], step[a], v, q); };
    std::vector<uint32_t> innerNumber(n, 0);
    uint32_t nInner = 0;
    for (uint64_t k = 0; k < n; k++)
        if (nodes[k].link % 2 == 1)
            innerNumber[k] = nInner++;
    auto linkOf = [&](uint64_t c) -> uint32_t {
        if (nodes[c].link % 2 == 1)
            return kInnerBit | innerNumber[c];
        const uint64_t sid = nodes[c].link / 2;
        if (3 * sid + 2 < surf.size() && __builtin_bit_cast(uint32_t, surf[3 * sid].w) == kTriStatic) {
            const float4 &A = surf[3 * sid], &B = surf[3 * sid + 1], &C = surf[3 * sid + 2];
            const float lo[3] = { std::min(A.x, std::min(B.x, C.x)), std::min(A.y, std::min(B.y, C.y)),
                                  std::min(A.z, std::min(B.z, C.z)) };
            const float hi[3] = { std::max(A.x, std::max(B.x, C.x)), std::max(A.y, std::max(B.y, C.y)),
                                  std::max(A.z, std::max(B.z, C.z)) };
            if (std::memcmp(lo, nodes[c].lo, 12) == 0 && std::memcmp(hi, nodes[c].hi, 12) == 0)
                return uint32_t(sid);
        }
        return kLeafRowBit | uint32_t(c);
    };
    out.assign(2 * size_t(nInner), make_uint4(0, 0, 0, 0));
    for (uint64_t k = 0; k < n; k++) {
        if (nodes[k].link % 2 == 0)
            continue;
        const uint64_t child[2] = { k + 1, nodes[k].link / 2 };
        uint32_t w[6];
        for (int c = 0; c < 2; c++) {
            uint32_t qlo[3], qhi[3];
            for (int a = 0; a < 3; a++)
                if (!below(a, nodes[child[c]].lo[a], qlo[a]) || !above(a, nodes[child[c]].hi[a], qhi[a]))
                    return false;
            w[3 * c + 0] = qlo[0] | (qlo[1] << 16);
            w[3 * c + 1] = qlo[2] | (qhi[0] << 16);
            w[3 * c + 2] = qhi[1] | (qhi[2] << 16);
        }
        const size_t r = 2 * size_t(innerNumber[k]);
        out[r + 0] = make_uint4(w[0], w[1], w[2], w[3]);
        out[r + 1] = make_uint4(w[4], w[5], linkOf(child[0]), linkOf(child[1]));
    }
    return true;
}

// The table of the four-wide form of that walk (DevScene::quad), from the device's node table.  A row stands for an
// inner node and holds up to four of its descendants: beginning with its two children, the inner node with the
// largest box surface among them is replaced by its own two children until there are four (or only leaves): the
// boxes a ray is most likely to enter anyway are the ones not worth a step of their own.  Rows are numbered in
// the order of the nodes they stand for, so a row's children rows come after it.  Same grid, same rounding and same
// three kinds of link as the two-wide rows; a slot that is not used holds kQuadNoLink and a box nothing enters.
// Returns false if a box cannot be put on the grid.
bool buildQuadRows(const float4* nodes, uint64_t n, const std::vector<float4>& surf, const float base[3], const float step[3],
                   std::vector<uint4>& out)
{
    out.clear();
    if (n == 0 || n >= kLeafRowBit)
        return n == 0;
    auto leafWord = [&](uint64_t k) { return __builtin_bit_cast(uint32_t, nodes[2 * k + 1].w); };
    auto lo = [&](uint64_t k, int a) { return a == 0 ? nodes[2 * k].x : a == 1 ? nodes[2 * k].y : nodes[2 * k].z; };
    auto hi = [&](uint64_t k, int a) { return a == 0 ? nodes[2 * k].w : a == 1 ? nodes[2 * k + 1].x : nodes[2 * k + 1].y; };
    auto area = [&](uint64_t k) {
        const double dx = double(hi(k, 0)) - double(lo(k, 0)), dy = double(hi(k, 1)) - double(lo(k, 1)), dz = double(hi(k, 2)) - double(lo(k, 2));
        return dx * dy + dy * dz + dz * dx;
    };
    if (!(leafWord(0) & kInnerBit))
        return true; // a single leaf: no row is ever read
    // pass 1: which nodes get rows, their numbers, their children
    struct Row {
        uint32_t node, child[4], nChildren;
    };
    std::vector<Row> rows;
    std::vector<uint32_t> rowOf(n, 0xFFFFFFFFu), todo;
    todo.push_back(0);
    while (!todo.empty()) {
        const uint32_t k = todo.back();
        todo.pop_back();
        Row r;
        r.node = k;
        r.child[0] = k + 1;
        r.child[1] = leafWord(k) & ~kInnerBit;
        r.nChildren = 2;
        while (r.nChildren < 4) {
            int pick = -1;
            double pickArea = -1.0;
            for (uint32_t c = 0; c < r.nChildren; c++)
                if ((leafWord(r.child[c]) & kInnerBit) && area(r.child[c]) > pickArea) {
                    pick = int(c);
                    pickArea = area(r.child[c]);
                }
            if (pick < 0)
                break;
            const uint32_t parent = r.child[pick];
            r.child[pick] = parent + 1;
            r.child[r.nChildren++] = leafWord(parent) & ~kInnerBit;
        }
        for (uint32_t c = 0; c < r.nChildren; c++)
            if (r.child[c] >= n)
                return false;
        rows.push_back(r);
        for (int c = int(r.nChildren) - 1; c >= 0; c--)
            if (leafWord(r.child[c]) & kInnerBit)
                todo.push_back(r.child[c]);
    }
    // rows in the order of the nodes they stand for
    std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.node < b.node; });
    for (size_t i = 0; i < rows.size(); i++)
        rowOf[rows[i].node] = uint32_t(i);
    // pass 2: the rows
    out.assign(4 * rows.size(), make_uint4(0, 0, 0, 0));
    for (size_t i = 0; i < rows.size(); i++) {
        const Row& r = rows[i];
        uint32_t w[12], link[4];
        for (uint32_t c = 0; c < 4; c++) {
            if (c >= r.nChildren) {
                w[3 * c + 0] = 0xFFFFFFFFu; // lo.x = lo.y = 65535
                w[3 * c + 1] = 0x0000FFFFu; // lo.z = 65535, hi.x = 0
                w[3 * c + 2] = 0u;          // hi.y = hi.z = 0
                link[c] = kQuadNoLink;
                continue;
            }
            const uint64_t at = r.child[c];
            uint32_t qlo[3], qhi[3];
            for (int a = 0; a < 3; a++)
                if (!gridCoordBelow(base[a], step[a], lo(at, a), qlo[a]) || !gridCoordAbove(base[a], step[a], hi(at, a), qhi[a]))
                    return false;
            w[3 * c + 0] = qlo[0] | (qlo[1] << 16);
            w[3 * c + 1] = qlo[2] | (qhi[0] << 16);
            w[3 * c + 2] = qhi[1] | (qhi[2] << 16);
            const uint32_t l = leafWord(at);
            if (l & kInnerBit) {
                if (rowOf[at] == 0xFFFFFFFFu || rowOf[at] <= i)
                    return false;
                link[c] = kInnerBit | rowOf[at];
            } else {
                link[c] = kLeafRowBit | uint32_t(at);
                if (3 * size_t(l) + 2 < surf.size() && __builtin_bit_cast(uint32_t, surf[3 * size_t(l)].w) == kTriStatic) {
                    const float4 &A = surf[3 * size_t(l)], &B = surf[3 * size_t(l) + 1], &C = surf[3 * size_t(l) + 2];
                    const float tlo[3] = { std::min(A.x, std::min(B.x, C.x)), std::min(A.y, std::min(B.y, C.y)),
                                           std::min(A.z, std::min(B.z, C.z)) };
                    const float thi[3] = { std::max(A.x, std::max(B.x, C.x)), std::max(A.y, std::max(B.y, C.y)),
                                           std::max(A.z, std::max(B.z, C.z)) };
                    bool same = true;
                    for (int a = 0; a < 3; a++)
                        same = same && __builtin_bit_cast(uint32_t, tlo[a]) == __builtin_bit_cast(uint32_t, lo(at, a))
                            && __builtin_bit_cast(uint32_t, thi[a]) == __builtin_bit_cast(uint32_t, hi(at, a));
                    if (same)
                        link[c] = l;
                }
            }
        }
        out[4 * i + 0] = make_uint4(w[0], w[1], w[2], w[3]);
        out[4 * i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        out[4 * i + 2] = make_uint4(w[8], w[9], w[10], w[11]);
        out[4 * i + 3] = make_uint4(link[0], link[1], link[2], link[3]);
    }
    return true;
}

int checkRender(const pt_scene* scene, const pt_camera* camera, const pt_render_params* p, const void* out, RenderArgs& a)
{
    if (!scene || !camera || !p || !out)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (p->width <= 0 || p->height <= 0 || p->sqrt_spp <= 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "width, height and sqrt_spp must be positive");
    if (p->sqrt_spp > 46340 || uint64_t(p->width) * uint64_t(p->height) > 0x7FFFFFFFull)
        return fail(PT_ERR_LIMIT, "image or sample count too large");
    int x0 = p->tile_x0, y0 = p->tile_y0, x1 = p->tile_x1, y1 = p->tile_y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) {
        x1 = p->width;
        y1 = p->height;
    }
    int s0 = p->sample_begin, s1 = p->sample_end;
    if (s0 == 0 && s1 == 0)
        s1 = p->sqrt_spp * p->sqrt_spp;
    if (x0 < 0 || y0 < 0 || x1 > p->width || y1 > p->height || x0 > x1 || y0 > y1)
        return fail(PT_ERR_INVALID_ARGUMENT, "tile rectangle outside the image");
    if (s0 < 0 || s1 > p->sqrt_spp * p->sqrt_spp || s0 > s1)
        return fail(PT_ERR_INVALID_ARGUMENT, "sample range outside [0, sqrt_spp^2]");
    // (whether the scene has a tree is looked at under the scene's lock, by the caller: requireTree)
    // (the path records keep the segment number in 12 bits and the number of random draws, at most eight per
    // segment, in 16; the reference stops at 128 segments)
    if (p->max_path_segments > 4095)
        return fail(PT_ERR_LIMIT, "max_path_segments above 4095");
    if (camera->has_animation
        && (camera->animation.type < PT_ANIM_IDENTITY || camera->animation.type > PT_ANIM_TWOKEY))
        return fail(PT_ERR_UNSUPPORTED, "camera animation has an unknown type");
    a.cam = *camera;
    a.p = *p;
    a.x0 = x0;
    a.y0 = y0;
    a.tw = x1 - x0;
    a.th = y1 - y0;
    a.s0 = s0;
    a.s1 = s1;
    a.pp.maxSegments = p->max_path_segments > 0 ? p->max_path_segments : 128;
    a.pp.rrStart = p->rr_start_segment == 0 ? 5 : (p->rr_start_segment < 0 ? 0 : p->rr_start_segment);
    a.pp.traceNullShadowRays = std::getenv("PT_TRACE_NULL_SHADOW_RAYS") != nullptr;
    return PT_OK;
}

// With the scene's lock held (SceneCall): a tree that another thread is rebuilding right now is waited for by taking
// the lock, not reported as missing.
int requireTree(const pt_scene* scene)
{
    if (scene->dev.n_nodes == 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "scene has no acceleration structure yet (pt_scene_build_bvh_device or pt_scene_update_bvh)");
    return PT_OK;
}

bool useMegakernel()
{
    const char* v = std::getenv("PT_RENDERER");
    return v && std::strcmp(v, "mega") == 0;
}

// The per-pixel megakernel: kept for instrumentation (work counters) and as the
// comparison point for the wavefront renderer.
int launchMegakernel(pt_scene* scene, const RenderArgs& a, float* d_out, hipStream_t stream, unsigned long long* d_counters)
{
    if (a.tw == 0 || a.th == 0)
        return PT_OK;
    const int tilesX = (a.tw + 15) / 16, tilesY = (a.th + 15) / 16;
    dim3 grid(unsigned(tilesX) * unsigned(tilesY)), block(256);
    HIP_TRY(hipEventRecord(scene->evStart, stream));
    if (d_counters)
        hipLaunchKernelGGL(renderPixelsKernel<true>, grid, block, 0, stream, scene->dev, a, d_out, d_counters);
    else
        hipLaunchKernelGGL(renderPixelsKernel<false>, grid, block, 0, stream, scene->dev, a, d_out, d_counters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(scene->evStop, stream));
    scene->timed = true;
    return PT_OK;
}

int launchRender(pt_scene* scene, const RenderArgs& a, float* d_out, hipStream_t stream, unsigned long long* d_counters)
{
    if (d_counters || useMegakernel())
        return launchMegakernel(scene, a, d_out, stream, d_counters);
    return ptWavefrontRender(scene, a, d_out, stream);
}

} // namespace

int ptBuildQuadRows(pt_scene* scene, hipStream_t stream)
{
    return ptBuildQuadRowsDevice(scene, stream);
}

extern "C" {

int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

void pt_scene_destroy(pt_scene* scene)
{
    if (!scene)
        return;
    DeviceGuard g(scene->device);
    if (scene->pending)
        (void)hipEventSynchronize(scene->evDone);
    ptWavefrontRelease(scene);
    ptMultiRelease(scene);
    for (void* p : scene->owned)
        (void)hipFree(p);
    if (scene->buildArea)
        (void)hipFree(scene->buildArea);
    if (scene->quadArea)
        (void)hipFree(scene->quadArea);
    if (scene->evStart)
        (void)hipEventDestroy(scene->evStart);
    if (scene->evStop)
        (void)hipEventDestroy(scene->evStop);
    if (scene->evDone)
        (void)hipEventDestroy(scene->evDone);
    delete scene;
}

int pt_scene_create(const pt_scene_desc* d, int device, pt_scene** out_scene)
try {
    if (!d || !out_scene)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *out_scene = nullptr;
    int rc = pt_scene_desc_validate(d);
    if (rc != PT_OK)
        return rc;
    // (no nodes at all: the tree is to be built on the device, pt_scene_build_bvh_device, before the first render)
    if (d->n_surfaces == 0 || (d->n_bvh_nodes != 0 && d->n_bvh_nodes != 2 * d->n_surfaces - 1))
        return fail(PT_ERR_INVALID_ARGUMENT, "scene needs at least one surface and a built acceleration structure or none");
    if (device < 0 || device >= pt_device_count())
        return fail(PT_ERR_DEVICE, "no such GPU: " + std::to_string(device));
    DeviceGuard g(device);
    if (!g.ok)
        return fail(PT_ERR_DEVICE, "cannot select GPU " + std::to_string(device));

    // host-side preparation: device nodes, per-surface rows, mesh table
    std::vector<float4> nodes;
    if (d->n_bvh_nodes > 0) {
        rc = buildDeviceNodes(d->bvh_nodes, d->n_bvh_nodes, nodes);
        if (rc != PT_OK)
            return rc;
    }
    nodes.resize(2 * (2 * size_t(d->n_surfaces) - 1), make_float4(0.0f, 0.0f, 0.0f, 0.0f)); // room for any tree

    std::vector<DevMesh> meshes(d->n_meshes);
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        const pt_mesh& m = d->meshes[i];
        DevMesh& dm = meshes[i];
        std::memset(&dm, 0, sizeof(dm));
        dm.first_vertex = m.first_vertex;
        dm.first_index = m.first_index;
        dm.material = m.material;
        dm.animation = m.animation;
        dm.flags = (m.has_normals ? 1u : 0u) | (m.has_texcoords ? 2u : 0u) | (m.has_tangents ? 4u : 0u);
    }

    auto bitsf = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    // the type of a surface's material travels with the surface (shading work is ordered by it)
    auto matClassOf = [&](int32_t material) {
        return uint32_t(std::min<int32_t>(std::max<int32_t>(d->materials[material].type, 0), 7));
    };
    // what the material of a surface will ask of a hit record: texture coordinates (16) if it looks up a texture
    // that is not constant, the tangent (32) if it has a normal map; a two-sided material asks what its sides ask
    std::function<uint32_t(int32_t, int)> materialNeeds = [&](int32_t material, int depth) -> uint32_t {
        const pt_material& m = d->materials[material];
        auto varies = [&](int k) { return m.tex[k] >= 0 && d->textures[m.tex[k]].type != PT_TEX_CONSTANT; };
        switch (m.type) {
        case PT_MAT_LAMBERTIAN:
        case PT_MAT_MIRROR:
            return varies(0) ? 16u : 0u;
        case PT_MAT_PHONG:
            return ((varies(0) || varies(1) || varies(2) || m.tex[3] >= 0 || m.tex[4] >= 0) ? 16u : 0u) | (m.tex[4] >= 0 ? 32u : 0u);
        case PT_MAT_TWOSIDED:
            return depth < 10 ? (materialNeeds(m.front, depth + 1) | materialNeeds(m.back, depth + 1)) : 48u;
        default:
            return 0u;
        }
    };
    bool hasDynamic = false;
    std::vector<float4> surf(3 * d->n_surfaces);
    std::vector<uint4> surfAttr(d->n_surfaces);
    std::vector<float4> faceN(d->n_surfaces, make_float4(0.0f, 0.0f, 0.0f, -1.0f));
    if (d->n_materials >= (1u << 24))
        return fail(PT_ERR_LIMIT, "more than 2^24 materials");
    for (uint64_t i = 0; i < d->n_surfaces; i++) {
        const pt_surface& s = d->surfaces[i];
        if (s.type == PT_SURF_TRIANGLE) {
            const pt_mesh& m = d->meshes[s.owner];
            const uint32_t* idx = d->indices + m.first_index + 3 * size_t(s.index);
            V3 A = v3(d->positions + 3 * size_t(m.first_vertex + idx[0]));
            V3 B = v3(d->positions + 3 * size_t(m.first_vertex + idx[1]));
            V3 C = v3(d->positions + 3 * size_t(m.first_vertex + idx[2]));
            uint32_t kind = kTriStatic;
            if (m.animation >= 0) {
                const pt_animation& an = d->animations[m.animation];
                if (an.type == PT_ANIM_TWOKEY) {
                    kind = kTriDynamic;
                    hasDynamic = true;
                } else {
                    // a time-independent pose: apply it once, with the arithmetic of
                    // Transformation::operator* (reference src/15/transformation.hpp:34-37)
                    Xform T = an.type == PT_ANIM_CONSTANT ? xformOf(an.T0) : xformIdentity();
                    A = apply(T, A);
                    B = apply(T, B);
                    C = apply(T, C);
                }
            }
            surf[3 * i + 0] = make_float4(A.x, A.y, A.z, bitsf(kind));
            if (uint32_t(s.owner) > kMeshMask)
                return fail(PT_ERR_LIMIT, "more than 2^24 meshes");
            surf[3 * i + 1] = make_float4(B.x, B.y, B.z, bitsf(uint32_t(s.owner) | (matClassOf(m.material) << 24)));
            surf[3 * i + 2] = make_float4(C.x, C.y, C.z, bitsf(s.index));
            uint32_t flags = (m.has_normals ? 1u : 0u) | (m.has_texcoords ? 2u : 0u) | (m.has_tangents ? 4u : 0u)
                | (m.animation >= 0 ? 8u : 0u) | materialNeeds(m.material, 0);
            // (a mesh with a constant pose has been posed above; what its pose would still turn when shading are
            // vertex normals and tangents, so it qualifies if it has neither)
            if (kind == kTriStatic && !m.has_normals && (m.animation < 0 || !m.has_tangents)) {
                // the normal of the triangle's plane and its area, with the operations of the hit record
                // (reference src/15/surface_triangle.hpp:151-156) and of Surface::p (:201-204)
                const V3 edgeCross = cross(B - A, C - A);
                const float edgeCrossLength = ::sqrtf(dot(edgeCross, edgeCross));
                const V3 n = edgeCross / edgeCrossLength;
                faceN[i] = make_float4(n.x, n.y, n.z, 0.5f * edgeCrossLength);
                flags |= 64u;
            }
            surfAttr[i] = make_uint4(m.first_vertex + idx[0], m.first_vertex + idx[1], m.first_vertex + idx[2],
                                     uint32_t(m.material) | (flags << 24));
        } else {
            const pt_sphere& q = d->spheres[s.owner];
            V3 c = v3(q.center);
            float r = q.radius;
            uint32_t kind = kSphStatic;
            if (q.animation >= 0) {
                const pt_animation& an = d->animations[q.animation];
                if (an.type == PT_ANIM_TWOKEY) {
                    kind = kSphDynamic;
                    hasDynamic = true;
                } else {
                    Xform T = an.type == PT_ANIM_CONSTANT ? xformOf(an.T0) : xformIdentity();
                    c = apply(T, c);
                    r *= T.scaling.x;
                }
            }
            surf[3 * i + 0] = make_float4(c.x, c.y, c.z, bitsf(kind));
            // (radius, sphere number, animation number -- what the test of a moving sphere looks up --, type of material)
            surf[3 * i + 1] = make_float4(r, bitsf(s.owner), bitsf(uint32_t(q.animation)), bitsf(matClassOf(q.material) << 24));
            surf[3 * i + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            surfAttr[i] = make_uint4(s.owner, 0u, 0u, uint32_t(q.material) | (materialNeeds(q.material, 0) << 24));
        }
    }

    // constant textures of the first three texture slots, resolved once
    std::vector<float4> matConst(2 * d->n_materials);
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const pt_material& m = d->materials[i];
        uint32_t mask = 0;
        float c[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
        const int slots = m.type == PT_MAT_PHONG ? 3 : ((m.type == PT_MAT_LAMBERTIAN || m.type == PT_MAT_MIRROR) ? 1 : 0);
        for (int k = 0; k < slots; k++) {
            if (m.tex[k] >= 0 && d->textures[m.tex[k]].type == PT_TEX_CONSTANT) {
                mask |= 1u << k;
                for (int j = 0; j < 3; j++)
                    c[k][j] = d->textures[m.tex[k]].v[j];
            }
        }
        matConst[2 * i + 0] = make_float4(c[0][0], c[0][1], c[0][2], bitsf(mask));
        matConst[2 * i + 1] = make_float4(c[1][0], c[1][1], c[1][2], c[2][0]);
    }

    pt_scene* sc = new pt_scene;
    sc->device = device;
    sc->n_surfaces = d->n_surfaces;
    sc->nMaterials = uint32_t(std::min<uint64_t>(d->n_materials, 0xFFFFFFFFu));
    sc->nTextures = uint32_t(std::min<uint64_t>(d->n_textures, 0xFFFFFFFFu));
    sc->nMeshes = uint32_t(std::min<uint64_t>(d->n_meshes, 0xFFFFFFFFu));
    sc->hasDynamic = hasDynamic;
    std::memset(&sc->dev, 0, sizeof(sc->dev));
    DevScene& dev = sc->dev;
    auto bail = [&](int code) {
        pt_scene_destroy(sc);
        return code;
    };
    float4* dNodes = nullptr;
    uint4* dWide = nullptr;
    float4* dSurf = nullptr;
    uint4* dSurfAttr = nullptr;
    float4* dMatConst = nullptr;
    float4 *dFaceN = nullptr, *dLightGeo = nullptr;
    float *dPos = nullptr, *dNrm = nullptr, *dTc = nullptr, *dTng = nullptr, *dTexels = nullptr;
    uint32_t *dIdx = nullptr, *dLights = nullptr;
    DevMesh* dMeshes = nullptr;
    pt_sphere* dSpheres = nullptr;
    pt_material* dMats = nullptr;
    pt_texture* dTex = nullptr;
    pt_animation* dAnims = nullptr;
    if ((rc = upload(nodes.data(), nodes.size(), &dNodes, sc->owned)) != PT_OK) return bail(rc);
    // (room for a row per inner node even if this tree's boxes do not fit the grid: a later tree's may)
    std::vector<uint4> wide;
    const bool haveWide = buildWideNodes(d->bvh_nodes, d->n_bvh_nodes, surf, wide, dev.qbase, dev.qstep);
    wide.resize(2 * size_t(d->n_surfaces > 0 ? d->n_surfaces - 1 : 0), make_uint4(0, 0, 0, 0));
    if ((rc = upload(wide.data(), wide.size(), &dWide, sc->owned)) != PT_OK) return bail(rc);
    sc->hostSurf = surf;
    if ((rc = upload(surf.data(), surf.size(), &dSurf, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(surfAttr.data(), surfAttr.size(), &dSurfAttr, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(matConst.data(), matConst.size(), &dMatConst, sc->owned)) != PT_OK) return bail(rc);
    std::vector<float4> lightGeo(d->n_lights);
    for (uint64_t i = 0; i < d->n_lights; i++)
        lightGeo[i] = faceN[d->lights[i]];
    if ((rc = upload(faceN.data(), faceN.size(), &dFaceN, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(lightGeo.data(), lightGeo.size(), &dLightGeo, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->positions, 3 * d->n_vertices, &dPos, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->normals, 3 * d->n_vertices, &dNrm, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->texcoords, 2 * d->n_vertices, &dTc, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->tangents, 3 * d->n_vertices, &dTng, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->indices, d->n_indices, &dIdx, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(meshes.data(), meshes.size(), &dMeshes, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->spheres, d->n_spheres, &dSpheres, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->lights, d->n_lights, &dLights, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->materials, d->n_materials, &dMats, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->textures, d->n_textures, &dTex, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->texels, d->n_texels, &dTexels, sc->owned)) != PT_OK) return bail(rc);
    if ((rc = upload(d->animations, d->n_animations, &dAnims, sc->owned)) != PT_OK) return bail(rc);
    sc->nodes = dNodes;
    sc->wide = dWide;
    dev.nodes = dNodes;
    dev.wide = haveWide ? dWide : nullptr;
    dev.surf = dSurf;
    dev.surfAttr = dSurfAttr;
    dev.matConst = dMatConst;
    dev.positions = dPos;
    dev.normals = dNrm;
    dev.texcoords = dTc;
    dev.tangents = dTng;
    dev.indices = dIdx;
    dev.meshes = dMeshes;
    dev.spheres = dSpheres;
    dev.lights = dLights;
    dev.faceN = dFaceN;
    dev.lightGeo = dLightGeo;
    dev.materials = dMats;
    dev.textures = dTex;
    dev.texels = dTexels;
    dev.animations = dAnims;
    dev.n_animations = uint32_t(d->n_animations);
    dev.n_nodes = uint32_t(d->n_bvh_nodes);
    dev.n_surfaces = uint32_t(d->n_surfaces);
    dev.n_lights = uint32_t(d->n_lights);
    dev.envmap = d->envmap;
    // (the address of this GPU's copy of the fixed work area, for pt_render_device_noalloc: looked up here, at
    // set-up, so that the render call itself finds it in a table)
    if ((rc = ptWavefrontStaticSetup(device)) != PT_OK) return bail(rc);
    if (hipDeviceGetAttribute(&sc->computeUnits, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || sc->computeUnits <= 0)
        sc->computeUnits = 256;
    if (hipEventCreate(&sc->evStart) != hipSuccess || hipEventCreate(&sc->evStop) != hipSuccess
        || hipEventCreateWithFlags(&sc->evDone, hipEventDisableTiming) != hipSuccess) {
        pt_scene_destroy(sc);
        return fail(PT_ERR_DEVICE, "cannot create timing events");
    }
    if ((rc = ptWavefrontBuildWalkLinks(sc, 0.0f, nullptr)) != PT_OK) return bail(rc);
    if ((rc = ptBuildQuadRows(sc, nullptr)) != PT_OK) return bail(rc);
    *out_scene = sc;
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_bvh_quad_rows(const pt_bvh_node* nodes, uint64_t n_nodes, uint32_t* out_rows, uint64_t capacity_rows,
                     uint64_t* out_n_rows, float out_base[3], float out_step[3])
try {
    if (!nodes || !out_rows || !out_n_rows || !out_base || !out_step)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_nodes == 0 || n_nodes % 2 == 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "node count is not 2 * surfaces - 1");
    for (uint64_t i = 0; i < n_nodes; i++) {
        const uint64_t link = nodes[i].link;
        if (link % 2 == 0 ? link / 2 >= (n_nodes + 1) / 2 : (link / 2 <= i + 1 || link / 2 >= n_nodes))
            return fail(PT_ERR_INVALID_ARGUMENT, "node " + std::to_string(i) + " has a bad link");
    }
    std::vector<float4> dn;
    const int rc = buildDeviceNodes(nodes, n_nodes, dn);
    if (rc != PT_OK)
        return rc;
    std::vector<uint4> two, rows;
    const std::vector<float4> noSurfaces;
    if (!buildWideNodes(nodes, n_nodes, noSurfaces, two, out_base, out_step)
        || !buildQuadRows(dn.data(), n_nodes, noSurfaces, out_base, out_step, rows))
        return fail(PT_ERR_INVALID_ARGUMENT, "the boxes do not fit a grid over the root box");
    *out_n_rows = rows.size() / 4;
    if (rows.size() / 4 > capacity_rows)
        return fail(PT_ERR_LIMIT, "more rows than the output holds");
    if (!rows.empty())
        std::memcpy(out_rows, rows.data(), rows.size() * sizeof(uint4));
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_scene_update_bvh(pt_scene* scene, const pt_bvh_node* nodes, uint64_t n_nodes)
try {
    if (!scene || !nodes)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_nodes != 2 * scene->n_surfaces - 1)
        return fail(PT_ERR_INVALID_ARGUMENT, "node count is not 2 * surfaces - 1");
    for (uint64_t i = 0; i < n_nodes; i++) {
        uint64_t link = nodes[i].link;
        if (link % 2 == 0 ? link / 2 >= scene->n_surfaces : (link / 2 <= i + 1 || link / 2 >= n_nodes))
            return fail(PT_ERR_INVALID_ARGUMENT, "node " + std::to_string(i) + " has a bad link");
    }
    std::vector<float4> dn;
    int rc = buildDeviceNodes(nodes, n_nodes, dn);
    if (rc != PT_OK)
        return rc;
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(scene->nodes, dn.data(), dn.size() * sizeof(float4), hipMemcpyHostToDevice));
    scene->dev.n_nodes = uint32_t(n_nodes);
    std::vector<uint4> wide;
    if (buildWideNodes(nodes, n_nodes, scene->hostSurf, wide, scene->dev.qbase, scene->dev.qstep)) {
        if (!wide.empty())
            HIP_TRY(hipMemcpy(scene->wide, wide.data(), wide.size() * sizeof(uint4), hipMemcpyHostToDevice));
        scene->dev.wide = scene->wide;
    } else {
        scene->dev.wide = nullptr;
    }
    rc = ptWavefrontBuildWalkLinks(scene, 0.0f, nullptr);
    if (rc == PT_OK)
        rc = ptBuildQuadRows(scene, nullptr);
    if (rc != PT_OK) {
        scene->dev.n_nodes = 0;
        return rc;
    }
    HIP_TRY(call.done(nullptr));
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_render_device(pt_scene* scene, const pt_camera* camera, const pt_render_params* params, float* d_out_rgb,
                     void* hip_stream)
{
    RenderArgs a;
    int rc = checkRender(scene, camera, params, d_out_rgb, a);
    if (rc != PT_OK)
        return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    SceneCall call(scene, stream);
    HIP_TRY(call.status);
    if ((rc = requireTree(scene)) != PT_OK)
        return rc;
    rc = launchRender(scene, a, d_out_rgb, stream, nullptr);
    HIP_TRY(call.done(stream));
    return rc;
}

int pt_render_device_noalloc(pt_scene* scene, const pt_camera* camera, const pt_render_params* params, float* d_out_rgb,
                             void* hip_stream)
{
    RenderArgs a;
    int rc = checkRender(scene, camera, params, d_out_rgb, a);
    if (rc != PT_OK)
        return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    SceneCall call(scene, stream);
    HIP_TRY(call.status);
    if ((rc = requireTree(scene)) != PT_OK)
        return rc;
    rc = ptWavefrontRenderStatic(scene, a, d_out_rgb, stream);
    HIP_TRY(call.done(stream));
    return rc;
}

static int renderToHost(pt_scene* scene, const pt_camera* camera, const pt_render_params* params, float* out_rgb,
                        pt_counters* out_counters)
{
    RenderArgs a;
    int rc = checkRender(scene, camera, params, out_rgb, a);
    if (rc != PT_OK)
        return rc;
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    if ((rc = requireTree(scene)) != PT_OK)
        return rc;
    const size_t bytes = size_t(3) * a.p.width * a.p.height * sizeof(float);
    float* d_out = nullptr;
    unsigned long long* d_cnt = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), bytes));
    // pixels outside the tile keep the caller's values
    hipError_t e = hipMemcpy(d_out, out_rgb, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && out_counters) {
        e = hipMalloc(reinterpret_cast<void**>(&d_cnt), 6 * sizeof(unsigned long long));
        if (e == hipSuccess)
            e = hipMemset(d_cnt, 0, 6 * sizeof(unsigned long long));
    }
    if (e == hipSuccess) {
        rc = launchRender(scene, a, d_out, nullptr, d_cnt);
        if (rc == PT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess)
                e = hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost);
            if (e == hipSuccess && out_counters) {
                unsigned long long c[6];
                e = hipMemcpy(c, d_cnt, sizeof(c), hipMemcpyDeviceToHost);
                out_counters->path_samples = uint64_t(a.tw) * a.th * (a.s1 - a.s0);
                out_counters->rays = c[0];
                out_counters->node_visits = c[1];
                out_counters->leaf_tests = c[2];
                out_counters->light_tests = c[3];
                out_counters->vertices = c[4];
                out_counters->rng_draws = c[5];
            }
        }
    }
    (void)hipFree(d_out);
    if (d_cnt)
        (void)hipFree(d_cnt);
    if (rc != PT_OK)
        return rc;
    if (e != hipSuccess)
        return fail(PT_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_render(pt_scene* scene, const pt_camera* camera, const pt_render_params* params, float* out_rgb)
{
    return renderToHost(scene, camera, params, out_rgb, nullptr);
}

int pt_render_counted(pt_scene* scene, const pt_camera* camera, const pt_render_params* params, float* out_rgb,
                      pt_counters* out_counters)
{
    if (!out_counters)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    return renderToHost(scene, camera, params, out_rgb, out_counters);
}

float pt_last_kernel_ms(pt_scene* scene)
{
    if (!scene || !scene->timed)
        return -1.0f;
    DeviceGuard g(scene->device);
    if (hipEventSynchronize(scene->evStop) != hipSuccess)
        return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, scene->evStart, scene->evStop) != hipSuccess)
        return -1.0f;
    return ms;
}

int pt_scene_profile(pt_scene* scene, int enable)
{
    if (!scene)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_profile: null scene");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    HIP_TRY(hipDeviceSynchronize());
    scene->kt.collect();
    if (enable)
        scene->kt.reset();
    scene->kt.enabled = enable != 0;
    return PT_OK;
}

int pt_scene_kernel_times(pt_scene* scene, pt_kernel_time* out, int capacity, int* count)
{
    if (!scene || !count || (capacity > 0 && !out))
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_kernel_times: null argument");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    HIP_TRY(hipDeviceSynchronize());
    scene->kt.collect();
    static const char* const names[KernelTimes::kKinds] = { "generateKernel", "traceKernel", "shadeHitsKernel",
                                                            "resolveKernel", "classifyKernel", "traceOrderedKernel",
                                                            "traceRetryKernel", "finishPathsKernel" };
    *count = KernelTimes::kKinds;
    for (int k = 0; k < KernelTimes::kKinds && k < capacity; k++) {
        std::memset(&out[k], 0, sizeof(out[k]));
        std::strncpy(out[k].name, names[k], sizeof(out[k].name) - 1);
        out[k].launches = scene->kt.launches[k];
        out[k].total_ms = scene->kt.ms[k];
    }
    return PT_OK;
}

int pt_scene_count_walk(pt_scene* scene, int enable)
{
    if (!scene)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_count_walk: null scene");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    HIP_TRY(hipDeviceSynchronize());
    return ptWavefrontWalkStats(scene, enable);
}

int pt_scene_walk_counters(pt_scene* scene, pt_walk_counters* out)
{
    if (!scene || !out)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_walk_counters: null argument");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    HIP_TRY(hipDeviceSynchronize());
    return ptWavefrontReadWalkStats(scene, out);
}

// ---------------------------------------------------------------- output stage
namespace {

// reference src/15/color.hpp:36-39
PT_DEV float linearToNonlinear(float x)
{
    return x <= 0.0031308f ? (x * 12.92f) : (1.055f * ptm::powf(x, 1.0f / 2.4f) - 0.055f);
}

// reference src/15/color.hpp:66-70
PT_DEV unsigned char floatToByte(float x)
{
    x = (x < 0.0f ? 0.0f : x > 1.0f ? 1.0f : x);
    return (unsigned char)(::roundf(x * 255.0f));
}

// reference src/15/color.hpp:5-34,71-82
PT_DEV V3 rationalQuantization(V3 rgb, float maxVal, float brightness)
{
    V3 xyz = 100.0f * v3(0.412453f * rgb.x + 0.357580f * rgb.y + 0.180423f * rgb.z,
                         0.212671f * rgb.x + 0.715160f * rgb.y + 0.072169f * rgb.z,
                         0.019334f * rgb.x + 0.119193f * rgb.y + 0.950227f * rgb.z);
    float oldY = xyz.y;
    float newY = brightness * oldY / ((brightness - 1.0f) * oldY + maxVal);
    float target = 100.0f * newY;
    V3 adj = v3(0.0f);
    if (!(xyz.y <= 0.0f)) {
        float sum = xyz.x + xyz.y + xyz.z;
        if (!(sum <= 0.0f)) {
            float x = xyz.x / sum;
            float y = xyz.y / sum;
            float r = target / y;
            adj = v3(r * x, target, r * (1.0f - x - y));
        }
    }
    return 0.01f * v3(+3.240479f * adj.x - 1.537150f * adj.y - 0.498535f * adj.z,
                      -0.969256f * adj.x + 1.875991f * adj.y + 0.041556f * adj.z,
                      +0.055648f * adj.x - 0.204023f * adj.y + 1.057311f * adj.z);
}

__global__ void __launch_bounds__(256) to8BitKernel(const float* __restrict__ rgb, int width, int height, float urqMax,
                                                    float urqBrightness, int flip, unsigned char* __restrict__ out)
{
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= size_t(width) * size_t(height))
        return;
    V3 c = v3(rgb[3 * i + 0], rgb[3 * i + 1], rgb[3 * i + 2]);
    if (urqMax > 0.0f)
        c = rationalQuantization(c, urqMax, urqBrightness);
    const size_t y = i / size_t(width), x = i % size_t(width);
    const size_t o = ((flip ? size_t(height) - 1 - y : y) * size_t(width) + x) * 3;
    out[o + 0] = floatToByte(linearToNonlinear(c.x));
    out[o + 1] = floatToByte(linearToNonlinear(c.y));
    out[o + 2] = floatToByte(linearToNonlinear(c.z));
}

} // namespace

int pt_image_to_8bit_device(int device, const float* d_rgb, int width, int height, float urq_max_val,
                            float urq_brightness, int flip_rows, unsigned char* d_out, void* hip_stream)
{
    if (!d_rgb || !d_out || width <= 0 || height <= 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_image_to_8bit_device: null buffer or empty image");
    DeviceGuard g(device);
    if (!g.ok)
        return fail(PT_ERR_DEVICE, "cannot select GPU " + std::to_string(device));
    const size_t n = size_t(width) * size_t(height);
    hipLaunchKernelGGL(to8BitKernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       d_rgb, width, height, urq_max_val, urq_brightness, flip_rows, d_out);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_image_to_8bit(int device, const float* rgb, int width, int height, float urq_max_val, float urq_brightness,
                     int flip_rows, unsigned char* out)
{
    if (!rgb || !out || width <= 0 || height <= 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_image_to_8bit: null buffer or empty image");
    DeviceGuard g(device);
    if (!g.ok)
        return fail(PT_ERR_DEVICE, "cannot select GPU " + std::to_string(device));
    const size_t n = size_t(width) * size_t(height);
    float* dIn = nullptr;
    unsigned char* dOut = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dIn), n * 3 * sizeof(float)));
    if (hipMalloc(reinterpret_cast<void**>(&dOut), n * 3) != hipSuccess) {
        (void)hipFree(dIn);
        return fail(PT_ERR_DEVICE, "out of device memory");
    }
    int rc = PT_OK;
    if (hipMemcpy(dIn, rgb, n * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PT_ERR_DEVICE, "copy to the device failed");
    if (rc == PT_OK)
        rc = pt_image_to_8bit_device(device, dIn, width, height, urq_max_val, urq_brightness, flip_rows, dOut, nullptr);
    if (rc == PT_OK && hipMemcpy(out, dOut, n * 3, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(PT_ERR_DEVICE, "copy from the device failed");
    (void)hipFree(dIn);
    (void)hipFree(dOut);
    return rc;
}

int pt_render_samples(pt_scene* scene, const pt_camera* camera, const pt_render_params* params,
                      float* out_rgb_per_sample)
{
    RenderArgs a;
    int rc = checkRender(scene, camera, params, out_rgb_per_sample, a);
    if (rc != PT_OK)
        return rc;
    const uint64_t n = uint64_t(a.tw) * a.th * (a.s1 - a.s0);
    if (n == 0)
        return PT_OK;
    if (n > (uint64_t(1) << 31))
        return fail(PT_ERR_LIMIT, "too many samples for one per-sample call");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    if ((rc = requireTree(scene)) != PT_OK)
        return rc;
    if (!useMegakernel())
        return ptWavefrontSamples(scene, a, out_rgb_per_sample);
    float* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n * 3 * sizeof(float)));
    hipLaunchKernelGGL(renderSamplesKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, nullptr, scene->dev, a, n, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(out_rgb_per_sample, d_out, n * 3 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess)
        return fail(PT_ERR_DEVICE, std::string("render_samples: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_primary_hits(pt_scene* scene, const pt_camera* camera, const pt_render_params* params, pt_hit* out_hits)
{
    RenderArgs a;
    int rc = checkRender(scene, camera, params, out_hits, a);
    if (rc != PT_OK)
        return rc;
    const uint64_t n = uint64_t(a.tw) * a.th * (a.s1 - a.s0);
    if (n == 0)
        return PT_OK;
    if (n > (uint64_t(1) << 31))
        return fail(PT_ERR_LIMIT, "too many samples for one call");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    if ((rc = requireTree(scene)) != PT_OK)
        return rc;
    pt_hit* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n * sizeof(pt_hit)));
    hipLaunchKernelGGL(primaryHitsKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, nullptr, scene->dev, a, n, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(out_hits, d_out, n * sizeof(pt_hit), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess)
        return fail(PT_ERR_DEVICE, std::string("primary_hits: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_trace_rays(pt_scene* scene, uint64_t n, const float* origins, const float* directions, const float* times,
                  float amin, float amax, pt_hit* out_hits)
{
    if (!scene || !origins || !directions || !out_hits)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0)
        return PT_OK;
    if (n > (uint64_t(1) << 31))
        return fail(PT_ERR_LIMIT, "too many rays for one call");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    {
        const int rcTree = requireTree(scene);
        if (rcTree != PT_OK)
            return rcTree;
    }
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    pt_hit* d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), n * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_d), n * 3 * sizeof(float));
    if (e == hipSuccess && times) e = hipMalloc(reinterpret_cast<void**>(&d_t), n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), n * sizeof(pt_hit));
    if (e == hipSuccess) e = hipMemcpy(d_o, origins, n * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, directions, n * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && times) e = hipMemcpy(d_t, times, n * sizeof(float), hipMemcpyHostToDevice);
    int rc = PT_OK;
    if (e == hipSuccess) {
        // with the bounds of the render loop (reference src/15/pathtracer.cpp:60) the rays go through the tracing
        // kernels a render of this scene uses; other bounds, or PT_TRACE_RAYS=simple, take the one-lane-per-ray walk
        const char* how = std::getenv("PT_TRACE_RAYS");
        if (amin == 0.0001f && amax == kFltMax && !(how && std::strcmp(how, "simple") == 0)) {
            rc = ptWavefrontTraceRays(scene, n, d_o, d_d, d_t, d_out, nullptr);
        } else {
            hipLaunchKernelGGL(traceRaysKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, nullptr, scene->dev, n, d_o,
                               d_d, d_t, amin, amax, d_out);
            e = hipGetLastError();
        }
    }
    if (rc != PT_OK)
        e = hipErrorUnknown;
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out_hits, d_out, n * sizeof(pt_hit), hipMemcpyDeviceToHost);
    if (d_o) (void)hipFree(d_o);
    if (d_d) (void)hipFree(d_d);
    if (d_t) (void)hipFree(d_t);
    if (d_out) (void)hipFree(d_out);
    if (rc != PT_OK)
        return rc;
    if (e != hipSuccess)
        return fail(PT_ERR_DEVICE, std::string("trace_rays: ") + hipGetErrorString(e));
    return PT_OK;
}

} // extern "C"
