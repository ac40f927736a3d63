// pt_bvh_device.hip -- the acceleration structure built on the device.
//
// The reference builds its tree on one host thread: per node a sort of the node's surfaces
// along the longest axis and a surface-area sweep (reference src/15/bvh.hpp:36-108), then a
// pre-order flattening (:146-176).  An animation rebuilds it for every frame
// (src/07/pathtracer-frames.cpp:209-211).  Here the tree for the scene's surfaces over a time
// interval is built by a handful of kernels and written straight into the scene's tables
// (`nodes` in the reference's pre-order layout with escape links, and the 32-byte rows of the
// nearer-child-first walk); nothing but the root box crosses to the host.
//
//   1 surfaceBoxesKernel  Surface::aabb(t0, t1) per surface with the reference's arithmetic
//                         (surface_triangle.hpp:32-63, surface_sphere.hpp:50-69: 16 poses of a
//                         moving surface), and per block the bounds of the box centres
//   2 mortonKernel        63-bit Morton code of each box centre inside the centre bounds
//   3 rocprim radix sort  (code, surface) pairs
//   4 hierarchyKernel     the binary radix tree over the sorted codes (Karras 2012): node i
//                         finds its range of leaves and its split by counting common prefix
//                         bits; equal codes are told apart by their position
//   5 refitKernel         boxes of the inner nodes bottom-up: the second child to arrive at a
//                         node merges and goes on
//   6 layoutKernel        every node's place in pre-order: 2 * (first leaf of its range) + the
//                         number of left turns on the way down from the root; writes `nodes`
//   7 wideRowsKernel      the rows of the nearer-child-first walk (DevScene::wide)
//
// The tree is not the reference's tree (that one needs the sequential sort per node), so the
// walk order differs and with it which of two surfaces hit at exactly the same distance wins;
// apart from such ties every ray finds the same surface at the same distance, because leaf
// boxes and surface tests are the reference's.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pt_scene_host.hpp"

using namespace ptd;
using ptint::fail;

namespace {

struct BoxF {
    float lo[3], hi[3];
};

PT_DEV void growBox(V3& lo, V3& hi, V3 a, V3 b)
{
    // merge, reference src/15/aabb.hpp:63-71
    lo = v3(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z));
    hi = v3(fmaxf(hi.x, b.x), fmaxf(hi.y, b.y), fmaxf(hi.z, b.z));
}

// box of surface `sid` at one time
PT_DEV void surfaceBoxAt(const DevScene& sc, uint32_t sid, uint32_t kind, float t, V3& lo, V3& hi)
{
    if (kind <= kTriDynamic) {
        V3 A, B, C;
        triVertices(sc, sid, t, A, B, C);
        lo = v3(fminf(A.x, fminf(B.x, C.x)), fminf(A.y, fminf(B.y, C.y)), fminf(A.z, fminf(B.z, C.z)));
        hi = v3(fmaxf(A.x, fmaxf(B.x, C.x)), fmaxf(A.y, fmaxf(B.y, C.y)), fmaxf(A.z, fmaxf(B.z, C.z)));
    } else {
        V3 c;
        float r;
        Xform T;
        uint32_t sphereId;
        sphereCR(sc, sid, t, c, r, T, sphereId);
        lo = c - v3(r);
        hi = c + v3(r);
    }
}

constexpr uint32_t kBlock = 256;

__global__ void __launch_bounds__(kBlock) surfaceBoxesKernel(DevScene sc, float t0, float t1, float4* __restrict__ leafBox,
                                                             float4* __restrict__ blockBounds)
{
    __shared__ float red[6][kBlock / 64];
    const uint32_t sid = blockIdx.x * kBlock + threadIdx.x;
    const float inf = __builtin_huge_valf();
    V3 cLo = v3(inf), cHi = v3(-inf);
    if (sid < sc.n_surfaces) {
        const uint32_t kind = fbits(sc.surf[3 * size_t(sid)].w);
        V3 lo, hi;
        surfaceBoxAt(sc, sid, kind, t0, lo, hi);
        if (kind == kTriDynamic || kind == kSphDynamic) {
            const int steps = 16;
            for (int i = 1; i < steps; i++) {
                const float t = mixf(t0, t1, i / (steps - 1.0f)); // ends at t1
                V3 l, h;
                surfaceBoxAt(sc, sid, kind, t, l, h);
                growBox(lo, hi, l, h);
            }
        }
        leafBox[2 * size_t(sid) + 0] = make_float4(lo.x, lo.y, lo.z, hi.x);
        leafBox[2 * size_t(sid) + 1] = make_float4(hi.y, hi.z, 0.0f, 0.0f);
        const V3 c = v3(mixf(lo.x, hi.x, 0.5f), mixf(lo.y, hi.y, 0.5f), mixf(lo.z, hi.z, 0.5f));
        cLo = c;
        cHi = c;
    }
    // bounds of the block's centres: wave reduction by lane shuffles, then one lane per block
    float v[6] = { cLo.x, cLo.y, cLo.z, cHi.x, cHi.y, cHi.z };
    for (int k = 0; k < 6; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[k], off, 64);
            v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
        }
        if ((threadIdx.x & 63u) == 0)
            red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 6; k++)
            for (uint32_t w = 1; w < kBlock / 64; w++)
                red[k][0] = k < 3 ? fminf(red[k][0], red[k][w]) : fmaxf(red[k][0], red[k][w]);
        blockBounds[2 * size_t(blockIdx.x) + 0] = make_float4(red[0][0], red[1][0], red[2][0], red[3][0]);
        blockBounds[2 * size_t(blockIdx.x) + 1] = make_float4(red[4][0], red[5][0], 0.0f, 0.0f);
    }
}

// one block: bounds of all block bounds -> bounds[0..1]
__global__ void __launch_bounds__(kBlock) reduceBoundsKernel(const float4* __restrict__ blockBounds, uint32_t nBlocks,
                                                             float4* __restrict__ bounds)
{
    __shared__ float red[6][kBlock];
    const float inf = __builtin_huge_valf();
    float v[6] = { inf, inf, inf, -inf, -inf, -inf };
    for (uint32_t b = threadIdx.x; b < nBlocks; b += kBlock) {
        const float4 a = blockBounds[2 * size_t(b)], c = blockBounds[2 * size_t(b) + 1];
        v[0] = fminf(v[0], a.x);
        v[1] = fminf(v[1], a.y);
        v[2] = fminf(v[2], a.z);
        v[3] = fmaxf(v[3], a.w);
        v[4] = fmaxf(v[4], c.x);
        v[5] = fmaxf(v[5], c.y);
    }
    for (int k = 0; k < 6; k++)
        red[k][threadIdx.x] = v[k];
    __syncthreads();
    for (uint32_t s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 6; k++)
                red[k][threadIdx.x] = k < 3 ? fminf(red[k][threadIdx.x], red[k][threadIdx.x + s])
                                            : fmaxf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bounds[0] = make_float4(red[0][0], red[1][0], red[2][0], red[3][0]);
        bounds[1] = make_float4(red[4][0], red[5][0], 0.0f, 0.0f);
    }
}

// spread the low 21 bits of x so that there are two zero bits between each
PT_DEV uint64_t spread21(uint64_t x)
{
    x &= 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(kBlock) mortonKernel(const float4* __restrict__ leafBox, const float4* __restrict__ bounds,
                                                       uint32_t n, uint64_t* __restrict__ codes, uint32_t* __restrict__ order)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const float4 b0 = leafBox[2 * size_t(i)], b1 = leafBox[2 * size_t(i) + 1];
    const float4 g0 = bounds[0], g1 = bounds[1];
    const float c[3] = { mixf(b0.x, b0.w, 0.5f), mixf(b0.y, b1.x, 0.5f), mixf(b0.z, b1.y, 0.5f) };
    const float lo[3] = { g0.x, g0.y, g0.z }, hi[3] = { g0.w, g1.x, g1.y };
    uint64_t q[3];
    for (int a = 0; a < 3; a++) {
        const float ext = hi[a] - lo[a];
        float f = ext > 0.0f ? (c[a] - lo[a]) / ext : 0.0f;
        f = fminf(fmaxf(f, 0.0f), 1.0f); // (a NaN centre lands at 0)
        q[a] = uint64_t(f * 2097151.0f);
    }
    codes[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    order[i] = i;
}

// common leading bits of the codes at sorted positions i and j; equal codes are told apart by position
PT_DEV int commonPrefix(const uint64_t* __restrict__ codes, int n, int i, int j)
{
    if (j < 0 || j >= n)
        return -1;
    const uint64_t a = codes[i], b = codes[j];
    if (a != b)
        return __clzll(a ^ b);
    return 64 + __clz(uint32_t(i) ^ uint32_t(j));
}

// Node numbering while building: inner node i in [0, n-1), leaf at sorted position p is n - 1 + p.
__global__ void __launch_bounds__(kBlock) hierarchyKernel(const uint64_t* __restrict__ codes, int n, uint2* __restrict__ children,
                                                          uint2* __restrict__ range, uint32_t* __restrict__ parent)
{
    const int i = int(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n - 1)
        return;
    const int d = commonPrefix(codes, n, i, i + 1) - commonPrefix(codes, n, i, i - 1) >= 0 ? 1 : -1;
    const int dMin = commonPrefix(codes, n, i, i - d);
    int lMax = 2;
    while (commonPrefix(codes, n, i, i + lMax * d) > dMin)
        lMax *= 2;
    int l = 0;
    for (int t = lMax / 2; t >= 1; t /= 2)
        if (commonPrefix(codes, n, i, i + (l + t) * d) > dMin)
            l += t;
    const int j = i + l * d;
    const int dNode = commonPrefix(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (commonPrefix(codes, n, i, i + (s + t) * d) > dNode)
            s += t;
        if (t == 1)
            break;
    }
    const int split = i + s * d + (d < 0 ? -1 : 0);
    const int first = d > 0 ? i : j, last = d > 0 ? j : i;
    const uint32_t left = split == first ? uint32_t(n - 1 + split) : uint32_t(split);
    const uint32_t right = split + 1 == last ? uint32_t(n - 1 + split + 1) : uint32_t(split + 1);
    children[i] = make_uint2(left, right);
    range[i] = make_uint2(uint32_t(first), uint32_t(last));
    parent[left] = uint32_t(i);
    parent[right] = uint32_t(i) | 0x80000000u; // reached by a right turn
    if (i == 0)
        parent[0] = 0xFFFFFFFFu;
}

// boxes of the inner nodes, bottom-up from every leaf; the second arrival at a node does the merge.
// The boxes of inner nodes travel between threads of different blocks, possibly on different
// dies with their own L2: they are written and read with agent-scope atomic accesses (coherent per
// location, no cache-wide write-back or invalidate as a fence of that scope would cost on every
// step), and a thread's stores are complete (vmcnt) before it counts itself in.
// What this leans on is the memory system of gfx950 rather than the language's memory model (under which a
// relaxed counter orders nothing): agent-scope atomic loads and stores are carried out at the level all compute
// units share, past the per-unit caches; a wave's stores have been acknowledged by that level once its vmcnt is 0;
// and the counter's atomic is carried out at that same level afterwards.  The thread that sees the count at 1
// therefore reads, from that level, what the first arrival stored before counting itself in.  The "memory" clobber
// keeps the compiler from moving the accesses across the wait.  The release/acquire form of the counter is correct
// by the model too, and 25 times slower here (a write-back and an invalidate of a whole L2 per step).
PT_DEV void storeCoherent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PT_DEV float loadCoherent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(kBlock) refitKernel(const float4* __restrict__ leafBox, const uint32_t* __restrict__ order,
                                                      const uint2* __restrict__ children, const uint32_t* __restrict__ parent,
                                                      int n, float* innerBox /* 8 floats per inner node */, uint32_t* arrived)
{
    const int p = int(blockIdx.x * kBlock + threadIdx.x);
    if (p >= n)
        return;
    uint32_t node = parent[n - 1 + p];
    while (node != 0xFFFFFFFFu) {
        const uint32_t k = node & 0x7FFFFFFFu;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_fetch_add(&arrived[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            return; // the other child's thread will come by
        const uint2 ch = children[k];
        float a[6], b[6];
        const uint32_t child[2] = { ch.x, ch.y };
        for (int c = 0; c < 2; c++) {
            float* dst = c == 0 ? a : b;
            if (child[c] >= uint32_t(n - 1)) {
                const uint32_t s = order[child[c] - uint32_t(n - 1)];
                const float4 b0 = leafBox[2 * size_t(s)], b1 = leafBox[2 * size_t(s) + 1];
                dst[0] = b0.x; dst[1] = b0.y; dst[2] = b0.z; dst[3] = b0.w; dst[4] = b1.x; dst[5] = b1.y;
            } else {
                for (int i = 0; i < 6; i++)
                    dst[i] = loadCoherent(&innerBox[8 * size_t(child[c]) + i]);
            }
        }
        for (int i = 0; i < 3; i++)
            storeCoherent(&innerBox[8 * size_t(k) + i], fminf(a[i], b[i]));
        for (int i = 3; i < 6; i++)
            storeCoherent(&innerBox[8 * size_t(k) + i], fmaxf(a[i], b[i]));
        node = parent[k];
    }
}

// place of a build node in pre-order: 2 * first leaf + left turns on the way from the root
PT_DEV uint32_t preorderOf(const uint32_t* __restrict__ parent, uint32_t node, uint32_t firstLeaf)
{
    uint32_t turnsLeft = 0;
    for (uint32_t up = parent[node]; up != 0xFFFFFFFFu; up = parent[up & 0x7FFFFFFFu])
        turnsLeft += (up >> 31) ? 0u : 1u;
    return 2u * firstLeaf + turnsLeft;
}

// writes the scene's node table; place[] keeps every build node's pre-order place for the rows
__global__ void __launch_bounds__(kBlock) layoutKernel(const float4* __restrict__ leafBox, const float4* __restrict__ innerBox,
                                                       const uint32_t* __restrict__ order, const uint2* __restrict__ children,
                                                       const uint2* __restrict__ range, const uint32_t* __restrict__ parent,
                                                       int n, float4* __restrict__ nodes, uint32_t* __restrict__ place)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= uint32_t(2 * n - 1))
        return;
    if (t >= uint32_t(n - 1)) {
        const uint32_t p = t - uint32_t(n - 1), s = order[p];
        const uint32_t at = preorderOf(parent, t, p);
        const float4 b0 = leafBox[2 * size_t(s)], b1 = leafBox[2 * size_t(s) + 1];
        nodes[2 * size_t(at) + 0] = b0;
        nodes[2 * size_t(at) + 1] = make_float4(b1.x, b1.y, __uint_as_float(at + 1u), __uint_as_float(s));
        place[t] = at;
    } else {
        const uint2 r = range[t], ch = children[t];
        const uint32_t at = preorderOf(parent, t, r.x);
        const uint32_t leaves = r.y - r.x + 1u;
        // leaves under the first child: up to the split, which is where the second child's range begins
        const uint32_t secondFirst = ch.y >= uint32_t(n - 1) ? ch.y - uint32_t(n - 1) : range[ch.y].x;
        const uint32_t second = at + 2u * (secondFirst - r.x);
        const float4 b0 = innerBox[2 * size_t(t)], b1 = innerBox[2 * size_t(t) + 1];
        nodes[2 * size_t(at) + 0] = b0;
        nodes[2 * size_t(at) + 1] = make_float4(b1.x, b1.y, __uint_as_float(at + 2u * leaves - 1u), __uint_as_float(kInnerBit | second));
        place[t] = at;
    }
}

// grid coordinate at or below / at or above v, checked with the expression the walk decodes it with
PT_DEV uint32_t gridBelow(float v, float base, float step)
{
    float est = floorf((v - base) / step);
    int k = int(fminf(fmaxf(est, 0.0f), 65535.0f));
    while (k > 0 && fmaf(float(k), step, base) > v)
        k--;
    while (k < 65535 && fmaf(float(k + 1), step, base) <= v)
        k++;
    return uint32_t(k);
}
PT_DEV uint32_t gridAbove(float v, float base, float step)
{
    float est = ceilf((v - base) / step);
    int k = int(fminf(fmaxf(est, 0.0f), 65535.0f));
    while (k < 65535 && fmaf(float(k), step, base) < v)
        k++;
    while (k > 0 && fmaf(float(k - 1), step, base) >= v)
        k--;
    return uint32_t(k);
}

// the rows of the nearer-child-first walk, see DevScene::wide
__global__ void __launch_bounds__(kBlock) wideRowsKernel(DevScene sc, const float4* __restrict__ nodes, const uint2* __restrict__ children,
                                                         const uint2* __restrict__ range, const uint32_t* __restrict__ place,
                                                         const uint32_t* __restrict__ order, int n, V3 base, V3 step,
                                                         uint4* __restrict__ wide)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= uint32_t(n - 1))
        return;
    const uint2 ch = children[t];
    const uint32_t child[2] = { ch.x, ch.y };
    uint32_t w[6], link[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t at = place[child[c]];
        const float4 b0 = nodes[2 * size_t(at)], b1 = nodes[2 * size_t(at) + 1];
        const uint32_t qlx = gridBelow(b0.x, base.x, step.x), qly = gridBelow(b0.y, base.y, step.y), qlz = gridBelow(b0.z, base.z, step.z);
        const uint32_t qhx = gridAbove(b0.w, base.x, step.x), qhy = gridAbove(b1.x, base.y, step.y), qhz = gridAbove(b1.y, base.z, step.z);
        w[3 * c + 0] = qlx | (qly << 16);
        w[3 * c + 1] = qlz | (qhx << 16);
        w[3 * c + 2] = qhy | (qhz << 16);
        if (child[c] < uint32_t(n - 1)) {
            link[c] = kInnerBit | (at - range[child[c]].x); // inner nodes before it = place - leaves before it
        } else {
            const uint32_t s = order[child[c] - uint32_t(n - 1)];
            // a static triangle's box is that of its vertices, which is what surfaceBoxesKernel wrote
            link[c] = fbits(sc.surf[3 * size_t(s)].w) == kTriStatic ? s : (kLeafRowBit | at);
        }
    }
    const uint32_t row = place[t] - range[t].x;
    wide[2 * size_t(row) + 0] = make_uint4(w[0], w[1], w[2], w[3]);
    wide[2 * size_t(row) + 1] = make_uint4(w[4], w[5], link[0], link[1]);
}

__global__ void singleLeafKernel(const float4* __restrict__ leafBox, float4* __restrict__ nodes)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        nodes[0] = leafBox[0];
        nodes[1] = make_float4(leafBox[1].x, leafBox[1].y, __uint_as_float(1u), __uint_as_float(0u));
    }
}

template <typename T>
T* carve(char*& p, size_t count)
{
    T* r = reinterpret_cast<T*>(p);
    p += (count * sizeof(T) + 255) / 256 * 256;
    return r;
}

} // namespace

extern "C" {

int pt_scene_build_bvh_device(pt_scene* scene, float t0, float t1, void* hip_stream)
{
    if (!scene)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    const uint64_t n64 = scene->n_surfaces;
    if (n64 == 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "scene has no surfaces");
    if (2 * n64 - 1 >= kLeafRowBit)
        return fail(PT_ERR_LIMIT, "too many surfaces for the device builder");
    const int n = int(n64);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    SceneCall call(scene, stream);
    HIP_TRY(call.status);
    // until every step below has succeeded the scene has no tree: a call that fails half way leaves a scene that
    // refuses to render rather than one with a partly rewritten tree
    scene->dev.n_nodes = 0;
    scene->dev.wide = nullptr;
    const DevScene& sc = scene->dev;
    const uint32_t leafBlocks = (uint32_t(n) + kBlock - 1) / kBlock;

    // work area: one allocation, kept with the scene for the next frame
    size_t sortBytes = 0;
    uint64_t* nullKeys = nullptr;
    uint32_t* nullVals = nullptr;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, sortBytes, nullKeys, nullKeys, nullVals, nullVals, size_t(n), 0, 63, stream));
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = pad(size_t(n) * 32) + pad(size_t(leafBlocks) * 32) + pad(64) + 2 * pad(size_t(n) * 8) + 2 * pad(size_t(n) * 4)
        + 2 * pad(size_t(n) * 8) + pad(size_t(2 * n) * 4) + pad(size_t(n) * 32) + pad(size_t(n) * 4) + pad(size_t(2 * n) * 4)
        + pad(sortBytes) + 4096;
    if (scene->buildCapacity < need) {
        if (scene->buildArea)
            HIP_TRY(hipFree(scene->buildArea));
        scene->buildArea = nullptr;
        scene->buildCapacity = 0;
        HIP_TRY(hipMalloc(&scene->buildArea, need));
        scene->buildCapacity = need;
    }
    char* p = static_cast<char*>(scene->buildArea);
    float4* leafBox = carve<float4>(p, 2 * size_t(n));
    float4* blockBounds = carve<float4>(p, 2 * size_t(leafBlocks));
    float4* bounds = carve<float4>(p, 4);
    uint64_t* codes = carve<uint64_t>(p, size_t(n));
    uint64_t* codesSorted = carve<uint64_t>(p, size_t(n));
    uint32_t* order = carve<uint32_t>(p, size_t(n));
    uint32_t* orderSorted = carve<uint32_t>(p, size_t(n));
    uint2* children = carve<uint2>(p, size_t(n));
    uint2* range = carve<uint2>(p, size_t(n));
    uint32_t* parent = carve<uint32_t>(p, 2 * size_t(n));
    float4* innerBox = carve<float4>(p, 2 * size_t(n));
    uint32_t* arrived = carve<uint32_t>(p, size_t(n));
    uint32_t* place = carve<uint32_t>(p, 2 * size_t(n));
    void* sortTemp = p;

    hipLaunchKernelGGL(surfaceBoxesKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, sc, t0, t1, leafBox, blockBounds);
    if (n == 1) {
        hipLaunchKernelGGL(singleLeafKernel, dim3(1), dim3(64), 0, stream, leafBox, scene->nodes);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        scene->dev.n_nodes = 1;
        scene->dev.wide = scene->wide; // no inner node: no row is ever read
        HIP_TRY(call.done(stream));
        return PT_OK;
    }
    hipLaunchKernelGGL(reduceBoundsKernel, dim3(1), dim3(kBlock), 0, stream, blockBounds, leafBlocks, bounds);
    hipLaunchKernelGGL(mortonKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, leafBox, bounds, uint32_t(n), codes, order);
    HIP_TRY(hipGetLastError());
    HIP_TRY(rocprim::radix_sort_pairs(sortTemp, sortBytes, codes, codesSorted, order, orderSorted, size_t(n), 0, 63, stream));
    const uint32_t innerBlocks = (uint32_t(n - 1) + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(hierarchyKernel, dim3(innerBlocks), dim3(kBlock), 0, stream, codesSorted, n, children, range, parent);
    HIP_TRY(hipMemsetAsync(arrived, 0, size_t(n) * 4, stream));
    hipLaunchKernelGGL(refitKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, leafBox, orderSorted, children, parent, n,
                       reinterpret_cast<float*>(innerBox), arrived);
    const uint32_t allBlocks = (uint32_t(2 * n - 1) + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(layoutKernel, dim3(allBlocks), dim3(kBlock), 0, stream, leafBox, innerBox, orderSorted, children, range, parent,
                       n, scene->nodes, place);
    HIP_TRY(hipGetLastError());

    // the grid of the rows spans the root box: the only values that cross to the host
    float root[8];
    HIP_TRY(hipMemcpyAsync(root, scene->nodes, sizeof(root), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const float rlo[3] = { root[0], root[1], root[2] }, rhi[3] = { root[3], root[4], root[5] };
    bool gridOk = true;
    for (int a = 0; a < 3; a++) {
        if (!std::isfinite(rlo[a]) || !std::isfinite(rhi[a]) || !(rlo[a] <= rhi[a])) {
            gridOk = false;
            break;
        }
        float st = (rhi[a] - rlo[a]) / 65535.0f;
        if (!std::isfinite(st)) {
            gridOk = false;
            break;
        }
        if (!(st > 1.0e-30f))
            st = 1.0e-30f;
        for (int guard = 0; guard < 64 && std::fmaf(65535.0f, st, rlo[a]) < rhi[a]; guard++)
            st = std::nextafterf(st, std::numeric_limits<float>::infinity());
        if (std::fmaf(65535.0f, st, rlo[a]) < rhi[a])
            gridOk = false;
        scene->dev.qbase[a] = rlo[a];
        scene->dev.qstep[a] = st;
    }
    if (gridOk) {
        hipLaunchKernelGGL(wideRowsKernel, dim3(innerBlocks), dim3(kBlock), 0, stream, sc, scene->nodes, children, range, place,
                           orderSorted, n, v3(scene->dev.qbase[0], scene->dev.qbase[1], scene->dev.qbase[2]),
                           v3(scene->dev.qstep[0], scene->dev.qstep[1], scene->dev.qstep[2]), scene->wide);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(call.done(stream));
    scene->dev.wide = gridOk ? scene->wide : nullptr; // without rows: storage-order walk only
    scene->dev.n_nodes = uint32_t(2 * n - 1);
    return PT_OK;
}

int pt_scene_get_bvh(pt_scene* scene, pt_bvh_node* out_nodes, uint64_t capacity, uint64_t* out_count)
try {
    if (!scene || !out_count)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    const uint64_t n = scene->dev.n_nodes;
    *out_count = n;
    if (!out_nodes || capacity < n)
        return out_nodes ? fail(PT_ERR_INVALID_ARGUMENT, "node buffer too small") : PT_OK;
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float4> rows(2 * n);
    if (n > 0)
        HIP_TRY(hipMemcpy(rows.data(), scene->nodes, rows.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint64_t k = 0; k < n; k++) {
        const float4 &a = rows[2 * k], &b = rows[2 * k + 1];
        pt_bvh_node& o = out_nodes[k];
        o.lo[0] = a.x;
        o.lo[1] = a.y;
        o.lo[2] = a.z;
        o.hi[0] = a.w;
        o.hi[1] = b.x;
        o.hi[2] = b.y;
        const uint32_t leaf = __builtin_bit_cast(uint32_t, b.w);
        o.link = (leaf & kInnerBit) ? (uint64_t(leaf & ~kInnerBit) * 2 + 1) : uint64_t(leaf) * 2;
    }
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

} // extern "C"
