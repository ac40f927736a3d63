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
//   surfaceBoxesKernel   Surface::aabb(t0, t1) per surface with the reference's arithmetic
//                        (surface_triangle.hpp:32-63, surface_sphere.hpp:50-69: 16 poses of a
//                        moving surface), and per block the bounds of the boxes (or of their centres)
//
// Two builders share that step and the tables they fill.  The default is the surface-area builder
// (the reference's criterion, bvh.hpp:60-96, applied to 32 bins per axis for large nodes and to every
// split of the sorted order for nodes of up to 64 surfaces; see "the surface-area builder" below).
// PT_BVH_BUILDER=morton selects the radix tree, which is five times quicker to build and up to half
// as fast to walk:
//   mortonKernel        63-bit Morton code of each box centre inside the centre bounds
//   rocprim radix sort  (code, surface) pairs
//   hierarchyKernel     the binary radix tree over the sorted codes (Karras 2012): node i
//                       finds its range of leaves and its split by counting common prefix
//                       bits; equal codes are told apart by their position
//   refitKernel         boxes of the inner nodes bottom-up: the second child to arrive at a
//                       node merges and goes on
//   layoutKernel        every node's place in pre-order: 2 * (first leaf of its range) + the
//                       number of left turns on the way down from the root; writes `nodes`
//   wideRowsKernel      the rows of the nearer-child-first walk (DevScene::wide)
//
// Neither tree is the reference's tree (that one needs the sequential sort per node), so the
// walk order differs and with it which of two surfaces hit at exactly the same distance wins;
// apart from such ties every ray finds the same surface at the same distance, because leaf
// boxes and surface tests are the reference's.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cstdlib>

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
                                                             float4* __restrict__ blockBounds, int boundsOfBoxes)
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
        cLo = boundsOfBoxes ? lo : c;
        cHi = boundsOfBoxes ? hi : c;
    }
    // bounds of the block's centres (or boxes): wave reduction by lane shuffles, then one lane per block
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

// ---- the surface-area builder -------------------------------------------------------------------------------
// Top-down, one tree level per pass, every node written straight to its final place: a node at place P over the
// surfaces at places [b, b + m) of the work order has its first child at P + 1 and, with mL surfaces in the first
// child, its second child at P + 2 mL; its row of the `wide` table is P - b.
//   nodes of more than 64 surfaces: per pass, every surface adds its box to one of 32 bins per axis of its node
//     (binNodesKernel, one wave per node of up to 2048 surfaces with its bins in LDS; binPlacesKernel for larger
//     nodes, many blocks per node, per-block bins in LDS), the node picks the plane between bins
//     with the smallest  area(left) * count(left) + area(right) * count(right)  (splitKernel), and a stable
//     partition by a prefix sum over the whole order moves the surfaces (flagKernel, scan, scatterKernel);
//   nodes of up to 64 surfaces: one wave each (smallSubtreesKernel) builds the whole subtree in LDS, trying every
//     split of the surfaces sorted by centre along every axis.
// Equal inputs give equal trees: the bins are filled by order-independent integer atomics, the partition is
// stable, and the order in which nodes are numbered inside a pass does not reach the output.
constexpr int kBins = 32;
constexpr int kBinWords = 7; // count, lo xyz, hi xyz (the latter as order-preserving keys)
constexpr int kNodeBinWords = 3 * kBins * kBinWords;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kSmallNode = 64;
constexpr uint32_t kByPosition = 3; // SplitRec::axis of a node that is halved by place

struct ActiveNode {
    uint32_t place, begin, count, pad;
    float lo[3], hi[3]; // what the bins span: the node's box, or a box around it
    uint32_t pad2[2];
};
struct SplitRec {
    uint32_t axis, plane, nLeft;
    uint32_t swap; // the surfaces beyond the plane go to the first child
    uint32_t child[2]; // number of the child in the next pass's list, or kNone
    uint32_t pad2[2];
};
struct SmallTask {
    uint32_t place, begin, count, pad;
};

// floats as unsigned keys of the same order (for integer atomic min / max)
PT_DEV uint32_t orderKey(float f)
{
    const uint32_t u = fbits(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
PT_DEV float keyFloat(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

PT_DEV int binOf(float c, float lo, float hi)
{
    const float ext = hi - lo;
    float f = ext > 0.0f ? (c - lo) * (float(kBins) / ext) : 0.0f;
    f = fminf(fmaxf(f, 0.0f), float(kBins - 1)); // (NaN lands in bin 0)
    return int(f);
}

PT_DEV float halfArea(const float* lo, const float* hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(kBlock) sahStartKernel(const float4* __restrict__ bounds, uint32_t n, ActiveNode* __restrict__ active,
                                                         uint32_t* __restrict__ idx, uint32_t* __restrict__ nodeAt,
                                                         SmallTask* __restrict__ small, uint32_t* __restrict__ counters)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) {
        idx[p] = p;
        nodeAt[p] = n > kSmallNode ? 0u : kNone;
    }
    if (p == 0 && n <= kSmallNode) {
        const SmallTask t = { 0u, 0u, n, 0u };
        small[0] = t;
        counters[2] = 1u;
    } else if (p == 0) {
        const float4 a = bounds[0], b = bounds[1];
        ActiveNode r = {};
        r.count = n;
        r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z;
        r.hi[0] = a.w; r.hi[1] = b.x; r.hi[2] = b.y;
        active[0] = r;
    }
}

__global__ void __launch_bounds__(kBlock) clearBinsKernel(uint32_t* __restrict__ bins, size_t words)
{
    const size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i < words) {
        const uint32_t w = uint32_t(i % kBinWords);
        bins[i] = w == 0 ? 0u : (w <= 3 ? 0xFFFFFFFFu : 0u);
    }
}

template <typename Add, typename Min, typename Max>
PT_DEV void addToBins(uint32_t* bins, const ActiveNode& nd, float4 b0, float4 b1, Add add, Min mn, Max mx)
{
    const float lo[3] = { b0.x, b0.y, b0.z }, hi[3] = { b0.w, b1.x, b1.y };
    uint32_t klo[3], khi[3];
    for (int a = 0; a < 3; a++) {
        klo[a] = orderKey(lo[a]);
        khi[a] = orderKey(hi[a]);
    }
    for (int a = 0; a < 3; a++) {
        const int bin = binOf(mixf(lo[a], hi[a], 0.5f), nd.lo[a], nd.hi[a]);
        uint32_t* w = bins + (a * kBins + bin) * kBinWords;
        add(w);
        for (int k = 0; k < 3; k++) {
            mn(w + 1 + k, klo[k]);
            mx(w + 4 + k, khi[k]);
        }
    }
}

// Nodes of more than kWaveNodeMax surfaces are binned by place (binPlacesKernel: many blocks per node), the others
// by one wave per node (binNodesKernel: no global atomics at all).
constexpr uint32_t kWaveNodeMax = 2048;
constexpr uint32_t kBinChunks = 8; // runs of kBlock places per block of binPlacesKernel

PT_DEV void initBins(uint32_t* sBins, uint32_t thread, uint32_t threads)
{
    for (uint32_t i = thread; i < uint32_t(kNodeBinWords); i += threads) {
        const uint32_t w = i % kBinWords;
        sBins[i] = w == 0 ? 0u : (w <= 3 ? 0xFFFFFFFFu : 0u);
    }
}

PT_DEV void flushBins(const uint32_t* sBins, uint32_t* __restrict__ g)
{
    for (uint32_t i = threadIdx.x; i < uint32_t(kNodeBinWords); i += kBlock) {
        const uint32_t w = i % kBinWords, v = sBins[i];
        if (w == 0) {
            if (v)
                atomicAdd(g + i, v);
        } else if (w <= 3) {
            if (v != 0xFFFFFFFFu)
                atomicMin(g + i, v);
        } else if (v) {
            atomicMax(g + i, v);
        }
    }
}

// A block takes kBinChunks runs of kBlock places.  A node owns a run of places: the same node at both ends of a
// run means the run lies inside it, and its surfaces go to bins in LDS that are added to the node's bins in memory
// when the block moves on to another node or ends; the few runs that straddle nodes add to memory directly.
__global__ void __launch_bounds__(kBlock) binPlacesKernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ nodeAt,
                                                          const ActiveNode* __restrict__ active, const float4* __restrict__ leafBox,
                                                          uint32_t* __restrict__ bins)
{
    __shared__ uint32_t sBins[kNodeBinWords];
    uint32_t held = kNone; // the node whose bins are in LDS (the same in every thread)
    for (uint32_t c = 0; c < kBinChunks; c++) {
        const uint32_t first = (blockIdx.x * kBinChunks + c) * kBlock;
        if (first >= n)
            break;
        const uint32_t last = min(first + kBlock, n) - 1u, p = first + threadIdx.x;
        const uint32_t aFirst = nodeAt[first];
        const bool inOneNode = aFirst != kNone && aFirst == nodeAt[last] && active[aFirst].count > kWaveNodeMax;
        if (inOneNode) {
            if (held != aFirst) {
                __syncthreads();
                if (held != kNone)
                    flushBins(sBins, bins + size_t(held) * kNodeBinWords);
                __syncthreads();
                initBins(sBins, threadIdx.x, kBlock);
                __syncthreads();
                held = aFirst;
            }
            if (p < n) {
                const uint32_t s = idx[p];
                addToBins(sBins, active[aFirst], leafBox[2 * size_t(s)], leafBox[2 * size_t(s) + 1],
                          [](uint32_t* w) { atomicAdd(w, 1u); }, [](uint32_t* w, uint32_t v) { atomicMin(w, v); },
                          [](uint32_t* w, uint32_t v) { atomicMax(w, v); });
            }
        } else if (p < n) {
            const uint32_t a = nodeAt[p];
            if (a != kNone && active[a].count > kWaveNodeMax) {
                const uint32_t s = idx[p];
                addToBins(bins + size_t(a) * kNodeBinWords, active[a], leafBox[2 * size_t(s)], leafBox[2 * size_t(s) + 1],
                          [](uint32_t* w) { atomicAdd(w, 1u); }, [](uint32_t* w, uint32_t v) { atomicMin(w, v); },
                          [](uint32_t* w, uint32_t v) { atomicMax(w, v); });
            }
        }
    }
    __syncthreads();
    if (held != kNone)
        flushBins(sBins, bins + size_t(held) * kNodeBinWords);
}

// One wave per live node of up to kWaveNodeMax surfaces: bins in LDS, written to memory whole.
__global__ void __launch_bounds__(kBlock) binNodesKernel(uint32_t nActive, const uint32_t* __restrict__ idx,
                                                         const ActiveNode* __restrict__ active, const float4* __restrict__ leafBox,
                                                         uint32_t* __restrict__ bins)
{
    __shared__ uint32_t sBins[kBlock / 64][kNodeBinWords];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t a = blockIdx.x * (kBlock / 64) + wave;
    ActiveNode nd = {};
    if (a < nActive)
        nd = active[a];
    const bool mine = a < nActive && nd.count <= kWaveNodeMax;
    if (mine)
        initBins(sBins[wave], lane, 64u);
    __syncthreads();
    if (mine)
        for (uint32_t k = lane; k < nd.count; k += 64u) {
            const uint32_t s = idx[nd.begin + k];
            addToBins(sBins[wave], nd, leafBox[2 * size_t(s)], leafBox[2 * size_t(s) + 1],
                      [](uint32_t* w) { atomicAdd(w, 1u); }, [](uint32_t* w, uint32_t v) { atomicMin(w, v); },
                      [](uint32_t* w, uint32_t v) { atomicMax(w, v); });
        }
    __syncthreads();
    if (mine) {
        uint32_t* g = bins + size_t(a) * kNodeBinWords;
        for (uint32_t i = lane; i < uint32_t(kNodeBinWords); i += 64u)
            g[i] = sBins[wave][i];
    }
}

struct BinBox {
    float lo[3], hi[3];
    uint32_t count;
};
PT_DEV void clearBox(BinBox& b)
{
    const float inf = __builtin_huge_valf();
    for (int k = 0; k < 3; k++) {
        b.lo[k] = inf;
        b.hi[k] = -inf;
    }
    b.count = 0;
}
PT_DEV void addBin(BinBox& b, const uint32_t* __restrict__ w)
{
    const uint32_t c = w[0];
    if (c == 0)
        return;
    b.count += c;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = fminf(b.lo[k], keyFloat(w[1 + k]));
        b.hi[k] = fmaxf(b.hi[k], keyFloat(w[4 + k]));
    }
}

// A block of 128 threads takes kSplitNodes live nodes in turn; thread (axis, plane) prices the split between bins
// plane - 1 and plane.  The children are collected in LDS and get their numbers in the lists of the next pass with
// one reservation per block and list (an atomic on one address completes at about 90 per microsecond: one per node
// was a third of a millisecond per pass at 30 000 nodes).
constexpr uint32_t kSplitNodes = 8;
__global__ void __launch_bounds__(128) splitKernel(const ActiveNode* __restrict__ active, uint32_t nActive,
                                                   uint32_t perBlock /* nodes per block, at most kSplitNodes */, const uint32_t* __restrict__ bins,
                                                   int halveAll, SplitRec* __restrict__ splits, ActiveNode* __restrict__ nextActive,
                                                   SmallTask* __restrict__ small,
                                                   uint32_t* __restrict__ counters /* [0] live nodes of the next pass, [1] the largest of them, [2] small */,
                                                   float4* __restrict__ nodes, uint32_t* __restrict__ firstLeaf)
{
    __shared__ unsigned long long best[kSplitNodes];
    __shared__ ActiveNode sChild[2 * kSplitNodes]; // count 0: no such child
    __shared__ uint32_t sBase[2];
    const uint32_t axis = threadIdx.x / kBins, plane = threadIdx.x % kBins;
    if (threadIdx.x < kSplitNodes)
        best[threadIdx.x] = ~0ull;
    if (threadIdx.x < 2 * kSplitNodes)
        sChild[threadIdx.x].count = 0;
    __syncthreads();
    for (uint32_t g = 0; g < perBlock; g++) {
        const uint32_t a = blockIdx.x * perBlock + g;
        if (a >= nActive)
            break;
        const uint32_t* nb = bins + size_t(a) * kNodeBinWords;
        BinBox L, R;
        clearBox(L);
        clearBox(R);
        unsigned long long key = ~0ull;
        if (axis < 3 && plane > 0 && !halveAll) {
            for (uint32_t k = 0; k < uint32_t(kBins); k++)
                addBin(k < plane ? L : R, nb + (axis * kBins + k) * kBinWords);
            if (L.count > 0 && R.count > 0) {
                const float cost = halfArea(L.lo, L.hi) * float(L.count) + halfArea(R.lo, R.hi) * float(R.count);
                if (cost >= 0.0f && cost < __builtin_huge_valf())
                    key = ((unsigned long long) fbits(cost) << 32) | (axis << 8) | plane;
            }
        }
        if (key != ~0ull)
            atomicMin(&best[g], key);
        __syncthreads();
        const unsigned long long win = best[g];
        const bool byPlace = win == ~0ull;
        if (byPlace ? threadIdx.x != 0 : key != win)
            continue;
        const ActiveNode nd = active[a];
        SplitRec sp = {};
        uint32_t nLeft;
        if (byPlace) {
            // no plane separates the centres (or the tree is too deep already): halve the run of places; the halves
            // take this node's span for their bins, and their boxes come out of their own bins in the next pass
            clearBox(L);
            clearBox(R);
            for (uint32_t k = 0; k < uint32_t(kBins); k++)
                addBin(L, nb + k * kBinWords);
            nLeft = nd.count / 2;
        } else {
            // the half with more surfaces becomes the first child: the storage-order walk goes through the first
            // child's subtree before it looks at the second, and the sooner it has a hit the more it can skip
            // (measured on the scenes that are walked that way: tracing -1.5 % on the Cornell box, -4 % on the 100
            // spheres; larger area first: +8 %, -4 %)
            const bool swap = R.count > L.count;
            if (swap) {
                const BinBox t = L;
                L = R;
                R = t;
            }
            sp.swap = swap ? 1u : 0u;
            nLeft = L.count;
        }
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(L.lo[k], R.lo[k]);
            hi[k] = fmaxf(L.hi[k], R.hi[k]);
        }
        const uint32_t second = nd.place + 2u * nLeft;
        nodes[2 * size_t(nd.place) + 0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        nodes[2 * size_t(nd.place) + 1] = make_float4(hi[1], hi[2], __uint_as_float(nd.place + 2u * nd.count - 1u), __uint_as_float(kInnerBit | second));
        firstLeaf[nd.place] = nd.begin;
        sp.axis = byPlace ? kByPosition : axis;
        sp.plane = plane;
        sp.nLeft = nLeft;
        sp.child[0] = sp.child[1] = kNone; // the live ones get their numbers below
        splits[a] = sp;
        for (int c = 0; c < 2; c++) {
            ActiveNode ch = {};
            ch.place = c == 0 ? nd.place + 1u : second;
            ch.begin = c == 0 ? nd.begin : nd.begin + nLeft;
            ch.count = c == 0 ? nLeft : nd.count - nLeft;
            const BinBox& bx = c == 0 ? L : R;
            for (int k = 0; k < 3; k++) {
                ch.lo[k] = byPlace ? nd.lo[k] : bx.lo[k];
                ch.hi[k] = byPlace ? nd.hi[k] : bx.hi[k];
            }
            sChild[2 * g + c] = ch;
        }
    }
    __syncthreads();
    // numbers for the children: one thread counts, one reservation per list
    if (threadIdx.x == 0) {
        uint32_t live = 0, little = 0, largest = 0;
        for (uint32_t k = 0; k < 2 * kSplitNodes; k++) {
            const uint32_t c = sChild[k].count;
            live += c > kSmallNode ? 1u : 0u;
            little += c > 0 && c <= kSmallNode ? 1u : 0u;
            largest = max(largest, c);
        }
        sBase[0] = live ? atomicAdd(&counters[0], live) : 0u;
        sBase[1] = little ? atomicAdd(&counters[2], little) : 0u;
        if (live)
            atomicMax(&counters[1], largest);
    }
    __syncthreads();
    if (threadIdx.x < 2 * kSplitNodes) {
        const ActiveNode ch = sChild[threadIdx.x];
        if (ch.count > 0) {
            uint32_t before = 0; // children of the same kind before this one
            const bool live = ch.count > kSmallNode;
            for (uint32_t k = 0; k < threadIdx.x; k++) {
                const uint32_t c = sChild[k].count;
                before += (live ? c > kSmallNode : (c > 0 && c <= kSmallNode)) ? 1u : 0u;
            }
            if (live) {
                const uint32_t slot = sBase[0] + before;
                nextActive[slot] = ch;
                splits[blockIdx.x * perBlock + threadIdx.x / 2].child[threadIdx.x % 2] = slot;
            } else {
                const SmallTask t = { ch.place, ch.begin, ch.count, 0u };
                small[sBase[1] + before] = t;
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock) flagKernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ nodeAt,
                                                     const ActiveNode* __restrict__ active, const SplitRec* __restrict__ splits,
                                                     const float4* __restrict__ leafBox, uint32_t* __restrict__ flags)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t a = nodeAt[p];
    uint32_t left = 0;
    if (a != kNone) {
        const SplitRec sp = splits[a];
        if (sp.axis == kByPosition) {
            left = p - active[a].begin < sp.nLeft ? 1u : 0u;
        } else {
            const uint32_t s = idx[p];
            const float4 b0 = leafBox[2 * size_t(s)], b1 = leafBox[2 * size_t(s) + 1];
            const float lo = sp.axis == 0 ? b0.x : (sp.axis == 1 ? b0.y : b0.z);
            const float hi = sp.axis == 0 ? b0.w : (sp.axis == 1 ? b1.x : b1.y);
            const ActiveNode& nd = active[a];
            left = (uint32_t(binOf(mixf(lo, hi, 0.5f), nd.lo[sp.axis], nd.hi[sp.axis])) < sp.plane ? 1u : 0u) ^ sp.swap;
        }
    }
    flags[p] = left;
}

__global__ void __launch_bounds__(kBlock) scatterKernel(uint32_t n, const uint32_t* __restrict__ idxIn, const uint32_t* __restrict__ nodeAtIn,
                                                        const ActiveNode* __restrict__ active, const SplitRec* __restrict__ splits,
                                                        const uint32_t* __restrict__ flags, const uint32_t* __restrict__ leftBefore,
                                                        uint32_t* __restrict__ idxOut, uint32_t* __restrict__ nodeAtOut)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t a = nodeAtIn[p], s = idxIn[p];
    if (a == kNone) {
        idxOut[p] = s;
        nodeAtOut[p] = kNone;
        return;
    }
    const uint32_t begin = active[a].begin;
    const SplitRec sp = splits[a];
    const uint32_t rank = leftBefore[p] - leftBefore[begin]; // surfaces of the node before this one that go left
    const bool left = flags[p] != 0;
    const uint32_t q = left ? begin + rank : begin + sp.nLeft + (p - begin - rank);
    idxOut[q] = s;
    nodeAtOut[q] = sp.child[left ? 0 : 1];
}

// One wave per node of up to 64 surfaces: the whole subtree, level by level, all nodes of a level at once.
// Lane = place inside the run.  A lane's candidate split of its node sends to the first child every surface
// whose centre comes before the lane's own along the axis (ties by place), so every way of cutting the sorted
// order is tried on every axis, as in a full sweep.
__global__ void __launch_bounds__(64) smallSubtreesKernel(const SmallTask* __restrict__ tasks, const uint32_t* __restrict__ idx,
                                                          const float4* __restrict__ leafBox, float4* __restrict__ nodes,
                                                          uint32_t* __restrict__ firstLeaf)
{
    __shared__ uint32_t sSwap[64];           // per node: the winner's second half goes first
    __shared__ float sBox[64][8];           // lo xyz, hi xyz, surface
    __shared__ unsigned long long sBest[64]; // per node (by its first lane): cheapest candidate
    __shared__ float sHalves[64][12];        // per node: boxes of the winner's two halves
    const SmallTask task = tasks[blockIdx.x];
    const uint32_t lane = threadIdx.x, m = task.count;
    const float inf = __builtin_huge_valf();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    uint32_t sid = 0;
    if (lane < m) {
        sid = idx[task.begin + lane];
        const float4 b0 = leafBox[2 * size_t(sid)], b1 = leafBox[2 * size_t(sid) + 1];
        lo[0] = b0.x; lo[1] = b0.y; lo[2] = b0.z;
        hi[0] = b0.w; hi[1] = b1.x; hi[2] = b1.y;
    }
    // my node: lanes [nb, nb + nm) at place nP, box nlo..nhi
    uint32_t nb = 0, nm = lane < m ? m : 0u, nP = task.place;
    float nlo[3], nhi[3];
    for (int k = 0; k < 3; k++) {
        float a = lo[k], b = hi[k];
        for (int off = 32; off > 0; off >>= 1) {
            a = fminf(a, __shfl_xor(a, off, 64));
            b = fmaxf(b, __shfl_xor(b, off, 64));
        }
        nlo[k] = a;
        nhi[k] = b;
    }
    for (;;) {
        const bool open = nm > 1;
        if (__ballot(open) == 0ull)
            break;
        for (int k = 0; k < 3; k++) {
            sBox[lane][k] = lo[k];
            sBox[lane][3 + k] = hi[k];
        }
        sBox[lane][6] = __uint_as_float(sid);
        if (open && lane == nb)
            sBest[nb] = ~0ull;
        __syncthreads();
        float c[3];
        for (int k = 0; k < 3; k++)
            c[k] = mixf(lo[k], hi[k], 0.5f);
        float Llo[3][3], Lhi[3][3], Rlo[3][3], Rhi[3][3];
        uint32_t nL[3] = { 0, 0, 0 };
        for (int a = 0; a < 3; a++)
            for (int k = 0; k < 3; k++) {
                Llo[a][k] = inf; Lhi[a][k] = -inf;
                Rlo[a][k] = inf; Rhi[a][k] = -inf;
            }
        for (uint32_t k = 0; __ballot(open && k < nm) != 0ull; k++) {
            if (open && k < nm) {
                const uint32_t j = nb + k;
                float jlo[3], jhi[3];
                for (int d = 0; d < 3; d++) {
                    jlo[d] = sBox[j][d];
                    jhi[d] = sBox[j][3 + d];
                }
                for (int a = 0; a < 3; a++) {
                    const float cj = mixf(jlo[a], jhi[a], 0.5f);
                    const bool left = cj < c[a] || (cj == c[a] && j < lane);
                    nL[a] += left ? 1u : 0u;
                    for (int d = 0; d < 3; d++) {
                        Llo[a][d] = left ? fminf(Llo[a][d], jlo[d]) : Llo[a][d];
                        Lhi[a][d] = left ? fmaxf(Lhi[a][d], jhi[d]) : Lhi[a][d];
                        Rlo[a][d] = left ? Rlo[a][d] : fminf(Rlo[a][d], jlo[d]);
                        Rhi[a][d] = left ? Rhi[a][d] : fmaxf(Rhi[a][d], jhi[d]);
                    }
                }
            }
        }
        unsigned long long key = ~0ull;
        if (open) {
            for (uint32_t a = 0; a < 3; a++) {
                if (nL[a] == 0)
                    continue;
                const float cost = halfArea(Llo[a], Lhi[a]) * float(nL[a]) + halfArea(Rlo[a], Rhi[a]) * float(nm - nL[a]);
                if (cost >= 0.0f && cost < inf) {
                    const unsigned long long k2 = ((unsigned long long) fbits(cost) << 32) | (lane << 2) | a;
                    key = k2 < key ? k2 : key;
                }
            }
            if (key != ~0ull)
                atomicMin(&sBest[nb], key);
        }
        __syncthreads();
        bool left = false;
        uint32_t nLeft = 0;
        if (open) {
            const unsigned long long win = sBest[nb];
            if (win == ~0ull) {
                // no candidate (centres that do not compare): halve the run of lanes; the first lane of the node
                // works out the two boxes
                nLeft = nm / 2;
                left = lane - nb < nLeft;
                if (lane == nb) {
                    float h[12];
                    for (int d = 0; d < 3; d++) {
                        h[d] = inf; h[3 + d] = -inf; h[6 + d] = inf; h[9 + d] = -inf;
                    }
                    for (uint32_t k = 0; k < nm; k++) {
                        const int o = k < nLeft ? 0 : 6;
                        for (int d = 0; d < 3; d++) {
                            h[o + d] = fminf(h[o + d], sBox[nb + k][d]);
                            h[o + 3 + d] = fmaxf(h[o + 3 + d], sBox[nb + k][3 + d]);
                        }
                    }
                    for (int d = 0; d < 12; d++)
                        sHalves[nb][d] = h[d];
                }
            } else {
                const uint32_t wl = uint32_t(win >> 2) & 63u, wa = uint32_t(win) & 3u;
                const float cw = mixf(sBox[wl][wa], sBox[wl][3 + wa], 0.5f);
                const float cm = wa == 0 ? c[0] : (wa == 1 ? c[1] : c[2]);
                left = cm < cw || (cm == cw && lane < wl);
                if (lane == wl) {
                    float h[12];
                    for (int d = 0; d < 3; d++) {
                        h[d] = wa == 0 ? Llo[0][d] : (wa == 1 ? Llo[1][d] : Llo[2][d]);
                        h[3 + d] = wa == 0 ? Lhi[0][d] : (wa == 1 ? Lhi[1][d] : Lhi[2][d]);
                        h[6 + d] = wa == 0 ? Rlo[0][d] : (wa == 1 ? Rlo[1][d] : Rlo[2][d]);
                        h[9 + d] = wa == 0 ? Rhi[0][d] : (wa == 1 ? Rhi[1][d] : Rhi[2][d]);
                    }
                    // the half with more surfaces goes first (see splitKernel)
                    const uint32_t cl = wa == 0 ? nL[0] : (wa == 1 ? nL[1] : nL[2]);
                    const bool swap = nm - cl > cl;
                    sSwap[nb] = swap ? 1u : 0u;
                    for (int d = 0; d < 6; d++) {
                        sHalves[nb][d] = swap ? h[6 + d] : h[d];
                        sHalves[nb][6 + d] = swap ? h[d] : h[6 + d];
                    }
                }
            }
        }
        __syncthreads();
        if (open && sBest[nb] != ~0ull && sSwap[nb])
            left = !left;
        // stable partition inside each node: new lane of every surface
        const unsigned long long leftMask = __ballot(open && left);
        uint32_t to = lane;
        if (open) {
            const unsigned long long nodeMask = (nm == 64u ? ~0ull : ((1ull << nm) - 1ull)) << nb;
            const unsigned long long below = (1ull << lane) - 1ull;
            nLeft = uint32_t(__popcll(leftMask & nodeMask));
            to = left ? nb + uint32_t(__popcll(leftMask & nodeMask & below))
                      : nb + nLeft + uint32_t(__popcll(~leftMask & nodeMask & below));
        }
        __syncthreads();
        if (open) {
            if (lane == nb) {
                const uint32_t second = nP + 2u * nLeft;
                nodes[2 * size_t(nP) + 0] = make_float4(nlo[0], nlo[1], nlo[2], nhi[0]);
                nodes[2 * size_t(nP) + 1] = make_float4(nhi[1], nhi[2], __uint_as_float(nP + 2u * nm - 1u), __uint_as_float(kInnerBit | second));
                firstLeaf[nP] = task.begin + nb;
            }
            // what this lane's place belongs to from now on
            const bool first = lane - nb < nLeft;
            for (int d = 0; d < 3; d++) {
                nlo[d] = sHalves[nb][(first ? 0 : 6) + d];
                nhi[d] = sHalves[nb][(first ? 3 : 9) + d];
            }
            nP = first ? nP + 1u : nP + 2u * nLeft;
            const uint32_t oldNb = nb;
            nb = first ? oldNb : oldNb + nLeft;
            nm = first ? nLeft : nm - nLeft;
        }
        __syncthreads();
        for (int k = 0; k < 3; k++) {
            sBox[to][k] = lo[k];
            sBox[to][3 + k] = hi[k];
        }
        sBox[to][6] = __uint_as_float(sid);
        __syncthreads();
        for (int k = 0; k < 3; k++) {
            lo[k] = sBox[lane][k];
            hi[k] = sBox[lane][3 + k];
        }
        sid = fbits(sBox[lane][6]);
        __syncthreads();
    }
    if (lane < m) {
        nodes[2 * size_t(nP) + 0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        nodes[2 * size_t(nP) + 1] = make_float4(hi[1], hi[2], __uint_as_float(nP + 1u), __uint_as_float(sid));
    }
}

// the rows of the nearer-child-first walk from a finished pre-order table; firstLeaf[P] = leaves before place P
__global__ void __launch_bounds__(kBlock) rowsFromNodesKernel(DevScene sc, const float4* __restrict__ nodes, const uint32_t* __restrict__ firstLeaf,
                                                              uint32_t nNodes, V3 base, V3 step, uint4* __restrict__ wide)
{
    const uint32_t P = blockIdx.x * kBlock + threadIdx.x;
    if (P >= nNodes)
        return;
    const uint32_t mine = fbits(nodes[2 * size_t(P) + 1].w);
    if (!(mine & kInnerBit))
        return;
    const uint32_t b = firstLeaf[P], second = mine & ~kInnerBit, nLeft = (second - P) / 2u;
    const uint32_t child[2] = { P + 1u, second }, childFirst[2] = { b, b + nLeft };
    uint32_t w[6], link[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t at = child[c];
        const float4 b0 = nodes[2 * size_t(at)], b1 = nodes[2 * size_t(at) + 1];
        const uint32_t qlx = gridBelow(b0.x, base.x, step.x), qly = gridBelow(b0.y, base.y, step.y), qlz = gridBelow(b0.z, base.z, step.z);
        const uint32_t qhx = gridAbove(b0.w, base.x, step.x), qhy = gridAbove(b1.x, base.y, step.y), qhz = gridAbove(b1.y, base.z, step.z);
        w[3 * c + 0] = qlx | (qly << 16);
        w[3 * c + 1] = qlz | (qhx << 16);
        w[3 * c + 2] = qhy | (qhz << 16);
        const uint32_t l = fbits(b1.w);
        if (l & kInnerBit)
            link[c] = kInnerBit | (at - childFirst[c]);
        else
            link[c] = fbits(sc.surf[3 * size_t(l)].w) == kTriStatic ? l : (kLeafRowBit | at);
    }
    const uint32_t row = P - b;
    wide[2 * size_t(row) + 0] = make_uint4(w[0], w[1], w[2], w[3]);
    wide[2 * size_t(row) + 1] = make_uint4(w[4], w[5], link[0], link[1]);
}

// ---- the rows of the four-wide walk (DevScene::quad) from a finished node table ---------------------------------
// A row stands for an inner node and holds up to four of its descendants: beginning with its two children, the inner
// node with the largest box surface among them (the first of equals) gives way to its own two children until there
// are four, or only leaves.  Which nodes get rows is found level by level from the root (quadMarkKernel); the rows
// are numbered in the order of the nodes they stand for (a scan over the marks), so a row's children rows come
// after it and a subtree's rows lie together; quadRowsKernel then writes them.

PT_DEV double boxSurface(const float4* __restrict__ nodes, uint32_t k)
{
    const float4 a = nodes[2 * size_t(k)], b = nodes[2 * size_t(k) + 1];
    const double dx = double(a.w) - double(a.x), dy = double(b.x) - double(a.y), dz = double(b.y) - double(a.z);
    return dx * dy + dy * dz + dz * dx;
}

PT_DEV uint32_t leafWordOf(const float4* __restrict__ nodes, uint32_t k) { return fbits(nodes[2 * size_t(k) + 1].w); }

// (four named slots and selects instead of an indexed array: the slots stay in registers)
struct QuadSlots {
    uint32_t c0, c1, c2, c3, n;
};

PT_DEV QuadSlots quadSlotsOf(const float4* __restrict__ nodes, uint32_t k, uint32_t nNodes)
{
    QuadSlots q;
    q.c0 = k + 1u;
    q.c1 = leafWordOf(nodes, k) & ~kInnerBit;
    q.c2 = q.c3 = 0u;
    q.n = 2u;
    for (int round = 0; round < 2; round++) {
        int pick = -1;
        double pickArea = -1.0;
        uint32_t parent = 0u;
        const uint32_t cand[4] = { q.c0, q.c1, q.c2, q.c3 };
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (uint32_t(c) < q.n && cand[c] < nNodes && (leafWordOf(nodes, cand[c]) & kInnerBit)) {
                const double area = boxSurface(nodes, cand[c]);
                if (area > pickArea) {
                    pick = c;
                    pickArea = area;
                    parent = cand[c];
                }
            }
        if (pick < 0)
            break;
        const uint32_t first = parent + 1u, second = leafWordOf(nodes, parent) & ~kInnerBit;
        q.c0 = pick == 0 ? first : q.c0;
        q.c1 = pick == 1 ? first : q.c1;
        q.c2 = pick == 2 ? first : q.c2;
        q.c3 = pick == 3 ? first : q.c3;
        q.c2 = q.n == 2u ? second : q.c2;
        q.c3 = q.n == 3u ? second : q.c3;
        q.n++;
    }
    return q;
}

// one level: the nodes of `in` get rows; their inner descendants that a row holds make up the next level
__global__ void __launch_bounds__(kBlock) quadMarkKernel(const float4* __restrict__ nodes, uint32_t nNodes, const uint32_t* __restrict__ in,
                                                         uint32_t nIn, uint32_t* __restrict__ out, uint32_t outRoom,
                                                         uint32_t* __restrict__ nOut, uint32_t* __restrict__ isRow)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nIn)
        return;
    const uint32_t k = in[i];
    if (k >= nNodes)
        return;
    isRow[k] = 1u;
    const QuadSlots q = quadSlotsOf(nodes, k, nNodes);
    const uint32_t child[4] = { q.c0, q.c1, q.c2, q.c3 };
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (uint32_t(c) < q.n && child[c] < nNodes && (leafWordOf(nodes, child[c]) & kInnerBit)) {
            const uint32_t at = atomicAdd(nOut, 1u);
            if (at < outRoom)
                out[at] = child[c];
        }
}

__global__ void __launch_bounds__(kBlock) quadRowsKernel(DevScene sc, const float4* __restrict__ nodes, uint32_t nNodes,
                                                         const uint32_t* __restrict__ isRow, const uint32_t* __restrict__ rowOf,
                                                         uint32_t rowRoom, V3 base, V3 step, uint4* __restrict__ quad)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nNodes || !isRow[k])
        return;
    const uint32_t row = rowOf[k];
    if (row >= rowRoom)
        return;
    const QuadSlots q = quadSlotsOf(nodes, k, nNodes);
    const uint32_t child[4] = { q.c0, q.c1, q.c2, q.c3 };
    uint32_t w[12], link[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // a slot not used: a box nothing enters
        w[3 * c + 0] = 0xFFFFFFFFu;
        w[3 * c + 1] = 0x0000FFFFu;
        w[3 * c + 2] = 0u;
        link[c] = kQuadNoLink;
        const uint32_t at = child[c];
        if (uint32_t(c) < q.n && at < nNodes) {
            const float4 b0 = nodes[2 * size_t(at)], b1 = nodes[2 * size_t(at) + 1];
            const uint32_t qlx = gridBelow(b0.x, base.x, step.x), qly = gridBelow(b0.y, base.y, step.y), qlz = gridBelow(b0.z, base.z, step.z);
            const uint32_t qhx = gridAbove(b0.w, base.x, step.x), qhy = gridAbove(b1.x, base.y, step.y), qhz = gridAbove(b1.y, base.z, step.z);
            w[3 * c + 0] = qlx | (qly << 16);
            w[3 * c + 1] = qlz | (qhx << 16);
            w[3 * c + 2] = qhy | (qhz << 16);
            const uint32_t l = fbits(b1.w);
            if (l & kInnerBit) {
                link[c] = kInnerBit | rowOf[at];
            } else {
                link[c] = kLeafRowBit | at;
                if (l < sc.n_surfaces && fbits(sc.surf[3 * size_t(l)].w) == kTriStatic) {
                    // a static triangle whose box is bit for bit that of its vertices is tested from its surface row alone
                    const float4 r0 = sc.surf[3 * size_t(l)], r1 = sc.surf[3 * size_t(l) + 1], r2 = sc.surf[3 * size_t(l) + 2];
                    const bool same = fbits(fminf(r0.x, fminf(r1.x, r2.x))) == fbits(b0.x) && fbits(fminf(r0.y, fminf(r1.y, r2.y))) == fbits(b0.y)
                        && fbits(fminf(r0.z, fminf(r1.z, r2.z))) == fbits(b0.z) && fbits(fmaxf(r0.x, fmaxf(r1.x, r2.x))) == fbits(b0.w)
                        && fbits(fmaxf(r0.y, fmaxf(r1.y, r2.y))) == fbits(b1.x) && fbits(fmaxf(r0.z, fmaxf(r1.z, r2.z))) == fbits(b1.y);
                    if (same)
                        link[c] = l;
                }
            }
        }
    }
    quad[4 * size_t(row) + 0] = make_uint4(w[0], w[1], w[2], w[3]);
    quad[4 * size_t(row) + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    quad[4 * size_t(row) + 2] = make_uint4(w[8], w[9], w[10], w[11]);
    quad[4 * size_t(row) + 3] = make_uint4(link[0], link[1], link[2], link[3]);
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

int ptBuildQuadRowsDevice(pt_scene* scene, hipStream_t stream)
{
    scene->dev.quad = nullptr;
    scene->nQuadRows = 0;
    const uint32_t nNodes = scene->dev.n_nodes;
    if (!scene->dev.wide || nNodes < 3 || scene->n_surfaces < 2)
        return PT_OK;
    const size_t rowRoom = size_t(scene->n_surfaces - 1);
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t scanBytes = 0;
    uint32_t* nullVals = nullptr;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scanBytes, nullVals, nullVals, 0u, size_t(nNodes), rocprim::plus<uint32_t>(), stream));
    const size_t need = 2 * pad(size_t(nNodes) * 4) + 2 * pad(rowRoom * 4) + pad(256) + pad(scanBytes);
    if (scene->quadCapacity < need) {
        if (scene->quadArea)
            HIP_TRY(hipFree(scene->quadArea));
        scene->quadArea = nullptr;
        scene->quadCapacity = 0;
        HIP_TRY(hipMalloc(&scene->quadArea, need));
        scene->quadCapacity = need;
    }
    if (!scene->quad) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, 4 * rowRoom * sizeof(uint4)));
        scene->owned.push_back(p);
        scene->quad = static_cast<uint4*>(p);
    }
    char* p = static_cast<char*>(scene->quadArea);
    uint32_t* isRow = carve<uint32_t>(p, size_t(nNodes));
    uint32_t* rowOf = carve<uint32_t>(p, size_t(nNodes));
    uint32_t* level[2] = { carve<uint32_t>(p, rowRoom), carve<uint32_t>(p, rowRoom) };
    uint32_t* counters = carve<uint32_t>(p, 64);
    void* scanTemp = p;
    HIP_TRY(hipMemsetAsync(isRow, 0, size_t(nNodes) * 4, stream));
    HIP_TRY(hipMemsetAsync(level[0], 0, 4, stream)); // the first level: the root
    uint32_t nIn = 1, total = 0;
    int cur = 0;
    // (a tree is at most as deep as it has inner nodes, and every level holds at least one of them)
    for (uint64_t guard = 0; nIn > 0 && guard < rowRoom; guard++) {
        total += nIn;
        if (total > rowRoom)
            return fail(PT_ERR_DEVICE, "four-wide rows: more rows than inner nodes");
        HIP_TRY(hipMemsetAsync(counters, 0, 4, stream));
        hipLaunchKernelGGL(quadMarkKernel, dim3((nIn + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, scene->nodes, nNodes, level[cur], nIn,
                           level[cur ^ 1], uint32_t(rowRoom), counters, isRow);
        HIP_TRY(hipGetLastError());
        uint32_t nOut = 0;
        HIP_TRY(hipMemcpyAsync(&nOut, counters, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (nOut > rowRoom)
            return fail(PT_ERR_DEVICE, "four-wide rows: a level with more nodes than the tree has");
        nIn = nOut;
        cur ^= 1;
    }
    HIP_TRY(rocprim::exclusive_scan(scanTemp, scanBytes, isRow, rowOf, 0u, size_t(nNodes), rocprim::plus<uint32_t>(), stream));
    const V3 base = v3(scene->dev.qbase[0], scene->dev.qbase[1], scene->dev.qbase[2]);
    const V3 step = v3(scene->dev.qstep[0], scene->dev.qstep[1], scene->dev.qstep[2]);
    hipLaunchKernelGGL(quadRowsKernel, dim3((nNodes + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, scene->dev, scene->nodes, nNodes, isRow, rowOf,
                       uint32_t(rowRoom), base, step, scene->quad);
    HIP_TRY(hipGetLastError());
    scene->nQuadRows = total;
    scene->dev.quad = scene->quad;
    return PT_OK;
}

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
    scene->dev.quad = nullptr;
    const DevScene& sc = scene->dev;
    const uint32_t leafBlocks = (uint32_t(n) + kBlock - 1) / kBlock;

    // PT_BVH_BUILDER=morton selects the radix tree over Morton codes; the default is the surface-area builder
    const char* which = std::getenv("PT_BVH_BUILDER");
    const bool morton = which && std::strcmp(which, "morton") == 0;
    if (which && !morton && std::strcmp(which, "sah") != 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "PT_BVH_BUILDER must be sah or morton");

    // work area: one allocation, kept with the scene for the next frame
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t sortBytes = 0, scanBytes = 0;
    const uint32_t maxActive = uint32_t(n) / (kSmallNode + 1u) + 1u;
    size_t need = pad(size_t(n) * 32) + pad(size_t(leafBlocks) * 32) + pad(64) + 4096;
    if (morton) {
        uint64_t* nullKeys = nullptr;
        uint32_t* nullVals = nullptr;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, sortBytes, nullKeys, nullKeys, nullVals, nullVals, size_t(n), 0, 63, stream));
        need += 2 * pad(size_t(n) * 8) + 2 * pad(size_t(n) * 4) + 2 * pad(size_t(n) * 8) + pad(size_t(2 * n) * 4) + pad(size_t(n) * 32)
            + pad(size_t(n) * 4) + pad(size_t(2 * n) * 4) + pad(sortBytes);
    } else {
        uint32_t* nullVals = nullptr;
        HIP_TRY(rocprim::exclusive_scan(nullptr, scanBytes, nullVals, nullVals, 0u, size_t(n), rocprim::plus<uint32_t>(), stream));
        need += 6 * pad(size_t(n) * 4) + 2 * pad(size_t(maxActive) * sizeof(ActiveNode)) + pad(size_t(maxActive) * sizeof(SplitRec))
            + pad(size_t(maxActive) * kNodeBinWords * 4) + pad(size_t(n) * sizeof(SmallTask)) + pad(256) + pad(size_t(2 * n) * 4)
            + pad(scanBytes);
    }
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

    hipLaunchKernelGGL(surfaceBoxesKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, sc, t0, t1, leafBox, blockBounds, morton ? 0 : 1);
    if (n == 1) {
        hipLaunchKernelGGL(singleLeafKernel, dim3(1), dim3(64), 0, stream, leafBox, scene->nodes);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        scene->dev.n_nodes = 1;
        scene->dev.wide = scene->wide; // no inner node: no row is ever read
        int rcLinks = ptWavefrontBuildWalkLinks(scene, t0, stream);
        if (rcLinks == PT_OK)
            rcLinks = ptBuildQuadRows(scene, stream);
        if (rcLinks != PT_OK) {
            scene->dev.n_nodes = 0;
            return rcLinks;
        }
        HIP_TRY(call.done(stream));
        return PT_OK;
    }
    hipLaunchKernelGGL(reduceBoundsKernel, dim3(1), dim3(kBlock), 0, stream, blockBounds, leafBlocks, bounds);
    const uint32_t innerBlocks = (uint32_t(n - 1) + kBlock - 1) / kBlock;
    const uint32_t allBlocks = (uint32_t(2 * n - 1) + kBlock - 1) / kBlock;
    // what the rows are made from afterwards
    uint2 *children = nullptr, *range = nullptr;
    uint32_t *orderSorted = nullptr, *place = nullptr, *firstLeaf = nullptr;
    if (morton) {
        uint64_t* codes = carve<uint64_t>(p, size_t(n));
        uint64_t* codesSorted = carve<uint64_t>(p, size_t(n));
        uint32_t* order = carve<uint32_t>(p, size_t(n));
        orderSorted = carve<uint32_t>(p, size_t(n));
        children = carve<uint2>(p, size_t(n));
        range = carve<uint2>(p, size_t(n));
        uint32_t* parent = carve<uint32_t>(p, 2 * size_t(n));
        float4* innerBox = carve<float4>(p, 2 * size_t(n));
        uint32_t* arrived = carve<uint32_t>(p, size_t(n));
        place = carve<uint32_t>(p, 2 * size_t(n));
        void* sortTemp = p;
        hipLaunchKernelGGL(mortonKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, leafBox, bounds, uint32_t(n), codes, order);
        HIP_TRY(hipGetLastError());
        HIP_TRY(rocprim::radix_sort_pairs(sortTemp, sortBytes, codes, codesSorted, order, orderSorted, size_t(n), 0, 63, stream));
        hipLaunchKernelGGL(hierarchyKernel, dim3(innerBlocks), dim3(kBlock), 0, stream, codesSorted, n, children, range, parent);
        HIP_TRY(hipMemsetAsync(arrived, 0, size_t(n) * 4, stream));
        hipLaunchKernelGGL(refitKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, leafBox, orderSorted, children, parent, n,
                           reinterpret_cast<float*>(innerBox), arrived);
        hipLaunchKernelGGL(layoutKernel, dim3(allBlocks), dim3(kBlock), 0, stream, leafBox, innerBox, orderSorted, children, range, parent,
                           n, scene->nodes, place);
        HIP_TRY(hipGetLastError());
    } else {
        uint32_t* idx[2] = { carve<uint32_t>(p, size_t(n)), carve<uint32_t>(p, size_t(n)) };
        uint32_t* nodeAt[2] = { carve<uint32_t>(p, size_t(n)), carve<uint32_t>(p, size_t(n)) };
        uint32_t* flags = carve<uint32_t>(p, size_t(n));
        uint32_t* leftBefore = carve<uint32_t>(p, size_t(n));
        ActiveNode* active[2] = { carve<ActiveNode>(p, maxActive), carve<ActiveNode>(p, maxActive) };
        SplitRec* splits = carve<SplitRec>(p, maxActive);
        uint32_t* bins = carve<uint32_t>(p, size_t(maxActive) * kNodeBinWords);
        SmallTask* small = carve<SmallTask>(p, size_t(n));
        uint32_t* counters = carve<uint32_t>(p, 64);
        firstLeaf = carve<uint32_t>(p, 2 * size_t(n));
        void* scanTemp = p;
        HIP_TRY(hipMemsetAsync(counters, 0, 256, stream));
        hipLaunchKernelGGL(sahStartKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, bounds, uint32_t(n), active[0], idx[0], nodeAt[0],
                           small, counters);
        HIP_TRY(hipGetLastError());
        uint32_t nActive = uint32_t(n) > kSmallNode ? 1u : 0u, nSmall = nActive ? 0u : 1u, largest = uint32_t(n);
        int cur = 0;
        for (int level = 0; nActive > 0; level++) {
            if (nActive > maxActive)
                return fail(PT_ERR_LIMIT, "surface-area builder: more live nodes than the work area holds");
            const size_t binWords = size_t(nActive) * kNodeBinWords;
            hipLaunchKernelGGL(clearBinsKernel, dim3(uint32_t((binWords + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, bins, binWords);
            if (largest > kWaveNodeMax)
                hipLaunchKernelGGL(binPlacesKernel, dim3((leafBlocks + kBinChunks - 1) / kBinChunks), dim3(kBlock), 0, stream, uint32_t(n),
                                   idx[cur], nodeAt[cur], active[cur], leafBox, bins);
            hipLaunchKernelGGL(binNodesKernel, dim3((nActive + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, stream, nActive, idx[cur],
                               active[cur], leafBox, bins);
            HIP_TRY(hipMemsetAsync(counters, 0, 8, stream));
            // a tree 48 levels deep over nodes of more than 64 surfaces is degenerate: halve from there on
            // (few nodes: a block each, for the latency; many: several per block, for fewer reservations)
            const uint32_t perBlock = nActive >= 2048u ? kSplitNodes : 1u;
            hipLaunchKernelGGL(splitKernel, dim3((nActive + perBlock - 1) / perBlock), dim3(128), 0, stream, active[cur], nActive, perBlock,
                               bins, level >= 48 ? 1 : 0, splits,
                               active[cur ^ 1], small, counters, scene->nodes, firstLeaf);
            hipLaunchKernelGGL(flagKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, uint32_t(n), idx[cur], nodeAt[cur], active[cur], splits,
                               leafBox, flags);
            HIP_TRY(hipGetLastError());
            HIP_TRY(rocprim::exclusive_scan(scanTemp, scanBytes, flags, leftBefore, 0u, size_t(n), rocprim::plus<uint32_t>(), stream));
            hipLaunchKernelGGL(scatterKernel, dim3(leafBlocks), dim3(kBlock), 0, stream, uint32_t(n), idx[cur], nodeAt[cur], active[cur],
                               splits, flags, leftBefore, idx[cur ^ 1], nodeAt[cur ^ 1]);
            HIP_TRY(hipGetLastError());
            uint32_t h[3] = { 0, 0, 0 };
            HIP_TRY(hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            nActive = h[0];
            largest = h[1];
            nSmall = h[2];
            cur ^= 1;
        }
        if (nSmall > uint32_t(n))
            return fail(PT_ERR_LIMIT, "surface-area builder: more subtrees than surfaces");
        hipLaunchKernelGGL(smallSubtreesKernel, dim3(nSmall), dim3(64), 0, stream, small, idx[cur], leafBox, scene->nodes, firstLeaf);
        HIP_TRY(hipGetLastError());
    }

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
        const V3 base = v3(scene->dev.qbase[0], scene->dev.qbase[1], scene->dev.qbase[2]);
        const V3 step = v3(scene->dev.qstep[0], scene->dev.qstep[1], scene->dev.qstep[2]);
        if (morton)
            hipLaunchKernelGGL(wideRowsKernel, dim3(innerBlocks), dim3(kBlock), 0, stream, sc, scene->nodes, children, range, place,
                               orderSorted, n, base, step, scene->wide);
        else
            hipLaunchKernelGGL(rowsFromNodesKernel, dim3(allBlocks), dim3(kBlock), 0, stream, sc, scene->nodes, firstLeaf,
                               uint32_t(2 * n - 1), base, step, scene->wide);
        HIP_TRY(hipGetLastError());
    }
    scene->dev.wide = gridOk ? scene->wide : nullptr; // without rows: storage-order walk only
    scene->dev.n_nodes = uint32_t(2 * n - 1);
    int rcLinks = ptWavefrontBuildWalkLinks(scene, t0, stream);
    if (rcLinks == PT_OK)
        rcLinks = ptBuildQuadRows(scene, stream);
    if (rcLinks != PT_OK) {
        scene->dev.n_nodes = 0;
        return rcLinks;
    }
    HIP_TRY(call.done(stream));
    return PT_OK;
}

int pt_scene_get_quad_rows(pt_scene* scene, uint32_t* out_rows, uint64_t capacity_rows, uint64_t* out_n_rows)
try {
    if (!scene || !out_n_rows)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    const uint64_t n = scene->dev.quad ? scene->nQuadRows : 0;
    *out_n_rows = n;
    if (!out_rows || capacity_rows < n)
        return out_rows ? fail(PT_ERR_INVALID_ARGUMENT, "row buffer too small") : PT_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (n > 0)
        HIP_TRY(hipMemcpy(out_rows, scene->quad, n * 4 * sizeof(uint4), hipMemcpyDeviceToHost));
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_scene_get_bvh(pt_scene* scene, pt_bvh_node* out_nodes, uint64_t capacity, uint64_t* out_count)
try {
    if (!scene || !out_count)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    // (the count is read under the scene's lock: a rebuild in another thread keeps it at zero while it runs)
    SceneCall call(scene, nullptr);
    HIP_TRY(call.status);
    const uint64_t n = scene->dev.n_nodes;
    *out_count = n;
    if (!out_nodes || capacity < n)
        return out_nodes ? fail(PT_ERR_INVALID_ARGUMENT, "node buffer too small") : PT_OK;
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
