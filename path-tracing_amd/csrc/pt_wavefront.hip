// pt_wavefront.hip -- the wavefront renderer: trace / classify / shade / generate
// kernels over two pools of path records in HBM, plus the resolve kernel.
//
// One path = one (pixel, sample).  Path records live in HBM as structure-of-arrays
// of 16-byte rows, in two pools that take turns.  Every iteration of the host loop
// runs, on the current pool,
//
//   traceKernel       closest hit (reference src/15/bvh.hpp:183-213) of the ray of
//   (or traceOrdered- every live record, and of every queued shadow ray: if that one
//   Kernel + trace-   reaches the chosen light, the prepared contribution is added
//   RetryKernel)      (reference src/15/pathtracer.cpp:99-107);
//   classifyKernel    puts the records whose ray hit a surface on a hit list and finishes
//                     the ones whose ray left the scene: environment added to the path's
//                     sample (reference src/15/pathtracer.cpp:60-62);
//   shadeHitsKernel   per hit: hit record, scatter, emission, light pdfs, light sample,
//                     MIS, Russian roulette (reference src/15/pathtracer.cpp:63-126);
//                     queues the shadow ray; a path that goes on is written densely
//                     into the other pool;
//   generateKernel    fills the rest of the other pool with new paths: path number,
//                     camera ray, fresh path state
//                     (reference src/15/pathtracer.cpp:202-215, camera.hpp:34-55);
//
// and then the pools change roles.  A path's radiance lives in its place of a sample buffer
// indexed [sample][pixel] from the start: emission, shadow-ray contributions and the
// environment are added there in the order the reference adds them.  resolveKernel then adds
// each pixel's samples in sample order -- the same sequence of float additions as the
// reference's accumulation loop (src/15/pathtracer.cpp:207-220), which keeps the image
// bit-identical to a sequential evaluation however the paths were scheduled.  All arithmetic
// per path is that of pt_device.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "pt_scene_host.hpp"

using namespace ptd;
using ptint::fail;

namespace {

enum : uint32_t { kEmpty = 0, kActive = 1 }; // state of a record: no path / a path with a ray to trace
enum : uint8_t { kClsEmpty = 0, kClsMiss = 1, kClsHit = 2 }; // a hit is kClsHit + the type of the surface's material
constexpr uint32_t kHitClasses = 8;
constexpr uint32_t kBackEmits = 0x40000000u;
constexpr uint32_t kLightMask = 0x3FFFFFFFu;
constexpr uint32_t kRetryIsShadow = 0x80000000u; // retry list entry: a shadow-queue entry rather than a slot
constexpr uint32_t kRetryFullWalk = 0x80000000u; // second word of the entry: the bound itself, not the closest distance
#ifndef PT_RETRY_PRUNE
#define PT_RETRY_PRUNE 1
#endif

struct Pool {
    float4* o_t;   // ray origin, time
    float4* d_s;   // ray direction, bits: random draws made << 16 | state << 12 | segment (stateWord)
    uint4* hit;    // surface | backside << 31, bits a, bits u, bits v
    float4* thr_p; // throughput, bits: index of the path in the sample buffer -- where its radiance so far lives: what
                   // a path gathers (emission at a hit, a shadow ray's contribution, the environment) is added to
                   // its sample in place, in the order the reference adds it, so radiance does not travel with the
                   // record (a sample that is never added to keeps the +0 its path started with, and x + (+-0) = x
                   // for every sum that began at +0)
    uint2* key;    // generator key of the path
    uint8_t* cls;  // what the tracing found for the record: kClsEmpty / kClsMiss / kClsHit + class (classifyKernel reads
                   // only this byte per record)
    float4* sh_o;  // shadow ray queue: origin, time
    float4* sh_d;  //   direction, bits: sample-buffer index of the path
    float4* sh_c;  //   what to add if the ray reaches the front side of the chosen light, bits: light
                   //   surface | kBackEmits
    float4* sh_b;  //   what to add if it reaches the back side (only written when kBackEmits)
    uint2* retry;  // rays to trace again in storage order: (slot, or kRetryIsShadow | shadow-queue entry; bits of
                   // the distance beyond which nothing needs to be looked at), kRetryShards sub-lists
    uint32_t retryCap; // entries per sub-list
    uint32_t* hitList;  // per shard: records whose ray hit a surface (to be shaded)
    uint32_t listCap;   // entries per shard of either list (= the records of the shard's blocks, rounded up)
    uint32_t nSlots;    // records per pool: a multiple of kClassifyRecords * kShards; shard q's region is [q, q + 1) * nSlots / kShards
};

// geometry of one batch of paths
// Division of 32-bit numbers by a divisor fixed for a launch: the quotient is the upper half of n * ceil(2^64 / d),
// exact for every 32-bit n (d = 1 has no such multiplier and is passed through).  A handful of instructions
// instead of the thirty of a general division: generateKernel makes four per path.
struct FastDiv {
    uint32_t d = 1, mhi = 0, mlo = 0;
    FastDiv() = default;
    explicit FastDiv(uint32_t divisor) : d(divisor ? divisor : 1u)
    {
        if (d > 1) {
            const uint64_t m = ~0ull / d + 1ull;
            mhi = uint32_t(m >> 32);
            mlo = uint32_t(m);
        }
    }
    __host__ __device__ uint32_t div(uint32_t n) const
    {
        if (d == 1)
            return n;
        const uint64_t carry = (uint64_t(mlo) * n) >> 32;
        return uint32_t((uint64_t(mhi) * n + carry) >> 32);
    }
};

struct Batch {
    uint32_t tilesX, tilesY; // 8x8 pixel tiles covering the tile rectangle
    FastDiv byTilesX, byNS, bySqrtSpp;
    uint32_t nS;             // samples per pixel in this batch
    uint32_t sFirst;         // first sample index of the batch
    uint32_t npix;           // pixels of the tile rectangle
    uint32_t totalIds;       // tilesX * tilesY * 64 * nS
};

// Counters, one per 128-byte line (word index = line * 32), kShards = S:
//   0..S-1        path numbers handed out by shard q (a shard = the blocks of classifyKernel with
//                 blockIdx % S == q, their lists, and region q of a pool)
//   S+1..3S       fill of the shadow-ray queue of shard q, double buffered by iteration parity
//   3S+1..3S+64   rays the nearer-child-first tracing left for the storage-order pass, per sub-list
//   then S each   entries of the hit list, paths written to region q of the other pool in this
//                 iteration, and records in use at the front of region q of the pool traced next
// Shards keep the returning atomics of one launch off a common address: one address serves
// only about 90 of them per microsecond.  (Measured: one atomic per block on 8 shards beats one
// per wave on 16..256 shards, which also fragments the shadow-ray queues.)
#ifndef PT_SHARDS
#define PT_SHARDS 8
#endif
constexpr uint32_t kShards = PT_SHARDS;
constexpr uint32_t kCtrStride = 32;
constexpr uint32_t kCtrShadow = kShards + 1;
constexpr uint32_t kRetryShards = 64;            // sub-lists of the retry list (a wave appends to list wave % 64)
constexpr uint32_t kCtrRetry = 3 * kShards + 1;  // .. + kRetryShards - 1: entries in the retry sub-lists
constexpr uint32_t kCtrHit = kCtrRetry + kRetryShards; // .. + S - 1: entries in the hit list of shard q
constexpr uint32_t kCtrLive = kCtrHit + kShards;        // .. + S - 1: continuing paths written to region q of the next pool
constexpr uint32_t kCtrOcc = kCtrLive + kShards;         // .. + S - 1: records in use at the front of region q of the pool traced next
constexpr uint32_t kCtrWork = kCtrOcc + kShards;        // .. + 2 S - 1: runs handed out of region q / of shadow queue q in this tracing launch
constexpr uint32_t kCtrWords = (kCtrWork + 2 * kShards) * kCtrStride;

PT_DEV uint32_t laneId() { return __lane_id(); }

// A sample = three floats at a 4-byte-aligned place of the sample buffer, moved with one 12-byte access.
struct __attribute__((packed, aligned(4))) Sample3 {
    float x, y, z;
};
PT_DEV V3 loadSample(const float* samples, size_t index)
{
    const Sample3 v = *reinterpret_cast<const Sample3*>(samples + 3 * index);
    return v3(v.x, v.y, v.z);
}
PT_DEV void storeSample(float* samples, size_t index, V3 v)
{
    *reinterpret_cast<Sample3*>(samples + 3 * index) = Sample3{ v.x, v.y, v.z };
}

// last word of a record's direction row: segments below 4096 (checkRender), at most eight draws per segment
PT_DEV uint32_t stateWord(uint32_t state, uint32_t segment, uint32_t draws) { return draws << 16 | state << 12 | segment; }
PT_DEV uint32_t stateOf(uint32_t w) { return (w >> 12) & 3u; }
PT_DEV uint32_t segmentOf(uint32_t w) { return w & 0xFFFu; }
PT_DEV uint32_t drawsOf(uint32_t w) { return w >> 16; }

// (the count of path numbers handed out by a shard keeps growing while a batch drains, a region's worth per iteration:
// it must stop at the top rather than start again)
PT_DEV uint32_t addSaturating(uint32_t a, uint32_t b) { return a > 0xFFFFFFFFu - b ? 0xFFFFFFFFu : a + b; }

// Two independent reservations for the threads of a block of WAVES waves: one unit for every
// thread that wants one, by wave ballots, a prefix sum over the wave counts in LDS, and ONE
// atomic per block and reservation (two barriers).
template <uint32_t WAVES>
PT_DEV void blockReserve2(uint32_t* counterA, bool wantA, uint32_t* counterB, bool wantB,
                          uint32_t* lds /* 2 * (WAVES + 1) words */, uint32_t& posA, uint32_t& posB)
{
    static_assert(WAVES >= 2, "threads 0 and 64 make the two reservations: a block needs two waves");
    const uint64_t maskA = __builtin_amdgcn_ballot_w64(wantA), maskB = __builtin_amdgcn_ballot_w64(wantB);
    const uint32_t wave = threadIdx.x >> 6, lane = laneId();
    uint32_t* ldsA = lds;
    uint32_t* ldsB = lds + WAVES + 1;
    if (lane == 0) {
        ldsA[wave] = uint32_t(__popcll(maskA));
        ldsB[wave] = uint32_t(__popcll(maskB));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < WAVES; w++)
            total += ldsA[w];
        ldsA[WAVES] = total ? atomicAdd(counterA, total) : 0u;
    }
    if (threadIdx.x == 64) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < WAVES; w++)
            total += ldsB[w];
        ldsB[WAVES] = total ? atomicAdd(counterB, total) : 0u;
    }
    __syncthreads();
    uint32_t baseA = ldsA[WAVES], baseB = ldsB[WAVES];
    for (uint32_t w = 0; w < wave; w++) {
        baseA += ldsA[w];
        baseB += ldsB[w];
    }
    const uint64_t below = (1ull << lane) - 1ull;
    posA = baseA + uint32_t(__popcll(maskA & below));
    posB = baseB + uint32_t(__popcll(maskB & below));
}

// A shadow ray has found its closest hit: if that is the chosen light, add the prepared
// contribution to the path's sample (reference
// src/15/pathtracer.cpp:102-107).  Which side of the light: triangles know from the hit test,
// spheres from the hit point (reference src/15/surface_sphere.hpp:29-34).
template <bool DYN>
PT_DEV void shadowArrive(const DevScene& sc, const float4* __restrict__ surf, const Pool& pool, float* __restrict__ samples,
                         uint32_t entry, const Ray& ray, const HitLite& best, float4 payload, uint32_t dest,
                         V3 destValue)
{
    if (best.sid == PT_NO_HIT || best.sid != (fbits(payload.w) & kLightMask))
        return;
    bool back = best.backside;
    const float4 l0 = surf[3 * best.sid + 0];
    if (fbits(l0.w) > kTriDynamic) {
        V3 c = v3(l0.x, l0.y, l0.z);
        if (DYN && fbits(l0.w) == kSphDynamic) {
            float r;
            Xform T;
            uint32_t sphereId;
            sphereCR(sc, best.sid, ray.time, c, r, T, sphereId);
        }
        V3 n = normalize(rayAt(ray, best.a) - c);
        back = dot(n, -ray.direction) < 0.0f;
    }
    V3 add = v3(payload.x, payload.y, payload.z);
    if (back) {
        if (!(fbits(payload.w) & kBackEmits))
            return;
        const float4 pb = pool.sh_b[entry];
        add = v3(pb.x, pb.y, pb.z);
    }
    storeSample(samples, dest, destValue + add);
}

// Closest-hit tracing of a stream of rays.  Every wave owns a contiguous run of
// items and keeps its 64 lanes busy: a lane whose ray has finished writes its
// result and takes the wave's next item, so lanes do not wait for the longest ray
// of a fixed group of 64.  The hand-out is a ballot and a popcount inside the wave;
// no memory atomics.  One launch covers both kinds of item:
//   waves [0, slotWaves)   items are path slots (rays of live paths; result = hit record)
//   waves [slotWaves, ..)  items are queued shadow rays (result = contribution added if
//                          the chosen light is hit)
// Within a lane the walk alternates two phases: node steps, repeated while at least
// kWalkLanes lanes of the wave are walking, and then the surface tests that became due
// (a leaf whose box was hit), done together.  The order of tests seen by one ray is
// unchanged.
#ifndef PT_ITEMS_PER_LANE
#define PT_ITEMS_PER_LANE 8
#endif
#ifndef PT_REFILL_LANES
#define PT_REFILL_LANES 32
#endif
#ifndef PT_WALK_LANES
#define PT_WALK_LANES 16
#endif
#ifndef PT_STEPS_PER_VOTE
#define PT_STEPS_PER_VOTE 4 // node steps of the LDS walk between two votes on going on
#endif
constexpr uint32_t kItemsPerLane = PT_ITEMS_PER_LANE;
constexpr uint32_t kRefillLanes = PT_REFILL_LANES; // refill once this many lanes are idle
#ifndef PT_REFILL_LANES_ORDERED
#define PT_REFILL_LANES_ORDERED 16
#endif
#ifndef PT_WALK_LANES_ORDERED
#define PT_WALK_LANES_ORDERED 32
#endif
constexpr uint32_t kRefillLanesOrdered = PT_REFILL_LANES_ORDERED; // the same for traceOrderedKernel
constexpr uint32_t kWalkLanesOrdered = PT_WALK_LANES_ORDERED;
constexpr uint32_t kWalkLanes = PT_WALK_LANES;     // keep walking nodes while this many lanes do

// A run of items inside a part (a region of a pool, a shard's shadow-ray queue) of `cap` items of which the first
// `fill` are in use: 64 * kItemsPerLane consecutive ones -- or 64 when the part is at most 1 / kItemsPerLane full, so
// that the few items at the front of a draining pool are spread over that many more waves (a wave works a run off
// 64 at a time: with whole runs the last paths of a batch kept a few hundred waves busy for eight rounds each while
// the rest of the chip had nothing to do).
PT_DEV uint32_t runLength(uint32_t cap, uint32_t fill, uint32_t itemsPerLane = kItemsPerLane)
{
    return fill * itemsPerLane <= cap ? 64u : 64u * itemsPerLane;
}
// (the runs that the waves of traceOrderedKernel take from a counter are shorter: 256 against 512 items measured 4 %
// faster on both scenes in memory, the hand-out at the end of the launch is finer; the LDS kernel's stay at 512: with
// its short walks 256 cost 1.5 %, 128 18 %)
#ifndef PT_TAKEN_ITEMS_PER_LANE
#define PT_TAKEN_ITEMS_PER_LANE 4
#endif
constexpr uint32_t kTakenItemsPerLane = PT_TAKEN_ITEMS_PER_LANE;

// Two ways of giving the waves of a tracing launch their runs.  Fixed: a wave per run, the launch covers all parts
// (the hardware hands out blocks as others finish).  Taken: as many waves as the chip holds at once, which take run
// after run from counters, one per region of the pool (kind 0) and one per shadow-ray queue (kind 1), cleared by
// classifyKernel between two tracing launches: a wave that has worked off a run takes the next one and keeps its
// lanes filled across the seam, so the thinning out at the end of a run happens once per wave and kind instead of
// once per 512 rays.  Measured per kernel (see launchTrace): taken runs win where rays are long (scenes in memory,
// moving surfaces: -5 to -10 %), fixed runs for the short walks of small static scenes (taken: +9 %).
struct RunTaker {
    uint32_t kind, part; // what the wave works on, and where it looks for its next run first
};

// items in use in each part: [kind * kShards + q]
PT_DEV void fillsOfParts(const uint32_t* __restrict__ counters, int parity, uint32_t* sFill)
{
    if (threadIdx.x < 2u * kShards) {
        const uint32_t q = threadIdx.x % kShards;
        sFill[threadIdx.x] = threadIdx.x < kShards ? counters[(kCtrOcc + q) * kCtrStride]
                                                   : counters[(kCtrShadow + uint32_t(parity ^ 1) * kShards + q) * kCtrStride];
    }
}

// The next run of the wave's kind: from its part on, the first part with a run left; false when there is none.
// (Everything here is the same in every lane.)
PT_DEV bool takeRun(uint32_t* __restrict__ counters, const uint32_t* sFill, RunTaker& t, uint32_t regionCap, uint32_t shadowCap,
                    uint32_t itemsPerLane, uint32_t& pos, uint32_t& endPos)
{
    for (uint32_t tries = 0; tries < kShards; tries++, t.part = (t.part + 1u) % kShards) {
        const uint32_t cap = t.kind == 0 ? regionCap : shadowCap;
        const uint32_t fill = sFill[t.kind * kShards + t.part];
        const uint32_t len = runLength(cap, fill, itemsPerLane);
        const uint32_t runs = (fill + len - 1u) / len;
        uint32_t* next = &counters[(kCtrWork + t.kind * kShards + t.part) * kCtrStride];
        if (__hip_atomic_load(next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= runs)
            continue;
        uint32_t r = 0;
        if (laneId() == 0)
            r = atomicAdd(next, 1u);
        r = uint32_t(__builtin_amdgcn_readfirstlane(int(r)));
        if (r < runs) {
            // (items are numbered through the whole pool, or through all queues)
            pos = t.part * cap + r * len;
            endPos = t.part * cap + min(r * len + len, fill);
            return true;
        }
    }
    return false;
}

// The one run of wave `waveInKind` of a launch with fixed runs; false if its part has no such run.
PT_DEV bool fixedRun(const uint32_t* sFill, uint32_t kind, uint32_t waveInKind, uint32_t regionCap, uint32_t shadowCap, uint32_t& pos,
                     uint32_t& endPos)
{
    const uint32_t cap = kind == 0 ? regionCap : shadowCap;
    const uint32_t wavesPerPart = cap / (64u * kItemsPerLane), part = waveInKind / wavesPerPart;
    if (part >= kShards)
        return false;
    const uint32_t fill = sFill[kind * kShards + part], len = runLength(cap, fill);
    const uint32_t first = (waveInKind % wavesPerPart) * len;
    pos = part * cap + first;
    endPos = part * cap + min(first + len, fill);
    return first < fill;
}

// Work of the walk a tracing kernel actually executes, summed per launch when STATS is on (an instrumented
// build of the kernel, chosen at launch; the plain one carries none of this).  Index into the scene's block of
// 64-bit sums:
enum : uint32_t {
    kStatPathRays = 0, // rays of paths traced by the first pass
    kStatShadowRays,   // queued shadow rays traced by the first pass
    kStatBoxRows,      // 32-byte rows fetched for box tests: one node (storage-order walk) or the two children of an
                       // inner node (nearer-child-first walk)
    kStatLeafRows,     // nearer-child-first walk: exact leaf boxes read from the node table (32 bytes)
    kStatSurfaceTests, // triangle / sphere tests (48-byte surface rows)
    kStatRetryRays,    // rays handed to the storage-order second pass
    kStatRetryNodes,   //   its node visits (32 bytes each)
    kStatRetrySurfaceTests,
    kStatNullShadowRays, // shadow rays the shading kernel did not queue because they could only add zeros
    kStatWords
};

template <bool STATS>
PT_DEV void addStats(unsigned long long* __restrict__ stats, uint32_t which, uint32_t mine)
{
    if (!STATS)
        return;
    // sum over the wave, one atomic per wave and counter
    uint32_t v = mine;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    if (laneId() == 0 && v)
        atomicAdd(&stats[which], (unsigned long long)v);
}

// BLOCK = threads per block: 256, or one wave where the scene's tables are small enough for 32 blocks per compute
// unit to hold their own copy in LDS (a block keeps its LDS until its slowest wave is through: one-wave blocks
// measured 2 % faster on the Cornell box, and 30 % slower with the 12.5 KB of the 100 spheres).
// OCT: the node table in LDS holds a copy per combination of direction signs (nodeStepOctant); for static scenes of
// a few dozen nodes, whose eight-fold table still leaves room for eight blocks per compute unit.
template <bool DYN, bool LDS, bool STATS, uint32_t BLOCK, bool OCT = false>
__global__ void __launch_bounds__(BLOCK) traceKernel(DevScene sc, Pool pool, uint32_t* __restrict__ counters, int parity,
                                                   float* __restrict__ samples, uint32_t shadowCap, uint32_t slotWaves,
                                                   int taken, unsigned long long* __restrict__ stats)
{
    uint32_t nRays = 0, nShadowRays = 0, nRows = 0, nTests = 0;
    // the queues this iteration's shadeKernel fills were read by the previous launch of this
    // kernel; this launch reads the other set
    if (blockIdx.x == 0)
        for (uint32_t q = threadIdx.x; q < kShards; q += BLOCK) {
            counters[(kCtrShadow + parity * kShards + q) * kCtrStride] = 0;
            counters[(kCtrHit + q) * kCtrStride] = 0; // the list of the previous iteration has been worked off
            // the generateKernel before this launch gave path numbers to the places of region q that the
            // continuing paths had left free
            counters[q * kCtrStride] = addSaturating(counters[q * kCtrStride], pool.nSlots / kShards - counters[(kCtrLive + q) * kCtrStride]);
            counters[(kCtrLive + q) * kCtrStride] = 0;
        }
    // small scenes: node and surface tables are copied into LDS once per block, which takes the
    // per-lane gathers of the walk off the vector memory path
    extern __shared__ float4 ldsGeo[];
    __shared__ uint32_t sFill[2 * kShards];
    fillsOfParts(counters, parity, sFill);
    const float4* nodes = sc.nodes;
    const float4* surf = sc.surf;
    const pt_animation* anims = sc.animations;
    __syncthreads();
    const uint32_t waveIdx = uint32_t(__builtin_amdgcn_readfirstlane(int((blockIdx.x * BLOCK + threadIdx.x) >> 6)));
    const uint32_t regionCap = pool.nSlots / kShards;
    uint32_t pos = 0, endPos = 0; // the rest of the run in hand
    bool fixedRunHere = false;
    if (!taken) {
        // fixed runs: a block none of whose waves has a run (most of them while the pool drains) is through before it
        // copies any tables
        fixedRunHere = fixedRun(sFill, waveIdx >= slotWaves ? 1u : 0u, waveIdx >= slotWaves ? waveIdx - slotWaves : waveIdx, regionCap,
                                shadowCap, pos, endPos);
        if (!__syncthreads_or(fixedRunHere ? 1 : 0))
            return;
    }
    if (LDS) {
        static_assert(!OCT || (LDS && !DYN), "copies per direction sign: static scenes in LDS only");
        const uint32_t nNode4 = (OCT ? kOctantRow4 : 2u) * sc.n_nodes, nSurf4 = 3u * sc.n_surfaces;
        // the two link words of every node as this walk wants them: from the scene's table, which may leave inner
        // nodes out of the walk (DevScene::walkLinks), else made here
        if (OCT) {
            for (uint32_t node = threadIdx.x; node < sc.n_nodes; node += BLOCK) {
                const float4 n0 = sc.nodes[2u * node];
                float4 n1 = sc.nodes[2u * node + 1u];
                if (sc.walkLinks) {
                    const uint2 l = sc.walkLinks[node];
                    n1.z = __uint_as_float(l.x);
                    n1.w = __uint_as_float(l.y);
                } else {
                    packNodeLinks(node, n1);
                }
#pragma unroll
                for (uint32_t octant = 0; octant < 8u; octant++) {
                    float4 r0, r1;
                    octantRow(octant, n0, n1, r0, r1);
                    ldsGeo[2u * (octant * sc.n_nodes + node)] = r0;
                    ldsGeo[2u * (octant * sc.n_nodes + node) + 1u] = r1;
                }
            }
        } else {
            for (uint32_t i = threadIdx.x; i < nNode4; i += BLOCK) {
                float4 v = sc.nodes[i];
                if (i & 1u) {
                    if (sc.walkLinks) {
                        const uint2 l = sc.walkLinks[i >> 1];
                        v.z = __uint_as_float(l.x);
                        v.w = __uint_as_float(l.y);
                    } else {
                        packNodeLinks(i >> 1, v);
                    }
                }
                ldsGeo[i] = v;
            }
        }
        for (uint32_t i = threadIdx.x; i < nSurf4; i += BLOCK)
            ldsGeo[nNode4 + i] = sc.surf[i];
        if (DYN) {
            // moving surfaces: the poses they move between as well (a test of one reads its two keys)
            static_assert(sizeof(pt_animation) % 4 == 0, "copied by words");
            uint32_t* to = reinterpret_cast<uint32_t*>(ldsGeo + nNode4 + nSurf4);
            const uint32_t* from = reinterpret_cast<const uint32_t*>(sc.animations);
            const uint32_t nWords = sc.n_animations * uint32_t(sizeof(pt_animation) / 4);
            for (uint32_t i = threadIdx.x; i < nWords; i += BLOCK)
                to[i] = from[i];
            anims = reinterpret_cast<const pt_animation*>(to);
        }
        __syncthreads();
        nodes = ldsGeo;
        surf = ldsGeo + nNode4;
    }
    const uint32_t lane = laneId();
    const uint64_t below = (1ull << lane) - 1ull;
    // Tables in LDS: where the walk begins.  Without a link table, at the root's first child: the root's box is the
    // merge of its children's, and the slab arithmetic is monotone in the planes, so a ray that misses it misses
    // both children too -- the same surfaces get tested, one box test fewer for every ray that hits it (which is
    // every ray that starts inside the scene).  A link table names its own first node, by the same argument.
    uint32_t walkStart = 0;
    if (LDS) {
        walkStart = sc.n_nodes > 1u ? 1u : 0u;
        if (sc.walkLinks)
            walkStart = uint32_t(__builtin_amdgcn_readfirstlane(int(sc.walkLinks[sc.n_nodes].x)));
    }
    // 0: rays of paths (items are pool records), 1: shadow rays (items are queue entries); half of the waves begin
    // with either kind and go over to the other when theirs is worked off and their lanes are empty (see takeRun)
    // taken runs: half of the waves begin with either kind and go over to the other when theirs is worked off and
    // their lanes are empty; fixed runs: the first slotWaves waves have the pool's regions, the others the queues
    RunTaker taker = { taken ? (waveIdx / kShards) & 1u : (waveIdx >= slotWaves ? 1u : 0u), waveIdx % kShards };
    bool kindDone = false, otherKindTried = false, allDone = false;
    if (!taken) {
        kindDone = otherKindTried = allDone = true;
        if (!fixedRunHere)
            return; // (no barrier follows)
    }

    bool have = false;
    uint32_t item = 0;
    Ray ray = makeRay(v3(0.0f), v3(1.0f), 0.0f);
    float amax = 0.0f;
    // Tables in LDS: the links are packed (packNodeLinks) and `cur` carries the lane's state (nodeStepPacked), so
    // that "does this lane walk" is one comparison.  Tables in memory: `pending` = the leaf surface whose box was
    // hit and whose test is due.
    uint32_t cur = LDS ? kNoWalk : 0u;
    uint32_t pending = kInner;
    const float4* rows = nodes; // OCT: the ray's copy of the node table
    HitLite best = noHit();
    // shadow rays: what to add on arrival, where, and the current value there
    float4 payload = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    V3 destValue = v3(0.0f);
    uint32_t dest = 0;

    for (;;) {
        const uint64_t idle = __builtin_amdgcn_ballot_w64(!have);
        const uint32_t nIdle = uint32_t(__popcll(idle));
        if (pos >= endPos && !allDone && nIdle >= kRefillLanes) {
            if (!kindDone && !takeRun(counters, sFill, taker, regionCap, shadowCap, kItemsPerLane, pos, endPos))
                kindDone = true;
            if (kindDone && nIdle == 64u) {
                if (!otherKindTried) {
                    otherKindTried = true;
                    taker.kind ^= 1u;
                    taker.part = waveIdx % kShards;
                    kindDone = !takeRun(counters, sFill, taker, regionCap, shadowCap, kItemsPerLane, pos, endPos);
                }
                allDone = kindDone;
            }
        }
        const bool shadow = taker.kind != 0u; // the same for every lane of the wave, and for every ray it has under way
        if (nIdle == 64u && pos >= endPos && allDone)
            break;
        if (pos < endPos && (nIdle >= kRefillLanes || nIdle == 64u)) {
            const uint32_t cand = pos + uint32_t(__popcll(idle & below));
            if (!have && cand < endPos) {
                item = cand;
                if (!shadow) {
                    const float4 d_s = pool.d_s[item];
                    if (stateOf(fbits(d_s.w)) == kActive) {
                        const float4 o_t = pool.o_t[item];
                        ray = makeRay(v3(o_t.x, o_t.y, o_t.z), v3(d_s.x, d_s.y, d_s.z), o_t.w);
                        have = true;
                    }
                } else {
                    const float4 so = pool.sh_o[item], sd = pool.sh_d[item];
                    payload = pool.sh_c[item];
                    dest = fbits(sd.w);
                    destValue = loadSample(samples, dest);
                    ray = makeRay(v3(so.x, so.y, so.z), v3(sd.x, sd.y, sd.z), so.w);
                    have = true;
                }
                if (have) {
                    if (OCT)
                        rows = nodes + 2u * sc.n_nodes * octantOf(ray);
                    amax = kFltMax;
                    cur = walkStart;
                    pending = kInner;
                    best = noHit();
                    if (STATS) {
                        nRays += shadow ? 0u : 1u;
                        nShadowRays += shadow ? 1u : 0u;
                    }
                }
            }
            pos = min(pos + nIdle, endPos);
        }
        // walk: node steps for every lane that still has nodes to visit and no surface waiting,
        // repeated while enough lanes are walking; then the waiting surfaces are tested together
        if (LDS) {
            uint64_t walkers = __builtin_amdgcn_ballot_w64(cur < sc.n_nodes);
            while (walkers != 0) {
#pragma unroll
                for (int k = 0; k < PT_STEPS_PER_VOTE; k++) {
                    if (cur < sc.n_nodes) {
                        if (OCT)
                            nodeStepOctant(rows, ray, 0.0001f, amax, cur, pending);
                        else
                            nodeStepPacked(nodes, ray, 0.0001f, amax, cur, pending);
                        if (STATS)
                            nRows++;
                    }
                }
                walkers = __builtin_amdgcn_ballot_w64(cur < sc.n_nodes);
                if (__builtin_popcountll(walkers) < int(kWalkLanes))
                    break;
            }
            if (cur & kLeafDue) {
                if (surfaceTest<DYN>(sc, surf, anims, (pending >> 16) - 1u, ray, 0.0001f, amax, best))
                    amax = best.a;
                cur &= ~kLeafDue;
                if (STATS)
                    nTests++;
            }
        } else {
            bool walking = have && pending == kInner && cur < sc.n_nodes;
            while (__builtin_amdgcn_ballot_w64(walking) != 0) {
                if (walking) {
                    nodeStep<!LDS>(nodes, ray, 0.0001f, amax, cur, pending);
                    if (STATS)
                        nRows++;
                }
                walking = have && pending == kInner && cur < sc.n_nodes;
                // (two steps per vote: the votes and branches cost more than the lanes idling in the second)
                if (walking) {
                    nodeStep<!LDS>(nodes, ray, 0.0001f, amax, cur, pending);
                    if (STATS)
                        nRows++;
                }
                walking = have && pending == kInner && cur < sc.n_nodes;
                if (uint32_t(__popcll(__builtin_amdgcn_ballot_w64(walking))) < kWalkLanes)
                    break;
            }
            if (have && pending != kInner) {
                if (surfaceTest<DYN>(sc, surf, anims, pending, ray, 0.0001f, amax, best))
                    amax = best.a;
                pending = kInner;
                if (STATS)
                    nTests++;
            }
        }
        if (have && cur >= sc.n_nodes) {
            have = false;
            if (LDS)
                cur = kNoWalk;
            if (!shadow) {
                pool.hit[item] = make_uint4(best.sid == PT_NO_HIT ? PT_NO_HIT : (best.sid | (best.backside ? 0x80000000u : 0u)),
                                            fbits(best.a), fbits(best.u), fbits(best.v));
                pool.cls[item] = best.sid == PT_NO_HIT ? uint8_t(kClsMiss) : uint8_t(kClsHit + best.matClass);
            } else {
                shadowArrive<DYN>(sc, surf, pool, samples, item, ray, best, payload, dest, destValue);
            }
        }
    }
    addStats<STATS>(stats, kStatPathRays, nRays);
    addStats<STATS>(stats, kStatShadowRays, nShadowRays);
    addStats<STATS>(stats, kStatBoxRows, nRows);
    addStats<STATS>(stats, kStatSurfaceTests, nTests);
}

// The same job for scenes whose tree lives in memory, visiting the nearer child first.
//
// The reference walks the tree in storage order (traceKernel does the same), so a ray tests
// every box it passes through until the surfaces in front happen to come up.  Going to the
// nearer child first and skipping what lies beyond the closest hit so far tests a fraction of
// the boxes.  The result is the reference's whenever it is unambiguous.  The reference accepts
// a surface when it reaches it, unless a box above it is entered (by the box test's
// arithmetic) beyond the closest distance accepted before; and after the closest surface it
// accepts only an exact tie.  So the closest surface found here is the reference's result if
// no other surface is hit before
//     limit = max(its distance, entry distance of the boxes above it) * (1 + kTieBand);
// the band also covers the rounding difference between a box's entry distance and a surface's
// hit distance inside that box, which is what allows skipping subtrees entered beyond the limit.
// A ray whose result is ambiguous by this rule is put on the retry list and traced in storage
// order by traceRetryKernel; so is a ray that had to leave out a subtree for lack of stack,
// unless that subtree is entered beyond the limit.
//
// Per lane: the node it stands on, as its `link` (see DevScene::wide), `gate` = the largest
// entry distance on the way down, and a stack of (link, gate) in LDS.  A step at an inner node
// reads one 32-byte row with the boxes of both children on a 16-bit grid, rounded outwards, and
// their links: half the bytes of exact boxes, and the kernel runs at what HBM delivers.  Grid
// boxes are entered no later than the exact ones, so they may be used to skip and to order
// subtrees (and for the stack-overflow rule, which errs towards a second pass).  The entry
// distance that the rule above needs is the exact one of the boxes above the surface; entry
// distances grow from a box to a box inside it, so that of the leaf's own exact box is the
// largest of them.  A leaf step therefore tests the exact box first (as the reference does) --
// computed from the vertices of a static triangle, read from the leaf's row of `nodes`
// otherwise -- and takes its entry distance as the gate.
#ifndef PT_STACK_DEPTH
#define PT_STACK_DEPTH 12
#endif
constexpr uint32_t kStackDepth = PT_STACK_DEPTH;
constexpr float kTieBand = 1.0e-4f;

#ifndef PT_RETRY_BUF
#define PT_RETRY_BUF 64
#endif
constexpr uint32_t kRetryBuf = PT_RETRY_BUF;

// Move a wave's buffered retry entries to its sub-list of the retry list: one atomic per call.
PT_DEV void flushRetries(const Pool& pool, uint32_t* counters, uint32_t waveIdx, const uint2* buf, uint32_t n)
{
    if (n == 0)
        return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the entries were written by other lanes of the wave
    const uint32_t list = waveIdx % kRetryShards;
    uint32_t base = 0;
    if (laneId() == 0)
        base = atomicAdd(&counters[(kCtrRetry + list) * kCtrStride], n);
    base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
    for (uint32_t i = laneId(); i < n; i += 64u)
        pool.retry[size_t(list) * pool.retryCap + base + i] = buf[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

#ifndef PT_ORDERED_WAVES
#define PT_ORDERED_WAVES 6
#endif
#ifndef PT_ORDERED_STEPS_PER_VOTE
#define PT_ORDERED_STEPS_PER_VOTE 2 // inner-node steps (each followed by the pops that became due) between two votes
#endif
#ifndef PT_ORDERED_BLOCK
#define PT_ORDERED_BLOCK 256
#endif
constexpr uint32_t kOrderedBlock = PT_ORDERED_BLOCK; // threads per block of traceOrderedKernel
#ifndef PT_STACK_DEPTH_QUAD
#define PT_STACK_DEPTH_QUAD 16
#endif
#ifndef PT_QUAD_GATE16
#define PT_QUAD_GATE16 1
#endif
#ifndef PT_QUAD_STEPS_PER_VOTE
#define PT_QUAD_STEPS_PER_VOTE 1 // (a four-wide step is twice as long: one per vote measured 1.5 % faster than two)
#endif

// one box of a row: three words on the grid (see DevScene::wide) against the ray
PT_DEV bool gridBoxEntry(uint32_t w0, uint32_t w1, uint32_t w2, V3 base, V3 step, const Ray& ray, float cull, float& entry)
{
    const V3 lo = v3(fmaf(float(w0 & 0xFFFFu), step.x, base.x), fmaf(float(w0 >> 16), step.y, base.y), fmaf(float(w1 & 0xFFFFu), step.z, base.z));
    const V3 hi = v3(fmaf(float(w1 >> 16), step.x, base.x), fmaf(float(w2 & 0xFFFFu), step.y, base.y), fmaf(float(w2 >> 16), step.z, base.z));
    return aabbEntry(lo, hi, ray, 0.0001f, cull, entry);
}

// put two (entry distance, link) pairs in order of distance
PT_DEV void nearerFirst(float& ea, uint32_t& la, float& eb, uint32_t& lb)
{
    const bool swap = eb < ea;
    const float e0 = swap ? eb : ea, e1 = swap ? ea : eb;
    const uint32_t l0 = swap ? lb : la, l1 = swap ? la : lb;
    ea = e0;
    eb = e1;
    la = l0;
    lb = l1;
}

// WIDTH 2: rows of DevScene::wide; WIDTH 4: rows of DevScene::quad (up to four boxes per step: half the steps, each
// one trip to memory, for the same number of box tests)
template <int WIDTH, bool DYN, bool STATS>
__global__ void __launch_bounds__(kOrderedBlock, PT_ORDERED_WAVES) traceOrderedKernel(DevScene sc, Pool pool, uint32_t* __restrict__ counters, int parity,
                                                          float* __restrict__ samples, uint32_t shadowCap, uint32_t slotWaves,
                                                          int taken, unsigned long long* __restrict__ stats)
{
    uint32_t nRays = 0, nShadowRays = 0, nRows = 0, nLeafRows = 0, nTests = 0, nAmbiguous = 0;
    constexpr uint32_t kDepth = WIDTH == 4 ? uint32_t(PT_STACK_DEPTH_QUAD) : kStackDepth;
    // A kept node is (link, gate).  The four-wide walk keeps up to three per step, so its stacks are deeper and an
    // entry is 6 bytes: the gate cut to its upper 16 bits, which rounds it down -- a kept node's gate only decides
    // whether the node is still visited when its turn comes (a lower gate visits it a little more often), and the
    // gate that the rule for ambiguous results uses is set anew from the exact box of every leaf.
    constexpr bool kGate16 = WIDTH == 4 && PT_QUAD_GATE16 != 0;
    __shared__ uint2 stack[kGate16 ? 1 : kDepth][kGate16 ? 1 : kOrderedBlock];
    __shared__ uint32_t stackLink[kGate16 ? kDepth : 1][kGate16 ? kOrderedBlock : 1];
    __shared__ uint16_t stackGate[kGate16 ? kDepth : 1][kGate16 ? kOrderedBlock : 1];
    auto keep = [&](uint32_t at, uint32_t l, float g) {
        if (kGate16) {
            stackLink[at][threadIdx.x] = l;
            stackGate[at][threadIdx.x] = uint16_t(fbits(g) >> 16);
        } else {
            stack[at][threadIdx.x] = make_uint2(l, fbits(g));
        }
    };
    auto kept = [&](uint32_t at) {
        if (kGate16)
            return make_uint2(stackLink[at][threadIdx.x], uint32_t(stackGate[at][threadIdx.x]) << 16);
        return stack[at][threadIdx.x];
    };
    __shared__ uint2 retryBuf[kOrderedBlock / 64u][kRetryBuf]; // per wave: ambiguous rays waiting to go to the retry list
    uint32_t nRetry = 0;                     // entries in this wave's part (the same in every lane)
    if (blockIdx.x == 0)
        for (uint32_t q = threadIdx.x; q < kShards; q += kOrderedBlock) {
            counters[(kCtrShadow + parity * kShards + q) * kCtrStride] = 0;
            counters[(kCtrHit + q) * kCtrStride] = 0; // the list of the previous iteration has been worked off
            // the generateKernel before this launch gave path numbers to the places of region q that the
            // continuing paths had left free
            counters[q * kCtrStride] = addSaturating(counters[q * kCtrStride], pool.nSlots / kShards - counters[(kCtrLive + q) * kCtrStride]);
            counters[(kCtrLive + q) * kCtrStride] = 0;
        }
    __shared__ uint32_t sFill[2 * kShards];
    fillsOfParts(counters, parity, sFill);
    __syncthreads();
    const float4* __restrict__ nodes = sc.nodes;
    const uint4* __restrict__ wide = WIDTH == 4 ? sc.quad : sc.wide;
    const float4* __restrict__ surf = sc.surf;
    const uint32_t waveIdx = uint32_t(__builtin_amdgcn_readfirstlane(int((blockIdx.x * kOrderedBlock + threadIdx.x) >> 6)));
    const uint32_t lane = laneId();
    const uint64_t below = (1ull << lane) - 1ull;
    const uint32_t regionCap = pool.nSlots / kShards;
    // 0: rays of paths (items are pool records), 1: shadow rays (items are queue entries); half of the waves begin
    // with either kind and go over to the other when theirs is worked off and their lanes are empty
    // taken runs: half of the waves begin with either kind and go over to the other when theirs is worked off and
    // their lanes are empty; fixed runs: the first slotWaves waves have the pool's regions, the others the queues
    RunTaker taker = { taken ? (waveIdx / kShards) & 1u : (waveIdx >= slotWaves ? 1u : 0u), waveIdx % kShards };
    bool kindDone = false, otherKindTried = false, allDone = false;
    uint32_t pos = 0, endPos = 0;       // the rest of the run in hand
    if (!taken) {
        kindDone = otherKindTried = allDone = true;
        if (!fixedRun(sFill, taker.kind, taker.kind ? waveIdx - slotWaves : waveIdx, regionCap, shadowCap, pos, endPos))
            return;
    }

    bool have = false;
    uint32_t item = 0;
    Ray ray = makeRay(v3(0.0f), v3(1.0f), 0.0f);
    // `link` is also the lane's state: kInnerBit set -- it stands on an inner node; kNoLink -- nothing left to visit
    // (or no ray); else a leaf whose test is due
    constexpr uint32_t kNoLink = 0x7FFFFFFFu; // (as a leaf link this would be node 2^30 - 1: the builders stop below)
    uint32_t link = kNoLink, top = 0;
    float gate = 0.0f, cull = kFltMax, otherT = kFltMax;
    float dropGate = __builtin_huge_valf(); // nearest entry distance of a subtree that found no room on the stack
    HitLite best = noHit();
    // shadow rays: the chosen light, and the distance below which any other surface settles the question
    uint32_t lightSid = PT_NO_HIT;
    float settledBelow = -__builtin_huge_valf();

    for (;;) {
        const uint64_t idle = __builtin_amdgcn_ballot_w64(!have);
        const uint32_t nIdle = uint32_t(__popcll(idle));
        if (pos >= endPos && !allDone && nIdle >= kRefillLanesOrdered) {
            // the run is worked off: the next one of this kind; the other kind once this one is through and the
            // lanes are empty (the two kinds end differently)
            if (!kindDone && !takeRun(counters, sFill, taker, regionCap, shadowCap, kTakenItemsPerLane, pos, endPos))
                kindDone = true;
            if (kindDone && nIdle == 64u) {
                if (!otherKindTried) {
                    otherKindTried = true;
                    taker.kind ^= 1u;
                    taker.part = waveIdx % kShards;
                    kindDone = !takeRun(counters, sFill, taker, regionCap, shadowCap, kTakenItemsPerLane, pos, endPos);
                }
                allDone = kindDone;
            }
        }
        const bool shadow = taker.kind != 0u;
        if (nIdle == 64u && pos >= endPos && allDone)
            break;
        if (pos < endPos && (nIdle >= kRefillLanesOrdered || nIdle == 64u)) {
            const uint32_t cand = pos + uint32_t(__popcll(idle & below));
            if (!have && cand < endPos) {
                item = cand;
                if (!shadow) {
                    const float4 d_s = pool.d_s[item];
                    if (stateOf(fbits(d_s.w)) == kActive) {
                        const float4 o_t = pool.o_t[item];
                        ray = makeRay(v3(o_t.x, o_t.y, o_t.z), v3(d_s.x, d_s.y, d_s.z), o_t.w);
                        have = true;
                    }
                } else {
                    // (what to add on arrival is read when the ray arrives: fewer registers held
                    // through the walk, more waves in flight)
                    const float4 so = pool.sh_o[item], sd = pool.sh_d[item];
                    ray = makeRay(v3(so.x, so.y, so.z), v3(sd.x, sd.y, sd.z), so.w);
                    have = true;
                }
                float cullStart = kFltMax;
#ifndef PT_NO_SHADOW_BOUND
                if (shadow) {
                    // All that is asked of this ray is whether its closest hit is the chosen light (reference
                    // src/15/pathtracer.cpp:97-100).  The light's own test, the one the walk makes when it reaches
                    // the light's leaf, says where: a miss means no walk can find it; a hit at aL means nothing
                    // beyond aL can be the closest hit, and any other surface found well before aL ends the walk.
                    lightSid = fbits(pool.sh_c[item].w) & kLightMask;
                    HitLite lh;
                    if (STATS)
                        nTests++;
                    if (surfaceTest<DYN>(sc, surf, lightSid, ray, 0.0001f, kFltMax, lh)) {
                        cullStart = lh.a * (1.0f + kTieBand);
                        if (!(cullStart < kFltMax))
                            cullStart = kFltMax;
                        settledBelow = lh.a * (1.0f - kTieBand);
                    } else {
                        have = false;
                        if (STATS)
                            nShadowRays++;
                    }
                }
#endif
                if (have) {
                    if (STATS) {
                        nRays += shadow ? 0u : 1u;
                        nShadowRays += shadow ? 1u : 0u;
                        nRows++; // the root's box
                    }
                    best = noHit();
                    cull = cullStart;
                    otherT = kFltMax;
                    dropGate = __builtin_huge_valf();
                    top = 0;
                    // the root
                    link = kNoLink;
                    if (sc.n_nodes > 0) {
                        const float4 n0 = nodes[0], n1 = nodes[1];
                        if (aabbEntry(v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), ray, 0.0001f, cull, gate))
                            link = (fbits(n1.w) & kInnerBit) ? kInnerBit : kLeafRowBit; // inner number 0, or node 0
                    }
                }
            }
            pos = min(pos + nIdle, endPos);
        }
        // inner nodes: test both children, go to the nearer one that is hit, keep the other;
        // repeated while enough lanes stand on inner nodes
        {
            uint64_t walkers = __builtin_amdgcn_ballot_w64(int32_t(link) < 0);
            while (walkers != 0) {
#pragma unroll
                for (int rep = 0; rep < (WIDTH == 4 ? PT_QUAD_STEPS_PER_VOTE : PT_ORDERED_STEPS_PER_VOTE); rep++) {
                if (WIDTH == 4 && int32_t(link) < 0) {
                    // test the boxes of the row, go to the nearest one that is hit, keep the others, the nearest on top
                    const uint32_t k = link & ~kInnerBit;
                    const uint4 q0 = wide[4 * size_t(k) + 0], q1 = wide[4 * size_t(k) + 1], q2 = wide[4 * size_t(k) + 2];
                    const uint4 q3 = wide[4 * size_t(k) + 3];
                    if (STATS)
                        nRows++;
                    const V3 base = v3(sc.qbase[0], sc.qbase[1], sc.qbase[2]), step = v3(sc.qstep[0], sc.qstep[1], sc.qstep[2]);
                    float e0, e1, e2, e3;
                    uint32_t l0 = q3.x, l1 = q3.y, l2 = q3.z, l3 = q3.w;
                    const bool h0 = gridBoxEntry(q0.x, q0.y, q0.z, base, step, ray, cull, e0) && l0 != kQuadNoLink;
                    const bool h1 = gridBoxEntry(q0.w, q1.x, q1.y, base, step, ray, cull, e1) && l1 != kQuadNoLink;
                    const bool h2 = gridBoxEntry(q1.z, q1.w, q2.x, base, step, ray, cull, e2) && l2 != kQuadNoLink;
                    const bool h3 = gridBoxEntry(q2.y, q2.z, q2.w, base, step, ray, cull, e3) && l3 != kQuadNoLink;
                    // a box that is not hit goes to the end
                    e0 = h0 ? e0 : __builtin_huge_valf();
                    e1 = h1 ? e1 : __builtin_huge_valf();
                    e2 = h2 ? e2 : __builtin_huge_valf();
                    e3 = h3 ? e3 : __builtin_huge_valf();
                    l0 = h0 ? l0 : kNoLink;
                    l1 = h1 ? l1 : kNoLink;
                    l2 = h2 ? l2 : kNoLink;
                    l3 = h3 ? l3 : kNoLink;
                    nearerFirst(e0, l0, e1, l1);
                    nearerFirst(e2, l2, e3, l3);
                    nearerFirst(e0, l0, e2, l2);
                    nearerFirst(e1, l1, e3, l3);
                    nearerFirst(e1, l1, e2, l2); // (without this last exchange: 4 % slower, the kept nodes come up out of order)
                    // when room runs out the farthest ones are left out (the result stands if they turn out to lie
                    // beyond the limit anyway)
                    const uint32_t room = kDepth - top;
                    if (l3 != kNoLink) {
                        const float g = fmaxf(gate, e3);
                        if (room >= 3u) {
                            keep(top, l3, g);
                            top++;
                        } else {
                            dropGate = fminf(dropGate, g);
                        }
                    }
                    if (l2 != kNoLink) {
                        const float g = fmaxf(gate, e2);
                        if (room >= 2u) {
                            keep(top, l2, g);
                            top++;
                        } else {
                            dropGate = fminf(dropGate, g);
                        }
                    }
                    if (l1 != kNoLink) {
                        const float g = fmaxf(gate, e1);
                        if (room >= 1u) {
                            keep(top, l1, g);
                            top++;
                        } else {
                            dropGate = fminf(dropGate, g);
                        }
                    }
                    if (l0 != kNoLink)
                        gate = fmaxf(gate, e0);
                    link = l0; // kNoLink: nothing below, take the next kept node
                }
                if (WIDTH == 2 && int32_t(link) < 0) {
                    const uint32_t k = link & ~kInnerBit;
                    const uint4 q0 = wide[2 * k + 0], q1 = wide[2 * k + 1];
                    if (STATS)
                        nRows++;
                    const V3 base = v3(sc.qbase[0], sc.qbase[1], sc.qbase[2]), step = v3(sc.qstep[0], sc.qstep[1], sc.qstep[2]);
                    const V3 lo1 = v3(fmaf(float(q0.x & 0xFFFFu), step.x, base.x), fmaf(float(q0.x >> 16), step.y, base.y),
                                      fmaf(float(q0.y & 0xFFFFu), step.z, base.z));
                    const V3 hi1 = v3(fmaf(float(q0.y >> 16), step.x, base.x), fmaf(float(q0.z & 0xFFFFu), step.y, base.y),
                                      fmaf(float(q0.z >> 16), step.z, base.z));
                    const V3 lo2 = v3(fmaf(float(q0.w & 0xFFFFu), step.x, base.x), fmaf(float(q0.w >> 16), step.y, base.y),
                                      fmaf(float(q1.x & 0xFFFFu), step.z, base.z));
                    const V3 hi2 = v3(fmaf(float(q1.x >> 16), step.x, base.x), fmaf(float(q1.y & 0xFFFFu), step.y, base.y),
                                      fmaf(float(q1.y >> 16), step.z, base.z));
                    float e1, e2;
                    const bool h1 = aabbEntry(lo1, hi1, ray, 0.0001f, cull, e1);
                    const bool h2 = aabbEntry(lo2, hi2, ray, 0.0001f, cull, e2);
                    const float g1 = fmaxf(gate, e1), g2 = fmaxf(gate, e2);
                    const bool firstNear = e1 <= e2;
                    if (h1 && h2) {
                        if (top < kDepth) {
                            keep(top, firstNear ? q1.w : q1.z, firstNear ? g2 : g1);
                            top++;
                        } else {
                            // no room: leave the farther subtree out; the result stands if it turns
                            // out to lie beyond the limit anyway
                            dropGate = fminf(dropGate, firstNear ? g2 : g1);
                        }
                    }
                    const bool take1 = h1 && (!h2 || firstNear);
                    if (h1 || h2) {
                        gate = take1 ? g1 : g2;
                        link = take1 ? q1.z : q1.w;
                    } else {
                        link = kNoLink; // nothing below: take the next kept node
                    }
                }
                // a lane with nothing below pops the nearest kept node still within the limit (a lane without a ray
                // has nothing kept)
                if (link == kNoLink) {
                    while (top > 0) {
                        top--;
                        const uint2 e = kept(top);
                        if (!(__uint_as_float(e.y) > cull)) {
                            link = e.x;
                            gate = __uint_as_float(e.y);
                            break;
                        }
                    }
                }
                }
                walkers = __builtin_amdgcn_ballot_w64(int32_t(link) < 0);
                if (__builtin_popcountll(walkers) < int(kWalkLanesOrdered))
                    break;
            }
        }
        // leaves: test the surface, then take the next kept node
        if (link < kNoLink) {
            HitLite h;
            bool found = false;
            if (link & kLeafRowBit) {
                const uint32_t k = link & ~kLeafRowBit;
                const float4 n0 = nodes[2 * k + 0], n1 = nodes[2 * k + 1];
                float e;
                if (STATS)
                    nLeafRows++;
                if (aabbEntry(v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), ray, 0.0001f, cull, e)) {
                    gate = e;
                    found = surfaceTest<DYN>(sc, surf, fbits(n1.w), ray, 0.0001f, cull, h);
                    if (STATS)
                        nTests++;
                }
            } else {
                // a static triangle: its box is that of its vertices (reference src/15/surface_triangle.hpp:44-45)
                const float4 r0 = surf[3 * link + 0], r1 = surf[3 * link + 1], r2 = surf[3 * link + 2];
                const V3 A = v3(r0.x, r0.y, r0.z), B = v3(r1.x, r1.y, r1.z), C = v3(r2.x, r2.y, r2.z);
                const V3 lo = v3(fminf(A.x, fminf(B.x, C.x)), fminf(A.y, fminf(B.y, C.y)), fminf(A.z, fminf(B.z, C.z)));
                const V3 hi = v3(fmaxf(A.x, fmaxf(B.x, C.x)), fmaxf(A.y, fmaxf(B.y, C.y)), fmaxf(A.z, fmaxf(B.z, C.z)));
                if (STATS)
                    nTests++; // (the surface row is fetched whether or not its box is hit)
                // the reference tests the box, then the surface (src/15/bvh.hpp:191-193): both must say yes, so the
                // surface may be asked first and the box only for the one test in six that hits
                float a, u, v, e;
                bool bs;
                if (triTest(A, B, C, ray, 0.0001f, cull, a, u, v, bs) && aabbEntry(lo, hi, ray, 0.0001f, cull, e)) {
                    gate = e;
                    h.sid = link;
                    h.matClass = fbits(r1.w) >> 24;
                    h.a = a;
                    h.u = u;
                    h.v = v;
                    h.backside = bs;
                    found = true;
                }
            }
            if (found) {
                if (best.sid == PT_NO_HIT || h.a < best.a) {
                    if (best.sid != PT_NO_HIT)
                        otherT = fminf(otherT, best.a);
                    best = h;
                    cull = fmaxf(best.a, gate) * (1.0f + kTieBand);
                    if (!(cull < kFltMax))
                        cull = kFltMax;
                } else {
                    otherT = fminf(otherT, h.a);
                }
#ifndef PT_NO_SHADOW_BOUND
                if (shadow && h.sid != lightSid && fmaxf(h.a, gate) < settledBelow) {
                    // another surface, reached through boxes entered before the light is hit, and hit before it:
                    // whatever the closest hit is, it is not the light
                    best = noHit();
                    otherT = kFltMax;
                    dropGate = __builtin_huge_valf();
                    top = 0;
                }
#endif
            }
            link = kNoLink;
            while (top > 0) {
                top--;
                const uint2 e = kept(top);
                if (!(__uint_as_float(e.y) > cull)) {
                    link = e.x;
                    gate = __uint_as_float(e.y);
                    break;
                }
            }
        }
        // rays that are through: ambiguous ones are collected per wave in LDS and go to the retry list
        // in bulk (whole wave takes part in the bookkeeping)
        const bool through = have && link == kNoLink;
        const bool ambiguous = through && ((best.sid != PT_NO_HIT && otherT <= cull) || dropGate <= cull);
        {
            const uint64_t am = __builtin_amdgcn_ballot_w64(ambiguous);
            if (am != 0) {
                if (nRetry + 64u > kRetryBuf) {
                    flushRetries(pool, counters, waveIdx, retryBuf[threadIdx.x >> 6], nRetry);
                    nRetry = 0;
                }
                if (ambiguous) {
                    // what was found bounds the search of the second pass (same band once more)
                    float bound = best.sid != PT_NO_HIT ? cull * (1.0f + kTieBand) : kFltMax;
                    if (!(bound < kFltMax))
                        bound = kFltMax;
                    // Usually the limit is the closest distance found times the band (the boxes above that surface
                    // are entered before it) and no subtree was left out: then the entry carries that distance, from
                    // which the second pass works out the same bound and also where its search may begin (see
                    // traceRetryKernel); otherwise it carries the bound, marked by the sign bit.
                    uint32_t word = fbits(bound) | kRetryFullWalk;
#if PT_RETRY_PRUNE
                    if (best.sid != PT_NO_HIT && !(dropGate <= cull) && cull == best.a * (1.0f + kTieBand) && cull < kFltMax)
                        word = fbits(best.a);
#endif
                    retryBuf[threadIdx.x >> 6][nRetry + uint32_t(__popcll(am & below))]
                        = make_uint2(shadow ? (kRetryIsShadow | (item)) : item, word);
                }
                nRetry += uint32_t(__popcll(am));
                if (STATS && ambiguous)
                    nAmbiguous++;
            }
        }
        if (through) {
            have = false;
            if (ambiguous) {
                // settled by traceRetryKernel
            } else if (!shadow) {
                pool.hit[item] = make_uint4(best.sid == PT_NO_HIT ? PT_NO_HIT : (best.sid | (best.backside ? 0x80000000u : 0u)),
                                            fbits(best.a), fbits(best.u), fbits(best.v));
                pool.cls[item] = best.sid == PT_NO_HIT ? uint8_t(kClsMiss) : uint8_t(kClsHit + best.matClass);
            } else if (best.sid != PT_NO_HIT) {
                const float4 payload = pool.sh_c[item];
                if (best.sid == (fbits(payload.w) & kLightMask)) {
                    const uint32_t dest = fbits(pool.sh_d[item].w);
                    const V3 destValue = loadSample(samples, dest);
                    shadowArrive<DYN>(sc, surf, pool, samples, item, ray, best, payload, dest, destValue);
                }
            }
        }
    }
    flushRetries(pool, counters, waveIdx, retryBuf[threadIdx.x >> 6], nRetry);
    addStats<STATS>(stats, kStatPathRays, nRays);
    addStats<STATS>(stats, kStatShadowRays, nShadowRays);
    addStats<STATS>(stats, kStatBoxRows, nRows);
    addStats<STATS>(stats, kStatLeafRows, nLeafRows);
    addStats<STATS>(stats, kStatSurfaceTests, nTests);
    addStats<STATS>(stats, kStatRetryRays, nAmbiguous);
}

// The rays traceOrderedKernel left over, each traced to completion in storage order (reference
// src/15/bvh.hpp:183-213) by one lane, starting from the distance bound the first pass found:
// boxes entered beyond it hold nothing that could be accepted after the surfaces within it, so
// the walk sees the same surfaces win as one that starts unbounded.  They are few (a percent
// of the rays or less), so the list fits the chip many times over.
template <bool DYN, bool STATS>
__global__ void __launch_bounds__(256) traceRetryKernel(DevScene sc, Pool pool, const uint32_t* __restrict__ counters,
                                                        float* __restrict__ samples, unsigned long long* __restrict__ stats)
{
    Counters cnt = { 0, 0, 0, 0, 0, 0 };
    // block b works on sub-list b % kRetryShards (the grid is a multiple of kRetryShards blocks)
    const uint32_t list = blockIdx.x % kRetryShards;
    const uint32_t n = counters[(kCtrRetry + list) * kCtrStride];
    for (uint32_t i = (blockIdx.x / kRetryShards) * 256u + threadIdx.x; i < n; i += (gridDim.x / kRetryShards) * 256u) {
        const uint2 entry = pool.retry[size_t(list) * pool.retryCap + i];
        const uint32_t e = entry.x;
        // The first pass visited every subtree entered within its limit and found nothing closer than its closest
        // distance a: a box that the ray leaves before a * (1 - 2 bands) holds no surface that is hit (a surface's
        // distance and the exit distance of a box around it differ by rounding, which the band covers as it does on
        // the entry side), so the walk may skip it -- most of the boxes between the ray's origin and the hit.
        float bound = __uint_as_float(entry.y & ~kRetryFullWalk), boxAmin = 0.0001f;
        if (!(entry.y & kRetryFullWalk)) {
            const float a = bound;
            bound = a * (1.0f + kTieBand) * (1.0f + kTieBand);
            if (!(bound < kFltMax))
                bound = kFltMax;
            boxAmin = fmaxf(a * (1.0f - 2.0f * kTieBand), 0.0001f);
        }
        if (!(e & kRetryIsShadow)) {
            const float4 o_t = pool.o_t[e], d_s = pool.d_s[e];
            const Ray ray = makeRay(v3(o_t.x, o_t.y, o_t.z), v3(d_s.x, d_s.y, d_s.z), o_t.w);
            const HitLite best = closestHitFrom<STATS, DYN>(sc, ray, 0.0001f, boxAmin, bound, &cnt);
            pool.hit[e] = make_uint4(best.sid == PT_NO_HIT ? PT_NO_HIT : (best.sid | (best.backside ? 0x80000000u : 0u)),
                                     fbits(best.a), fbits(best.u), fbits(best.v));
            pool.cls[e] = best.sid == PT_NO_HIT ? uint8_t(kClsMiss) : uint8_t(kClsHit + best.matClass);
        } else {
            const uint32_t q = e & ~kRetryIsShadow;
            const float4 so = pool.sh_o[q], sd = pool.sh_d[q], payload = pool.sh_c[q];
            const uint32_t dest = fbits(sd.w);
            const V3 destValue = loadSample(samples, dest);
            const Ray ray = makeRay(v3(so.x, so.y, so.z), v3(sd.x, sd.y, sd.z), so.w);
            const HitLite best = closestHitFrom<STATS, DYN>(sc, ray, 0.0001f, boxAmin, bound, &cnt);
            shadowArrive<DYN>(sc, sc.surf, pool, samples, q, ray, best, payload, dest, destValue);
        }
    }
    addStats<STATS>(stats, kStatRetryNodes, cnt.nodeVisits);
    addStats<STATS>(stats, kStatRetrySurfaceTests, cnt.leafTests);
}

#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 5
#endif
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 256
#endif
constexpr uint32_t kShadeBlock = PT_SHADE_BLOCK; // threads per block of shadeHitsKernel
// ---- the shading step over compact lists, from one pool of path records into the other ----
// Two pools of records take turns.  After the tracing of pool A:
//   classifyKernel  one lane per record of A: records whose ray hit a surface go to the hit list of
//                   their shard; a record whose ray left the scene is finished on the spot
//                   (environment, sample);
//   shadeHitsKernel one lane per hit-list entry: the turn of the pathSample loop for a hit; a path
//                   that goes on is written to the next free place of its shard's region of pool
//                   B (dense, whole cache lines), a path that ends writes its sample;
//   generateKernel  fills the rest of every region of B with new paths (the free places of a region
//                   are one run, so path numbers need no reservation; camera ray), or marks it
//                   empty when no paths are left.
// Then B is traced.  Every wave of the heavy kernels is full of lanes doing the same kind of work,
// records that need no shading are not loaded by the shading kernel, and all state stores are
// dense.  New paths lie together in path-number order (neighbouring pixels), which also keeps the
// camera rays of a wave and their first hits coherent.
// A block of 1024 threads takes kClassifyChunks runs of 1024 records, so the hits of 8192 records are ordered
// together and one reservation serves them all.  What gains is the shading kernel (fewer waves with mixed
// materials: Cornell 647 -> 610 us per launch); this kernel's own time hardly depends on the number of runs
// (about half of it is the work for the misses).  8 runs measured best overall, 2/4/16/32 slower.
constexpr uint32_t kClassifyBlock = 1024;
#ifndef PT_CLASSIFY_CHUNKS
#define PT_CLASSIFY_CHUNKS 8
#endif
constexpr uint32_t kClassifyChunks = PT_CLASSIFY_CHUNKS;
constexpr uint32_t kClassifyRecords = kClassifyBlock * kClassifyChunks; // records per block
// The grid has blocksPerRegion blocks for the front of each region: all of a region (regionCap / kClassifyRecords), or
// fewer once the host knows that no region has more than so many records in use (see runBatch).
__global__ void __launch_bounds__(kClassifyBlock, 8) classifyKernel(DevScene sc, Pool pool, uint32_t* __restrict__ counters,
                                                                 float* __restrict__ samples, FastDiv blocksPerRegion)
{
    // hits go to the list ordered by the type of their material within the block's records, so that the
    // 64 entries a wave of shadeHitsKernel takes mostly run the same material code
    constexpr uint32_t kWaves = kClassifyBlock / 64;
    __shared__ uint32_t count[kHitClasses][kWaves]; // hits of class c in wave w; then their first place in the block's run
    __shared__ uint32_t blockBase, present;
    if (blockIdx.x == 0 && threadIdx.x < kRetryShards)
        counters[(kCtrRetry + threadIdx.x) * kCtrStride] = 0; // the retry lists of this iteration have been worked off
    if (blockIdx.x == 0 && threadIdx.x < 2u * kShards)
        counters[(kCtrWork + threadIdx.x) * kCtrStride] = 0; // the next tracing launch hands out its runs from the start
    // only the records at the front of a region are in use (a region is a whole number of blocks); what lies behind
    // them is whatever an earlier iteration left there
    const uint32_t regionCap = pool.nSlots / kShards, region = blocksPerRegion.div(blockIdx.x);
    const uint32_t inRegion0 = (blockIdx.x - region * blocksPerRegion.d) * kClassifyRecords + threadIdx.x;
    const uint32_t slot0 = region * regionCap + inRegion0;
    const uint32_t inUse = counters[(kCtrOcc + region) * kCtrStride];
    if (inRegion0 - threadIdx.x >= inUse)
        return;
    uint8_t cls[kClassifyChunks];
#pragma unroll
    for (uint32_t j = 0; j < kClassifyChunks; j++)
        cls[j] = inRegion0 + j * kClassifyBlock < inUse ? pool.cls[slot0 + j * kClassifyBlock] : uint8_t(kClsEmpty);
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t wave = threadIdx.x >> 6, lane = laneId();
    const uint64_t below = (1ull << lane) - 1ull;
    // the classes that occur among the block's records (a scene has a few material types, not sixteen): the ranking
    // below runs once per class that is there
    if (threadIdx.x < kHitClasses * kWaves)
        (&count[0][0])[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        present = 0;
    __syncthreads();
    {
        uint32_t bits = 0;
#pragma unroll
        for (uint32_t j = 0; j < kClassifyChunks; j++)
            if (cls[j] >= kClsHit)
                bits |= 1u << min(uint32_t(cls[j] - kClsHit), kHitClasses - 1u);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            bits |= __shfl_xor(bits, off, 64);
        if (lane == 0 && bits)
            atomicOr(&present, bits);
    }
    __syncthreads();
    uint32_t rank[kClassifyChunks]; // among the wave's hits of the same class: run by run, lane by lane
#pragma unroll
    for (uint32_t j = 0; j < kClassifyChunks; j++)
        rank[j] = 0;
    for (uint32_t left = present; left != 0; left &= left - 1u) {
        const uint32_t c = uint32_t(__ffs(int(left))) - 1u;
        uint32_t sofar = 0;
#pragma unroll
        for (uint32_t j = 0; j < kClassifyChunks; j++) {
            const bool mine = cls[j] >= kClsHit && min(uint32_t(cls[j] - kClsHit), kHitClasses - 1u) == c;
            const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
            if (mine)
                rank[j] = sofar + uint32_t(__popcll(m & below));
            sofar += uint32_t(__popcll(m));
        }
        if (lane == 0)
            count[c][wave] = sofar;
    }
    __syncthreads();
    if (wave == 0) {
        // class-major running sum over the kHitClasses * kWaves = 128 counts by one wave (two per lane), then one
        // reservation for the block
        static_assert(kHitClasses * kWaves == 128, "two counts per lane");
        uint32_t* flat = &count[0][0];
        const uint32_t v0 = flat[lane], v1 = flat[64 + lane];
        uint32_t s0 = v0, s1 = v1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a = __shfl_up(s0, off, 64), b = __shfl_up(s1, off, 64);
            if (int(lane) >= off) {
                s0 += a;
                s1 += b;
            }
        }
        const uint32_t total0 = __shfl(s0, 63, 64), total = total0 + __shfl(s1, 63, 64);
        flat[lane] = s0 - v0;
        flat[64 + lane] = total0 + s1 - v1;
        if (lane == 0)
            blockBase = total ? atomicAdd(&counters[(kCtrHit + shard) * kCtrStride], total) : 0u;
    }
    __syncthreads();
    uint64_t misses = 0; // bit 8 * j: run j's record is a miss
#pragma unroll
    for (uint32_t j = 0; j < kClassifyChunks; j++) {
        if (cls[j] >= kClsHit) {
            const uint32_t cl = min(uint32_t(cls[j] - kClsHit), kHitClasses - 1u);
            pool.hitList[size_t(shard) * pool.listCap + blockBase + count[cl][wave] + rank[j]] = slot0 + j * kClassifyBlock;
        }
        misses |= uint64_t(cls[j] == kClsMiss) << (8 * j);
    }
    static_assert(kClassifyChunks <= 8, "one byte per run in `misses`");
    // the rays that left the scene (reference src/15/pathtracer.cpp:60-62): add the environment, done
    if (sc.envmap.type == PT_ENV_NONE)
        return; // (nothing to add: the path's sample holds its radiance already)
    // The environment's value is a few hundred instructions (two inverse trigonometric functions in double
    // precision, a texture lookup), and a quarter of the records or fewer are misses: each wave first collects its
    // misses in LDS and then works them off with all lanes.
    __shared__ uint32_t missSlots[kWaves][kClassifyChunks * 64u];
    uint32_t nMisses = 0; // of the wave (the same in every lane)
#pragma unroll
    for (uint32_t j = 0; j < kClassifyChunks; j++) {
        const bool mine = (misses >> (8 * j)) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
        if (mine)
            missSlots[wave][nMisses + uint32_t(__popcll(m & below))] = slot0 + j * kClassifyBlock;
        nMisses += uint32_t(__popcll(m));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the entries are read by other lanes of the wave
#pragma unroll 1
    for (uint32_t i = lane; i < nMisses; i += 64u) {
        const uint32_t slot = missSlots[wave][i];
        const float4 thr_p = pool.thr_p[slot], d_s = pool.d_s[slot];
        storeSample(samples, fbits(thr_p.w),
                    loadSample(samples, fbits(thr_p.w)) + v3(thr_p.x, thr_p.y, thr_p.z) * envmapValue(sc, v3(d_s.x, d_s.y, d_s.z)));
    }
}

// The material-side tables of a scene -- materials, their constant textures, the texture table, the meshes -- that
// the lanes of the shading kernel index each with a number of their own: copies in LDS, made once per block, where
// they fit (a few dozen materials: a few KB).  Every such look-up from memory is a gather that the address unit of a
// compute unit works off a cache line per clock, and a material's record alone is several of them; from LDS they
// cost a few clocks per wave.  Measured per launch of 32 M records: Cornell box 1200 -> 1142 us, the textured scene
// 2030 -> 1884, the mesh 1032 -> 988, the animation 1612 -> 1598.  (The per-surface tables as well -- attributes,
// plane normals, surface rows, spheres, animations, the list of lights -- made the kernel slower where they fit:
// Cornell box 1203 -> 1321 us at 107 registers, the animation unchanged; those gathers fall on a handful of cache
// lines and were cheap already, the copy per block and the registers cost more.  Tables that the whole wave reads
// at one index, the loop over the lights, stay in memory: those are scalar loads.)
struct ShadeTableCounts {
    uint32_t materials, textures, meshes;
};

constexpr uint32_t padWords(uint32_t bytes) { return ((bytes / 4u) + 3u) & ~3u; } // in 4-byte words, whole 16-byte rows

__host__ __device__ inline uint32_t shadeTableWords(const ShadeTableCounts& n)
{
    return padWords(n.materials * uint32_t(sizeof(pt_material))) + padWords(n.materials * 32u)
        + padWords(n.textures * uint32_t(sizeof(pt_texture))) + padWords(n.meshes * uint32_t(sizeof(DevMesh)));
}

PT_DEV void shadeTablesToLds(const DevScene& sc, const ShadeTableCounts& n, uint32_t* lds, DevScene& ls)
{
    uint32_t off = 0;
    auto take = [&](const void* src, uint32_t bytes) {
        uint32_t* dst = lds + off;
        const uint32_t* from = static_cast<const uint32_t*>(src);
        for (uint32_t i = threadIdx.x; i < bytes / 4u; i += kShadeBlock)
            dst[i] = from[i];
        off += padWords(bytes);
        return dst;
    };
    ls.materials = reinterpret_cast<const pt_material*>(take(sc.materials, n.materials * uint32_t(sizeof(pt_material))));
    ls.matConst = reinterpret_cast<const float4*>(take(sc.matConst, n.materials * 32u));
    ls.textures = reinterpret_cast<const pt_texture*>(take(sc.textures, n.textures * uint32_t(sizeof(pt_texture))));
    ls.meshes = reinterpret_cast<const DevMesh*>(take(sc.meshes, n.meshes * uint32_t(sizeof(DevMesh))));
}

// The grid has blocksPerShard blocks for the front of each shard's list: the whole list (listCap / kShadeBlock), or
// fewer once the host knows that no list can be longer (see runBatch).
template <bool MAT>
__global__ void __launch_bounds__(kShadeBlock, PT_SHADE_WAVES) shadeHitsKernel(DevScene sc, Pool pool, Pool next, PathParams pp,
                                                                       uint32_t* __restrict__ counters, int parity,
                                                                       float* __restrict__ samples, uint32_t shadowCap,
                                                                       FastDiv blocksPerShard,
                                                                       unsigned long long* __restrict__ stats,
                                                                       ShadeTableCounts tableCounts)
{
    __shared__ uint32_t lds[2 * (kShadeBlock / 64 + 1)];
    extern __shared__ uint4 ldsTableRows[]; // (rows of 16 bytes: the tables are read with 16-byte LDS loads)
    uint32_t* ldsTables = reinterpret_cast<uint32_t*>(ldsTableRows);
    // the block's entries lie inside one shard's list
    const uint32_t shard = blocksPerShard.div(blockIdx.x);
    const uint32_t first = (blockIdx.x - shard * blocksPerShard.d) * kShadeBlock; // within the list
    const uint32_t fill = counters[(kCtrHit + shard) * kCtrStride];
    if (first >= fill)
        return; // the whole block lies beyond the list
    const uint32_t entry = first + threadIdx.x;
    const bool valid = entry < fill;
    const uint32_t slot = valid ? pool.hitList[size_t(shard) * pool.listCap + entry] : 0u;
    // `ls`: the scene with the material-side tables in LDS; `sc`: everything in memory
    DevScene ls = sc;
    if (MAT) {
        shadeTablesToLds(sc, tableCounts, ldsTables, ls);
        __syncthreads();
    }
    bool finished = false;
    bool haveShadow = false, nullShadow = false;
    V3 shOrigin, shDir, shC, shBack = v3(0.0f);
    float shTime = 0.0f;
    uint32_t shLight = 0;
    uint32_t pathLocal = 0;
    float4 nO = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nD = nO, nT = nO;
    uint2 kw = make_uint2(0u, 0u);

    if (valid) {
        const float4 d_s = pool.d_s[slot];
        const float4 o_t = pool.o_t[slot];
        const float4 thr_p = pool.thr_p[slot];
        const uint4 hw = pool.hit[slot];
        kw = pool.key[slot];
        const int segment = int(segmentOf(fbits(d_s.w)));
        Ray ray = makeRay(v3(o_t.x, o_t.y, o_t.z), v3(d_s.x, d_s.y, d_s.z), o_t.w);
        V3 throughput = v3(thr_p.x, thr_p.y, thr_p.z);
        pathLocal = fbits(thr_p.w);
        Rng<false> rng;
        rng.key = uint64_t(kw.x) | (uint64_t(kw.y) << 32);
        rng.n = drawsOf(fbits(d_s.w));
        rng.cnt = nullptr;

        // one turn of the loop of pathSample for a hit, reference src/15/pathtracer.cpp:63-127
        HitLite hl;
        hl.sid = hw.x & 0x7FFFFFFFu;
        hl.backside = (hw.x >> 31) != 0;
        hl.a = __uint_as_float(hw.y);
        hl.u = __uint_as_float(hw.z);
        hl.v = __uint_as_float(hw.w);
        Hit hr = makeHit(ls, ray, hl);
        MaterialValues mv;
        Scatter sr = materialScatter<false>(ls, hr.material, ray, hr, rng, &mv);
        {
            // radiance += throughput * Le, on the path's sample: nothing to do for a surface that does not emit --
            // unless the throughput is no longer finite, when 0 * throughput is NaN and the reference's sum becomes NaN
            const V3 le = materialLe(ls, hr.material, hr.backside);
            const bool emits = le.x != 0.0f || le.y != 0.0f || le.z != 0.0f;
            const bool finite = fabs_(throughput.x) < __builtin_huge_valf() && fabs_(throughput.y) < __builtin_huge_valf()
                && fabs_(throughput.z) < __builtin_huge_valf();
            if (emits || !finite) {
                storeSample(samples, pathLocal, loadSample(samples, pathLocal) + throughput * le);
            }
        }
        if (sr.type == ScatterNone) {
            finished = true;
        } else {
            V3 nextThroughput = throughput * sr.attenuation / sr.p;
            if (sr.type == ScatterRandom && sc.n_lights > 0) {
                uint32_t lightIndex = uint32_t(rng.in01() * sc.n_lights);
                if (lightIndex >= sc.n_lights)
                    lightIndex = sc.n_lights - 1;
                const uint32_t lightSurf = ls.lights[lightIndex];
                V3 lightDir = surfaceDirection<false>(ls, lightSurf, hr.position, ray.time, rng);
                // (the two sums over the lights draw no random numbers, so they can be taken together)
                float lightsP, lightDirP;
                uint32_t lightSide;
                lightsPdf2(sc, hr.position, sr.direction, lightDir, ray.time, lightIndex, lightsP, lightDirP, lightSide);
                nextThroughput = nextThroughput * powerHeuristicMIS(sr.p, lightsP);
                if (lightDirP > 0.0f) {
                    // (a two-sided material scattered as one of its sides and answers scatterToDirection itself)
                    Scatter lightSR = materialScatterToDirection(ls, hr.material, ray, hr, lightDir,
                                                                 ls.materials[hr.material].type == PT_MAT_TWOSIDED ? nullptr : &mv);
                    if (lightSR.p > 0.0f) {
                        float weight = powerHeuristicMIS(lightDirP, lightSR.p);
                        // everything of the contribution but the light's emission, which the
                        // shadow ray's hit decides
                        shC = throughput * lightSR.attenuation / lightDirP * weight;
                        shOrigin = hr.position;
                        shDir = lightDir;
                        shTime = ray.time;
                        shLight = lightSurf;
                        haveShadow = true;
                        // emission of the chosen light, looked up here so that the tracing kernel
                        // needs no material tables when the shadow ray arrives: the ray carries
                        // contribution * emission for the light's front and, if that side emits
                        // too, for its back
                        int lmat;
                        if (fbits(ls.surf[3 * size_t(lightSurf) + 0].w) <= kTriDynamic)
                            lmat = int(ls.surfAttr[lightSurf].w & 0xFFFFFFu);
                        else
                            lmat = ls.spheres[ls.surfAttr[lightSurf].x].material;
                        // (a black back side still matters if the product is NaN: NaN != 0)
                        shBack = shC * materialLe(ls, lmat, true);
                        if (shBack.x != 0.0f || shBack.y != 0.0f || shBack.z != 0.0f)
                            shLight |= kBackEmits;
                        shC = shC * materialLe(ls, lmat, false);
                        // A shadow ray adds its contribution if its closest hit is the chosen light, by the side it
                        // arrives at.  The test it would make at that light's leaf has just been made for the light's
                        // pdf (same ray, same vertices, a wider interval): if that missed, or found the side whose
                        // contribution is zero, the ray can add nothing but zeros and is not traced.
                        // (PT_TRACE_NULL_SHADOW_RAYS traces them all the same: the tracing kernels then do exactly
                        // the reference's work.)
                        const bool frontNull = shC.x == 0.0f && shC.y == 0.0f && shC.z == 0.0f;
                        const bool backNull = !(shLight & kBackEmits);
                        nullShadow = !pp.traceNullShadowRays
                            && (lightSide == kSideMiss || (lightSide == kSideFront && frontNull) || (lightSide == kSideBack && backNull));
                        haveShadow = !nullShadow;
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
                if (rng.in01() < q) {
                    finished = true;
                } else {
                    float rrWeight = 1.0f / (1.0f - q);
                    throughput = throughput * rrWeight;
                }
            }
            if (segment + 1 >= pp.maxSegments)
                finished = true;
        }
        if (!finished) {
            nO = make_float4(ray.origin.x, ray.origin.y, ray.origin.z, ray.time);
            nD = make_float4(ray.direction.x, ray.direction.y, ray.direction.z,
                             __uint_as_float(stateWord(kActive, uint32_t(segment + 1), rng.n)));
            nT = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(pathLocal));
        }
    }

    if (stats) { // (instrumented renders only; the same for the whole launch)
        const uint32_t n = uint32_t(__popcll(__builtin_amdgcn_ballot_w64(nullShadow)));
        if (laneId() == 0 && n)
            atomicAdd(&stats[kStatNullShadowRays], (unsigned long long)n);
    }
    // two reservations with one atomic per block each: a place in the shard's shadow-ray queue,
    // and the next free place in the shard's region of the other pool for every path that goes on
    const bool goesOn = valid && !finished;
    uint32_t pos, posN;
    blockReserve2<kShadeBlock / 64>(&counters[(kCtrShadow + parity * kShards + shard) * kCtrStride], haveShadow,
                     &counters[(kCtrLive + shard) * kCtrStride], goesOn, lds, pos, posN);
    const uint32_t place = shard * (next.nSlots / kShards) + posN;
    if (goesOn) {
        next.o_t[place] = nO;
        next.d_s[place] = nD;
        next.thr_p[place] = nT;
        next.key[place] = kw;
    }
    if (haveShadow) {
        const uint32_t e = shard * shadowCap + pos;
        pool.sh_o[e] = make_float4(shOrigin.x, shOrigin.y, shOrigin.z, shTime);
        pool.sh_d[e] = make_float4(shDir.x, shDir.y, shDir.z, __uint_as_float(pathLocal));
        pool.sh_c[e] = make_float4(shC.x, shC.y, shC.z, __uint_as_float(shLight));
        if (shLight & kBackEmits)
            pool.sh_b[e] = make_float4(shBack.x, shBack.y, shBack.z, 0.0f);
    }
}

// The last paths of a batch, one lane each, to their end.  Once every path of a batch has been started the pool only
// drains, for as many iterations as the longest path has segments left, and an iteration costs four launches and a
// wave's walk however few paths it holds.  When few are left (see runBatch) the pool is traced once more -- which also
// delivers the shadow rays queued last -- and this kernel takes every record in use with its hit and runs the rest of
// pathSample for it (reference src/15/pathtracer.cpp:50-130; pathSampleFrom: the very functions of the other kernels,
// the storage-order walk and the shadow rays in line), then writes the sample.
__global__ void __launch_bounds__(64) finishPathsKernel(DevScene sc, Pool pool, PathParams pp, const uint32_t* __restrict__ counters,
                                                      float* __restrict__ samples, uint32_t blocksPerRegion)
{
    const uint32_t region = blockIdx.x / blocksPerRegion;
    const uint32_t inRegion = (blockIdx.x - region * blocksPerRegion) * 64u + threadIdx.x;
    if (inRegion >= counters[(kCtrOcc + region) * kCtrStride])
        return;
    const uint32_t slot = region * (pool.nSlots / kShards) + inRegion;
    const float4 d_s = pool.d_s[slot];
    if (stateOf(fbits(d_s.w)) != kActive)
        return;
    const float4 o_t = pool.o_t[slot];
    const float4 thr_p = pool.thr_p[slot];
    const uint4 hw = pool.hit[slot];
    const uint2 kw = pool.key[slot];
    Rng<false> rng;
    rng.key = uint64_t(kw.x) | (uint64_t(kw.y) << 32);
    rng.n = drawsOf(fbits(d_s.w));
    rng.cnt = nullptr;
    HitLite hl = noHit();
    if (hw.x != PT_NO_HIT) {
        hl.sid = hw.x & 0x7FFFFFFFu;
        hl.backside = (hw.x >> 31) != 0;
        hl.a = __uint_as_float(hw.y);
        hl.u = __uint_as_float(hw.z);
        hl.v = __uint_as_float(hw.w);
    }
    const Ray ray = makeRay(v3(o_t.x, o_t.y, o_t.z), v3(d_s.x, d_s.y, d_s.z), o_t.w);
    const V3 radiance = pathSampleFrom<false>(sc, pp, ray, rng, nullptr, int(segmentOf(fbits(d_s.w))),
                                              loadSample(samples, fbits(thr_p.w)) /* the path's radiance so far */,
                                              v3(thr_p.x, thr_p.y, thr_p.z), true, hl);
    storeSample(samples, fbits(thr_p.w), radiance);
}

// `next` = the pool being filled: the places of every region behind the paths that go on
#ifndef PT_GENERATE_ROOT_TEST
#define PT_GENERATE_ROOT_TEST 1
#endif
// The grid has blocksPerRegion blocks for the front of each region: all of a region (regionCap / 256), or one once the
// host knows that every path has been started (the launch then only notes the records in use).
__global__ void __launch_bounds__(256) generateKernel(DevScene sc, Pool next, RenderArgs a, Batch b, uint32_t* __restrict__ counters,
                                                      float* __restrict__ samples, FastDiv blocksPerRegion)
{
    const uint32_t regionCap = next.nSlots / kShards; // a multiple of 256
    const uint32_t shard = blocksPerRegion.div(blockIdx.x);
    const uint32_t first = shard * regionCap + (blockIdx.x - shard * blocksPerRegion.d) * 256u;
    const uint32_t goingOn = counters[(kCtrLive + shard) * kCtrStride];
    const uint32_t inRegion = first - shard * regionCap + threadIdx.x;
    // records in use at the front of the region once this launch is through: the paths that go on and the new ones
    // (shard q has the groups g = q, q + S, ... of 64 path numbers; see below).  The tracing and classifying kernels
    // look at those only, so the places behind them are left as they are.
    const uint32_t groups = b.totalIds >> 6;
    const uint32_t mine = groups > shard ? (groups - shard + kShards - 1u) / kShards * 64u : 0u;
    const uint32_t handedOut = counters[shard * kCtrStride];
    const uint32_t left = mine > handedOut ? mine - handedOut : 0u;
    const uint32_t inUse = goingOn + min(regionCap - goingOn, left);
    if (inRegion == 0)
        counters[(kCtrOcc + shard) * kCtrStride] = inUse;
    if (inRegion >= inUse)
        return;
    if (inRegion < goingOn)
        return; // a path that goes on
    // the free places of a region are one run, so their path numbers need no reservation; the next
    // tracing launch moves the shard's counter on
    const uint32_t place = first + threadIdx.x;
    const uint32_t local = counters[shard * kCtrStride] + (inRegion - goingOn);
    // shard q owns the groups of 64 path numbers g with g % kShards == q
    const uint32_t group = (local >> 6) * kShards + shard, inTile = local & 63u;
    // path number -> (8x8 tile, sample, pixel in tile): 64 consecutive numbers are one tile at one sample
    const uint32_t tile = b.byNS.div(group), sLocal = group - tile * b.nS;
    const uint32_t tileRow = b.byTilesX.div(tile);
    const int lx = int((tile - tileRow * b.tilesX) * 8u + (inTile & 7u));
    const int ly = int(tileRow * 8u + (inTile >> 3));
    if (group >= b.totalIds >> 6 || lx >= a.tw || ly >= a.th) {
        // no paths left, or padding of an edge tile: the place stays empty in this iteration
        next.d_s[place] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(stateWord(kEmpty, 0u, 0u)));
        next.cls[place] = kClsEmpty;
        return;
    }
    const int x = a.x0 + lx, y = a.y0 + ly;
    const int smp = int(b.sFirst + sLocal);
    Rng<false> rng;
    rng.key = pt_rng_key(a.p.seed, uint64_t(y) * a.p.width + x, uint64_t(smp));
    rng.n = 0;
    rng.cnt = nullptr;
    const uint32_t stratumRow = b.bySqrtSpp.div(uint32_t(smp));
    Ray ray = sampleRayOfStratum<false>(a.cam, a.p, x, y, int(stratumRow), smp - int(stratumRow) * a.p.sqrt_spp, rng);
#if PT_GENERATE_ROOT_TEST
    // A camera ray that misses the box of the whole tree ends here as the tracing and classifying kernels would
    // end it (the walk's first step is this very test, reference src/15/bvh.hpp:183-213, and a miss adds the
    // environment to radiance 0 with throughput 1, src/15/pathtracer.cpp:60-62): its place stays empty in this
    // iteration instead of carrying a path through two more kernels.
    {
        bool rootHit = false;
        if (sc.n_nodes > 0) {
            const float4 n0 = sc.nodes[0], n1 = sc.nodes[1];
            rootHit = aabbHit(v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), ray, 0.0001f, kFltMax);
        }
        if (!rootHit) {
            V3 radiance = v3(0.0f);
            if (sc.envmap.type != PT_ENV_NONE)
                radiance = radiance + v3(1.0f) * envmapValue(sc, ray.direction);
            storeSample(samples, group * 64u + inTile, radiance);
            next.d_s[place] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(stateWord(kEmpty, 0u, 0u)));
            next.cls[place] = kClsEmpty;
            return;
        }
    }
#endif
    next.o_t[place] = make_float4(ray.origin.x, ray.origin.y, ray.origin.z, ray.time);
    // A path's place in the sample buffer is its number: the paths under way at any time are a window of
    // consecutive numbers, so what the kernels add to samples stays within a few hundred megabytes of the buffer
    // (tile by tile, and within a tile sample by sample: [tile][sample][pixel of the tile]).
    const uint32_t pathLocal = group * 64u + inTile;
    next.d_s[place] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, __uint_as_float(stateWord(kActive, 0u, rng.n)));
    next.thr_p[place] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(pathLocal));
    storeSample(samples, pathLocal, v3(0.0f)); // radiance so far: none
    next.key[place] = make_uint2(uint32_t(rng.key), uint32_t(rng.key >> 32));
}

// Add the samples of every pixel in sample order (reference src/15/pathtracer.cpp:207-220).  The sample buffer is
// ordered [tile][sample][pixel of the 8x8 tile] (see generateKernel): a wave takes a tile.
__global__ void __launch_bounds__(256) resolveKernel(const float* __restrict__ samples, float* __restrict__ accum,
                                                     float* __restrict__ out, RenderArgs a, Batch b, int firstBatch,
                                                     int lastBatch)
{
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6), inTile = threadIdx.x & 63u;
    if (tile >= b.tilesX * b.tilesY)
        return;
    const uint32_t tileRow = b.byTilesX.div(tile);
    const int lx = int((tile - tileRow * b.tilesX) * 8u + (inTile & 7u)), ly = int(tileRow * 8u + (inTile >> 3));
    if (lx >= a.tw || ly >= a.th)
        return;
    const uint32_t pix = uint32_t(ly * a.tw + lx);
    V3 acc = v3(0.0f);
    if (!firstBatch)
        acc = v3(accum[3 * size_t(pix) + 0], accum[3 * size_t(pix) + 1], accum[3 * size_t(pix) + 2]);
    for (uint32_t s = 0; s < b.nS; s++)
        acc = acc + loadSample(samples, (size_t(tile) * b.nS + s) * 64u + inTile);
    if (lastBatch) {
        if (!(a.p.flags & PT_RENDER_SUM_ONLY))
            acc = acc / float(a.p.sqrt_spp * a.p.sqrt_spp);
        const int x = a.x0 + lx, y = a.y0 + ly;
        float* o = out + 3 * (size_t(y) * a.p.width + x);
        o[0] = acc.x;
        o[1] = acc.y;
        o[2] = acc.z;
    } else {
        accum[3 * size_t(pix) + 0] = acc.x;
        accum[3 * size_t(pix) + 1] = acc.y;
        accum[3 * size_t(pix) + 2] = acc.z;
    }
}

// Caller-supplied rays through the tracing kernels of the renderer: every ray becomes the ray of a path
// record, one tracing step runs, and the hit words of the records are turned into pt_hit.
__global__ void __launch_bounds__(256) fillRaysKernel(Pool pool, uint64_t n, const float* __restrict__ origins,
                                                      const float* __restrict__ dirs, const float* __restrict__ times)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x; // the grid covers the records exactly
    if (slot < n) {
        pool.o_t[slot] = make_float4(origins[3 * size_t(slot) + 0], origins[3 * size_t(slot) + 1], origins[3 * size_t(slot) + 2],
                                     times ? times[slot] : 0.0f);
        pool.d_s[slot] = make_float4(dirs[3 * size_t(slot) + 0], dirs[3 * size_t(slot) + 1], dirs[3 * size_t(slot) + 2],
                                     __uint_as_float(stateWord(kActive, 0u, 0u)));
    } else {
        pool.d_s[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(stateWord(kEmpty, 0u, 0u)));
    }
    pool.cls[slot] = kClsEmpty;
}

__global__ void __launch_bounds__(256) collectHitsKernel(DevScene sc, Pool pool, uint64_t n, pt_hit* __restrict__ out)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n)
        return;
    pt_hit r;
    r.surface = PT_NO_HIT;
    r.material = PT_NO_HIT;
    r.a = 0.0f;
    r.backside = 0;
    if (pool.cls[slot] >= kClsHit) {
        const uint4 hw = pool.hit[slot];
        const float4 o_t = pool.o_t[slot], d_s = pool.d_s[slot];
        const Ray ray = makeRay(v3(o_t.x, o_t.y, o_t.z), v3(d_s.x, d_s.y, d_s.z), o_t.w);
        HitLite hl;
        hl.sid = hw.x & 0x7FFFFFFFu;
        hl.backside = (hw.x >> 31) != 0;
        hl.a = __uint_as_float(hw.y);
        hl.u = __uint_as_float(hw.z);
        hl.v = __uint_as_float(hw.w);
        hl.matClass = 0;
        const Hit hr = makeHit(sc, ray, hl);
        r.surface = hl.sid;
        r.material = uint32_t(hr.material);
        r.a = hl.a;
        r.backside = hr.backside ? 1u : 0u;
    }
    out[slot] = r;
}

uint64_t envU64(const char* name, uint64_t dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v)
        return dflt;
    return std::strtoull(v, nullptr, 10);
}

// capacity of one shard's shadow queue: the slots of the blocks of that shard, rounded up to whole wave runs
constexpr uint32_t shadowCapOf(uint32_t nSlots)
{
    const uint32_t blocks = nSlots / 256u;
    const uint32_t perShard = (blocks + kShards - 1) / kShards * 256u;
    const uint32_t run = 64u * kItemsPerLane;
    return (perShard + run - 1) / run * run;
}

// waves of one tracing launch: whole blocks of slot waves, then whole blocks of shadow-queue waves
constexpr void traceWavesOf(uint32_t nSlots, uint32_t& slotWaves, uint32_t& shadowWaves)
{
    const uint32_t run = 64u * kItemsPerLane;
    slotWaves = ((nSlots + run - 1) / run + 3u) / 4u * 4u;
    shadowWaves = (shadowCapOf(nSlots) * kShards / run + 3u) / 4u * 4u;
}

// entries per retry sub-list: every wave of a tracing launch may put its whole run on its sub-list
constexpr uint32_t retryCapOf(uint32_t nSlots)
{
    uint32_t slotWaves = 0, shadowWaves = 0;
    traceWavesOf(nSlots, slotWaves, shadowWaves);
    return (slotWaves + shadowWaves + kRetryShards - 1) / kRetryShards * (64u * kItemsPerLane);
}

constexpr size_t poolBytes(uint32_t n)
{
    return 2 * size_t(n) * (16 * 4 + 8 + 1) + size_t(shadowCapOf(n)) * kShards * (16 * 4 + 4)
        + size_t(retryCapOf(n)) * kRetryShards * 8;
}

// Two pools of path records that share the queues and lists.
void carvePools(void* base, uint32_t n, Pool pools[2])
{
    Pool p;
    char* c = static_cast<char*>(base);
    auto take = [&](size_t bytes) {
        void* r = c;
        c += bytes;
        return r;
    };
    const size_t q = size_t(shadowCapOf(n)) * kShards;
    p.sh_o = static_cast<float4*>(take(q * 16));
    p.sh_d = static_cast<float4*>(take(q * 16));
    p.sh_c = static_cast<float4*>(take(q * 16));
    p.sh_b = static_cast<float4*>(take(q * 16));
    p.retryCap = retryCapOf(n);
    p.retry = static_cast<uint2*>(take(size_t(p.retryCap) * kRetryShards * 8));
    p.listCap = shadowCapOf(n);
    p.hitList = static_cast<uint32_t*>(take(q * 4));
    p.nSlots = n;
    for (int k = 0; k < 2; k++) {
        p.o_t = static_cast<float4*>(take(size_t(n) * 16));
        p.d_s = static_cast<float4*>(take(size_t(n) * 16));
        p.hit = static_cast<uint4*>(take(size_t(n) * 16));
        p.thr_p = static_cast<float4*>(take(size_t(n) * 16));
        p.key = static_cast<uint2*>(take(size_t(n) * 8));
        p.cls = static_cast<uint8_t*>(take(size_t(n)));
        pools[k] = p;
    }
}

int ensureBuffers(pt_scene* sc, uint32_t nSlots, uint64_t nSamples, uint64_t npix)
{
    WavefrontBuffers& w = sc->wf;
    const size_t needPool = poolBytes(nSlots);
    if (w.poolCapacity < needPool) {
        if (w.pool)
            (void)hipFree(w.pool);
        w.pool = nullptr;
        w.poolCapacity = 0;
        HIP_TRY(hipMalloc(&w.pool, needPool));
        w.poolCapacity = needPool;
    }
    if (w.sampleCapacity < nSamples) {
        if (w.samples)
            (void)hipFree(w.samples);
        w.samples = nullptr;
        w.sampleCapacity = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.samples), nSamples * 3 * sizeof(float)));
        w.sampleCapacity = nSamples;
    }
    if (w.accumCapacity < npix) {
        if (w.accum)
            (void)hipFree(w.accum);
        w.accum = nullptr;
        w.accumCapacity = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.accum), npix * 3 * sizeof(float)));
        w.accumCapacity = npix;
    }
    if (!w.counters)
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.counters), kCtrWords * sizeof(uint32_t)));
    if (!w.hostCounters)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&w.hostCounters), kCtrWords * sizeof(uint32_t)));
    if (!w.walkStats) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.walkStats), kStatWords * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(w.walkStats, 0, kStatWords * sizeof(unsigned long long)));
    }
    return PT_OK;
}

// node + surface tables up to this size are traced from LDS; with moving surfaces the animation table goes there
// too, and all three may take twice as much
constexpr size_t kLdsGeoLimit = 16 * 1024;

#ifndef PT_OCTANT_BLOCK
#define PT_OCTANT_BLOCK 256
#endif
constexpr uint32_t kOctantBlock = PT_OCTANT_BLOCK;
constexpr size_t kOctantLdsLimit = 20 * 1024; // eight blocks per compute unit

// bytes of LDS that the tables of a scene traced from LDS take
size_t ldsTableBytes(const pt_scene* sc)
{
    return (size_t(2) * sc->dev.n_nodes + size_t(3) * sc->dev.n_surfaces) * 16
        + (sc->hasDynamic ? size_t(sc->dev.n_animations) * sizeof(pt_animation) : 0);
}

bool sceneFitsLds(const pt_scene* sc)
{
    return (size_t(2) * sc->dev.n_nodes + size_t(3) * sc->dev.n_surfaces) * 16 <= kLdsGeoLimit
        && ldsTableBytes(sc) <= 2 * kLdsGeoLimit && std::getenv("PT_NO_LDS_SCENE") == nullptr;
}

// ---- the link table of the walk from LDS (DevScene::walkLinks) ----
//
// The storage-order walk tests every box on its way; an inner node's box only decides whether the boxes inside it
// are looked at.  A ray that misses a box misses every box inside it (the slab arithmetic is monotone in the planes,
// see aabbEntry), and a leaf's own box is always tested before its surface, so taking an inner node OUT of the walk
// -- going straight on to its first child -- changes neither which surfaces are tested nor their order: only the
// number of box tests.  Leaving node P out saves one test whenever P's box is hit and costs the tests of its
// children's boxes whenever it is missed, so it pays for boxes that most rays hit: the upper levels of a tree seen
// from inside the scene, and every node whose first child has the very same box (the two triangles of an
// axis-parallel wall): a miss of that child is a miss of P, and the child's link for a miss is P's.
//
// How often a box is hit is measured, not guessed: one block traces a few thousand rays between random points of
// the scene's surfaces (chosen by area, directions weighted by the cosine at the start, as bounced rays are) through
// the full tree and counts visits and hits per node; one thread then decides bottom-up -- leave P out if
// (visits - hits) x (boxes tested instead after a miss) < visits -- and writes for every node the two link words of
// the walk (packNodeLinks' layout) and, behind them, the node the walk starts at.  Two tables are made: all nodes
// (start below the root, as a walk without table has it) and the pruned one.
constexpr uint32_t kWalkLinkNodes = 1024; // a tree that fits LDS has a few hundred nodes at most
constexpr uint32_t kWalkLinkRays = 8;     // per thread of the one block
__global__ void __launch_bounds__(1024) walkLinksKernel(DevScene sc, float time, uint2* __restrict__ pruned)
{
    __shared__ uint32_t visits[kWalkLinkNodes], hits[kWalkLinkNodes];
    __shared__ uint32_t esc[kWalkLinkNodes], leafWord[kWalkLinkNodes], escTo[kWalkLinkNodes];
    __shared__ uint16_t frontier[kWalkLinkNodes];
    __shared__ uint8_t sameBox[kWalkLinkNodes], out_[kWalkLinkNodes], chain[kWalkLinkNodes];
    __shared__ float cum[kWalkLinkNodes / 2 + 1];
    const uint32_t n = sc.n_nodes, nSurf = sc.n_surfaces;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        visits[i] = hits[i] = 0;
        const float4 a0 = sc.nodes[2u * i], a1 = sc.nodes[2u * i + 1u];
        esc[i] = fbits(a1.z);
        leafWord[i] = fbits(a1.w);
        // (bit patterns: a box is "the same" only if a test of it is the same computation)
        bool same = false;
        if ((fbits(a1.w) & kInnerBit) && i + 1u < n) {
            const float4 b0 = sc.nodes[2u * i + 2u], b1 = sc.nodes[2u * i + 3u];
            same = fbits(a0.x) == fbits(b0.x) && fbits(a0.y) == fbits(b0.y) && fbits(a0.z) == fbits(b0.z) && fbits(a0.w) == fbits(b0.w)
                && fbits(a1.x) == fbits(b1.x) && fbits(a1.y) == fbits(b1.y);
        }
        sameBox[i] = same ? 1 : 0;
    }
    // surfaces by area
    for (uint32_t s = threadIdx.x; s < nSurf; s += blockDim.x) {
        const float4 r0 = sc.surf[3u * s];
        float area;
        if (fbits(r0.w) <= kTriDynamic) {
            V3 A, B, C;
            triVertices(sc, s, time, A, B, C);
            const V3 cr = cross(B - A, C - A);
            area = 0.5f * ::sqrtf(dot(cr, cr));
        } else {
            V3 c;
            float r;
            Xform T;
            uint32_t sphereId;
            sphereCR(sc, s, time, c, r, T, sphereId);
            area = 4.0f * kPi * r * r;
        }
        cum[s + 1u] = (area == area && area > 0.0f && area < 1.0e30f) ? area : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cum[0] = 0.0f;
        for (uint32_t s = 0; s < nSurf; s++)
            cum[s + 1u] += cum[s];
    }
    __syncthreads();
    const float total = cum[nSurf];
    auto pick = [&](float u) {
        const float x = u * total;
        uint32_t lo = 0, hi = nSurf; // first s with cum[s + 1] > x
        while (lo + 1u < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cum[mid] <= x)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    };
    auto pointOn = [&](uint32_t s, float u0, float u1, V3& normal) {
        const float4 r0 = sc.surf[3u * s];
        if (fbits(r0.w) <= kTriDynamic) {
            V3 A, B, C;
            triVertices(sc, s, time, A, B, C);
            const V3 b = uniformInTriangle(u0, u1);
            normal = cross(B - A, C - A);
            return b.x * A + b.y * B + b.z * C;
        }
        V3 c;
        float r;
        Xform T;
        uint32_t sphereId;
        sphereCR(sc, s, time, c, r, T, sphereId);
        normal = uniformOnSphere(u0, u1);
        return c + r * normal;
    };
    if (total > 0.0f && nSurf > 1u) {
        for (uint32_t k = 0; k < kWalkLinkRays; k++) {
            const uint64_t key = pt_rng_key(0x6c696e6b73ull, threadIdx.x, k);
            const uint32_t sa = pick(pt_rng_in01(key, 0)), sb = pick(pt_rng_in01(key, 1));
            V3 na, nb;
            const V3 pa = pointOn(sa, pt_rng_in01(key, 2), pt_rng_in01(key, 3), na);
            const V3 pb = pointOn(sb, pt_rng_in01(key, 4), pt_rng_in01(key, 5), nb);
            const V3 d = pb - pa;
            const float len = ::sqrtf(dot(d, d)), nlen = ::sqrtf(dot(na, na));
            if (!(len > 0.0f) || !(nlen > 0.0f) || !(len < 1.0e30f))
                continue;
            const V3 dir = d / len;
            // a bounced ray leaves its surface with a density proportional to the cosine
            if (!(fabs_(dot(na, dir)) / nlen > pt_rng_in01(key, 6)))
                continue;
            const Ray ray = makeRay(pa, dir, time);
            float amax = kFltMax;
            HitLite best = noHit();
            uint32_t cur = 0;
            while (cur < n) {
                uint32_t pending;
                const uint32_t at = cur;
                nodeStep<false>(sc.nodes, ray, 0.0001f, amax, cur, pending);
                atomicAdd(&visits[at], 1u);
                if (pending != kInner || ((leafWord[at] & kInnerBit) && cur == at + 1u))
                    atomicAdd(&hits[at], 1u);
                if (pending != kInner && surfaceTest<true>(sc, sc.surf, pending, ray, 0.0001f, amax, best))
                    amax = best.a;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    // Bottom-up (children lie behind their parent): out_[i] = node i is left out; frontier[i] = boxes tested when the
    // walk arrives at i and misses everything: 1 for a node that is tested itself; chain[i] = the first box tested on
    // arrival at i is i's own box or equal to it.
    uint32_t nOut = 0;
    for (uint32_t i = n; i-- > 0;) {
        out_[i] = 0;
        frontier[i] = 1;
        chain[i] = 1;
        if (!(leafWord[i] & kInnerBit))
            continue;
        const uint32_t c1 = i + 1u, c2 = leafWord[i] & ~kInnerBit;
        const bool viaChain = sameBox[i] && chain[c1];
        const uint32_t afterMiss = viaChain ? 1u : uint32_t(frontier[c1]) + frontier[c2];
        const uint64_t v = visits[i], h = hits[i];
        if (v > 0 && (v - h) * afterMiss < v && afterMiss < 60000u) {
            out_[i] = 1;
            frontier[i] = uint16_t(afterMiss);
            chain[i] = viaChain ? 1 : 0;
            nOut++;
        }
    }
    // Top-down: where a miss of node i leads (before nodes left out are skipped): its escape link, or that of the
    // node above whose place it takes (same box, first thing tested there).
    escTo[0] = n;
    for (uint32_t i = 0; i < n; i++) {
        if (!(leafWord[i] & kInnerBit))
            continue;
        const uint32_t c1 = i + 1u, c2 = leafWord[i] & ~kInnerBit;
        escTo[c1] = (out_[i] && sameBox[i] && chain[c1]) ? escTo[i] : esc[c1];
        escTo[c2] = esc[c2];
    }
    auto resolve = [&](uint32_t t) {
        while (t < n && out_[t])
            t++; // (a node left out is an inner node: its first child follows it)
        return t;
    };
    for (uint32_t i = 0; i < n; i++) {
        const bool inner = (leafWord[i] & kInnerBit) != 0;
        pruned[i] = make_uint2(resolve(escTo[i]) | (inner ? 0u : (leafWord[i] + 1u) << 16),
                               inner ? resolve(i + 1u) : (kLeafDue | resolve(esc[i])));
    }
    pruned[n] = make_uint2(resolve(0u), nOut);
}

} // namespace

// Make the link table for the tree the scene has now (the caller holds the scene's lock; `time` = a moment at which
// moving surfaces are looked at).  Scenes that are not traced from LDS have none.
int ptWavefrontBuildWalkLinks(pt_scene* scene, float time, hipStream_t stream)
{
    scene->dev.walkLinks = nullptr;
    const uint32_t n = scene->dev.n_nodes;
    if (n == 0 || n > kWalkLinkNodes || n >= 0xFFFFu || scene->dev.n_surfaces > kWalkLinkNodes / 2
        || (size_t(2) * n + size_t(3) * scene->dev.n_surfaces) * 16 > kLdsGeoLimit)
        return PT_OK;
    if (!scene->walkLinks) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, (2 * size_t(scene->n_surfaces) + 1) * sizeof(uint2)));
        scene->owned.push_back(p);
        scene->walkLinks = static_cast<uint2*>(p);
    }
    hipLaunchKernelGGL(walkLinksKernel, dim3(1), dim3(1024), 0, stream, scene->dev, time, scene->walkLinks);
    HIP_TRY(hipGetLastError());
    scene->dev.walkLinks = scene->walkLinks;
    return PT_OK;
}

namespace {

// blocks of a kernel that a compute unit holds at once (at least 1: also the answer if the runtime cannot say)
template <typename K>
int residentBlocks(K kernel, uint32_t blockSize)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, int(blockSize), 0) != hipSuccess || n < 1)
        n = 1;
    return n;
}

// the tracing step of one iteration
template <bool DYN, bool LDS, bool STATS>
void launchTrace(pt_scene* sc, WavefrontBuffers& w, const Pool& pool, hipStream_t stream, int parity)
{
    const dim3 block(256);
    const uint32_t shadowCap = shadowCapOf(pool.nSlots);
    uint32_t slotWaves, shadowWaves;
    traceWavesOf(pool.nSlots, slotWaves, shadowWaves);
    const size_t lds = LDS ? ldsTableBytes(sc) : 0;
    // (PT_NO_WALK_PRUNE: the walk from LDS visits every node of the tree below the root, as the reference's does)
    DevScene dv = sc->dev;
    if (std::getenv("PT_NO_WALK_PRUNE") != nullptr)
        dv.walkLinks = nullptr;
    // scenes in memory: nearer child first, then the few rays that leaves open in storage order
    const char* traversal = std::getenv("PT_TRAVERSAL");
    const bool referenceOrder = traversal && std::strcmp(traversal, "reference") == 0;
    // (for scenes held in LDS the storage-order walk is the faster one: measured 353 against 610 us per launch)
    const bool ordered = !LDS && !referenceOrder && sc->dev.wide != nullptr;
    unsigned long long* stats = STATS ? w.walkStats : nullptr;
    const char* runs = std::getenv("PT_TRACE_RUNS");
    const bool fixedRuns = runs && std::strcmp(runs, "fixed") == 0, takenRuns = runs && std::strcmp(runs, "taken") == 0;
    if (ordered) {
        sc->kt.begin(KernelTimes::kTraceOrdered, stream);
        // runs taken from counters by as many waves as the chip holds at once (PT_TRACE_RUNS=fixed: a wave per run)
        const uint32_t allBlocks = (slotWaves + shadowWaves) / (kOrderedBlock / 64u);
        // (PT_NO_QUAD_ROWS: the two-wide rows, one box pair per step)
        DevScene dq = sc->dev;
        if (std::getenv("PT_NO_QUAD_ROWS") != nullptr)
            dq.quad = nullptr;
        const bool quad = dq.quad != nullptr;
        // (blocks per compute unit: what registers and the stacks in LDS allow)
        static const int blocksPerUnit[2] = { residentBlocks(traceOrderedKernel<2, DYN, STATS>, kOrderedBlock),
                                              residentBlocks(traceOrderedKernel<4, DYN, STATS>, kOrderedBlock) };
        const uint32_t held = uint32_t(sc->computeUnits) * uint32_t(blocksPerUnit[quad ? 1 : 0]);
        const bool taken = !fixedRuns && held < allBlocks;
        if (quad)
            hipLaunchKernelGGL((traceOrderedKernel<4, DYN, STATS>), dim3(taken ? held : allBlocks), dim3(kOrderedBlock), 0, stream, dq,
                               pool, w.counters, parity, w.samples, shadowCap, slotWaves, taken ? 1 : 0, stats);
        else
            hipLaunchKernelGGL((traceOrderedKernel<2, DYN, STATS>), dim3(taken ? held : allBlocks), dim3(kOrderedBlock), 0, stream, dq,
                               pool, w.counters, parity, w.samples, shadowCap, slotWaves, taken ? 1 : 0, stats);
        sc->kt.end(stream);
        // a multiple of kRetryShards blocks: a few per sub-list, more only for large pools
        const uint32_t perList = std::min<uint32_t>(32u, std::max<uint32_t>(1u, pool.retryCap / 1024u));
        const uint32_t retryBlocks = kRetryShards * perList;
        sc->kt.begin(KernelTimes::kTraceRetry, stream);
        hipLaunchKernelGGL((traceRetryKernel<DYN, STATS>), dim3(retryBlocks), block, 0, stream, sc->dev, pool, w.counters,
                           w.samples, stats);
        sc->kt.end(stream);
    } else {
        sc->kt.begin(KernelTimes::kTrace, stream);
        // Static scenes: a wave per run (their walks are short, and the finer hand-out by the hardware is worth more
        // than filled lanes across runs: 882 against 959 us on the Cornell box); moving surfaces: runs taken from
        // counters by as many waves as the chip holds (601 against 667 us on the 100 moving spheres) -- eight per
        // SIMD, fewer when the scene's tables leave room for fewer blocks in LDS.  PT_TRACE_RUNS=fixed|taken overrides.
        // The smallest static scenes: a copy of every node per combination of direction signs, in blocks of
        // kOctantBlock threads that share the eight-fold table (PT_NO_OCTANT_ROWS: the plain table).
        const size_t octLds = (size_t(kOctantRow4) * sc->dev.n_nodes + size_t(3) * sc->dev.n_surfaces) * 16;
        if (LDS && !DYN && octLds <= kOctantLdsLimit && std::getenv("PT_NO_OCTANT_ROWS") == nullptr) {
            const uint32_t blocks = (slotWaves + shadowWaves) / (kOctantBlock / 64u);
            const uint32_t heldOct = uint32_t(sc->computeUnits) * std::min<uint32_t>(32u / (kOctantBlock / 64u), uint32_t((160u * 1024u) / octLds));
            const bool takenOct = takenRuns && heldOct < blocks;
            hipLaunchKernelGGL((traceKernel<false, LDS && !DYN, STATS, kOctantBlock, LDS && !DYN>), dim3(takenOct ? heldOct : blocks), dim3(kOctantBlock),
                               octLds, stream, dv, pool, w.counters, parity, w.samples, shadowCap, slotWaves, takenOct ? 1 : 0, stats);
            sc->kt.end(stream);
            return;
        }
        const bool waveBlocks = LDS && !DYN && lds * 32u <= 160u * 1024u;
        const uint32_t wavesPerBlock = waveBlocks ? 1u : 4u, allBlocks = (slotWaves + shadowWaves) / wavesPerBlock;
        const uint32_t perCu = std::max<uint32_t>(1u, lds ? std::min<uint32_t>(32u / wavesPerBlock, uint32_t((160u * 1024u) / lds))
                                                          : 32u / wavesPerBlock);
        const uint32_t held = uint32_t(sc->computeUnits) * perCu;
        const bool taken = (takenRuns || (DYN && !fixedRuns)) && held < allBlocks;
        if (waveBlocks)
            hipLaunchKernelGGL((traceKernel<DYN, LDS, STATS, (LDS && !DYN) ? 64u : 256u>), dim3(taken ? held : allBlocks), dim3(64), lds, stream,
                               dv, pool, w.counters, parity, w.samples, shadowCap, slotWaves, taken ? 1 : 0, stats);
        else
            hipLaunchKernelGGL((traceKernel<DYN, LDS, STATS, 256u>), dim3(taken ? held : allBlocks), dim3(256), lds, stream, dv, pool,
                               w.counters, parity, w.samples, shadowCap, slotWaves, taken ? 1 : 0, stats);
        sc->kt.end(stream);
    }
}

void launchTraceFor(pt_scene* sc, WavefrontBuffers& w, const Pool& pool, hipStream_t stream, int parity, bool dyn, bool lds)
{
    // (the instrumented kernels only when the scene's walk statistics are switched on and their block exists)
    const bool stats = sc->walkStatsOn && w.walkStats != nullptr;
    const int which = (dyn ? 4 : 0) | (lds ? 2 : 0) | (stats ? 1 : 0);
    switch (which) {
    case 0: launchTrace<false, false, false>(sc, w, pool, stream, parity); break;
    case 1: launchTrace<false, false, true>(sc, w, pool, stream, parity); break;
    case 2: launchTrace<false, true, false>(sc, w, pool, stream, parity); break;
    case 3: launchTrace<false, true, true>(sc, w, pool, stream, parity); break;
    case 4: launchTrace<true, false, false>(sc, w, pool, stream, parity); break;
    case 5: launchTrace<true, false, true>(sc, w, pool, stream, parity); break;
    case 6: launchTrace<true, true, false>(sc, w, pool, stream, parity); break;
    default: launchTrace<true, true, true>(sc, w, pool, stream, parity); break;
    }
}

// The shading launch, with the scene's material-side tables in LDS where they fit (PT_NO_SHADE_LDS: never).
constexpr uint32_t kShadeLdsLimit = 24 * 1024;

size_t shadeTableBytes(const pt_scene* sc)
{
    if (std::getenv("PT_NO_SHADE_LDS") != nullptr)
        return 0;
    // (32-bit sizes inside: a scene anywhere near the limit of those is far beyond what fits)
    if (sc->nMaterials >= (1u << 20) || sc->nTextures >= (1u << 20) || sc->nMeshes >= (1u << 20))
        return 0;
    const ShadeTableCounts n = { sc->nMaterials, sc->nTextures, sc->nMeshes };
    const uint64_t bytes = uint64_t(shadeTableWords(n)) * 4u;
    return bytes <= kShadeLdsLimit ? size_t(bytes) : 0;
}

void launchShade(pt_scene* sc, size_t ldsBytes, dim3 grid, hipStream_t stream, const Pool& cur, const Pool& next,
                 const PathParams& pp, uint32_t* counters, int parity, float* samples, uint32_t shadowCap, FastDiv blocksPerShard,
                 unsigned long long* stats)
{
    const ShadeTableCounts n = { sc->nMaterials, sc->nTextures, sc->nMeshes };
    if (ldsBytes > 0)
        hipLaunchKernelGGL(shadeHitsKernel<true>, grid, dim3(kShadeBlock), ldsBytes, stream, sc->dev, cur, next, pp, counters, parity,
                           samples, shadowCap, blocksPerShard, stats, n);
    else
        hipLaunchKernelGGL(shadeHitsKernel<false>, grid, dim3(kShadeBlock), 0, stream, sc->dev, cur, next, pp, counters, parity,
                           samples, shadowCap, blocksPerShard, stats, n);
}

// Run one batch of samples [b.sFirst, b.sFirst + b.nS) of every pixel of the tile into the sample buffer.
int runBatch(pt_scene* sc, WavefrontBuffers& w, const RenderArgs& a, const Batch& b, const Pool pools[2], hipStream_t stream)
{
    const uint32_t nSlots = pools[0].nSlots;
    // (no paths yet: no records in use at the front of any region, so the first tracing and classifying launches
    // look at nothing and the pools need no clearing)
    sc->kt.unchain(); // (what follows on the stream is not a timed launch)
    HIP_TRY(hipMemsetAsync(w.counters, 0, kCtrWords * sizeof(uint32_t), stream));
    const uint64_t maxIters = (uint64_t(b.totalIds) / nSlots + 2) * uint64_t(a.pp.maxSegments + 2) + 64;
    const bool dyn = sc->hasDynamic;
    const bool lds = sceneFitsLds(sc);
    const uint32_t shadowCap = shadowCapOf(nSlots);
    const dim3 block(256);
    const size_t shadeLds = shadeTableBytes(sc);
    // the first tracing launch has no generateKernel before it: nothing to add to the path numbers
    for (uint32_t q = 0; q < kShards; q++)
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.counters + (kCtrLive + q) * kCtrStride),
                                  int(nSlots / kShards), 1, stream));
    bool exhaustedBefore = false; // every path number had been handed out at the previous check already
    // paths left below which the batch is finished by finishPathsKernel (0: never)
    // (4 K to 1 M measured alike: what decides is when the host next looks at the counters)
    const uint64_t finishLimit = envU64("PT_FINISH_PATHS", 32768);
    uint64_t iter = 0;
    // Blocks per region (per list) of the classifying, shading and generating launches: all of it while paths are
    // being started.  Once the last one has been, the pool only drains -- with Russian roulette for a hundred
    // iterations and more, the last ones with a handful of paths -- and the counts read back bound what any later
    // iteration can hold, so the launches shrink to that: an iteration on an almost empty pool costs 80 us with
    // grids over 16 M records, most of it the hardware handing out blocks that find nothing to do.
    const uint32_t regionCap = nSlots / kShards;
    uint32_t classifyBlocks = regionCap / kClassifyRecords, shadeBlocks = pools[0].listCap / kShadeBlock, generateBlocks = regionCap / 256u;
    // iterations between two looks at the counters (a batch that fits the pool has started all its paths at once)
    int checkEvery = b.totalIds <= nSlots ? 2 : 16;
    for (;;) {
        for (int k = 0; k < checkEvery; k++, iter++) {
            const int parity = int(iter & 1);
            const Pool& cur = pools[iter & 1];
            const Pool& next = pools[(iter + 1) & 1];
            launchTraceFor(sc, w, cur, stream, parity, dyn, lds);
            sc->kt.begin(KernelTimes::kClassify, stream);
            hipLaunchKernelGGL(classifyKernel, dim3(classifyBlocks * kShards), dim3(kClassifyBlock), 0, stream, sc->dev, cur,
                               w.counters, w.samples, FastDiv(classifyBlocks));
            sc->kt.end(stream);
            sc->kt.begin(KernelTimes::kShade, stream);
            launchShade(sc, shadeLds, dim3(shadeBlocks * kShards), stream, cur, next, a.pp, w.counters, parity, w.samples, shadowCap,
                        FastDiv(shadeBlocks), sc->walkStatsOn ? w.walkStats : nullptr);
            sc->kt.end(stream);
            sc->kt.begin(KernelTimes::kGenerate, stream);
            hipLaunchKernelGGL(generateKernel, dim3(generateBlocks * kShards), block, 0, stream, sc->dev, next, a, b, w.counters,
                               w.samples, FastDiv(generateBlocks));
            sc->kt.end(stream);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(w.hostCounters, w.counters, kCtrWords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        // (only this stream has been waited for: spans of an earlier profiled call on another stream stay where they
        // are while this call records none)
        if (sc->kt.enabled)
            sc->kt.collect();
        // finished when the last iteration shaded nothing and every path number had been handed out
        // well before it (so that its generateKernel cannot have started a path)
        uint64_t alive = 0, goingOnAll = 0;
        bool exhausted = true;
        for (uint32_t q = 0; q < kShards; q++) {
            const uint32_t goingOn = w.hostCounters[(kCtrLive + q) * kCtrStride];
            alive += w.hostCounters[(kCtrHit + q) * kCtrStride] + goingOn;
            goingOnAll += goingOn;
            // path numbers of shard q: the counter as of the last tracing launch + the last generateKernel's
            const uint64_t handedOut = uint64_t(w.hostCounters[q * kCtrStride]) + (regionCap - goingOn);
            if ((handedOut >> 6) * kShards + q < (b.totalIds >> 6))
                exhausted = false;
        }
        if (exhaustedBefore && alive == 0)
            break;
        exhaustedBefore = exhausted;
        if (exhausted && finishLimit > 0 && !sc->walkStatsOn) {
            // few paths left: one more tracing launch (their hits, and the shadow rays queued last), then every path
            // to its end in one kernel instead of an iteration per segment of the longest
            uint32_t mostInUse = 0;
            uint64_t inUseAll = 0;
            for (uint32_t q = 0; q < kShards; q++) {
                const uint32_t inUse = w.hostCounters[(kCtrOcc + q) * kCtrStride];
                mostInUse = std::max(mostInUse, inUse);
                inUseAll += inUse;
            }
            if (inUseAll <= finishLimit) {
                const Pool& cur = pools[iter & 1];
                launchTraceFor(sc, w, cur, stream, int(iter & 1), dyn, lds);
                if (mostInUse > 0) {
                    const uint32_t perRegion = (mostInUse + 63u) / 64u;
                    sc->kt.begin(KernelTimes::kFinish, stream);
                    hipLaunchKernelGGL(finishPathsKernel, dim3(perRegion * kShards), dim3(64), 0, stream, sc->dev, cur, a.pp, w.counters,
                                       w.samples, perRegion);
                    sc->kt.end(stream);
                }
                HIP_TRY(hipGetLastError());
                return PT_OK;
            }
        }
        if (exhausted) {
            // No launch starts a path any more.  A list of hits is made from the records of every region, and the paths
            // of a list go on in one region, so what bounds a region and a list from here on is the number of all paths
            // that go on; generateKernel is left with noting the records in use, which the first block of a region does.
            classifyBlocks = uint32_t(std::min<uint64_t>(classifyBlocks, std::max<uint64_t>(1u, (goingOnAll + kClassifyRecords - 1u) / kClassifyRecords)));
            shadeBlocks = uint32_t(std::min<uint64_t>(shadeBlocks, std::max<uint64_t>(1u, (goingOnAll + kShadeBlock - 1u) / kShadeBlock)));
            generateBlocks = 1;
            // the paths that are left halve every few iterations: look again sooner the fewer they are (a look costs
            // the stream a pause, an iteration on an empty pool a few launches)
            checkEvery = goingOnAll > 100000 ? 16 : goingOnAll > 1000 ? 8 : goingOnAll > 30 ? 4 : 2;
        }
        if (iter > maxIters)
            return fail(PT_ERR_DEVICE, "wavefront renderer did not finish within its iteration bound");
    }
    // the last paths may have queued shadow rays in the very last shading launch: trace those
    launchTraceFor(sc, w, pools[iter & 1], stream, int(iter & 1), dyn, lds);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int renderBatches(pt_scene* sc, WavefrontBuffers& w, const RenderArgs& a, float* d_out, hipStream_t stream, bool keepSamples)
{
    const uint64_t npix = uint64_t(a.tw) * uint64_t(a.th);
    const uint32_t nsTotal = uint32_t(a.s1 - a.s0);
    if (npix == 0 || nsTotal == 0)
        return PT_OK;
    const uint32_t tilesX = uint32_t(a.tw + 7) / 8u, tilesY = uint32_t(a.th + 7) / 8u;
    const uint64_t idsPerSample = uint64_t(tilesX) * tilesY * 64u;
    // samples per batch: bounded by the sample-buffer budget and by 32-bit path numbers
    const uint64_t budgetBytes = w.fixed ? w.sampleCapacity * 12 : envU64("PT_SAMPLE_BUFFER_MB", 32768) << 20;
    uint64_t perBatch = std::max<uint64_t>(1, budgetBytes / (idsPerSample * 12)); // (a sample per path number: whole tiles)
    perBatch = std::min<uint64_t>(perBatch, (0xFFFFFFFFull - (1ull << 24)) / idsPerSample);
    if (perBatch == 0)
        return fail(PT_ERR_LIMIT, "tile too large for the wavefront renderer");
    perBatch = std::min<uint64_t>(perBatch, nsTotal);
    if (keepSamples && perBatch < nsTotal)
        return fail(PT_ERR_LIMIT, "too many samples for one per-sample call");
    // (32 M records per pool: 4 % more than 16 M on the Cornell box, 5 % on the scenes in memory, 64 M no more)
    uint64_t nSlots = std::min<uint64_t>(w.fixed ? w.fixedSlots : envU64("PT_WAVEFRONT_SLOTS", 1u << 25), idsPerSample * perBatch);
    const uint64_t slotUnit = uint64_t(kClassifyRecords) * kShards; // every shard's region and list is whole blocks
    nSlots = std::max<uint64_t>((nSlots + slotUnit - 1) / slotUnit * slotUnit, slotUnit);
    if (w.fixed) {
        // work areas of a fixed size (nothing is allocated here): the caller has checked the pixel count
        if (nSlots > w.fixedSlots || poolBytes(uint32_t(nSlots)) > w.poolCapacity || idsPerSample * perBatch > w.sampleCapacity
            || npix > w.accumCapacity)
            return fail(PT_ERR_LIMIT, "render does not fit the fixed work area");
    } else {
        int rc = ensureBuffers(sc, uint32_t(nSlots), idsPerSample * perBatch, npix);
        if (rc != PT_OK)
            return rc;
    }
    int rc = PT_OK;
    Pool pools[2];
    carvePools(w.pool, uint32_t(nSlots), pools);
    sc->kt.unchain();
    HIP_TRY(hipEventRecord(sc->evStart, stream));
    for (uint32_t s = 0; s < nsTotal; s += uint32_t(perBatch)) {
        Batch b;
        b.tilesX = tilesX;
        b.tilesY = tilesY;
        b.nS = std::min<uint32_t>(uint32_t(perBatch), nsTotal - s);
        b.byTilesX = FastDiv(tilesX);
        b.byNS = FastDiv(b.nS);
        b.bySqrtSpp = FastDiv(uint32_t(a.p.sqrt_spp));
        b.sFirst = uint32_t(a.s0) + s;
        b.npix = uint32_t(npix);
        b.totalIds = uint32_t(idsPerSample * b.nS);
        rc = runBatch(sc, w, a, b, pools, stream);
        if (rc != PT_OK)
            return rc;
        if (d_out) {
            sc->kt.begin(KernelTimes::kResolve, stream);
            hipLaunchKernelGGL(resolveKernel, dim3((tilesX * tilesY + 3u) / 4u), dim3(256), 0, stream, w.samples, w.accum,
                               d_out, a, b, s == 0 ? 1 : 0, s + b.nS >= nsTotal ? 1 : 0);
            sc->kt.end(stream);
            HIP_TRY(hipGetLastError());
        }
    }
    sc->kt.unchain();
    HIP_TRY(hipEventRecord(sc->evStop, stream));
    sc->timed = true;
    return PT_OK;
}

} // namespace

int ptWavefrontRender(pt_scene* scene, const RenderArgs& a, float* d_out, hipStream_t stream)
{
    return renderBatches(scene, scene->wf, a, d_out, stream, false);
}

// Per-sample radiance in pixel-major, sample-minor order (host buffer).
int ptWavefrontSamples(pt_scene* scene, const RenderArgs& a, float* h_out)
{
    int rc = renderBatches(scene, scene->wf, a, nullptr, nullptr, true);
    if (rc != PT_OK)
        return rc;
    const uint64_t npix = uint64_t(a.tw) * uint64_t(a.th);
    const uint64_t ns = uint64_t(a.s1 - a.s0);
    // the buffer is ordered [8x8 tile][sample][pixel of the tile]
    const uint64_t tilesX = uint64_t(a.tw + 7) / 8u, tilesY = uint64_t(a.th + 7) / 8u;
    std::vector<float> tmp(tilesX * tilesY * 64u * ns * 3);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(tmp.data(), scene->wf.samples, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (uint64_t p = 0; p < npix; p++) {
        const uint64_t lx = p % uint64_t(a.tw), ly = p / uint64_t(a.tw);
        const uint64_t tile = (ly / 8u) * tilesX + lx / 8u, inTile = (ly % 8u) * 8u + lx % 8u;
        for (uint64_t s = 0; s < ns; s++)
            std::memcpy(h_out + 3 * (p * ns + s), tmp.data() + 3 * ((tile * ns + s) * 64u + inTile), 12);
    }
    return PT_OK;
}

// BVHTreeLinear::hit (reference src/15/bvh.hpp:183-213) of n rays in device memory with the renderer's own
// bounds (amin 1e-4, amax FLT_MAX), through the tracing kernels a render of this scene uses.
int ptWavefrontTraceRays(pt_scene* scene, uint64_t n, const float* d_origins, const float* d_dirs, const float* d_times,
                         pt_hit* d_out, hipStream_t stream)
{
    const uint64_t slotUnit = uint64_t(kClassifyRecords) * kShards;
    const uint64_t nSlots = (n + slotUnit - 1) / slotUnit * slotUnit;
    if (nSlots > (1ull << 31))
        return fail(PT_ERR_LIMIT, "too many rays for one call");
    int rc = ensureBuffers(scene, uint32_t(nSlots), 1, 1);
    if (rc != PT_OK)
        return rc;
    WavefrontBuffers& w = scene->wf;
    Pool pools[2];
    carvePools(w.pool, uint32_t(nSlots), pools);
    HIP_TRY(hipMemsetAsync(w.counters, 0, kCtrWords * sizeof(uint32_t), stream));
    // (no generateKernel ran: nothing to add to the path numbers, see runBatch; every record of the pool is in use)
    for (uint32_t q = 0; q < kShards; q++) {
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.counters + (kCtrLive + q) * kCtrStride),
                                  int(nSlots / kShards), 1, stream));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.counters + (kCtrOcc + q) * kCtrStride),
                                  int(nSlots / kShards), 1, stream));
    }
    hipLaunchKernelGGL(fillRaysKernel, dim3(uint32_t(nSlots / 256u)), dim3(256), 0, stream, pools[0], n, d_origins, d_dirs,
                       d_times);
    scene->kt.unchain();
    launchTraceFor(scene, w, pools[0], stream, 0, scene->hasDynamic, sceneFitsLds(scene));
    scene->kt.unchain();
    hipLaunchKernelGGL(collectHitsKernel, dim3(uint32_t((n + 255u) / 256u)), dim3(256), 0, stream, scene->dev, pools[0], n,
                       d_out);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// ---- rendering without allocating: a work area that is part of the library's image ----
//
// `scratch` is a device array in the code object, so every GPU the library runs on holds one copy
// of it from the moment the code object is loaded there until the process ends.  Its layout is
// fixed: path pools for kStaticSlots records, the counters, running sums for kStaticMaxPixels
// pixels, and the rest as sample buffer.  In host code the name `scratch` is only a handle for
// hipGetSymbolAddress: the address of the copy on a device is looked up once per device, with
// that device current, and host code passes that pointer -- never the name -- to kernels and
// memsets.  All calls on one GPU share the GPU's copy: a call that arrives while another
// thread is inside a call on that GPU fails rather than waits (`inUse`), and calls one after
// the other -- whatever their scenes and streams -- are ordered on the device by an event that
// each call records behind its last kernel and the next one makes its stream wait for.  The
// area keeps the last call's contents: runBatch clears what has to start at zero.
constexpr uint32_t kStaticSlots = 2u * kClassifyRecords * kShards; // 131 072 records per pool
constexpr uint64_t kStaticMaxPixels = 512u * 1024u;
constexpr size_t kStaticBytes = size_t(64) << 20;
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
struct StaticArea {
    unsigned char* ptr = nullptr; // this device's copy of `scratch`
    bool inUse = false;           // a thread is inside a call that uses the area
    hipEvent_t done = nullptr;    // behind the last kernel of the last call (made at set-up, with the device current)
    bool pending = false;
    uint32_t* hostCounters = nullptr; // where the counters are read back to: pinned host memory, made at set-up
};
StaticArea g_static[kMaxDevices];
std::mutex g_staticMutex;

int ptWavefrontStaticSetup(int device)
{
    if (device < 0 || device >= kMaxDevices)
        return fail(PT_ERR_LIMIT, "device number beyond the table of work areas");
    std::lock_guard<std::mutex> lock(g_staticMutex);
    if (g_static[device].ptr)
        return PT_OK;
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p || (reinterpret_cast<uintptr_t>(p) & 255u) != 0)
        return fail(PT_ERR_DEVICE, "work area of the library image has no usable address on this GPU");
    StaticArea& area = g_static[device];
    if (!area.done)
        HIP_TRY(hipEventCreateWithFlags(&area.done, hipEventDisableTiming));
    if (!area.hostCounters)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&area.hostCounters), kCtrWords * sizeof(uint32_t)));
    area.ptr = static_cast<unsigned char*>(p);
    return PT_OK;
}

int ptWavefrontRenderStatic(pt_scene* scene, const RenderArgs& a, float* d_out, hipStream_t stream)
{
    // (the rectangle is worked on in 8x8 tiles, and samples are kept for whole tiles)
    const uint64_t npix = (uint64_t(a.tw + 7) / 8u) * (uint64_t(a.th + 7) / 8u) * 64u;
    if (npix > kStaticMaxPixels)
        return fail(PT_ERR_LIMIT, "pt_render_device_noalloc renders at most 524288 pixels per call, counted in whole 8x8 tiles (use tile rectangles)");
    const int device = scene->device;
    int rc = ptWavefrontStaticSetup(device); // (a lookup in a table after the first call on this GPU)
    if (rc != PT_OK)
        return rc;
    StaticArea& area = g_static[device];
    {
        std::lock_guard<std::mutex> lock(g_staticMutex);
        if (area.inUse)
            return fail(PT_ERR_INVALID_ARGUMENT, "another thread is inside a call that uses the fixed work area of this GPU");
        area.inUse = true;
    }
    // behind whatever the previous call on this GPU left running in the area, on any stream
    if (area.pending) {
        const hipError_t e = hipStreamWaitEvent(stream, area.done, 0);
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> lock(g_staticMutex);
            area.inUse = false;
            return fail(PT_ERR_DEVICE, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
        }
    }
    // the parts of the area
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    WavefrontBuffers w;
    w.fixed = true;
    w.fixedSlots = kStaticSlots;
    size_t off = 0;
    w.pool = area.ptr;
    w.poolCapacity = pad(poolBytes(kStaticSlots));
    off += w.poolCapacity;
    w.counters = reinterpret_cast<uint32_t*>(area.ptr + off);
    off += pad(kCtrWords * sizeof(uint32_t));
    w.accum = reinterpret_cast<float*>(area.ptr + off);
    w.accumCapacity = kStaticMaxPixels;
    off += pad(kStaticMaxPixels * 12);
    w.samples = reinterpret_cast<float*>(area.ptr + off);
    w.sampleCapacity = (kStaticBytes - off) / 12;
    w.hostCounters = area.hostCounters;
    static_assert((poolBytes(kStaticSlots) + 255) / 256 * 256 + (kCtrWords * 4 + 255) / 256 * 256 + 4 * kStaticMaxPixels * 12
                      <= kStaticBytes,
                  "the fixed work area holds the pools, the counters, the sums and at least three samples per pixel");
    // (no per-kernel event timing here: its events are created on demand)
    const bool timing = scene->kt.enabled;
    scene->kt.enabled = false;
    rc = renderBatches(scene, w, a, d_out, stream, false);
    scene->kt.enabled = timing;
    // the next call on this GPU waits for this one's kernels on the device
    const hipError_t recorded = hipEventRecord(area.done, stream);
    {
        std::lock_guard<std::mutex> lock(g_staticMutex);
        area.pending = area.pending || recorded == hipSuccess;
        area.inUse = false;
    }
    if (rc == PT_OK && recorded != hipSuccess)
        return fail(PT_ERR_DEVICE, std::string("hipEventRecord: ") + hipGetErrorString(recorded));
    return rc;
}

// Switch the instrumented tracing kernels on (clearing the sums) or off; read the sums.
int ptWavefrontWalkStats(pt_scene* scene, int enable)
{
    WavefrontBuffers& w = scene->wf;
    if (enable) {
        if (!w.walkStats)
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.walkStats), kStatWords * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(w.walkStats, 0, kStatWords * sizeof(unsigned long long)));
    }
    scene->walkStatsOn = enable != 0;
    return PT_OK;
}

int ptWavefrontReadWalkStats(pt_scene* scene, pt_walk_counters* out)
{
    static_assert(sizeof(pt_walk_counters) == kStatWords * sizeof(unsigned long long), "one field per sum");
    std::memset(out, 0, sizeof(*out));
    if (scene->wf.walkStats)
        HIP_TRY(hipMemcpy(out, scene->wf.walkStats, sizeof(*out), hipMemcpyDeviceToHost));
    return PT_OK;
}

void ptWavefrontRelease(pt_scene* scene)
{
    WavefrontBuffers& w = scene->wf;
    if (w.pool)
        (void)hipFree(w.pool);
    if (w.samples)
        (void)hipFree(w.samples);
    if (w.accum)
        (void)hipFree(w.accum);
    if (w.counters)
        (void)hipFree(w.counters);
    if (w.hostCounters)
        (void)hipHostFree(w.hostCounters);
    if (w.walkStats)
        (void)hipFree(w.walkStats);
    w = WavefrontBuffers();
    scene->kt.release();
}
