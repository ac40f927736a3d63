// pt_reindex.hip -- the importer's vertex re-indexing, on the device.
//
// For every (material, shape) pair the reference's importer walks the shape's face corners in file order,
// keeps those of faces with that material, and gives every distinct (position, normal, texcoord) index
// triple a new vertex number in order of first appearance, through a std::map per pair (reference
// src/15/import.hpp:233-289).  For a mesh of millions of faces that map is where import time goes.
// The same result without a sequential pass:
//
//   1 keep      flag the corners of faces with the wanted material; an exclusive scan numbers the kept corners
//               in file order (their place in the new index list)
//   2 group     three stable radix sorts of (triple, place) by texcoord, normal and position index bring equal
//               triples together, each group in order of place: the group's first entry is the triple's first
//               appearance
//   3 number    a flag at the place of every first appearance, scanned over the places, numbers the distinct
//               triples in order of first appearance -- the reference's newIndex
//   4 emit      every first appearance writes its vertex (normal normalised as the importer does,
//               import.hpp:265-267); every entry of a group writes the group's number to its place
//
// Normals (texcoords) are delivered only if every distinct triple of the pair has one (import.hpp:260-263,
// 281-284).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <vector>

#include "pt_scene_host.hpp"

using namespace ptd;
using ptint::fail;

namespace {

constexpr uint32_t kBlock = 256;

struct Buffers {
    std::vector<void*> held;
    ~Buffers()
    {
        for (void* p : held)
            (void)hipFree(p);
    }
    template <typename T>
    int get(T** p, size_t count)
    {
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
        held.push_back(q);
        *p = static_cast<T*>(q);
        return PT_OK;
    }
};

__global__ void __launch_bounds__(kBlock) keepFlagsKernel(const int32_t* __restrict__ triMaterial, int32_t material, uint32_t nCorners,
                                                          uint32_t* __restrict__ keep)
{
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c < nCorners)
        keep[c] = triMaterial[c / 3u] == material ? 1u : 0u;
}

// kept corner c -> entry `place`: the three indices (biased by one, so that "no index" = -1 sorts first) and its place
__global__ void __launch_bounds__(kBlock) gatherKeptKernel(const uint32_t* __restrict__ keep, const uint32_t* __restrict__ placeOf,
                                                           const int32_t* __restrict__ vi, const int32_t* __restrict__ ni,
                                                           const int32_t* __restrict__ ti, uint32_t nCorners, uint32_t* __restrict__ kv,
                                                           uint32_t* __restrict__ kn, uint32_t* __restrict__ kt, uint32_t* __restrict__ place)
{
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= nCorners || !keep[c])
        return;
    const uint32_t p = placeOf[c];
    kv[p] = uint32_t(vi[c] + 1);
    kn[p] = uint32_t(max(ni[c], -1) + 1);
    kt[p] = uint32_t(max(ti[c], -1) + 1);
    place[p] = p;
}

// key[i] = table[order[i]]: the next sort key in the current order
__global__ void __launch_bounds__(kBlock) permuteKernel(const uint32_t* __restrict__ table, const uint32_t* __restrict__ order, uint32_t n,
                                                        uint32_t* __restrict__ key)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n)
        key[i] = table[order[i]];
}

// entry i of the grouped order starts a group if its triple differs from the one before; the group's first entry
// has the smallest place (stable sorts), i.e. it is the triple's first appearance
__global__ void __launch_bounds__(kBlock) groupHeadsKernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ kv,
                                                           const uint32_t* __restrict__ kn, const uint32_t* __restrict__ kt, uint32_t n,
                                                           uint32_t* __restrict__ headSlot, uint32_t* __restrict__ firstAt,
                                                           uint32_t* __restrict__ missing)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t p = order[i];
    bool head = i == 0;
    if (!head) {
        const uint32_t q = order[i - 1];
        head = kv[p] != kv[q] || kn[p] != kn[q] || kt[p] != kt[q];
    }
    headSlot[i] = head ? i : 0u; // (an inclusive max-scan then gives every entry the slot of its group's head)
    if (head) {
        firstAt[p] = 1u;
        if (kn[p] == 0u)
            atomicOr(&missing[0], 1u);
        if (kt[p] == 0u)
            atomicOr(&missing[1], 1u);
    }
}

__global__ void __launch_bounds__(kBlock) emitKernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ headSlot,
                                                     const uint32_t* __restrict__ newIndexAt, const uint32_t* __restrict__ kv,
                                                     const uint32_t* __restrict__ kn, const uint32_t* __restrict__ kt, uint32_t n,
                                                     const float* __restrict__ vertices, const float* __restrict__ normals,
                                                     const float* __restrict__ texcoords, const uint32_t* __restrict__ missing,
                                                     uint32_t* __restrict__ outIndices, float* __restrict__ outPos,
                                                     float* __restrict__ outNrm, float* __restrict__ outTc)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t p = order[i];
    const uint32_t hs = headSlot[i];
    const uint32_t idx = newIndexAt[order[hs]];
    outIndices[p] = idx;
    if (hs != i)
        return;
    // a first appearance: the vertex itself (reference src/15/import.hpp:259-271)
    const size_t v = size_t(kv[p] - 1u);
    outPos[3 * size_t(idx) + 0] = vertices[3 * v + 0];
    outPos[3 * size_t(idx) + 1] = vertices[3 * v + 1];
    outPos[3 * size_t(idx) + 2] = vertices[3 * v + 2];
    if (!missing[0]) {
        const size_t k = size_t(kn[p] - 1u);
        const V3 nn = normalize(v3(normals[3 * k + 0], normals[3 * k + 1], normals[3 * k + 2]));
        outNrm[3 * size_t(idx) + 0] = nn.x;
        outNrm[3 * size_t(idx) + 1] = nn.y;
        outNrm[3 * size_t(idx) + 2] = nn.z;
    }
    if (!missing[1]) {
        const size_t k = size_t(kt[p] - 1u);
        outTc[2 * size_t(idx) + 0] = texcoords[2 * k + 0];
        outTc[2 * size_t(idx) + 1] = texcoords[2 * k + 1];
    }
}

struct MaxOp {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

} // namespace

extern "C" {

int pt_mesh_reindex(int device, const int32_t* corner_vi, const int32_t* corner_ni, const int32_t* corner_ti,
                    const int32_t* tri_material, uint64_t n_triangles, int32_t material, const float* vertices,
                    uint64_t n_vertices, const float* normals, uint64_t n_normals, const float* texcoords, uint64_t n_texcoords,
                    uint32_t* out_indices, float* out_positions, float* out_normals, float* out_texcoords,
                    uint64_t* out_n_indices, uint64_t* out_n_vertices, int* out_have_normals, int* out_have_texcoords)
try {
    if (!corner_vi || !corner_ni || !corner_ti || !tri_material || !vertices || !out_indices || !out_positions || !out_normals
        || !out_texcoords || !out_n_indices || !out_n_vertices || !out_have_normals || !out_have_texcoords)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_mesh_reindex: null argument");
    if (n_triangles > (1ull << 30) / 3 || n_vertices >= (1ull << 31) || n_normals >= (1ull << 31) || n_texcoords >= (1ull << 31))
        return fail(PT_ERR_LIMIT, "pt_mesh_reindex: more than 2^30 face corners or 2^31 attributes");
    *out_n_indices = 0;
    *out_n_vertices = 0;
    *out_have_normals = 1;
    *out_have_texcoords = 1;
    const uint32_t nCorners = uint32_t(3 * n_triangles);
    if (nCorners == 0)
        return PT_OK;
    // every index the corners use must exist (a bad file must not become a bad address on the GPU)
    for (uint32_t c = 0; c < nCorners; c++) {
        if (corner_vi[c] < 0 || uint64_t(corner_vi[c]) >= n_vertices || (corner_ni[c] >= 0 && uint64_t(corner_ni[c]) >= n_normals)
            || (corner_ti[c] >= 0 && uint64_t(corner_ti[c]) >= n_texcoords))
            return fail(PT_ERR_INVALID_ARGUMENT, "pt_mesh_reindex: corner " + std::to_string(c) + " refers to a missing attribute");
    }
    if (device < 0 || device >= pt_device_count())
        return fail(PT_ERR_DEVICE, "no such GPU: " + std::to_string(device));
    DeviceGuard g(device);
    Buffers b;
    int rc;
    int32_t *dVi, *dNi, *dTi, *dMat;
    float *dV, *dN, *dT;
    uint32_t *keep, *placeOf, *kv, *kn, *kt, *order, *order2, *key, *key2, *headSlot, *firstAt, *newIndexAt, *missing;
    if ((rc = b.get(&dVi, nCorners)) || (rc = b.get(&dNi, nCorners)) || (rc = b.get(&dTi, nCorners)) || (rc = b.get(&dMat, n_triangles))
        || (rc = b.get(&dV, 3 * n_vertices)) || (rc = b.get(&dN, 3 * n_normals)) || (rc = b.get(&dT, 2 * n_texcoords))
        || (rc = b.get(&keep, nCorners)) || (rc = b.get(&placeOf, nCorners)))
        return rc;
    HIP_TRY(hipMemcpy(dVi, corner_vi, size_t(nCorners) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dNi, corner_ni, size_t(nCorners) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dTi, corner_ti, size_t(nCorners) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dMat, tri_material, size_t(n_triangles) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dV, vertices, size_t(n_vertices) * 12, hipMemcpyHostToDevice));
    if (n_normals)
        HIP_TRY(hipMemcpy(dN, normals, size_t(n_normals) * 12, hipMemcpyHostToDevice));
    if (n_texcoords)
        HIP_TRY(hipMemcpy(dT, texcoords, size_t(n_texcoords) * 8, hipMemcpyHostToDevice));
    const dim3 block(kBlock), gridC((nCorners + kBlock - 1) / kBlock);

    // 1 keep
    hipLaunchKernelGGL(keepFlagsKernel, gridC, block, 0, nullptr, dMat, material, nCorners, keep);
    size_t tempBytes = 0, need = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, keep, placeOf, 0u, size_t(nCorners), rocprim::plus<uint32_t>()));
    tempBytes = std::max(tempBytes, need);
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keep, keep, keep, keep, size_t(nCorners)));
    tempBytes = std::max(tempBytes, need);
    HIP_TRY(rocprim::inclusive_scan(nullptr, need, keep, keep, size_t(nCorners), MaxOp()));
    tempBytes = std::max(tempBytes, need);
    char* temp;
    if ((rc = b.get(&temp, tempBytes)))
        return rc;
    need = tempBytes;
    HIP_TRY(rocprim::exclusive_scan(temp, need, keep, placeOf, 0u, size_t(nCorners), rocprim::plus<uint32_t>()));
    uint32_t lastKeep = 0, lastPlace = 0;
    HIP_TRY(hipMemcpy(&lastKeep, keep + nCorners - 1, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&lastPlace, placeOf + nCorners - 1, 4, hipMemcpyDeviceToHost));
    const uint32_t n = lastPlace + lastKeep; // corners kept = entries of the new index list
    if (n == 0)
        return PT_OK;
    if ((rc = b.get(&kv, n)) || (rc = b.get(&kn, n)) || (rc = b.get(&kt, n)) || (rc = b.get(&order, n)) || (rc = b.get(&order2, n))
        || (rc = b.get(&key, n)) || (rc = b.get(&key2, n)) || (rc = b.get(&headSlot, n)) || (rc = b.get(&firstAt, n))
        || (rc = b.get(&newIndexAt, n)) || (rc = b.get(&missing, 2)))
        return rc;
    hipLaunchKernelGGL(gatherKeptKernel, gridC, block, 0, nullptr, keep, placeOf, dVi, dNi, dTi, nCorners, kv, kn, kt, order);
    const dim3 gridN((n + kBlock - 1) / kBlock);

    // 2 group: least significant key first; every sort is stable, so equal triples end up in order of place
    const uint32_t* tables[3] = { kt, kn, kv };
    for (int pass = 0; pass < 3; pass++) {
        hipLaunchKernelGGL(permuteKernel, gridN, block, 0, nullptr, tables[pass], order, n, key);
        need = tempBytes;
        HIP_TRY(rocprim::radix_sort_pairs(temp, need, key, key2, order, order2, size_t(n)));
        std::swap(order, order2);
    }

    // 3 number
    HIP_TRY(hipMemsetAsync(firstAt, 0, size_t(n) * 4, nullptr));
    HIP_TRY(hipMemsetAsync(missing, 0, 8, nullptr));
    hipLaunchKernelGGL(groupHeadsKernel, gridN, block, 0, nullptr, order, kv, kn, kt, n, headSlot, firstAt, missing);
    need = tempBytes;
    HIP_TRY(rocprim::inclusive_scan(temp, need, headSlot, headSlot, size_t(n), MaxOp()));
    need = tempBytes;
    HIP_TRY(rocprim::exclusive_scan(temp, need, firstAt, newIndexAt, 0u, size_t(n), rocprim::plus<uint32_t>()));
    uint32_t lastFirst = 0, lastIndex = 0, miss[2] = { 0, 0 };
    HIP_TRY(hipMemcpy(&lastFirst, firstAt + n - 1, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&lastIndex, newIndexAt + n - 1, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(miss, missing, 8, hipMemcpyDeviceToHost));
    const uint32_t nUnique = lastIndex + lastFirst;

    // 4 emit
    uint32_t* dIdx;
    float *dPos, *dNrm, *dTc;
    if ((rc = b.get(&dIdx, n)) || (rc = b.get(&dPos, 3 * size_t(nUnique))) || (rc = b.get(&dNrm, 3 * size_t(nUnique)))
        || (rc = b.get(&dTc, 2 * size_t(nUnique))))
        return rc;
    hipLaunchKernelGGL(emitKernel, gridN, block, 0, nullptr, order, headSlot, newIndexAt, kv, kn, kt, n, dV, dN, dT, missing, dIdx,
                       dPos, dNrm, dTc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_indices, dIdx, size_t(n) * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_positions, dPos, size_t(nUnique) * 12, hipMemcpyDeviceToHost));
    if (!miss[0])
        HIP_TRY(hipMemcpy(out_normals, dNrm, size_t(nUnique) * 12, hipMemcpyDeviceToHost));
    if (!miss[1])
        HIP_TRY(hipMemcpy(out_texcoords, dTc, size_t(nUnique) * 8, hipMemcpyDeviceToHost));
    *out_n_indices = n;
    *out_n_vertices = nUnique;
    *out_have_normals = miss[0] ? 0 : 1;
    *out_have_texcoords = miss[1] ? 0 : 1;
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

} // extern "C"
