// pt_multi.hip -- one image rendered by several GPUs of a node behind the C ABI.
//
// The reference shards a render over operating-system processes and files: image tiles written
// as raw float rows and stitched by a second program (reference src/07/pathtracer-tiles.cpp:72-139,
// tile-composer.cpp:10-37); in its animation program the unit is a whole frame
// (src/07/pathtracer-frames.cpp:205-236).  Here a caller holds one scene handle per GPU (the
// scene is replicated) and pt_render_multi runs one host thread per handle:
//
//   tiles    the tiles of the image go round robin to the handles (tile t of the row-major tile
//            grid to handle t % n).  A pixel's value does not depend on who renders it -- the
//            generator is keyed by the pixel's index in the whole image -- so the combined image
//            is the single-GPU image bit for bit.
//   samples  every handle renders its contiguous range of the sample indices as plain sums; the
//            first handle's GPU adds the partial images in handle order (a fixed order, so the
//            result does not depend on timing) and divides by the sample count.  The sum order
//            differs from a sequential render, so the image equals it to rounding.
//
// The first handle renders straight into the result (a buffer on its GPU); every other handle
// renders into an image of its own GPU that is kept with the handle from call to call.  Combine,
// per peer and all on the first handle's GPU: the rows of the rectangle come over with ONE
// asynchronous peer copy on a stream of that peer's (behind an event the peer recorded after its
// last kernel, so the copies of all peers run side by side, each on its own xGMI link), into a
// staging buffer kept with the first handle; then ONE kernel on the caller's stream picks the
// peer's tiles out of it, or adds its sums, in handle order.  Nothing is allocated once the
// buffers have grown to the image size.  Where the GPUs can reach each other's memory, peer
// access is switched on so that the copy goes over the link; where they cannot, the same call
// stages through host memory by itself.
//
// The plan itself (which call renders what) is pt_multi_plan: pure host code, testable without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pt_scene_host.hpp"

using ptint::fail;

namespace {

constexpr int kTile = 256;

struct Rect {
    int x0, y0, x1, y1;
};

// One pixel of the rectangle per thread.  `part` holds whole image rows from row rect.y0 on.
// Tiles: the pixels of the tiles that share `share` of `nShares` rendered are copied.
__global__ void __launch_bounds__(256) gatherTilesKernel(float* __restrict__ out, const float* __restrict__ part, int width, Rect r,
                                                         int tilesPerRow, int nShares, int share)
{
    const int rw = r.x1 - r.x0;
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= size_t(rw) * size_t(r.y1 - r.y0))
        return;
    const int dx = int(i % size_t(rw)), dy = int(i / size_t(rw));
    const int tile = (dy / kTile) * tilesPerRow + dx / kTile;
    if (tile % nShares != share)
        return;
    const size_t to = (size_t(r.y0 + dy) * width + r.x0 + dx) * 3, from = (size_t(dy) * width + r.x0 + dx) * 3;
    out[to + 0] = part[from + 0];
    out[to + 1] = part[from + 1];
    out[to + 2] = part[from + 2];
}

// Samples: out = (first ? part : out + part) [/ divisor] on the rectangle.  `part` as above, or the result itself
// (partRowsFromRect = 0: indexed like `out`).
__global__ void __launch_bounds__(256) addRectKernel(float* __restrict__ out, const float* __restrict__ part, int width, Rect r,
                                                     int partRowsFromRect, int first, float divisor)
{
    const int rw = r.x1 - r.x0;
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= size_t(rw) * size_t(r.y1 - r.y0))
        return;
    const int dx = int(i % size_t(rw)), dy = int(i / size_t(rw));
    const size_t to = (size_t(r.y0 + dy) * width + r.x0 + dx) * 3;
    const size_t from = partRowsFromRect ? (size_t(dy) * width + r.x0 + dx) * 3 : to;
    for (int c = 0; c < 3; c++) {
        float v = first ? part[from + c] : out[to + c] + part[from + c];
        if (divisor != 0.0f)
            v = v / divisor;
        out[to + c] = v;
    }
}

struct Share {
    pt_scene* scene = nullptr;
    std::vector<pt_render_params> calls;
    float* image = nullptr; // where the share renders: the result (first handle) or the handle's own image
    int rc = PT_OK;
    std::string error;
};

int grow(float** buffer, size_t* capacity, size_t floats)
{
    if (*capacity >= floats)
        return PT_OK;
    if (*buffer)
        (void)hipFree(*buffer);
    *buffer = nullptr;
    *capacity = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(buffer), floats * sizeof(float)));
    *capacity = floats;
    return PT_OK;
}

// joins whatever threads have been started, also when starting one of them throws
struct Joiner {
    std::vector<std::thread>& threads;
    ~Joiner()
    {
        for (std::thread& t : threads)
            if (t.joinable())
                t.join();
    }
};

int renderMulti(pt_scene* const* scenes, int n_scenes, const pt_camera* camera, const pt_render_params* params, int mode,
                float* d_out, hipStream_t stream)
{
    pt_scene* rootScene = scenes[0];
    MultiState& rootState = rootScene->multi;
    std::lock_guard<std::mutex> oneCall(rootState.mu);
    const int width = params->width;
    std::vector<Share> shares{ size_t(n_scenes) };
    for (int i = 0; i < n_scenes; i++) {
        shares[i].scene = scenes[i];
        int count = 0;
        int rc = pt_multi_plan(params, mode, n_scenes, i, nullptr, 0, &count);
        if (rc != PT_OK)
            return rc;
        shares[i].calls.resize(size_t(count));
        if (count > 0) {
            rc = pt_multi_plan(params, mode, n_scenes, i, shares[i].calls.data(), count, &count);
            if (rc != PT_OK)
                return rc;
        }
    }
    Rect rect = { params->tile_x0, params->tile_y0, params->tile_x1, params->tile_y1 };
    if (rect.x0 == 0 && rect.y0 == 0 && rect.x1 == 0 && rect.y1 == 0) {
        rect.x1 = params->width;
        rect.y1 = params->height;
    }
    if (rect.x1 <= rect.x0 || rect.y1 <= rect.y0)
        return PT_OK;
    const size_t nFloats = size_t(3) * size_t(params->width) * size_t(params->height);
    const size_t spanFloats = size_t(3) * size_t(width) * size_t(rect.y1 - rect.y0); // the rows of the rectangle
    const size_t spanOffset = size_t(3) * size_t(width) * size_t(rect.y0);

    // one host thread per handle (a render call reads counters back while it enqueues): the first handle renders
    // into the result, the others into their own images, and every peer records an event behind its last kernel
    auto work = [&](Share& s, bool isRoot) {
        try {
            DeviceGuard g(s.scene->device);
            if (!g.ok) {
                s.rc = PT_ERR_DEVICE;
                s.error = "cannot select GPU " + std::to_string(s.scene->device);
                return;
            }
            MultiState& m = s.scene->multi;
            if (isRoot) {
                s.image = d_out;
            } else {
                if (s.calls.empty())
                    return;
                s.rc = grow(&m.image, &m.imageCapacity, nFloats);
                if (s.rc == PT_OK && !m.rendered && hipEventCreateWithFlags(&m.rendered, hipEventDisableTiming) != hipSuccess)
                    s.rc = fail(PT_ERR_DEVICE, "pt_render_multi: cannot create an event");
                if (s.rc != PT_OK) {
                    s.error = pt_last_error(); // (the message is per thread: carry it to the caller's)
                    return;
                }
                s.image = m.image;
            }
            hipStream_t st = isRoot ? stream : nullptr;
            for (const pt_render_params& p : s.calls) {
                s.rc = pt_render_device(s.scene, camera, &p, s.image, st);
                if (s.rc != PT_OK) {
                    s.error = pt_last_error();
                    return;
                }
            }
            if (!isRoot && hipEventRecord(m.rendered, st) != hipSuccess) {
                s.rc = PT_ERR_DEVICE;
                s.error = "pt_render_multi: cannot record an event";
            }
        } catch (const std::exception& e) {
            s.rc = PT_ERR_INVALID_ARGUMENT;
            s.error = std::string("pt_render_multi: ") + e.what();
        } catch (...) {
            s.rc = PT_ERR_INVALID_ARGUMENT;
            s.error = "pt_render_multi: unknown exception in a render thread";
        }
    };
    {
        std::vector<std::thread> threads;
        Joiner joiner{ threads };
        threads.reserve(size_t(n_scenes));
        for (int i = 1; i < n_scenes; i++)
            threads.emplace_back(work, std::ref(shares[i]), false);
        work(shares[0], true);
    }
    for (Share& s : shares)
        if (s.rc != PT_OK)
            return fail(s.rc, s.error);

    // combine on the first handle's GPU
    DeviceGuard g(rootScene->device);
    if (!g.ok)
        return fail(PT_ERR_DEVICE, "cannot select GPU " + std::to_string(rootScene->device));
    rootState.peers.resize(size_t(n_scenes));
    rootState.lastKernels = rootState.lastCopies = rootState.lastDirectCopies = 0;
    if (!rootState.combineStart) {
        HIP_TRY(hipEventCreate(&rootState.combineStart));
        HIP_TRY(hipEventCreate(&rootState.combineStop));
    }
    HIP_TRY(hipEventRecord(rootState.combineStart, stream));
    // the copies of all peers first (each on its own stream), then the kernels in handle order
    for (int i = 1; i < n_scenes; i++) {
        if (shares[i].calls.empty())
            continue;
        MultiPeer& peer = rootState.peers[size_t(i)];
        int rc = grow(&peer.staging, &peer.stagingCapacity, spanFloats);
        if (rc != PT_OK)
            return rc;
        if (!peer.stream) {
            HIP_TRY(hipStreamCreateWithFlags(&peer.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&peer.copied, hipEventDisableTiming));
        }
        const int from = shares[i].scene->device, to = rootScene->device;
        bool direct = from == to;
        if (!direct) {
            // over the link if this GPU can reach the peer's memory; else the copy call stages through the host
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, to, from) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(from, 0);
                direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                (void)hipGetLastError(); // ("already enabled" is not an error of this call)
            }
        }
        HIP_TRY(hipStreamWaitEvent(peer.stream, shares[i].scene->multi.rendered, 0));
        HIP_TRY(hipMemcpyPeerAsync(peer.staging, to, shares[i].image + spanOffset, from, spanFloats * sizeof(float), peer.stream));
        HIP_TRY(hipEventRecord(peer.copied, peer.stream));
        rootState.lastCopies++;
        rootState.lastDirectCopies += direct ? 1 : 0;
    }
    const size_t rectPixels = size_t(rect.x1 - rect.x0) * size_t(rect.y1 - rect.y0);
    const dim3 grid(uint32_t((rectPixels + 255) / 256));
    if (mode == PT_SHARD_TILES) {
        const int tilesPerRow = (rect.x1 - rect.x0 + kTile - 1) / kTile;
        for (int i = 1; i < n_scenes; i++) {
            if (shares[i].calls.empty())
                continue;
            HIP_TRY(hipStreamWaitEvent(stream, rootState.peers[size_t(i)].copied, 0));
            hipLaunchKernelGGL(gatherTilesKernel, grid, dim3(256), 0, stream, d_out, rootState.peers[size_t(i)].staging, width, rect,
                               tilesPerRow, n_scenes, i);
            rootState.lastKernels++;
        }
    } else {
        const bool sumOnly = (params->flags & PT_RENDER_SUM_ONLY) != 0;
        // the library's mean is over all sqrt_spp^2 strata, as in a single render of the same parameters
        const float divisor = sumOnly ? 0.0f : float(params->sqrt_spp * params->sqrt_spp);
        int last = -1;
        for (int i = 0; i < n_scenes; i++)
            if (!shares[i].calls.empty())
                last = i;
        bool first = true;
        for (int i = 0; i < n_scenes; i++) {
            if (shares[i].calls.empty())
                continue;
            const float lastDivisor = i == last ? divisor : 0.0f;
            if (i == 0) {
                // the first handle's sums are in the result already
                if (lastDivisor != 0.0f) {
                    hipLaunchKernelGGL(addRectKernel, grid, dim3(256), 0, stream, d_out, d_out, width, rect, 0, 1, lastDivisor);
                    rootState.lastKernels++;
                }
            } else {
                HIP_TRY(hipStreamWaitEvent(stream, rootState.peers[size_t(i)].copied, 0));
                hipLaunchKernelGGL(addRectKernel, grid, dim3(256), 0, stream, d_out, rootState.peers[size_t(i)].staging, width, rect, 1,
                                   first ? 1 : 0, lastDivisor);
                rootState.lastKernels++;
            }
            first = false;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(rootState.combineStop, stream));
    rootState.timed = true;
    return PT_OK;
}

} // namespace

void ptMultiRelease(pt_scene* scene)
{
    MultiState& m = scene->multi;
    if (m.image)
        (void)hipFree(m.image);
    if (m.result)
        (void)hipFree(m.result);
    if (m.rendered)
        (void)hipEventDestroy(m.rendered);
    if (m.combineStart)
        (void)hipEventDestroy(m.combineStart);
    if (m.combineStop)
        (void)hipEventDestroy(m.combineStop);
    for (MultiPeer& p : m.peers) {
        if (p.staging)
            (void)hipFree(p.staging);
        if (p.stream)
            (void)hipStreamDestroy(p.stream);
        if (p.copied)
            (void)hipEventDestroy(p.copied);
    }
    m.peers.clear();
    m.image = m.result = nullptr;
    m.imageCapacity = m.resultCapacity = 0;
    m.rendered = m.combineStart = m.combineStop = nullptr;
}

extern "C" {

int pt_multi_plan(const pt_render_params* params, int mode, int n_shares, int share, pt_render_params* out_calls,
                  int capacity, int* out_count)
{
    if (!params || !out_count || (capacity > 0 && !out_calls))
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_multi_plan: null argument");
    if (n_shares <= 0 || share < 0 || share >= n_shares)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_multi_plan: share outside [0, n_shares)");
    if (params->width <= 0 || params->height <= 0 || params->sqrt_spp <= 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "width, height and sqrt_spp must be positive");
    int x0 = params->tile_x0, y0 = params->tile_y0, x1 = params->tile_x1, y1 = params->tile_y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) {
        x1 = params->width;
        y1 = params->height;
    }
    int s0 = params->sample_begin, s1 = params->sample_end;
    if (s0 == 0 && s1 == 0)
        s1 = params->sqrt_spp * params->sqrt_spp;
    if (x0 < 0 || y0 < 0 || x1 > params->width || y1 > params->height || x0 > x1 || y0 > y1 || s0 < 0 || s0 > s1)
        return fail(PT_ERR_INVALID_ARGUMENT, "tile rectangle or sample range outside the image");
    int count = 0;
    auto emit = [&](const pt_render_params& p) {
        if (count < capacity)
            out_calls[count] = p;
        count++;
    };
    if (mode == PT_SHARD_TILES) {
        // tiles of the rectangle in row-major order, tile t to share t % n_shares
        int t = 0;
        for (int ty = y0; ty < y1; ty += kTile) {
            for (int tx = x0; tx < x1; tx += kTile, t++) {
                if (t % n_shares != share)
                    continue;
                pt_render_params p = *params;
                p.tile_x0 = tx;
                p.tile_y0 = ty;
                p.tile_x1 = std::min(tx + kTile, x1);
                p.tile_y1 = std::min(ty + kTile, y1);
                emit(p);
            }
        }
    } else if (mode == PT_SHARD_SAMPLES) {
        // contiguous ranges whose sizes differ by at most one
        const int n = s1 - s0, base = n / n_shares, extra = n % n_shares;
        const int begin = s0 + share * base + std::min(share, extra);
        const int end = begin + base + (share < extra ? 1 : 0);
        if (end > begin) {
            pt_render_params p = *params;
            p.tile_x0 = x0;
            p.tile_y0 = y0;
            p.tile_x1 = x1;
            p.tile_y1 = y1;
            p.sample_begin = begin;
            p.sample_end = end;
            p.flags |= PT_RENDER_SUM_ONLY;
            // (a range that starts at sample 0 and ends at 0 would read as "all samples")
            emit(p);
        }
    } else {
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_multi_plan: unknown sharding mode");
    }
    *out_count = count;
    return PT_OK;
}

int pt_render_multi_device(pt_scene* const* scenes, int n_scenes, const pt_camera* camera, const pt_render_params* params,
                           int mode, float* d_out_rgb, void* hip_stream)
try {
    if (!scenes || n_scenes <= 0 || !camera || !params || !d_out_rgb)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: null argument");
    for (int i = 0; i < n_scenes; i++)
        if (!scenes[i])
            return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: null scene handle");
    if (mode != PT_SHARD_TILES && mode != PT_SHARD_SAMPLES)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: unknown sharding mode");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (n_scenes == 1)
        return pt_render_device(scenes[0], camera, params, d_out_rgb, hip_stream);
    if (mode == PT_SHARD_SAMPLES && (params->flags & PT_RENDER_SUM_ONLY) == 0 && params->sample_begin == 0
        && params->sample_end == 0 && params->sqrt_spp * params->sqrt_spp < n_scenes) {
        // fewer samples than handles: nothing to split
        return pt_render_device(scenes[0], camera, params, d_out_rgb, hip_stream);
    }
    return renderMulti(scenes, n_scenes, camera, params, mode, d_out_rgb, stream);
} catch (...) {
    return ptint::caught();
}

int pt_render_multi(pt_scene* const* scenes, int n_scenes, const pt_camera* camera, const pt_render_params* params,
                    int mode, float* out_rgb)
try {
    if (!scenes || n_scenes <= 0 || !scenes[0] || !camera || !params || !out_rgb)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: null argument");
    if (params->width <= 0 || params->height <= 0)
        return fail(PT_ERR_INVALID_ARGUMENT, "width, height and sqrt_spp must be positive");
    if (n_scenes == 1)
        return pt_render(scenes[0], camera, params, out_rgb);
    // the result on the first handle's GPU (a buffer kept with the handle), then the rectangle to the caller's image
    pt_scene* root = scenes[0];
    DeviceGuard g(root->device);
    if (!g.ok)
        return fail(PT_ERR_DEVICE, "cannot select GPU " + std::to_string(root->device));
    const size_t nFloats = size_t(3) * size_t(params->width) * size_t(params->height);
    {
        std::lock_guard<std::mutex> lock(root->multi.mu);
        const int rc = grow(&root->multi.result, &root->multi.resultCapacity, nFloats);
        if (rc != PT_OK)
            return rc;
    }
    float* d_out = root->multi.result;
    int rc = pt_render_multi_device(scenes, n_scenes, camera, params, mode, d_out, nullptr);
    if (rc != PT_OK)
        return rc;
    int x0 = params->tile_x0, y0 = params->tile_y0, x1 = params->tile_x1, y1 = params->tile_y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) {
        x1 = params->width;
        y1 = params->height;
    }
    // (pixels outside the rectangle keep the caller's values)
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (x1 > x0 && y1 > y0) {
        const size_t pitch = size_t(params->width) * 12;
        const size_t off = (size_t(y0) * params->width + x0) * 3;
        HIP_TRY(hipMemcpy2D(out_rgb + off, pitch, d_out + off, pitch, size_t(x1 - x0) * 12, size_t(y1 - y0), hipMemcpyDeviceToHost));
    }
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

int pt_multi_last_combine(pt_scene* first_scene, float* out_ms, int* out_kernel_launches, int* out_peer_copies,
                          int* out_direct_copies)
try {
    if (!first_scene)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    MultiState& m = first_scene->multi;
    std::lock_guard<std::mutex> lock(m.mu);
    if (!m.timed)
        return fail(PT_ERR_INVALID_ARGUMENT, "no pt_render_multi call has combined an image on this handle yet");
    DeviceGuard g(first_scene->device);
    HIP_TRY(hipEventSynchronize(m.combineStop));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, m.combineStart, m.combineStop));
    if (out_ms)
        *out_ms = ms;
    if (out_kernel_launches)
        *out_kernel_launches = m.lastKernels;
    if (out_peer_copies)
        *out_peer_copies = m.lastCopies;
    if (out_direct_copies)
        *out_direct_copies = m.lastDirectCopies;
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

} // extern "C"
