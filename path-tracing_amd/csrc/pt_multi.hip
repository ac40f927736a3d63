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
//            is the single-GPU image bit for bit.  Combine: every tile is copied from the GPU
//            that rendered it into the first handle's image (peer copies over xGMI).
//   samples  every handle renders its contiguous range of the sample indices as plain sums; the
//            first handle's GPU adds the partial images in handle order (a fixed order, so the
//            result does not depend on timing) and divides by the sample count.  The sum order
//            differs from a sequential render, so the image equals it to rounding.
//
// The plan itself (which call renders what) is pt_multi_plan: pure host code, testable without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "pt_scene_host.hpp"

using ptint::fail;

namespace {

constexpr int kTile = 256;

__global__ void __launch_bounds__(256) addImageKernel(float* __restrict__ acc, const float* __restrict__ part, size_t n,
                                                      int first, float divisor)
{
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= n)
        return;
    float v = first ? part[i] : acc[i] + part[i];
    if (divisor != 0.0f)
        v = v / divisor;
    acc[i] = v;
}

struct Share {
    pt_scene* scene = nullptr;
    std::vector<pt_render_params> calls;
    float* image = nullptr; // full-size image on the share's GPU
    int rc = PT_OK;
    std::string error;
};

} // namespace

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

int pt_render_multi(pt_scene* const* scenes, int n_scenes, const pt_camera* camera, const pt_render_params* params,
                    int mode, float* out_rgb)
try {
    if (!scenes || n_scenes <= 0 || !camera || !params || !out_rgb)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: null argument");
    for (int i = 0; i < n_scenes; i++)
        if (!scenes[i])
            return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: null scene handle");
    if (n_scenes == 1)
        return pt_render(scenes[0], camera, params, out_rgb);
    if (mode == PT_SHARD_SAMPLES && (params->flags & PT_RENDER_SUM_ONLY) == 0 && params->sample_begin == 0
        && params->sample_end == 0 && params->sqrt_spp * params->sqrt_spp < n_scenes) {
        // fewer samples than handles: nothing to split
        return pt_render(scenes[0], camera, params, out_rgb);
    }
    const size_t nFloats = size_t(3) * size_t(params->width) * size_t(params->height);
    std::vector<Share> shares(n_scenes);
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
    // one host thread per handle: allocate the share's image on its GPU, render the share's calls, wait
    auto work = [&](Share& s) {
        DeviceGuard g(s.scene->device);
        if (!g.ok) {
            s.rc = PT_ERR_DEVICE;
            s.error = "cannot select GPU " + std::to_string(s.scene->device);
            return;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s.image), nFloats * sizeof(float));
        if (e == hipSuccess && &s == &shares[0] && mode == PT_SHARD_TILES)
            e = hipMemcpy(s.image, out_rgb, nFloats * sizeof(float), hipMemcpyHostToDevice); // pixels outside stay the caller's
        if (e != hipSuccess) {
            s.rc = PT_ERR_DEVICE;
            s.error = std::string("pt_render_multi: ") + hipGetErrorString(e);
            return;
        }
        for (const pt_render_params& p : s.calls) {
            s.rc = pt_render_device(s.scene, camera, &p, s.image, nullptr);
            if (s.rc != PT_OK) {
                s.error = pt_last_error(); // (the message is per thread: carry it to the caller's)
                return;
            }
        }
        e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            s.rc = PT_ERR_DEVICE;
            s.error = std::string("pt_render_multi: ") + hipGetErrorString(e);
        }
    };
    std::vector<std::thread> threads;
    for (int i = 1; i < n_scenes; i++)
        threads.emplace_back(work, std::ref(shares[i]));
    work(shares[0]);
    for (std::thread& t : threads)
        t.join();
    int rc = PT_OK;
    std::string error;
    for (Share& s : shares)
        if (s.rc != PT_OK && rc == PT_OK) {
            rc = s.rc;
            error = s.error;
        }
    // combine on the first handle's GPU
    if (rc == PT_OK) {
        Share& root = shares[0];
        DeviceGuard g(root.scene->device);
        hipError_t e = hipSuccess;
        float* staging = nullptr; // a peer's partial image, copied over before it is added
        if (mode == PT_SHARD_TILES) {
            for (int i = 1; i < n_scenes && e == hipSuccess; i++) {
                for (const pt_render_params& p : shares[i].calls) {
                    const size_t pitch = size_t(params->width) * 12;
                    const size_t off = (size_t(p.tile_y0) * params->width + p.tile_x0) * 3;
                    e = hipMemcpy2D(root.image + off, pitch, shares[i].image + off, pitch, size_t(p.tile_x1 - p.tile_x0) * 12,
                                    size_t(p.tile_y1 - p.tile_y0), hipMemcpyDeviceToDevice);
                    if (e != hipSuccess)
                        break;
                }
            }
        } else {
            int s0 = params->sample_begin, s1 = params->sample_end;
            if (s0 == 0 && s1 == 0)
                s1 = params->sqrt_spp * params->sqrt_spp;
            const bool sumOnly = (params->flags & PT_RENDER_SUM_ONLY) != 0;
            // the library's mean is over all sqrt_spp^2 strata, as in a single render of the same parameters
            const float divisor = sumOnly ? 0.0f : float(params->sqrt_spp * params->sqrt_spp);
            e = hipMalloc(reinterpret_cast<void**>(&staging), nFloats * sizeof(float));
            int last = 0;
            for (int i = 0; i < n_scenes; i++)
                if (!shares[i].calls.empty())
                    last = i;
            bool first = true;
            for (int i = 0; i < n_scenes && e == hipSuccess; i++) {
                if (shares[i].calls.empty())
                    continue;
                const float* part = shares[i].image;
                if (i > 0) {
                    e = hipMemcpyPeer(staging, root.scene->device, shares[i].image, shares[i].scene->device,
                                      nFloats * sizeof(float));
                    part = staging;
                }
                if (e != hipSuccess)
                    break;
                // only the pixels of the rectangle carry sums; the rest of every partial image is untouched memory
                // of its own allocation, so the rectangle is added row by row below
                const pt_render_params& p = shares[i].calls[0];
                for (int y = p.tile_y0; y < p.tile_y1; y++) {
                    const size_t off = (size_t(y) * params->width + p.tile_x0) * 3;
                    const size_t n = size_t(p.tile_x1 - p.tile_x0) * 3;
                    hipLaunchKernelGGL(addImageKernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, nullptr, root.image + off,
                                       part + off, n, first ? 1 : 0, i == last ? divisor : 0.0f);
                }
                e = hipDeviceSynchronize();
                first = false;
            }
        }
        if (e == hipSuccess) {
            if (mode == PT_SHARD_TILES) {
                e = hipMemcpy(out_rgb, root.image, nFloats * sizeof(float), hipMemcpyDeviceToHost);
            } else {
                // pixels outside the rectangle keep the caller's values
                const pt_render_params& p = shares[0].calls.empty() ? *params : shares[0].calls[0];
                int x0 = p.tile_x0, y0 = p.tile_y0, x1 = p.tile_x1, y1 = p.tile_y1;
                const size_t pitch = size_t(params->width) * 12;
                const size_t off = (size_t(y0) * params->width + x0) * 3;
                if (x1 > x0 && y1 > y0)
                    e = hipMemcpy2D(out_rgb + off, pitch, root.image + off, pitch, size_t(x1 - x0) * 12, size_t(y1 - y0),
                                    hipMemcpyDeviceToHost);
            }
        }
        if (staging)
            (void)hipFree(staging);
        if (e != hipSuccess) {
            rc = PT_ERR_DEVICE;
            error = std::string("pt_render_multi: ") + hipGetErrorString(e);
        }
    }
    for (Share& s : shares) {
        if (s.image) {
            DeviceGuard g(s.scene->device);
            (void)hipFree(s.image);
        }
    }
    if (rc != PT_OK)
        return fail(rc, error);
    return PT_OK;
} catch (...) {
    return ptint::caught();
}

} // extern "C"
