// pt_ingest.hip -- the two image passes of the reference's importer, on the device.
//
// importTexture (reference src/15/import.hpp:70-98) turns every texture file into float texels:
// TextureImage's constructor (texture_image.hpp:17-33) has stb_image convert the 8-bit samples
// with (float) pow(sample / 255.0f, gamma) (gamma 2.2, or 1 for data textures), bottom row first;
// a bump map then goes through bumpMapToNormalMap (import.hpp:47-67), four bilinear look-ups and a
// normalised cross product per texel.  Both are per-texel loops over images of millions of
// texels.  Here the file's bytes are uploaded as they are (a quarter of the floats), converted
// through a 256-entry table made with the host's pow -- the very values stb_image computes -- and
// the normal map is derived where the texels already are.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "pt_scene_host.hpp"

using namespace ptd;
using ptint::fail;

namespace {

__global__ void __launch_bounds__(256) bytesToTexelsKernel(const unsigned char* __restrict__ in, int width, int height, int flip,
                                                           const float* __restrict__ table, float* __restrict__ out)
{
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x; // one colour sample
    const size_t n = size_t(width) * size_t(height) * 3u;
    if (i >= n)
        return;
    const size_t rowLen = size_t(width) * 3u;
    const size_t y = i / rowLen, rest = i - y * rowLen;
    const size_t src = (flip ? size_t(height) - 1u - y : y) * rowLen + rest;
    out[i] = table[in[src]];
}

// TextureImage::value (reference src/15/texture_image.hpp:45-71), first component
PT_DEV float heightAt(const float* __restrict__ texels, int width, int height, float tu, float tv)
{
    const float u = fractf(tu), v = fractf(tv);
    float s = u * width - 0.5f;
    if (s < 0.0f)
        s = 0.0f;
    float t = v * height - 0.5f;
    if (t < 0.0f)
        t = 0.0f;
    const int x0 = int(s), y0 = int(t);
    int x1 = x0 + 1;
    if (x1 >= width)
        x1 = width - 1;
    int y1 = y0 + 1;
    if (y1 >= height)
        y1 = height - 1;
    const float alpha = s - x0, beta = t - y0;
    const float a = mixf(texels[3 * (size_t(y0) * width + x0)], texels[3 * (size_t(y0) * width + x1)], alpha);
    const float b = mixf(texels[3 * (size_t(y1) * width + x0)], texels[3 * (size_t(y1) * width + x1)], alpha);
    return mixf(a, b, beta);
}

__global__ void __launch_bounds__(256) bumpToNormalKernel(const float* __restrict__ bump, int width, int height, float bumpFactor,
                                                          float* __restrict__ out)
{
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= size_t(width) * size_t(height))
        return;
    const int y = int(i / size_t(width)), x = int(i - size_t(y) * size_t(width));
    const float texelW = 1.0f / width, texelH = 1.0f / height;
    const float v = (y + 0.5f) / height, u = (x + 0.5f) / width;
    // (a coordinate of exactly 1 wraps to 0 in the look-up, as in the reference)
    const float heightT = heightAt(bump, width, height, u, fminf(1.0f, v + texelH));
    const float heightB = heightAt(bump, width, height, u, fmaxf(0.0f, v - texelH));
    const float heightL = heightAt(bump, width, height, fmaxf(0.0f, u - texelW), v);
    const float heightR = heightAt(bump, width, height, fminf(1.0f, u + texelW), v);
    const V3 tx = v3(2.0f, 0.0f, bumpFactor * (heightR - heightL));
    const V3 ty = v3(0.0f, 2.0f, bumpFactor * (heightT - heightB));
    const V3 n = normalize(cross(tx, ty));
    const V3 c = 0.5f * (n + v3(1.0f));
    out[3 * i + 0] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
}

struct DeviceBuffers {
    std::vector<void*> held;
    ~DeviceBuffers()
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

int checkImage(int device, const void* in, int width, int height, const void* out)
{
    if (!in || !out)
        return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (width <= 0 || height <= 0 || uint64_t(width) * uint64_t(height) > (1ull << 30))
        return fail(PT_ERR_INVALID_ARGUMENT, "image size out of range");
    if (device < 0 || device >= pt_device_count())
        return fail(PT_ERR_DEVICE, "no such GPU: " + std::to_string(device));
    return PT_OK;
}

} // namespace

extern "C" {

int pt_texture_from_bytes_device(int device, const unsigned char* d_rgb8, int width, int height, int linearize, int flip_rows,
                                 float* d_out_texels, void* hip_stream)
{
    int rc = checkImage(device, d_rgb8, width, height, d_out_texels);
    if (rc != PT_OK)
        return rc;
    DeviceGuard g(device);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    // stbi__ldr_to_hdr: (float) (pow(sample / 255.0f, gamma) * scale), gamma as set by
    // stbi_ldr_to_hdr_gamma (reference src/15/texture_image.hpp:19), scale 1
    const float gamma = linearize ? 2.2f : 1.0f, scale = 1.0f;
    float table[256];
    for (int i = 0; i < 256; i++)
        table[i] = float(std::pow(i / 255.0f, gamma) * scale);
    DeviceBuffers tmp;
    float* dTable = nullptr;
    if ((rc = tmp.get(&dTable, 256)) != PT_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(dTable, table, sizeof(table), hipMemcpyHostToDevice, stream));
    const size_t n = size_t(width) * size_t(height) * 3u;
    hipLaunchKernelGGL(bytesToTexelsKernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, stream, d_rgb8, width, height,
                       flip_rows, dTable, d_out_texels);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream)); // the table is freed on return
    return PT_OK;
}

int pt_texture_from_bytes(int device, const unsigned char* rgb8, int width, int height, int linearize, int flip_rows,
                          float* out_texels)
{
    int rc = checkImage(device, rgb8, width, height, out_texels);
    if (rc != PT_OK)
        return rc;
    DeviceGuard g(device);
    const size_t n = size_t(width) * size_t(height) * 3u;
    DeviceBuffers tmp;
    unsigned char* dIn = nullptr;
    float* dOut = nullptr;
    if ((rc = tmp.get(&dIn, n)) != PT_OK || (rc = tmp.get(&dOut, n)) != PT_OK)
        return rc;
    HIP_TRY(hipMemcpy(dIn, rgb8, n, hipMemcpyHostToDevice));
    if ((rc = pt_texture_from_bytes_device(device, dIn, width, height, linearize, flip_rows, dOut, nullptr)) != PT_OK)
        return rc;
    HIP_TRY(hipMemcpy(out_texels, dOut, n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_bump_to_normal_map_device(int device, const float* d_bump_texels, int width, int height, float bump_factor,
                                 float* d_out_texels, void* hip_stream)
{
    int rc = checkImage(device, d_bump_texels, width, height, d_out_texels);
    if (rc != PT_OK)
        return rc;
    if (d_bump_texels == d_out_texels)
        return fail(PT_ERR_INVALID_ARGUMENT, "the normal map cannot be written over the bump map");
    DeviceGuard g(device);
    const size_t n = size_t(width) * size_t(height);
    hipLaunchKernelGGL(bumpToNormalKernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       d_bump_texels, width, height, bump_factor, d_out_texels);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_bump_to_normal_map(int device, const float* bump_texels, int width, int height, float bump_factor, float* out_texels)
{
    int rc = checkImage(device, bump_texels, width, height, out_texels);
    if (rc != PT_OK)
        return rc;
    DeviceGuard g(device);
    const size_t n = size_t(width) * size_t(height) * 3u;
    DeviceBuffers tmp;
    float *dIn = nullptr, *dOut = nullptr;
    if ((rc = tmp.get(&dIn, n)) != PT_OK || (rc = tmp.get(&dOut, n)) != PT_OK)
        return rc;
    HIP_TRY(hipMemcpy(dIn, bump_texels, n * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = pt_bump_to_normal_map_device(device, dIn, width, height, bump_factor, dOut, nullptr)) != PT_OK)
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_texels, dOut, n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

} // extern "C"
