// Micro-benchmark: throughput of gathering random 64-byte rows from a table in memory,
//   (a) one lane per row: four 16-byte loads per lane (64 different rows per wave instruction),
//   (b) four lanes per row: one 16-byte load per lane (16 different rows per wave instruction).
// Build: hipcc --offload-arch=gfx950 -O3 gather_rows.hip -o gather_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) laneRows(const float4* __restrict__ t, uint32_t rows, int steps, float* out)
{
    uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t k = mix(id) % rows;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        float4 a = t[4 * k + 0], b = t[4 * k + 1], c = t[4 * k + 2], d = t[4 * k + 3];
        acc += a.x + b.y + c.z + d.w;
        k = (mix(k + id) + __float_as_uint(d.x)) % rows; // next row depends on this one
    }
    out[id] = acc;
}

__global__ void __launch_bounds__(256) quadRows(const float4* __restrict__ t, uint32_t rows, int steps, float* out)
{
    uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t ray = id >> 2, q = id & 3u;
    uint32_t k = mix(ray) % rows;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        float4 a = t[4 * k + q];
        acc += a.x + a.y + a.z + a.w;
        float dx = __shfl(a.x, (threadIdx.x & ~3u) | 3u); // lane 3 of the quad holds the "links"
        k = (mix(k + ray) + __float_as_uint(dx)) % rows;
    }
    out[id] = acc;
}

int main(int argc, char** argv)
{
    uint32_t rows = argc > 1 ? atoi(argv[1]) : (4u << 20); // 4 M rows = 256 MB
    int steps = 64;
    float4* t; float* out;
    hipMalloc(&t, size_t(rows) * 64);
    hipMemset(t, 0, size_t(rows) * 64);
    uint32_t threads = 1u << 22;
    hipMalloc(&out, threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(laneRows, dim3(threads / 256), dim3(256), 0, 0, t, rows, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("lane-per-row : %8.3f ms  %7.1f G rows/s\n", ms, double(threads) * steps / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(quadRows, dim3(threads / 256), dim3(256), 0, 0, t, rows, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("quad-per-row : %8.3f ms  %7.1f G rows/s\n", ms, double(threads / 4) * steps / ms / 1e6);
    }
    return 0;
}
