// Micro-benchmark: what a dependent chain of small random row fetches costs on this chip, as the tracing
// kernel for scenes in memory does them (one lane = one ray, next row known only when this one has arrived).
//   rowBytes   16: one dwordx4 per step; 32: two dwordx4 from one 32-byte row; 64: four (one 64-byte row)
//   share      lanes in groups of `share` fetch the same row (1 = all different: incoherent rays)
//   waves      resident waves per SIMD (limited through LDS)
// Prints G rows/s per lane-step and the implied time for 2.27 G lane-rows (one tracing launch of the 2 M-triangle
// mesh).  Build: hipcc --offload-arch=gfx950 -O3 gather_model.hip -o gather_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int ROWBYTES>
__global__ void __launch_bounds__(256) chase(const uint4* __restrict__ t, uint32_t rows, int steps, uint32_t share, uint32_t* out)
{
    extern __shared__ char pad[];
    uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t group = id / share;
    uint32_t k = mix(group) % rows;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4* p = t + size_t(k) * (ROWBYTES / 16);
        uint4 a = p[0];
        acc += a.x ^ a.y;
        if (ROWBYTES >= 32) { uint4 b = p[1]; acc += b.z; a.w ^= b.w; }
        if (ROWBYTES >= 64) { uint4 c = p[2], d = p[3]; acc += c.x + d.y; a.w ^= c.w ^ d.w; }
        k = (mix(k + group * 7u + uint32_t(s)) + a.w) % rows; // depends on the data
    }
    out[id] = acc + (pad[0] & 0);
}

int main(int argc, char** argv)
{
    const size_t tableKB = argc > 1 ? atoi(argv[1]) : 65536; const size_t tableMB = tableKB >> 10;
    const int steps = 128;
    uint4* t; uint32_t* out;
    hipMalloc(&t, tableKB << 10);
    hipMemset(t, 0, tableKB << 10);
    const uint32_t threads = 1u << 22;
    hipMalloc(&out, threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {6}) {
        const size_t lds = (160 * 1024 / waves) - 512; // one block (4 waves, one per SIMD) per `lds` bytes: waves blocks per CU
        for (uint32_t share : {1u, 4u, 16u, 64u}) {
            for (int rb : {16, 32, 64}) {
                const uint32_t rows = uint32_t((tableKB << 10) / rb);
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    if (rb == 16) hipLaunchKernelGGL(chase<16>, dim3(threads / 256), dim3(256), lds, 0, t, rows, steps, share, out);
                    if (rb == 32) hipLaunchKernelGGL(chase<32>, dim3(threads / 256), dim3(256), lds, 0, t, rows, steps, share, out);
                    if (rb == 64) hipLaunchKernelGGL(chase<64>, dim3(threads / 256), dim3(256), lds, 0, t, rows, steps, share, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double grows = double(threads) * steps / best / 1e6;
                printf("table %6zu KB waves/SIMD %d share %2u rowBytes %2d : %8.3f ms  %7.1f G lane-rows/s  (2.27 G lane-rows: %6.2f ms)\n",
                       tableKB, waves, share, rb, best, grows, 2.27 / grows * 1e3);
            }
        }
    }
    return 0;
}
