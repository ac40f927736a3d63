// Micro-benchmark: what a vector instruction costs when only some lanes of the wave are active -- does the SIMD
// skip the quarters (16 lanes) of a wave in which no lane is active?  Every wave runs the same chain of dependent
// multiply-adds (8 independent chains per lane) under an execution mask with `active` lanes, either packed into the
// lowest quarters or spread evenly over all four; 8 waves per SIMD, every SIMD of the chip busy.
// Prints the time per wave-instruction in SIMD cycles (4 = full price).
// Build (the flags of the library: no packed-f32 vectorisation): hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize lane_mask.hip -o lane_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) chain(float* out, int steps, int active, int spread)
{
    const int lane = threadIdx.x & 63;
    // spread: lanes 0, s, 2s, ... (s = 64 / active) are active, so every quarter has some; packed: lanes [0, active)
    const bool on = spread ? (lane % (64 / active)) == 0 : lane < active;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0000001f, c = 1e-7f;
    if (on) {
        for (int s = 0; s < steps; s++) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 8; // 8 blocks of 4 waves per compute unit: 8 waves per SIMD
    const int steps = 20000;
    float* out;
    hipMalloc(&out, size_t(blocks) * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double clockGHz = prop.clockRate * 1e-6;
    printf("%d compute units, clock %.2f GHz\n", cus, clockGHz);
    for (int spread = 0; spread < 2; spread++)
        for (int active = 64; active >= 1; active /= 2) {
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, 100, active, spread);
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, steps, active, spread);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per SIMD: 8 waves x steps x 8 fmas (+ loop overhead)
            const double perSimd = 8.0 * steps * 8.0;
            printf("%s %2d lanes active: %.3f ms, %.2f cycles per wave-instruction\n", spread ? "spread" : "packed", active, ms,
                   ms * 1e-3 * clockGHz * 1e9 / perSimd);
        }
    return 0;
}
