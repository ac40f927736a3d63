// Micro-benchmark: what the packed FP32 instructions cost against the plain ones.  Every wave runs chains of
// dependent operations (8 independent chains per lane): v_fma_f32, v_pk_fma_f32, v_pk_mul_f32 and v_pk_add_f32 with
// whole pairs as operands, and the last two with one operand's half picked by op_sel (the form a box test needs:
// two planes against one origin); then the double-precision and wide-integer operations of the shading kernel.  8 waves per SIMD, every SIMD of the chip busy.  Prints SIMD cycles per
// wave-instruction (4 = one issue slot).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ void __launch_bounds__(256) chain(float* out, int steps)
{
    const int lane = threadIdx.x & 63;
    if (KIND == 0) {
        float a[8];
        for (int i = 0; i < 8; i++)
            a[i] = float(lane + i);
        const float m = 1.0000001f, c = 1e-7f;
        for (int s = 0; s < steps; s++)
#pragma unroll
            for (int i = 0; i < 8; i++)
                a[i] = __builtin_fmaf(a[i], m, c);
        float r = 0;
        for (int i = 0; i < 8; i++)
            r += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    } else {
        unsigned long long a[8], m, c;
        unsigned lo[8];
        for (int i = 0; i < 8; i++)
            lo[i] = lane * 2654435761u + i;
        for (int i = 0; i < 8; i++)
            a[i] = (unsigned long long)__float_as_uint(float(lane + i)) | ((unsigned long long)__float_as_uint(float(lane + i) + 0.5f) << 32);
        m = (unsigned long long)__float_as_uint(1.0000001f) | ((unsigned long long)__float_as_uint(0.9999999f) << 32);
        c = (unsigned long long)__float_as_uint(1e-7f) | ((unsigned long long)__float_as_uint(2e-7f) << 32);
        for (int s = 0; s < steps; s++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 1)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 2)
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 3)
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 4)
                    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(a[i]) : "v"(m));
                if (KIND == 5)
                    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                if (KIND == 6)
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 7)
                    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 8)
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 9)
                    asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                if (KIND == 10)
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(lo[(i + 1) & 7]));
                if (KIND == 11)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(lo[i]), "v"(lo[(i + 1) & 7]) : "vcc");
            }
        float r = 0;
        for (int i = 0; i < 8; i++)
            r += __uint_as_float((unsigned)a[i]) + __uint_as_float((unsigned)(a[i] >> 32)) + float(lo[i]);
        out[blockIdx.x * 256 + threadIdx.x] = r;
    }
}

template <int KIND>
void run(const char* name, float* out, int blocks, double clockGHz)
{
    const int steps = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double perSimd = 8.0 * steps * 8.0; // 8 waves x steps x 8 instructions
    printf("%-44s %.3f ms, %.2f cycles per wave-instruction\n", name, ms, ms * 1e-3 * clockGHz * 1e9 / perSimd);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 8;
    float* out;
    hipMalloc(&out, size_t(blocks) * 256 * sizeof(float));
    const double clockGHz = prop.clockRate * 1e-6;
    printf("%d compute units, clock %.2f GHz\n", cus, clockGHz);
    run<0>("v_fma_f32", out, blocks, clockGHz);
    run<1>("v_pk_fma_f32", out, blocks, clockGHz);
    run<2>("v_pk_mul_f32", out, blocks, clockGHz);
    run<3>("v_pk_add_f32", out, blocks, clockGHz);
    run<4>("v_pk_mul_f32, one operand's high half twice", out, blocks, clockGHz);
    run<5>("v_pk_add_f32, minus one operand's low half", out, blocks, clockGHz);
    run<6>("v_fma_f64", out, blocks, clockGHz);
    run<7>("v_mul_f64", out, blocks, clockGHz);
    run<8>("v_add_f64", out, blocks, clockGHz);
    run<9>("v_rcp_f64", out, blocks, clockGHz);
    run<10>("v_mul_hi_u32", out, blocks, clockGHz);
    run<11>("v_mad_u64_u32", out, blocks, clockGHz);
    return 0;
}
