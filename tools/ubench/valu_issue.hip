// Microbenchmark: VALU issue rate of v_fmac_f32 with an SGPR multiplier vs a VGPR multiplier,
// at 1/2/4 waves per SIMD (256 blocks, one per CU).   hipcc -O3 --offload-arch=gfx950 valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <bool SGPR>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ s, float* out, int iters) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x * 0.001f + i;
    float m[16];
    if (SGPR) {
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = s[i];              // uniform -> SGPR
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = s[i] + threadIdx.x * 1e-9f;   // per-lane -> VGPR
    }
    float x = threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = fmaf(m[i & 15], x, acc[i]);
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = fmaf(m[(i + 3) & 15], x, acc[i]);
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float *s, *out;
    hipMalloc(&s, 64 * sizeof(float));
    hipMemset(s, 0, 64 * sizeof(float));
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode) hipLaunchKernelGGL(k<true>, dim3(256), dim3(threads), 0, 0, s, out, iters);
                else hipLaunchKernelGGL(k<false>, dim3(256), dim3(threads), 0, 0, s, out, iters);
                hipEventRecord(b);
                hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            double instr_per_simd = (double)iters * 64 * (threads / 64) / 4;      // wave-instructions per SIMD
            printf("waves/SIMD=%d %s: %.3f ms -> %.2f cycles/instr/SIMD @2.4GHz, %.1f TFLOP/s\n", threads / 256, mode ? "SGPR" : "VGPR",
                   ms, ms * 1e-3 * 2.4e9 / instr_per_simd, (double)iters * 64 * 2 * 256 * threads / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
