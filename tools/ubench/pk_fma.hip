// Microbenchmark: v_pk_fma_f32 issue rate, VGPR operands vs an SGPR-pair multiplier with op_sel
// broadcast, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ s, float* out, int iters) {
    f2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f2{threadIdx.x * 0.001f + i, 1.f};
    f2 x = f2{threadIdx.x * 0.5f, 0.25f};
    f2 m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = f2{s[2 * i], s[2 * i + 1]};          // uniform -> SGPR pairs
    f2 mv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) mv[i] = m[i] + f2{threadIdx.x * 1e-9f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(mv[(i + rep) & 7]), "v"(x));
                } else if (MODE == 1) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(m[(i + rep) & 7]), "v"(x));
                } else if (MODE == 2) {   // broadcast low half of the SGPR pair, swap + negate-lo on x
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]"
                                 : "+v"(acc[i]) : "s"(m[(i + rep) & 7]), "v"(x));
                } else {                  // plain v_fma pair for reference (2 instr = same flops)
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "s"(m[(i + rep) & 7].x), "v"(x.x));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].y) : "s"(m[(i + rep) & 7].y), "v"(x.y));
                }
            }
    }
    f2 r = f2{0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
}

template <int MODE>
void run(const float* s, float* out, const char* name) {
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, s, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        double flops = (double)iters * 64 * 4 * 256 * threads;      // 64 pk_fma (or 128 fma) per iter, 4 flop per pk
        printf("%-28s waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", name, threads / 256, ms, flops / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float *s, *out;
    hipMalloc(&s, 64 * sizeof(float));
    hipMemset(s, 0, 64 * sizeof(float));
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>(s, out, "pk_fma vgpr");
    run<1>(s, out, "pk_fma sgpr-pair");
    run<2>(s, out, "pk_fma sgpr-pair op_sel/neg");
    run<3>(s, out, "2x v_fmac sgpr");
    return 0;
}
