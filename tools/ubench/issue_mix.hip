// Microbenchmark: does a SIMD issue scalar / LDS instructions of one wavefront in the shadow of another wavefront's vector
// instructions, or does every instruction of every wavefront cost the SIMD an issue slot?  Loops of 32 independent packed
// FMAs with 0 / 16 / 32 interleaved s_add / s_nop / broadcast ds_read_b32 / v_mov, at 1 / 2 / 4 wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FMA4 "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
#define X(s) s
// one group: 4 FMAs + EXTRA
#define GROUP(EXTRA) FMA4 EXTRA

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    extern __shared__ float lds[];
    f32x2 a0 = {1.f, 2.f}, a1 = a0, a2 = a0, a3 = a0;
    f32x2 x = {threadIdx.x * 1e-6f, 1.f}, y = {0.5f, 0.25f};
    int sc = 0;
    float l = 0.f;
    const int ldsaddr = 0;
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile(GROUP("") GROUP("") GROUP("") GROUP("") GROUP("") GROUP("") GROUP("") GROUP("")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        } else if (MODE == 1) {          // + 16 s_add
            asm volatile(GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n") GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n")
                         GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n") GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n")
                         GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n") GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n")
                         GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n") GROUP("s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "s"(sc) : "scc");
        } else if (MODE == 2) {          // + 32 s_add
#define S4 "s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %6, %6, 1\n"
            asm volatile(GROUP(S4) GROUP(S4) GROUP(S4) GROUP(S4) GROUP(S4) GROUP(S4) GROUP(S4) GROUP(S4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "s"(sc) : "scc");
        } else if (MODE == 3) {          // + 32 s_nop
#define N4 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
            asm volatile(GROUP(N4) GROUP(N4) GROUP(N4) GROUP(N4) GROUP(N4) GROUP(N4) GROUP(N4) GROUP(N4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        } else if (MODE == 4) {          // + 16 broadcast ds_read_b32 (waited once at the end)
#define L2 "ds_read_b32 %6, %7\n ds_read_b32 %6, %7 offset:4\n"
            asm volatile(GROUP(L2) GROUP(L2) GROUP(L2) GROUP(L2) GROUP(L2) GROUP(L2) GROUP(L2) GROUP(L2) "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "v"(l), "v"(ldsaddr));
        } else if (MODE == 5) {          // + 32 v_mov (plain vector instructions: the reference for "costs a slot")
#define V4 "v_mov_b32 %6, %6\n v_mov_b32 %6, %6\n v_mov_b32 %6, %6\n v_mov_b32 %6, %6\n"
            asm volatile(GROUP(V4) GROUP(V4) GROUP(V4) GROUP(V4) GROUP(V4) GROUP(V4) GROUP(V4) GROUP(V4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "v"(l));
        } else if (MODE == 6) {          // + 16 broadcast ds_read_b128
#define L128 "ds_read_b128 %6, %7\n ds_read_b128 %6, %7 offset:16\n"
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 l4 = {0.f, 0.f, 0.f, 0.f};
            asm volatile(GROUP(L128) GROUP(L128) GROUP(L128) GROUP(L128) GROUP(L128) GROUP(L128) GROUP(L128) GROUP(L128) "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "v"(l4), "v"(ldsaddr));
        } else if (MODE == 7) {          // + 32 s_waitcnt (already satisfied)
#define W4 "s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n"
            asm volatile(GROUP(W4) GROUP(W4) GROUP(W4) GROUP(W4) GROUP(W4) GROUP(W4) GROUP(W4) GROUP(W4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + sc + l;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 4000;
    for (int threads : {256, 512, 1024}) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 4096, 0, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // cycles per loop iteration and SIMD at a nominal 2.4 GHz
        printf("%-34s waves/SIMD=%d  %.3f ms  %.1f cycles per iteration and SIMD (32 FMAs = 128 alone)\n", name, threads / 256, ms,
               ms * 1e-3 * 2.4e9 / iters);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("32 pk_fma", out);
    run<1>("32 pk_fma + 16 s_add", out);
    run<2>("32 pk_fma + 32 s_add", out);
    run<3>("32 pk_fma + 32 s_nop", out);
    run<7>("32 pk_fma + 32 s_waitcnt", out);
    run<5>("32 pk_fma + 32 v_mov", out);
    run<4>("32 pk_fma + 16 ds_read_b32 bcast", out);
    run<6>("32 pk_fma + 16 ds_read_b128 bcast", out);
    return 0;
}
