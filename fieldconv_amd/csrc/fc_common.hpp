// Shared device helpers for the gfx950 field-convolution kernels.
//
// Layout vocabulary used across the kernels:
//   tile      16 consecutive vertices (targets in the forward pass, sources in the backward
//             pass); one 1024-thread workgroup = 16 wavefronts owns one tile at a time.
//   slab      the (ring, channel) slice of the per-vertex stencil response for ONE angular
//             frequency f, staged in LDS as two planes (re, im) of [16 vertices][KS] floats.
//   fragment  the 16x4 operand of v_mfma_f32_16x16x4_f32: lane l supplies element
//             [l & 15][l >> 4].  Four consecutive k live in one float4 per lane, so k-step
//             s of a 16-wide k block uses k = 16*blk + 4*(l>>4) + s for BOTH operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace fc {

// Development switches (the FC_* variables of fieldconv_amd/_env.py: older kernel families for A/B runs, phase skipping, stamps) exist
// only in a library compiled with -DFC_DEV_SWITCHES (fieldconv_amd.build.build_dev: libfieldconv_hip_dev.so).  The product library never
// reads the environment -- a stray variable in a C-ABI consumer's process cannot change which kernels run -- and dev_env() folds to
// "not set", so every switch takes its default at compile time.  The arithmetic mode travels in the dims of every call (fc_dims::mode).
#ifdef FC_DEV_SWITCHES
inline const char* dev_env(const char* name) { return getenv(name); }
constexpr bool kDevSwitches = true;
#else
inline const char* dev_env(const char*) { return nullptr; }
constexpr bool kDevSwitches = false;
#endif

constexpr int kWave = 64;
constexpr int kTile = 16;        // vertices per tile == MFMA N
constexpr int kWaves = 16;       // wavefronts per workgroup
constexpr int kThreads = kWave * kWaves;
constexpr float kOriginEps = 1e-7f;   // reference utils/field.py:8

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }

// First tile of a persistent workgroup.  Workgroups are dealt round-robin to the 8 XCDs; with this
// remap the 32 workgroups of one XCD walk 32 CONSECUTIVE tiles per round, so the source rows that
// neighbouring tiles of the mesh share are served by that XCD's own L2.
__device__ __forceinline__ int first_tile_of_block() {
    const int b = blockIdx.x, g = gridDim.x;
    return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3);
}

// Maximum of a NON-NEGATIVE float over the wavefront, the same value in every lane.  Six v_max with DPP operands
// (row shifts, then row broadcasts) and one v_readlane instead of six ds_bpermute round trips through the LDS
// crossbar (__shfl_xor).  Non-negative floats order like their bit patterns, and the zero fill of the shifts is
// neutral for a maximum.
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true));      // row_shr:1
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true));      // row_shr:2
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true));      // row_shr:4
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true));      // row_shr:8  -> lane 15 of each row holds the row maximum
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xF, 0xF, true));      // row_bcast:15
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xF, 0xF, true));      // row_bcast:31 -> lane 63 holds the maximum
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// Row gather with a 32-bit byte offset: base (wave-uniform, SGPR pair) + vertex * row bytes + lane part fits the
// scalar-base form of global_load.  The offset is ONE v_mad_u32_u24 instead of v_mul_lo_u32 + v_add (measured: no visible
// change -- the walks are not bound by their vector instructions, tools/ubench/walk.hip).  The callers guarantee
// N * C * 8 < 2^32 and N < 2^24 (checked on the host, rows_fit_32bit).
__device__ __forceinline__ float2 gather_row(const float2* __restrict__ base, int vertex, uint32_t row_bytes, uint32_t lane_bytes) {
    uint32_t off;        // (hipcc lowers __umul24 to v_and + v_mul_lo_u32: hence the instruction itself)
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(vertex), "s"(row_bytes), "v"(lane_bytes));
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + off);
}

// Compile-time loop: fn(std::integral_constant<int, i>) for i in [BEGIN, END).
template <int BEGIN, int END, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (BEGIN < END) {
        fn(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(fn);
    }
}

// Floats per edge record of the factored stencil (see fc_forward.hip): header of 4 + F complex phases,
// rounded up to a multiple of 4 floats so records stay 16-byte aligned.
__host__ __device__ constexpr int factored_record_floats(int B) { return round_up(4 + 2 * (2 * B + 1), 4); }

typedef __attribute__((address_space(1))) const void* gptr_t;   // global_load_lds source
typedef __attribute__((address_space(3))) void* lptr_t;         // global_load_lds destination (wave-uniform base)

// LDS-DMA of 16 bytes per lane (1 KiB per wavefront) that the compiler does not track: after the builtin
// form hipcc drains vmcnt before the next LDS access that may alias the destination, which serialises a
// prefetch with the compute it is meant to overlap.  The caller owns the ordering: s_waitcnt vmcnt before
// the destination is read, a barrier before other wavefronts read it.
__device__ __forceinline__ void lds_dma16_untracked(const void* src_lane, const void* lds_dst_uniform) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_uniform);   // LDS byte address
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src_lane) : "memory", "m0");
}

// The same for the lanes of `mask` only (a wave-uniform 64-bit lane mask): lane l < popcount writes its 16 bytes at
// lds_dst_uniform + 16 l.  Uniform control flow around the call (the full EXEC mask is restored afterwards).
__device__ __forceinline__ void lds_dma16_untracked_lanes(const void* src_lane, const void* lds_dst_uniform, unsigned long long mask) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_uniform);
    asm volatile("s_mov_b64 exec, %2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, -1"
                 : : "s"(dst), "v"(src_lane), "s"(mask) : "memory", "m0");
}

// The same with the wave-uniform part of the source address in an SGPR pair and the per-lane part as a 32-bit offset
// (global_load_lds ... v_off, s[base]): no 64-bit pointer per lane to keep alive across a loop.
__device__ __forceinline__ void lds_dma16_saddr(const void* src_uniform, uint32_t lane_off, const void* lds_dst_uniform) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_uniform);
    const uint64_t base = (uint64_t)(uintptr_t)src_uniform;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint64_t sb = ((uint64_t)hi << 32) | lo;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(lane_off), "s"(sb) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma16_saddr_lanes(const void* src_uniform, uint32_t lane_off, const void* lds_dst_uniform,
                                                      unsigned long long mask) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_uniform);
    const uint64_t base = (uint64_t)(uintptr_t)src_uniform;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint64_t sb = ((uint64_t)hi << 32) | lo;
    asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                 : : "s"(dst), "v"(lane_off), "s"(sb), "s"(mask) : "memory", "m0");
}
// 4 bytes per lane (256 bytes per wavefront)
__device__ __forceinline__ void lds_dma4_saddr(const void* src_uniform, uint32_t lane_off, const void* lds_dst_uniform) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_uniform);
    const uint64_t base = (uint64_t)(uintptr_t)src_uniform;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint64_t sb = ((uint64_t)hi << 32) | lo;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(dst), "v"(lane_off), "s"(sb) : "memory", "m0");
}

// Records per LDS-ring chunk (log2): the largest power of two whose records fit one 1 KiB global_load_lds.
__host__ __device__ constexpr int factored_log_chunk_records(int B) {
    const int per_kib = 256 / factored_record_floats(B);
    return per_kib >= 32 ? 5 : per_kib >= 16 ? 4 : per_kib >= 8 ? 3 : per_kib >= 4 ? 2 : 1;
}
constexpr int kRunStride = 8;      // ints per vertex in fc_csr::runs (ring-run offsets, n_rings <= 8)
constexpr int kRingChunks = 4;     // 1 KiB chunks per wavefront in the record ring (power of two, >= 3)



// LDS row stride (floats) for a slab with KP (multiple of 16) k-entries per vertex:
// KP + 8 keeps the 16x4 float4 fragment reads bank-conflict free (stride = 8 mod 16).
__host__ __device__ constexpr int slab_stride(int KP) { return KP + 8; }

// Reference utils/field.py:10-16: both components strictly inside (-eps, eps).
__device__ __forceinline__ bool is_origin(float2 z) {
    return (fabsf(z.x) < kOriginEps) && (fabsf(z.y) < kOriginEps);
}

// u = exp(-i*softAngle(z)) = conj(z)/|z|, or 1 at the origin box (angle defined as 0 there,
// reference utils/field.py:40-48).  No trigonometry needed.
__device__ __forceinline__ float2 unit_conj(float2 z) {
    const bool org = is_origin(z);
    const float n2 = z.x * z.x + z.y * z.y;
    const float inv = org ? 0.f : __frsqrt_rn(n2);
    return org ? make_float2(1.f, 0.f) : make_float2(z.x * inv, -z.y * inv);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// Complex products on packed fp32, two instructions each: a v_pk_mul_f32 with the real part of one
// factor broadcast, then a v_pk_fma_f32 that takes the imaginary part broadcast, the other factor
// with its halves swapped (op_sel) and one half of the product negated (neg_lo / neg_hi).  hipcc does
// not fold the per-half negation into the modifier (it materialises (-im, re) with v_xor + v_mov,
// ~20 % of the gather loop), hence the inline instruction; it has no tied operands and no side effects.
// The two steps are also available separately so that a caller with several independent products can
// issue all first steps before the second ones (back-to-back dependent packed ops cost a wait state).
__device__ __forceinline__ f32x2 cmul_pk_step1(f32x2 a, f32x2 b) { return f32x2{a.x, a.x} * b; }
__device__ __forceinline__ f32x2 cmul_pk_step2(f32x2 a, f32x2 b, f32x2 t) {
    f32x2 z;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(z) : "v"(a), "v"(b), "v"(t));
    return z;       // (t.x - a.y b.y, t.y + a.y b.x)
}
__device__ __forceinline__ f32x2 cmul_pk(f32x2 a, f32x2 b) {          // a * b
    return cmul_pk_step2(a, b, cmul_pk_step1(a, b));
}
__device__ __forceinline__ f32x2 cmul_conj_pk_step1(f32x2 a, f32x2 b) { return f32x2{b.x, b.x} * a; }
__device__ __forceinline__ f32x2 cmul_conj_pk_step2(f32x2 a, f32x2 b, f32x2 t) {
    f32x2 z;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(z) : "v"(b), "v"(a), "v"(t));
    return z;       // (t.x + b.y a.y, t.y - b.y a.x)
}
__device__ __forceinline__ f32x2 cmul_conj_pk(f32x2 a, f32x2 b) {     // a * conj(b)
    return cmul_conj_pk_step2(a, b, cmul_conj_pk_step1(a, b));
}

// All 2B+1 rotated copies xt[f] = z * u^(f-B) (reference nn/field_conv.py:128-130), u = exp(-i angle(z)).
// Slot B+1 (m = +1) is z*u = |z|, stored as (|z|, 0); inside the origin box u = 1 and every copy is z
// itself.  Valid for |z|^2 within the fp32 range (|z| < 1.8e19).
template <int B>
__device__ __forceinline__ void rotate_all(f32x2 z, f32x2 (&xt)[2 * B + 1]) {
    const bool org = (fabsf(z.x) < kOriginEps) && (fabsf(z.y) < kOriginEps);
    const f32x2 sq = z * z;
    const float n2 = sq.x + sq.y;
    const float inv = __frsqrt_rn(n2);
    const f32x2 u = org ? f32x2{1.f, 0.f} : f32x2{z.x * inv, -z.y * inv};
    xt[B] = z;
    if (B >= 1) {
        xt[B + 1] = org ? z : f32x2{n2 * inv, 0.f};
        xt[B - 1] = cmul_conj_pk(z, u);
    }
#pragma unroll
    for (int m = 2; m <= B; ++m) {
        xt[B + m] = cmul_pk(xt[B + m - 1], u);
        xt[B - m] = cmul_conj_pk(xt[B - m + 1], u);
    }
}

// Geometric-phase form of the factored stencil: FCPrecomp's phases are ph_f = c * g^(f-B) with |g| = 1
// (fSten = exp(i m theta), reference transforms/fc_precomp.py:88-95).  Then z_f = ph_f * x * u^(f-B) =
// (c x) (g u)^(f-B): one product for c x, one for v = g u, and one per further frequency -- 2 + 2B complex
// products per edge and lane instead of 4B + 1, and no special m = +1 copy.  Inside the origin box u = 1.
template <int B>
__device__ __forceinline__ void rotate_geometric(f32x2 x, f32x2 c, f32x2 g, f32x2 (&z)[2 * B + 1]) {
    const bool org = (fabsf(x.x) < kOriginEps) && (fabsf(x.y) < kOriginEps);
    const f32x2 sq = x * x;
    const float inv = __frsqrt_rn(sq.x + sq.y);
    const f32x2 u = org ? f32x2{1.f, 0.f} : f32x2{x.x * inv, -x.y * inv};
    const f32x2 v = cmul_pk(g, u);
    z[B] = cmul_pk(c, x);
    if (B >= 1) {
        z[B + 1] = cmul_pk(z[B], v);
        z[B - 1] = cmul_conj_pk(z[B], v);
    }
#pragma unroll
    for (int m = 2; m <= B; ++m) {
        z[B + m] = cmul_pk(z[B + m - 1], v);
        z[B - m] = cmul_conj_pk(z[B - m + 1], v);
    }
}
constexpr int kGeoRecordFloats = 8;     // [q bits, w_q, w_{q+1}, other endpoint bits, Re c, Im c, Re g, Im g]
constexpr int kGeoLogChunkRecords = 5;  // 32 records per 1 KiB chunk

// Complex multiply-accumulate on packed fp32: acc (re,im) += s * x with s a WAVE-UNIFORM complex
// number in an SGPR pair and x a per-lane complex number.  Plain wave64 FMAs issue every 4 cycles on
// gfx950 (measured 76 TFLOP/s with an SGPR operand); v_pk_fma_f32 does two per lane in the same
// slot (146 TFLOP/s, tools/ubench/pk_fma.hip).  Written with the generic vector fma so the
// register allocator sees ordinary values (inline-asm FMAs with tied operands made hipcc copy
// every accumulator once per edge); hipcc folds the broadcasts into op_sel / one s_mov:
//   acc += (s.re, s.re) * (x.re, x.im);   acc += (s.im, s.im) * (-x.im, x.re)
// `xs` = (-x.im, x.re) is formed once per rotated feature, not per ring.
__device__ __forceinline__ void cmac_sx(f32x2& acc, f32x2 s, f32x2 x, f32x2 xs) {
    acc = __builtin_elementwise_fma(f32x2{s.x, s.x}, x, acc);
    acc = __builtin_elementwise_fma(f32x2{s.y, s.y}, xs, acc);
}
// acc += g * conj(s) = (s.re, s.re)*(g.re, g.im) + (s.im, s.im)*(g.im, -g.re); gs = (g.im, -g.re)
__device__ __forceinline__ void cmac_gconjs(f32x2& acc, f32x2 s, f32x2 g, f32x2 gs) {
    acc = __builtin_elementwise_fma(f32x2{s.x, s.x}, g, acc);
    acc = __builtin_elementwise_fma(f32x2{s.y, s.y}, gs, acc);
}

// One rotated copy for a run-time frequency m (|m| <= B): returns u^m, the caller multiplies.
__device__ __forceinline__ float2 unit_power(float2 u, int m) {
    float2 p = make_float2(1.f, 0.f);
    const int am = m < 0 ? -m : m;
    for (int k = 0; k < am; ++k) p = (m > 0) ? cmul(p, u) : cmul_conj(p, u);
    return p;
}

// gx[j,i] = sum_f gxt_f conj(u^m) + [x != 0] (i x / |x|^2) sum_f m Im(conj(gxt_f) x u^m),  u = exp(-i angle(x))  (1 inside the origin box):
// the input gradient from the F slices gxt_f (complex numbers `stride` apart) of the adjoint of the rotated copies
// (autograd of reference nn/field_conv.py:128-130)
// ks: slices per frequency (the streaming arrangement's halves of the k range, fc_backward_stream.hpp), summed first
template <int B>
__device__ __forceinline__ float2 gx_from_slices(const float2 x, const float2* __restrict__ gxt, const size_t idx, const size_t stride,
                                                 const int ks = 1) {
    constexpr int F = 2 * B + 1;
    float2 z[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        z[f] = gxt[(size_t)(f * ks) * stride + idx];
        if (ks == 2) {
            const float2 z2 = gxt[(size_t)(f * ks + 1) * stride + idx];
            z[f].x += z2.x;
            z[f].y += z2.y;
        }
    }
    const float2 u1 = unit_conj(x);
    float2 up[B + 1];
    up[0] = make_float2(1.f, 0.f);
#pragma unroll
    for (int q = 1; q <= B; ++q) up[q] = cmul(up[q - 1], u1);
    const float inv2 = is_origin(x) ? 0.f : 1.f / (x.x * x.x + x.y * x.y);
    float2 acc = make_float2(0.f, 0.f);
    float eq = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int mm = f - B;
        float2 c = up[mm < 0 ? -mm : mm];
        if (mm < 0) c.y = -c.y;
        const float2 xtv = cmul(x, c);
        const float2 out = cmul_conj(z[f], c);
        acc.x += out.x;
        acc.y += out.y;
        eq += (float)mm * (z[f].x * xtv.y - z[f].y * xtv.x);
    }
    const float q = eq * inv2;
    acc.x += -x.y * q;
    acc.y += x.x * q;
    return acc;
}

// In-kernel time stamps (development): lane 0 of every wavefront of workgroup 0 appends (label << 56 | s_memtime).
struct Stamper {
    unsigned long long* p;      // wave-uniform: this wavefront's 256 slots, or nullptr
    int n;
    __device__ __forceinline__ void operator()(int label) {
        if (p) {
            if (n < 256 && (threadIdx.x & 63) == 0)
                p[n] = ((unsigned long long)label << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);
            ++n;
        }
    }
    // constant 100 MHz clock: with the shader-cycle stamps around it, the clock the launch actually ran at
    __device__ __forceinline__ void realtime(int label) {
        if (p) {
            if (n < 256 && (threadIdx.x & 63) == 0)
                p[n] = ((unsigned long long)label << 56) | (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull);
            ++n;
        }
    }
};

}  // namespace fc
