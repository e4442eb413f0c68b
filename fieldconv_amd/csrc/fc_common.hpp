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

namespace fc {

constexpr int kWave = 64;
constexpr int kTile = 16;        // vertices per tile == MFMA N
constexpr int kWaves = 16;       // wavefronts per workgroup
constexpr int kThreads = kWave * kWaves;
constexpr float kOriginEps = 1e-7f;   // reference utils/field.py:8

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }

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

// All 2B+1 rotated copies xt[f] = z * u^(f-B) (reference nn/field_conv.py:128-130).
template <int B>
__device__ __forceinline__ void rotate_all(float2 z, float2 (&xt)[2 * B + 1]) {
    const float2 u = unit_conj(z);
    xt[B] = z;
#pragma unroll
    for (int m = 1; m <= B; ++m) {
        xt[B + m] = cmul(xt[B + m - 1], u);
        xt[B - m] = cmul_conj(xt[B - m + 1], u);
    }
}

// One rotated copy for a run-time frequency m (|m| <= B): returns u^m, the caller multiplies.
__device__ __forceinline__ float2 unit_power(float2 u, int m) {
    float2 p = make_float2(1.f, 0.f);
    const int am = m < 0 ? -m : m;
    for (int k = 0; k < am; ++k) p = (m > 0) ? cmul(p, u) : cmul_conj(p, u);
    return p;
}

}  // namespace fc
