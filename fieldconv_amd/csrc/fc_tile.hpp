// Workgroup-level pieces shared by the dense and the factored FieldConv kernels: the LDS slab
// hand-off between the per-wavefront gather phase and the MFMA contraction, the contraction
// itself, and the fixed-order combination of the k-partials.
#pragma once
#include "fc_common.hpp"

namespace fc {

// Contraction geometry of one pass: out[M x 16 vertices] = Wpk[M x K] * slab[16 vertices x K].
//   forward : M = O (output channels),  K = R*I
//   backward: M = I (input channels),   K = R*O
struct MmaGeom {
    int MP;     // ceil16(M)
    int KP;     // ceil16(K)
    int KS;     // LDS slab row stride (floats), slab_stride(KP)
    int NMT;    // MP / 16 output tiles
    int NKP;    // k partitions (wavefronts per output tile)
    int KST;    // KP / 16 k blocks
};

__host__ __device__ inline MmaGeom make_mma_geom(int M, int K) {
    MmaGeom g;
    g.MP = round_up(M, 16);
    g.KP = round_up(K, 16);
    g.KS = slab_stride(g.KP);
    g.NMT = g.MP / 16;
    g.KST = g.KP / 16;
    g.NKP = kWaves / g.NMT;
    if (g.NKP > g.KST) g.NKP = g.KST;
    if (g.NKP < 1) g.NKP = 1;
    return g;
}

// One slab (one angular frequency) through the MFMA pipe for this wavefront's (output tile mt,
// k partition kp): acc (re, im) += W[mt*16.., k] * slab[vertex, k], complex via four real products
// on two accumulators.  wre / wim: this frequency's packed planes, [MP][KP] floats each.
// Filter fragments come straight from L2 as 16 rows x 64 B per instruction and are
// double-buffered in registers; slab fragments are conflict-free float4 LDS reads.
__device__ __forceinline__ void mma_slab(const float* __restrict__ wre_plane, const float* __restrict__ wim_plane,
                                         const float* sre, const float* sim, const MmaGeom& g, int mt, int kp, int lane,
                                         f32x4& acc_re, f32x4& acc_im) {
    const int fr = lane & 15, fq = lane >> 4;
    const float* wre = wre_plane + (size_t)(mt * 16 + fr) * g.KP + 4 * fq;
    const float* wim = wim_plane + (size_t)(mt * 16 + fr) * g.KP + 4 * fq;
    const float* bre = sre + fr * g.KS + 4 * fq;
    const float* bim = sim + fr * g.KS + 4 * fq;
    float4 wr = *reinterpret_cast<const float4*>(wre + 16 * kp);
    float4 wi = *reinterpret_cast<const float4*>(wim + 16 * kp);
    for (int kb = kp; kb < g.KST; kb += g.NKP) {
        const int kn = min(kb + g.NKP, g.KST - 1);          // next block (clamped re-read at the end)
        const float4 wr_n = *reinterpret_cast<const float4*>(wre + 16 * kn);
        const float4 wi_n = *reinterpret_cast<const float4*>(wim + 16 * kn);
        const float4 br = *reinterpret_cast<const float4*>(bre + 16 * kb);
        const float4 bi = *reinterpret_cast<const float4*>(bim + 16 * kb);
        // re += Wre*Sre - Wim*Sim ; im += Wim*Sre + Wre*Sim
        acc_re = mfma16(wr.x, br.x, acc_re); acc_im = mfma16(wi.x, br.x, acc_im);
        acc_re = mfma16(-wi.x, bi.x, acc_re); acc_im = mfma16(wr.x, bi.x, acc_im);
        acc_re = mfma16(wr.y, br.y, acc_re); acc_im = mfma16(wi.y, br.y, acc_im);
        acc_re = mfma16(-wi.y, bi.y, acc_re); acc_im = mfma16(wr.y, bi.y, acc_im);
        acc_re = mfma16(wr.z, br.z, acc_re); acc_im = mfma16(wi.z, br.z, acc_im);
        acc_re = mfma16(-wi.z, bi.z, acc_re); acc_im = mfma16(wr.z, bi.z, acc_im);
        acc_re = mfma16(wr.w, br.w, acc_re); acc_im = mfma16(wi.w, br.w, acc_im);
        acc_re = mfma16(-wi.w, bi.w, acc_re); acc_im = mfma16(wr.w, bi.w, acc_im);
        wr = wr_n;
        wi = wi_n;
    }
}

// k-partials in LDS: [kp][row m][vertex v] complex, row stride kPartStride floats.  The accumulator tile
// (D layout: column = vertex = lane&15, row = 4*(lane>>4)+j) is written with one 8-byte store per
// lane and row: the 16 lanes of a row cover 32 consecutive floats (all banks once) and the +2 pad
// keeps the four lane groups of an instruction on different banks.
constexpr int kPartStride = 2 * kTile + 2;

__host__ __device__ constexpr int partial_floats(int NKP, int MP) { return NKP * MP * kPartStride; }

__device__ __forceinline__ void store_partial(float* part, const MmaGeom& g, int mt, int kp, int lane, const f32x4& acc_re,
                                              const f32x4& acc_im) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = mt * 16 + 4 * fq + j;
        *reinterpret_cast<float2*>(part + (size_t)(kp * g.MP + m) * kPartStride + 2 * fr) = make_float2(acc_re[j], acc_im[j]);
    }
}

// Fixed-order sum over the k partitions of entry (vertex v, row m).
__device__ __forceinline__ float2 sum_partials(const float* part, const MmaGeom& g, int v, int m) {
    float re = 0.f, im = 0.f;
    for (int q = 0; q < g.NKP; ++q) {
        const float2 p = *reinterpret_cast<const float2*>(part + (size_t)(q * g.MP + m) * kPartStride + 2 * v);
        re += p.x;
        im += p.y;
    }
    return make_float2(re, im);
}

}  // namespace fc
