// FieldConv forward for gfx950: gather -> rotate -> stencil-multiply -> segmented reduce ->
// filter contraction, fused in one kernel (replaces reference nn/field_conv.py:128-137).
//
// One 16-wavefront workgroup owns a tile of 16 target vertices, one wavefront per target.
//
//  Phase A (VALU).  Lane c of the wavefront owns input channel c.  The wavefront walks the
//    target's in-edges (CSR by target, so no atomics and a fixed summation order): the source
//    row x[src,:] is one coalesced 8-byte-per-lane load, the R*F complex stencil entries of the
//    edge are wave-uniform and arrive through the scalar cache into SGPRs, and the per-target
//    response contrib[c, r, f] (R*F complex numbers per lane) stays in registers.
//  Phase B (MFMA).  For each angular frequency f the wavefronts drop their contrib[:, :, f]
//    slab into LDS ([vertex][k = r*I + c], re and im planes) and the workgroup multiplies it by
//    the packed filter on v_mfma_f32_16x16x4_f32: out^T[o, vertex] += W[o, k] * contrib[vertex, k].
//    Wavefront w takes output tile (w % NOT) and every NKP-th 16-wide k block; the k-partials
//    are combined through LDS in a fixed order (bitwise reproducible).
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

// Pointers travel as separate __restrict__ kernel parameters (not inside this struct) so that the
// compiler may treat the index and stencil streams as read-only and fetch the wave-uniform ones
// through the scalar cache (s_load) instead of per-lane vector loads.
struct FwdArgs {
    int N, I, O;
    int OP, KP, KS;     // OP = ceil16(O), KP = ceil16(R*I), KS = LDS slab row stride
    int NOT, NKP, KST;  // output tiles, k-partitions, 16-wide k blocks per slab
    int ntiles;
};

// Frequencies are processed in NG groups of at most MG so that the per-lane response
// (R*MG complex numbers) stays within the 128-VGPR budget of a 16-wavefront workgroup.
template <int R, int B>
__global__ __launch_bounds__(kThreads) void fc_forward_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ gsten, const int32_t* __restrict__ growptr,
    const int32_t* __restrict__ gnbr, const int32_t* __restrict__ geid, const float* __restrict__ gwpk,
    float2* __restrict__ gy_, const FwdArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int NG = (F * R + 31) / 32;
    constexpr int MG = (F + NG - 1) / NG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const cre = reinterpret_cast<float*>(smem);          // [16][KS]
    float* const cim = cre + kTile * a.KS;                       // [16][KS]
    float* const part = cim + kTile * a.KS;                      // [NKP][16][OP][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = a.I, O = a.O, KS = a.KS, KP = a.KP, OP = a.OP;

    // zero the slab once: the k padding [R*I, KP) is never written again and must not hold NaNs
    for (int idx = tid; idx < 2 * kTile * KS; idx += kThreads) cre[idx] = 0.f;
    __syncthreads();

    const int ot = wave % a.NOT;
    const int kp = wave / a.NOT;
    const bool mma_active = kp < a.NKP;
    const int fr = lane & 15;       // fragment row/col
    const int fq = lane >> 4;       // fragment k quarter

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int t = tile * kTile + wave;
        int beg = 0, end = 0;
        if (t < a.N) { beg = growptr[t]; end = growptr[t + 1]; }
        const bool has_c = lane < I;
        const int cl = has_c ? lane : 0;

        f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;

#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int MGc = MG;
            const int f0 = g * MGc;
            // -------------------------------------------------------------- phase A (group g)
            float cr[R][MG], ci[R][MG];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int ff = 0; ff < MG; ++ff) { cr[r][ff] = 0.f; ci[r][ff] = 0.f; }

            int s_next = 0, row_next = 0;
            float2 xv_next = make_float2(0.f, 0.f);
            if (beg < end) {
                s_next = gnbr[beg];
                row_next = geid ? geid[beg] : beg;
                xv_next = gx_[(size_t)s_next * I + cl];
            }
            for (int e = beg; e < end; ++e) {
                const int row = __builtin_amdgcn_readfirstlane(row_next);
                float2 xv = xv_next;
                if (e + 1 < end) {
                    s_next = gnbr[e + 1];
                    row_next = geid ? geid[e + 1] : e + 1;
                    xv_next = gx_[(size_t)s_next * I + cl];
                }
                if (!has_c) xv = make_float2(0.f, 0.f);
                float2 xt[F];
                rotate_all<B>(xv, xt);
                const float* __restrict__ S = gsten + (size_t)row * (2 * R * F);   // wave-uniform -> s_load
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff) {
                        const int f = f0 + ff;
                        if (f < F) {
                            const float sr = S[2 * (r * F + f)];
                            const float si = S[2 * (r * F + f) + 1];
                            cr[r][ff] = fmaf(sr, xt[f].x, cr[r][ff]);
                            cr[r][ff] = fmaf(-si, xt[f].y, cr[r][ff]);
                            ci[r][ff] = fmaf(sr, xt[f].y, ci[r][ff]);
                            ci[r][ff] = fmaf(si, xt[f].x, ci[r][ff]);
                        }
                    }
            }

            // -------------------------------------------------------------- phase B (group g)
#pragma unroll
            for (int ff = 0; ff < MG; ++ff) {
                const int f = f0 + ff;
                if (f < F) {
                    if (has_c) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            cre[wave * KS + r * I + lane] = cr[r][ff];
                            cim[wave * KS + r * I + lane] = ci[r][ff];
                        }
                    }
                    __syncthreads();
                    if (mma_active) {
                        const float* wre = gwpk + ((size_t)(f * 2 + 0) * OP + ot * 16 + fr) * KP + 4 * fq;
                        const float* wim = gwpk + ((size_t)(f * 2 + 1) * OP + ot * 16 + fr) * KP + 4 * fq;
                        const float* bre = cre + fr * KS + 4 * fq;
                        const float* bim = cim + fr * KS + 4 * fq;
                        for (int kb = kp; kb < a.KST; kb += a.NKP) {
                            const float4 wr = *reinterpret_cast<const float4*>(wre + 16 * kb);
                            const float4 wi = *reinterpret_cast<const float4*>(wim + 16 * kb);
                            const float4 br = *reinterpret_cast<const float4*>(bre + 16 * kb);
                            const float4 bi = *reinterpret_cast<const float4*>(bim + 16 * kb);
                            // re += Wre*Cre - Wim*Cim ; im += Wim*Cre + Wre*Cim
                            acc_re = mfma16(wr.x, br.x, acc_re); acc_im = mfma16(wi.x, br.x, acc_im);
                            acc_re = mfma16(-wi.x, bi.x, acc_re); acc_im = mfma16(wr.x, bi.x, acc_im);
                            acc_re = mfma16(wr.y, br.y, acc_re); acc_im = mfma16(wi.y, br.y, acc_im);
                            acc_re = mfma16(-wi.y, bi.y, acc_re); acc_im = mfma16(wr.y, bi.y, acc_im);
                            acc_re = mfma16(wr.z, br.z, acc_re); acc_im = mfma16(wi.z, br.z, acc_im);
                            acc_re = mfma16(-wi.z, bi.z, acc_re); acc_im = mfma16(wr.z, bi.z, acc_im);
                            acc_re = mfma16(wr.w, br.w, acc_re); acc_im = mfma16(wi.w, br.w, acc_im);
                            acc_re = mfma16(-wi.w, bi.w, acc_re); acc_im = mfma16(wr.w, bi.w, acc_im);
                        }
                    }
                    __syncthreads();
                }
            }
        }

        // k-partials -> LDS (D layout: column = vertex = lane&15, row = output 4*(lane>>4)+j)
        if (mma_active) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = ot * 16 + 4 * fq + j;
                float* p = part + ((size_t)(kp * kTile + fr) * OP + o) * 2;
                p[0] = acc_re[j];
                p[1] = acc_im[j];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < kTile * O; idx += kThreads) {
            const int v = idx / O, o = idx - v * O;
            const int n = tile * kTile + v;
            float re = 0.f, im = 0.f;
            for (int q = 0; q < a.NKP; ++q) {
                const float* p = part + ((size_t)(q * kTile + v) * OP + o) * 2;
                re += p[0];
                im += p[1];
            }
            if (n < a.N) gy_[(size_t)n * O + o] = make_float2(re, im);
        }
        // the next tile's first LDS write (slab f=0) happens after this tile's last slab read
        // (barrier above); `part` is rewritten only after F more barriers.
    }
}

template <int R, int B>
static int launch_forward(const float2* x, const float* sten, const fc_csr* g, const float* wpk, float2* y,
                          const FwdArgs& a, size_t lds_bytes, int grid, hipStream_t stream) {
    auto kern = fc_forward_kernel<R, B>;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return FC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds_bytes, stream, x, sten, g->rowptr, g->nbr, g->eid, wpk, y, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int forward_impl(const float* x, const float* sten, const fc_csr* g, const float* wpk, float* y,
                 const fc_dims* d, hipStream_t stream) {
    FwdArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.OP = round_up(d->O, 16);
    a.KP = round_up(d->R * d->I, 16);
    a.KS = slab_stride(a.KP);
    a.NOT = a.OP / 16;
    a.NKP = kWaves / a.NOT;
    a.KST = a.KP / 16;
    if (a.NKP > a.KST) a.NKP = a.KST;
    a.ntiles = (d->N + kTile - 1) / kTile;
    const size_t lds = (size_t)(2 * kTile * a.KS + a.NKP * kTile * a.OP * 2) * sizeof(float);
    if (lds > kMaxLds) return FC_ERR_UNSUPPORTED;
    const int grid = a.ntiles;
#define FC_CASE(RR, BB) if (d->R == RR && d->B == BB) return launch_forward<RR, BB>(reinterpret_cast<const float2*>(x), sten, g, wpk, reinterpret_cast<float2*>(y), a, lds, grid, stream);
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    return FC_ERR_UNSUPPORTED;
}

}  // namespace fc
