// FieldConv backward for gfx950 (the reference has no backward code: it relies on torch autograd
// through nn/field_conv.py:128-137, i.e. the saved (E,C,R,F) product and index_select/scatter
// twins).  Here the adjoint is evaluated source-centrically and never touches an edge-sized
// temporary or an atomic:
//
//   H[j,o,r,f]  = sum_{e: src_e = j} gy[dst_e,o] conj(S[e,r,f])            (gather, CSR by source)
//   gxt[j,i,f]  = 1/F sum_{o,r} H[j,o,r,f] conj(W[o,i,r,f])                (MFMA, K = R*O)
//   gx[j,i]     = sum_f gxt_f conj(u_f) + [x != 0] (i x/|x|^2) sum_f m_f Im(conj(gxt_f) xt_f)
//   gW[o,i,r,f] = 1/F sum_j    H[j,o,r,f] conj(xt[j,i,f])                  (MFMA, K = vertices)
//
// Two kernels.
//  fc_backward_data_kernel is the forward kernel transposed: one wavefront per SOURCE vertex gathers
//  H[j,:,:,:] (lane = output channel o, all frequencies, dense or factored stencil exactly as in
//  fc_forward.hip), drops one LDS slab per frequency, the workgroup contracts it with the packed
//  conjugated filter on MFMA, and every thread folds its (vertex, channel) entry of gxt_f into a
//  running gx.  Each slab is also copied to HBM (`hdump`, 2*16*KS floats per tile and frequency).
//  fc_backward_filter_kernel needs every tile's H for one frequency and 553 KB of accumulators in
//  total -- more than a CU's register file -- so blockIdx.y = f: a persistent workgroup keeps
//  gW[:,:,:,f] (KP x IP complex, spread over its wavefronts' MFMA accumulators) in registers,
//  streams the dumped slabs of its frequency back with LDS-DMA (double-buffered, no gather at all)
//  and writes one partial at the end; fc_backward_finish sums the partials in a fixed order.
//  The 2 x 237 MB of slab traffic at config 2 replace a five-fold repetition of the gather.
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

constexpr int kMaxGwTiles = 8;   // 16x16 complex gW tiles a wavefront can own

// (pointers are separate __restrict__ kernel parameters, see fc_forward.hip)
struct BwdArgs {
    int N, I, O;
    MmaGeom g;           // M = I (rows of gxt), K = R*O
    int ntiles;
    int ngw;             // KST * NMT 16x16 gW tiles per frequency
    int KD;              // row stride (floats) of the H slabs kept for the filter kernel: KP + 4, so that the
                         // filter kernel's 4-byte A-fragment reads (rows 4 apart per lane group) hit distinct banks
    int slab_floats;     // 2 * 16 * KD
    int slab_stride;     // floats between consecutive (tile, f) slabs in hdump (multiple of 256)
    int dbg;             // development only: bit0 skip gather, bit1 skip gxt MFMA, bit2 skip gW MFMA
};

template <int R, int B>
struct BwdShape {
    static constexpr int F = 2 * B + 1;
    static constexpr int NG = (F * R + 31) / 32;
    static constexpr int MG = (F + NG - 1) / NG;
};

// ------------------------------------------------------------------------------------ data gradient
template <int R, int B, bool FACTORED>
__global__ __launch_bounds__(kThreads) void fc_backward_data_kernel(
    const float2* __restrict__ gx_, const float2* __restrict__ ggy, const float* __restrict__ gsten,
    const int32_t* __restrict__ growptr, const int32_t* __restrict__ gnbr, const float* __restrict__ gwpk,
    float2* __restrict__ ggx, float* __restrict__ hdump, const BwdArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int NG = BwdShape<R, B>::NG;
    constexpr int MG = BwdShape<R, B>::MG;
    constexpr int ROWF = 2 * R * F;
    constexpr int RECF = factored_record_floats(B);
    constexpr int LOG_CR = factored_log_chunk_records(B);
    constexpr int CR = 1 << LOG_CR;
    constexpr int NR = kRingChunks;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.g;
    const int KS = mg.KS, KP = mg.KP, IP = mg.MP, I = a.I, O = a.O;
    float* const hre = reinterpret_cast<float*>(smem);     // [16][KS]
    float* const him = hre + kTile * KS;                    // [16][KS]
    float* const part = him + kTile * KS;                   // [NKP][16][IP][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ring = part + partial_floats(mg.NKP, IP) + wave * NR * 256;   // factored: [NR][256] floats per wavefront

    for (int idx = tid; idx < 2 * kTile * KS; idx += kThreads) hre[idx] = 0.f;
    __syncthreads();

    const int it = wave % mg.NMT;
    const int kp = wave / mg.NMT;
    const bool mma_active = kp < mg.NKP;
    const int ol = lane < O ? lane : 0;       // lanes >= O gather channel 0 and are never stored
    // gx epilogue: thread -> (vertex v, channel i) of the tile, fixed for the whole kernel
    const int ev = tid / I, ei = tid - ev * I;
    const bool e_active = tid < kTile * I;

    auto dma_chunk = [&](const int first, const int ch) {
        const float* src = gsten + ((size_t)first + ((size_t)ch << LOG_CR)) * RECF + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (ch & (NR - 1)) * 256), 16, 0, 0);
    };

    int beg = 0, end = 0, ro[R];      // ro: ring-run offsets of my source (factored), see fc_forward.hip
#pragma unroll
    for (int q = 0; q < R; ++q) ro[q] = 0;
    {
        const int j0 = blockIdx.x * kTile + wave;
        if (blockIdx.x < a.ntiles && j0 < a.N) {
            beg = growptr[j0];
            end = growptr[j0 + 1];
            if (FACTORED) {
#pragma unroll
                for (int q = 0; q < R; ++q) ro[q] = gnbr[(size_t)j0 * kRunStride + q];
            }
        }
        if (FACTORED) {
            const int nch = (end - beg + CR - 1) >> LOG_CR;
            for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
        }
    }

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int nbeg = 0, nend = 0, nro[R];      // my source in the next tile
#pragma unroll
        for (int q = 0; q < R; ++q) nro[q] = 0;
        {
            const int jn = (tile + gridDim.x) * kTile + wave;
            if (tile + gridDim.x < a.ntiles && jn < a.N) {
                nbeg = growptr[jn];
                nend = growptr[jn + 1];
                if (FACTORED) {
#pragma unroll
                    for (int q = 0; q < R; ++q) nro[q] = gnbr[(size_t)jn * kRunStride + q];
                }
            }
        }
        const int nslots = end - beg;
        const int nch = (nslots + CR - 1) >> LOG_CR;
        // my (vertex, channel) entry of x for the gx epilogue: issued now, consumed after the first slab
        const int ejn = tile * kTile + ev;
        float2 exs = make_float2(0.f, 0.f);
        if (e_active && ejn < a.N) exs = gx_[(size_t)ejn * I + ei];
        float2 gxacc = make_float2(0.f, 0.f);

#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int MGc = MG;
            const int f0 = g * MGc;
            f32x2 h[R][MG];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int ff = 0; ff < MG; ++ff) h[r][ff] = f32x2{0.f, 0.f};

            // ---------------------------------------------------------------- gather H for my source
            if constexpr (FACTORED) {
                if (g > 0) for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);   // walk the slots again
                auto rec_ptr = [&](const int s) { return ring + ((s >> LOG_CR) & (NR - 1)) * 256 + (s & (CR - 1)) * RECF; };
                float2 ga = make_float2(0.f, 0.f), gb = ga;
                if (nslots > 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
                    const int d0 = __float_as_int(rec_ptr(0)[3]);
                    const int d1 = __float_as_int(rec_ptr(min(1, nslots - 1))[3]);
                    ga = ggy[(size_t)d0 * O + ol];
                    gb = ggy[(size_t)d1 * O + ol];
                }
                // one slot with compile-time lower ring Q: z_f = g conj(ph_f); h[Q] += w0 z; h[Q+1] += w1 z
                auto slot = [&](auto qc, const int s, float2& gcur) {
                    constexpr int Q = decltype(qc)::value;
                    if ((s & (CR - 1)) == 0 && s > 0) {
                        const int ch = s >> LOG_CR;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (ch - 1 + NR < nch) dma_chunk(beg, ch - 1 + NR);
                    }
                    const float* rp = rec_ptr(s);
                    const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                    const int d2 = __float_as_int(rec_ptr(min(s + 2, nslots - 1))[3]);
                    const f32x2 gv = f32x2{gcur.x, gcur.y}, gs = f32x2{gcur.y, -gcur.x};
                    gcur = ggy[(size_t)d2 * O + ol];
                    const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff) {
                        const int f = f0 + ff;
                        if (f < F) {
                            const f32x2 ph = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                            f32x2 z = f32x2{ph.x, ph.x} * gv;
                            z = __builtin_elementwise_fma(f32x2{ph.y, ph.y}, gs, z);
                            h[Q][ff] = __builtin_elementwise_fma(w0v, z, h[Q][ff]);
                            h[Q + 1][ff] = __builtin_elementwise_fma(w1v, z, h[Q + 1][ff]);
                        }
                    }
                };
                if (!(a.dbg & 1)) {
                    static_for<0, R - 1>([&](auto qc) {
                        constexpr int Q = decltype(qc)::value;
                        int s = ro[Q];
                        const int run_end = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                        for (; s + 1 < run_end; s += 2) {
                            slot(qc, s, ga);
                            slot(qc, s + 1, gb);
                        }
                        if (s < run_end) {
                            slot(qc, s, ga);
                            const float2 t = ga; ga = gb; gb = t;
                        }
                    });
                }
                if (g + 1 == NG) {
                    // my source is done: stream the first record chunks of my next tile's source
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    const int nnch = (nend - nbeg + CR - 1) >> LOG_CR;
                    for (int ch = 0; ch < min(nnch, NR); ++ch) dma_chunk(nbeg, ch);
                }
            } else {
                const int last = end - 1;
                int nx = 0;
                float2 ga = make_float2(0.f, 0.f), gb = ga;
                if (beg < end) {
                    const int d0 = gnbr[beg];
                    const int d1 = gnbr[min(beg + 1, last)];
                    nx = gnbr[min(beg + 2, last)];
                    ga = ggy[(size_t)d0 * O + ol];
                    gb = ggy[(size_t)d1 * O + ol];
                }
                auto slot = [&](const int e, float2& gcur) {
                    const f32x2* __restrict__ Se = reinterpret_cast<const f32x2*>(gsten + (size_t)e * ROWF);   // wave-uniform
                    const int n3 = gnbr[min(e + 3, last)];
                    const f32x2 gv = f32x2{gcur.x, gcur.y}, gs = f32x2{gcur.y, -gcur.x};
                    gcur = ggy[(size_t)nx * O + ol];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int ff = 0; ff < MG; ++ff)
                            if (f0 + ff < F) cmac_gconjs(h[r][ff], Se[r * F + f0 + ff], gv, gs);
                    nx = n3;
                };
                if (!(a.dbg & 1))
                    for (int e = beg; e < end; e += 2) {
                        slot(e, ga);
                        if (e + 1 < end) slot(e + 1, gb);
                    }
            }

            // ---------------------------------------------------------------- slabs -> gxt_f -> gx
#pragma unroll
            for (int ff = 0; ff < MG; ++ff) {
                const int f = f0 + ff;
                if (f < F) {
                    if (lane < O) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            hre[wave * KS + r * O + lane] = h[r][ff].x;
                            him[wave * KS + r * O + lane] = h[r][ff].y;
                        }
                    }
                    __syncthreads();
                    {   // keep this slab for the filter-gradient kernel (16 B per thread, rows re-strided to KD)
                        float* dst = hdump + ((size_t)tile * F + f) * a.slab_stride;
                        const int k4n = KP / 4;
                        for (int idx = tid; idx < 2 * kTile * k4n; idx += kThreads) {
                            const int row = idx / k4n, k4 = idx - row * k4n;          // row = plane * 16 + vertex
                            *reinterpret_cast<float4*>(dst + row * a.KD + 4 * k4) = *reinterpret_cast<const float4*>(hre + row * KS + 4 * k4);
                        }
                    }
                    if (mma_active) {
                        f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;
                        if (!(a.dbg & 2))
                            mma_slab(gwpk + (size_t)(f * 2 + 0) * IP * KP, gwpk + (size_t)(f * 2 + 1) * IP * KP, hre, him, mg, it,
                                     kp, lane, acc_re, acc_im);
                        store_partial(part, mg, it, kp, lane, acc_re, acc_im);
                    }
                    __syncthreads();
                    if (e_active) {
                        const int m = f - B;
                        const float2 z = sum_partials(part, mg, ev, ei);
                        const float2 c = unit_power(unit_conj(exs), m);
                        const float2 xtv = cmul(exs, c);
                        float2 out = cmul_conj(z, c);
                        if (m != 0 && !is_origin(exs)) {
                            const float n2 = exs.x * exs.x + exs.y * exs.y;
                            const float q = (float)m * (z.x * xtv.y - z.y * xtv.x) / n2;
                            out.x += -exs.y * q;
                            out.y += exs.x * q;
                        }
                        gxacc.x += out.x;
                        gxacc.y += out.y;
                    }
                    // (`part` is next written after the following slab's first barrier)
                }
            }
        }
        if (e_active && ejn < a.N) ggx[(size_t)ejn * I + ei] = gxacc;
        beg = nbeg;
        end = nend;
#pragma unroll
        for (int q = 0; q < R; ++q) ro[q] = nro[q];
    }
}

// ---------------------------------------------------------------------------------- filter gradient
// T = 16x16 complex gW tiles owned by each wavefront (ceil(ngw / 16)); 8*T accumulator VGPRs.
template <int T>
__global__ __launch_bounds__(kThreads) void fc_backward_filter_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ hdump, float2* __restrict__ ggwp /* [P][F][KP][IP] */,
    const BwdArgs a, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.g;
    const int KD = a.KD, KP = mg.KP, IP = mg.MP, I = a.I;
    float* const slab0 = reinterpret_cast<float*>(smem);               // [2][slab_stride]: (hre | him) of a tile
    float* const xtr = slab0 + 2 * a.slab_stride;                      // [IP][16]
    float* const xti = xtr + IP * kTile;                               // [IP][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int m = f - B;
    const int fr = lane & 15, fq = lane >> 4;

    for (int idx = tid; idx < 2 * IP * kTile; idx += kThreads) xtr[idx] = 0.f;

    // my gW tiles: u = wave + 16 n  ->  (row tile rt over k = (r,o), column tile ct over i)
    int gw_h[T], gw_x[T];       // wave-uniform LDS offsets, -1 when the slot is unused
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int u = wave + kWaves * n;
        const int rt = u / mg.NMT, ct = u - rt * mg.NMT;
        gw_h[n] = (u < a.ngw) ? rt * 16 : -1;
        gw_x[n] = ct * 16 * kTile;
    }
    const int h_lane = (4 * fq) * KD + fr;        // A fragment: H[vertex 4fq+s][k = rt*16 + fr]
    const int x_lane = fr * kTile + 4 * fq;       // B fragment: xt[i = ct*16 + fr][vertex 4fq..4fq+3]

    f32x4 gre[T], gim[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }

    const int npieces = a.slab_stride / 256;      // 1 KiB DMA pieces per slab
    auto dma_slab = [&](const int tile, const int buf) {
        const float* src = hdump + ((size_t)tile * F + f) * a.slab_stride;
        for (int p = wave; p < npieces; p += kWaves)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 256 + lane * 4), (lptr_t)(slab0 + buf * a.slab_stride + p * 256), 16,
                                             0, 0);
    };

    if (blockIdx.x < a.ntiles) dma_slab(blockIdx.x, 0);
    float2 xv = make_float2(0.f, 0.f);
    {
        const int j = blockIdx.x * kTile + wave;
        if (blockIdx.x < a.ntiles && j < a.N && lane < I) xv = gx_[(size_t)j * I + lane];
    }
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        // rotated feature xt_f of my source row (B operand)
        const float2 xt = cmul(xv, unit_power(unit_conj(xv), m));
        __syncthreads();            // slab `buf` has landed (the barrier drains the DMA); previous tile's reads are done
        if (lane < IP) {
            xtr[lane * kTile + wave] = xt.x;
            xti[lane * kTile + wave] = xt.y;
        }
        const int tn = tile + gridDim.x;
        if (tn < a.ntiles) {
            dma_slab(tn, buf ^ 1);
            const int jn = tn * kTile + wave;
            xv = (jn < a.N && lane < I) ? gx_[(size_t)jn * I + lane] : make_float2(0.f, 0.f);
        }
        // the xt stores must be visible to every wavefront; LDS only, the DMA just issued stays in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        const float* hre = slab0 + buf * a.slab_stride;
        const float* him = hre + kTile * KD;
        // gW += H^T . conj(xt):  re += Hre*Xre + Him*Xim ; im += Him*Xre - Hre*Xim
        if (!(a.dbg & 4)) {
#pragma unroll
            for (int n = 0; n < T; ++n) {
                if (gw_h[n] >= 0) {
                    const float4 xr4 = *reinterpret_cast<const float4*>(xtr + gw_x[n] + x_lane);
                    const float4 xi4 = *reinterpret_cast<const float4*>(xti + gw_x[n] + x_lane);
                    const float* ha = hre + gw_h[n] + h_lane;
                    const float* hb = him + gw_h[n] + h_lane;
                    const float a0 = ha[0], a1 = ha[KD], a2 = ha[2 * KD], a3 = ha[3 * KD];
                    const float b0 = hb[0], b1 = hb[KD], b2 = hb[2 * KD], b3 = hb[3 * KD];
                    gre[n] = mfma16(a0, xr4.x, gre[n]); gim[n] = mfma16(b0, xr4.x, gim[n]);
                    gre[n] = mfma16(b0, xi4.x, gre[n]); gim[n] = mfma16(-a0, xi4.x, gim[n]);
                    gre[n] = mfma16(a1, xr4.y, gre[n]); gim[n] = mfma16(b1, xr4.y, gim[n]);
                    gre[n] = mfma16(b1, xi4.y, gre[n]); gim[n] = mfma16(-a1, xi4.y, gim[n]);
                    gre[n] = mfma16(a2, xr4.z, gre[n]); gim[n] = mfma16(b2, xr4.z, gim[n]);
                    gre[n] = mfma16(b2, xi4.z, gre[n]); gim[n] = mfma16(-a2, xi4.z, gim[n]);
                    gre[n] = mfma16(a3, xr4.w, gre[n]); gim[n] = mfma16(b3, xr4.w, gim[n]);
                    gre[n] = mfma16(b3, xi4.w, gre[n]); gim[n] = mfma16(-a3, xi4.w, gim[n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // flush my gW partial
#pragma unroll
    for (int n = 0; n < T; ++n) {
        if (gw_h[n] >= 0) {
            const int ct16 = gw_x[n] / kTile;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = gw_h[n] + 4 * fq + jj;
                const int i = ct16 + fr;
                ggwp[(((size_t)blockIdx.x * F + f) * KP + k) * IP + i] = make_float2(gre[n][jj], gim[n][jj]);
            }
        }
    }
}

// gw_eff[o][i][r][f] = 1/F sum_p gwp[p][f][r*O+o][i]
__global__ void fc_reduce_gw_kernel(const float2* __restrict__ gwp, float2* __restrict__ gw, int P, int F, int R,
                                    int O, int I, int KP, int IP) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over (f, k<R*O, i<I), i fastest
    const int total = F * R * O * I;
    if (idx >= total) return;
    const int i = idx % I;
    const int k = (idx / I) % (R * O);
    const int f = idx / (I * R * O);
    float2 s = make_float2(0.f, 0.f);
    for (int p = 0; p < P; ++p) {
        const float2 v = gwp[(((size_t)p * F + f) * KP + k) * IP + i];
        s.x += v.x;
        s.y += v.y;
    }
    const float sc = 1.f / (float)F;
    const int r = k / O, o = k - r * O;
    gw[(((size_t)o * I + i) * R + r) * F + f] = make_float2(s.x * sc, s.y * sc);
}

struct BwdPlan {
    MmaGeom g;
    int IP, KP, KD, ntiles, ngw, P, F, slab_floats, slab_stride;
    size_t lds_data, lds_data_factored, lds_filter, hdump_bytes, gwp_bytes;
    bool ok, ok_factored;
};

static BwdPlan plan_backward(const fc_dims* d) {
    BwdPlan p;
    p.F = 2 * d->B + 1;
    p.g = make_mma_geom(d->I, d->R * d->O);
    p.IP = p.g.MP;
    p.KP = p.g.KP;
    p.ntiles = (d->N + kTile - 1) / kTile;
    p.ngw = p.g.KST * p.g.NMT;
    int P = kNumCUs / p.F;                   // filter kernel: one workgroup per CU across the F frequency slices
    if (P < 1) P = 1;
    if (P > p.ntiles) P = p.ntiles;
    p.P = P;
    p.KD = p.g.KP + 4;
    p.slab_floats = 2 * kTile * p.KD;
    p.slab_stride = round_up(p.slab_floats, 256);
    p.lds_data = (size_t)(2 * kTile * p.g.KS + partial_floats(p.g.NKP, p.IP)) * sizeof(float);
    p.lds_data_factored = p.lds_data + (size_t)kWaves * kRingChunks * 1024;
    p.lds_filter = (size_t)(2 * p.slab_stride + 2 * p.IP * kTile) * sizeof(float);
    p.hdump_bytes = ((size_t)p.ntiles * p.F * p.slab_stride + 256) * sizeof(float);
    p.gwp_bytes = (size_t)p.P * p.F * p.KP * p.IP * sizeof(float2);
    p.ok = p.lds_data <= kMaxLds && p.lds_filter <= kMaxLds && p.ngw <= kMaxGwTiles * kWaves && p.g.NMT <= kWaves &&
           kTile * d->I <= kThreads;
    p.ok_factored = p.ok && p.lds_data_factored <= kMaxLds;
    return p;
}

size_t backward_workspace_bytes(const fc_dims* d) {
    const BwdPlan p = plan_backward(d);
    return p.hdump_bytes + p.gwp_bytes + 256;
}

static BwdArgs make_args(const fc_dims* d, const BwdPlan& p) {
    BwdArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.g = p.g;
    a.ntiles = p.ntiles;
    a.ngw = p.ngw;
    a.KD = p.KD;
    a.slab_floats = p.slab_floats;
    a.slab_stride = p.slab_stride;
    { const char* e = getenv("FC_DEBUG_BWD"); a.dbg = e ? atoi(e) : 0; }
    return a;
}

template <int R, int B, bool FACTORED>
static int launch_backward_data(const float2* x, const float2* gy, const float* sten, const fc_csr* g, const float* wpk,
                                float2* gx, float* hdump, const BwdArgs& a, const BwdPlan& p, hipStream_t stream) {
    auto kern = fc_backward_data_kernel<R, B, FACTORED>;
    const size_t lds = FACTORED ? p.lds_data_factored : p.lds_data;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return FC_ERR_LAUNCH;
    }
    const int grid = FACTORED ? (p.ntiles < kNumCUs ? p.ntiles : kNumCUs) : p.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, x, gy, sten, g->rowptr, FACTORED ? g->runs : g->nbr, wpk, gx,
                       hdump, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_data_impl(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk, float* gx,
                       void* ws, size_t ws_bytes, const fc_dims* d, bool factored, hipStream_t stream) {
    const BwdPlan p = plan_backward(d);
    if (!(factored ? p.ok_factored : p.ok)) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const BwdArgs a = make_args(d, p);
    float* hdump = reinterpret_cast<float*>(ws);
#define FC_CASE(RR, BB)                                                                                                \
    if (d->R == RR && d->B == BB)                                                                                      \
        return factored ? launch_backward_data<RR, BB, true>(reinterpret_cast<const float2*>(x),                       \
                                                             reinterpret_cast<const float2*>(gy), sten, g, wpk,        \
                                                             reinterpret_cast<float2*>(gx), hdump, a, p, stream)       \
                        : launch_backward_data<RR, BB, false>(reinterpret_cast<const float2*>(x),                      \
                                                              reinterpret_cast<const float2*>(gy), sten, g, wpk,       \
                                                              reinterpret_cast<float2*>(gx), hdump, a, p, stream);
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    return FC_ERR_UNSUPPORTED;
}

template <int T>
static int launch_backward_filter(const float2* x, const float* hdump, float2* gwp, const BwdArgs& a, const BwdPlan& p,
                                  int B, hipStream_t stream) {
    auto kern = fc_backward_filter_kernel<T>;
    if (p.lds_filter > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)p.lds_filter) != hipSuccess)
            return FC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.P, p.F), dim3(kThreads), p.lds_filter, stream, x, hdump, gwp, a, p.F, B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_filter_impl(const float* x, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BwdPlan p = plan_backward(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const BwdArgs a = make_args(d, p);
    const float* hdump = reinterpret_cast<const float*>(ws);
    float2* gwp = reinterpret_cast<float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const float2* x2 = reinterpret_cast<const float2*>(x);
    const int need = (p.ngw + kWaves - 1) / kWaves;
    if (need <= 2) return launch_backward_filter<2>(x2, hdump, gwp, a, p, d->B, stream);
    if (need <= 4) return launch_backward_filter<4>(x2, hdump, gwp, a, p, d->B, stream);
    if (need <= kMaxGwTiles) return launch_backward_filter<kMaxGwTiles>(x2, hdump, gwp, a, p, d->B, stream);
    return FC_ERR_UNSUPPORTED;
}

// Fixed-order sum of the per-workgroup filter-gradient partials left in the workspace.
int backward_finish_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BwdPlan p = plan_backward(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float2* gwp = reinterpret_cast<const float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const int total = p.F * d->R * d->O * d->I;
    hipLaunchKernelGGL(fc_reduce_gw_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gwp,
                       reinterpret_cast<float2*>(gw_eff), p.P, p.F, d->R, d->O, d->I, p.KP, p.IP);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc
