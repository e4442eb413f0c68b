// FieldConv backward for gfx950 (the reference has no backward code: it relies on torch autograd
// through nn/field_conv.py:128-137, i.e. the saved (E,C,R,F) product and index_select/scatter
// twins).  Here the adjoint is evaluated source-centrically and never touches an edge-sized
// temporary, an atomic, or the (N,C,R,F) response:
//
//   H[j,o,r,f]  = sum_{e: src_e = j} gy[dst_e,o] conj(S[e,r,f])            (gather, CSR by source)
//   gxt[j,i,f]  = 1/F sum_{o,r} H[j,o,r,f] conj(W[o,i,r,f])                (MFMA, K = R*O)
//   gW[o,i,r,f] = 1/F sum_j    H[j,o,r,f] conj(xt[j,i,f])                  (MFMA, K = vertices)
//   gx[j,i]     = sum_f gxt_f conj(u_f) + [x != 0] (i x/|x|^2) sum_f m_f Im(conj(gxt_f) xt_f)
//
// Everything is block-diagonal in the angular frequency f, so blockIdx.y = f: a workgroup keeps
// only the R complex H values per lane (lane = output channel o) and its share of gW[:,:,:,f]
// (KP x IP complex, spread over the 16 wavefronts' MFMA accumulators) in registers while it
// walks its source tiles persistently; gW partials (one per workgroup) and the per-frequency gx
// terms are summed by two small reduction kernels in a fixed order (bitwise reproducible).
//
// As in the forward pass the gather has a dense variant (stencil rows through the scalar cache)
// and a factored one (per-edge records through a per-wavefront LDS ring, see fc_forward.hip).
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
    int dbg;             // development only: bit0 skip gather, bit1 skip gxt MFMA, bit2 skip gW MFMA
};

// T = 16x16 complex gW tiles owned by each wavefront (ceil(ngw / 16)); 8*T accumulator VGPRs.
template <int R, int B, int T, bool FACTORED>
__global__ __launch_bounds__(kThreads) void fc_backward_kernel(
    const float2* __restrict__ gx_, const float2* __restrict__ ggy, const float* __restrict__ gsten,
    const int32_t* __restrict__ growptr, const int32_t* __restrict__ gnbr,
    const float* __restrict__ gwpk, float2* __restrict__ ggxp /* [F][N][I] */,
    float2* __restrict__ ggwp /* [P][F][KP][MP] */, const BwdArgs a) {
    constexpr int F = 2 * B + 1;
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
    float* const xtr = him + kTile * KS;                    // [IP][16]
    float* const xti = xtr + IP * kTile;                    // [IP][16]
    float* const part = xti + IP * kTile;                   // [NKP][16][IP][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ring = part + mg.NKP * kTile * IP * 2 + wave * NR * 256;   // factored: [NR][256] floats per wavefront
    const int f = blockIdx.y;
    const int m = f - B;

    for (int idx = tid; idx < 2 * kTile * KS + 2 * IP * kTile; idx += kThreads) hre[idx] = 0.f;
    __syncthreads();

    const int it = wave % mg.NMT;
    const int kp = wave / mg.NMT;
    const bool mma_active = kp < mg.NKP;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const bool has_o = lane < O;
    const int ol = has_o ? lane : 0;       // lanes >= O gather channel 0 and are never stored

    // my gW tiles: u = wave + 16 n  ->  (row tile rt over k = (r,o), column tile ct over i)
    int gw_h[T], gw_x[T];       // wave-uniform LDS offsets, -1 when the slot is unused
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int u = wave + kWaves * n;
        const int rt = u / mg.NMT, ct = u - rt * mg.NMT;
        gw_h[n] = (u < a.ngw) ? rt * 16 : -1;
        gw_x[n] = ct * 16 * kTile;
    }
    const int h_lane = (4 * fq) * KS + fr;        // A fragment: H[vertex 4fq+s][k = rt*16 + fr]
    const int x_lane = fr * kTile + 4 * fq;       // B fragment: xt[i = ct*16 + fr][vertex 4fq..4fq+3]

    f32x4 gre[T], gim[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }

    auto dma_chunk = [&](const int first, const int ch) {
        const float* src = gsten + ((size_t)first + ((size_t)ch << LOG_CR)) * RECF + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (ch & (NR - 1)) * 256), 16, 0, 0);
    };

    int beg = 0, end = 0;
    {
        const int j0 = blockIdx.x * kTile + wave;
        if (blockIdx.x < a.ntiles && j0 < a.N) { beg = growptr[j0]; end = growptr[j0 + 1]; }
        if (FACTORED) {
            const int nch = (end - beg + CR - 1) >> LOG_CR;
            for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
        }
    }

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int j = tile * kTile + wave;
        int nbeg = 0, nend = 0;      // my source in the next tile
        {
            const int jn = (tile + gridDim.x) * kTile + wave;
            if (tile + gridDim.x < a.ntiles && jn < a.N) { nbeg = growptr[jn]; nend = growptr[jn + 1]; }
        }
        // ------------------------------------------------------------ gather H[:, f] for my source
        f32x2 h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) h[r] = f32x2{0.f, 0.f};
        const int nslots = end - beg;

        if constexpr (FACTORED) {
            const int nch = (nslots + CR - 1) >> LOG_CR;
            auto rec_ptr = [&](const int s) { return ring + ((s >> LOG_CR) & (NR - 1)) * 256 + (s & (CR - 1)) * RECF; };
            auto ring_of = [&](const int s) { return __builtin_amdgcn_readfirstlane(__float_as_int(rec_ptr(s)[0])); };
            float2 ga = make_float2(0.f, 0.f), gb = ga;
            if (nslots > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
                const int d0 = __float_as_int(rec_ptr(0)[3]);
                const int d1 = __float_as_int(rec_ptr(min(1, nslots - 1))[3]);
                ga = ggy[(size_t)d0 * O + ol];
                gb = ggy[(size_t)d1 * O + ol];
            }
            // one slot with compile-time lower ring Q: z = g conj(ph_f); h[Q] += w0 z; h[Q+1] += w1 z
            auto slot = [&](auto qc, const int s, float2& gcur) {
                constexpr int Q = decltype(qc)::value;
                if ((s & (CR - 1)) == 0 && s > 0) {
                    const int ch = s >> LOG_CR;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (ch - 1 + NR < nch) dma_chunk(beg, ch - 1 + NR);
                }
                const float* rp = rec_ptr(s);
                const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                const f32x2 ph = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                const int d2 = __float_as_int(rec_ptr(min(s + 2, nslots - 1))[3]);
                const f32x2 g = f32x2{gcur.x, gcur.y};
                gcur = ggy[(size_t)d2 * O + ol];
                f32x2 z = f32x2{ph.x, ph.x} * g;
                z = __builtin_elementwise_fma(f32x2{ph.y, ph.y}, f32x2{g.y, -g.x}, z);
                h[Q] = __builtin_elementwise_fma(f32x2{head.y, head.y}, z, h[Q]);
                h[Q + 1] = __builtin_elementwise_fma(f32x2{head.z, head.z}, z, h[Q + 1]);
            };
            if (!(a.dbg & 1)) {
                int s = 0;
                static_for<0, R - 1>([&](auto qc) {
                    constexpr int Q = decltype(qc)::value;
                    while (s < nslots && ring_of(s) == Q) {
                        const bool two = (s + 1 < nslots) && ring_of(s + 1) == Q;
                        slot(qc, s, ga);
                        if (two) {
                            slot(qc, s + 1, gb);
                            s += 2;
                        } else {
                            const float2 t = ga; ga = gb; gb = t;
                            s += 1;
                        }
                    }
                });
            }
            // my source is done: stream the first record chunks of my next tile's source; they land
            // while the MFMAs below run
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const int nnch = (nend - nbeg + CR - 1) >> LOG_CR;
            for (int ch = 0; ch < min(nnch, NR); ++ch) dma_chunk(nbeg, ch);
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
                const f32x2* __restrict__ Se = reinterpret_cast<const f32x2*>(gsten + (size_t)e * ROWF) + f;   // wave-uniform
                const int n3 = gnbr[min(e + 3, last)];
                const f32x2 g = f32x2{gcur.x, gcur.y}, gs = f32x2{gcur.y, -gcur.x};
                gcur = ggy[(size_t)nx * O + ol];
#pragma unroll
                for (int r = 0; r < R; ++r) cmac_gconjs(h[r], Se[r * F], g, gs);
                nx = n3;
            };
            if (!(a.dbg & 1))
                for (int e = beg; e < end; e += 2) {
                    slot(e, ga);
                    if (e + 1 < end) slot(e + 1, gb);
                }
        }

        // rotated feature xt_f of my source row (B operand of the gW product)
        {
            float2 xv = make_float2(0.f, 0.f);
            if (j < a.N && lane < I) xv = gx_[(size_t)j * I + lane];
            const float2 xt = cmul(xv, unit_power(unit_conj(xv), m));
            if (lane < IP) {
                xtr[lane * kTile + wave] = xt.x;
                xti[lane * kTile + wave] = xt.y;
            }
        }
        if (has_o) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                hre[wave * KS + r * O + lane] = h[r].x;
                him[wave * KS + r * O + lane] = h[r].y;
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ (a) gxt = H . conj(W)/F
        if (mma_active) {
            f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;
            if (!(a.dbg & 2))
                mma_slab(gwpk + (size_t)(f * 2 + 0) * IP * KP, gwpk + (size_t)(f * 2 + 1) * IP * KP, hre, him, mg, it, kp, lane,
                         acc_re, acc_im);
            store_partial(part, mg, it, kp, lane, acc_re, acc_im);
        }
        __builtin_amdgcn_sched_barrier(0);

        // ------------------------------------------------------------ (b) gW += H^T . conj(xt)
        // re += Hre*Xre + Him*Xim ; im += Him*Xre - Hre*Xim
        if (!(a.dbg & 4)) {
#pragma unroll
            for (int n = 0; n < T; ++n) {
                if (gw_h[n] >= 0) {
                    const float4 xr4 = *reinterpret_cast<const float4*>(xtr + gw_x[n] + x_lane);
                    const float4 xi4 = *reinterpret_cast<const float4*>(xti + gw_x[n] + x_lane);
                    const float* ha = hre + gw_h[n] + h_lane;
                    const float* hb = him + gw_h[n] + h_lane;
                    const float a0 = ha[0], a1 = ha[KS], a2 = ha[2 * KS], a3 = ha[3 * KS];
                    const float b0 = hb[0], b1 = hb[KS], b2 = hb[2 * KS], b3 = hb[3 * KS];
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
        __syncthreads();

        // ------------------------------------------------------------ gx term of this frequency
        for (int idx = tid; idx < kTile * I; idx += kThreads) {
            const int v = idx / I, i = idx - v * I;
            const int jn = tile * kTile + v;
            if (jn >= a.N) continue;
            const float2 z = sum_partials(part, mg, v, i);
            const float2 xs = gx_[(size_t)jn * I + i];
            const float2 c = unit_power(unit_conj(xs), m);
            const float2 xtv = cmul(xs, c);
            float2 out = cmul_conj(z, c);
            if (m != 0 && !is_origin(xs)) {
                const float n2 = xs.x * xs.x + xs.y * xs.y;
                const float q = (float)m * (z.x * xtv.y - z.y * xtv.x) / n2;
                out.x += -xs.y * q;
                out.y += xs.x * q;
            }
            ggxp[((size_t)f * a.N + jn) * I + i] = out;
        }
        __syncthreads();
        beg = nbeg;
        end = nend;
    }

    // ---------------------------------------------------------------- flush my gW partial
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

// gx[n,i] = sum_f gxp[f][n][i]
__global__ void fc_reduce_gx_kernel(const float2* __restrict__ gxp, float2* __restrict__ gx, size_t NI, int F) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NI) return;
    float2 s = make_float2(0.f, 0.f);
    for (int f = 0; f < F; ++f) {
        const float2 v = gxp[(size_t)f * NI + idx];
        s.x += v.x;
        s.y += v.y;
    }
    gx[idx] = s;
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
    int IP, KP, ntiles, ngw, P, F;
    size_t lds, lds_factored, gxp_bytes, gwp_bytes;
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
    int P = kNumCUs / p.F;                   // one workgroup per CU across the F frequency slices
    if (P < 1) P = 1;
    if (P > p.ntiles) P = p.ntiles;
    p.P = P;
    p.lds = (size_t)(2 * kTile * p.g.KS + 2 * p.IP * kTile + p.g.NKP * kTile * p.IP * 2) * sizeof(float);
    p.lds_factored = p.lds + (size_t)kWaves * kRingChunks * 1024;
    p.gxp_bytes = (((size_t)p.F * d->N * d->I * sizeof(float2) + 255) / 256) * 256;
    p.gwp_bytes = (size_t)p.P * p.F * p.KP * p.IP * sizeof(float2);
    p.ok = p.lds <= kMaxLds && p.ngw <= kMaxGwTiles * kWaves && p.g.NMT <= kWaves;
    p.ok_factored = p.ok && p.lds_factored <= kMaxLds;
    return p;
}

size_t backward_workspace_bytes(const fc_dims* d) {
    const BwdPlan p = plan_backward(d);
    return p.gxp_bytes + p.gwp_bytes + 256;
}

struct BwdPtrs {
    const float2* x; const float2* gy; const float* sten; const fc_csr* g; const float* wpk; float2* gxp; float2* gwp;
};

template <int R, int B, int T, bool FACTORED>
static int launch_backward_t(const BwdPtrs& q, const BwdArgs& a, const BwdPlan& p, hipStream_t stream) {
    auto kern = fc_backward_kernel<R, B, T, FACTORED>;
    const size_t lds = FACTORED ? p.lds_factored : p.lds;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return FC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.P, p.F), dim3(kThreads), lds, stream, q.x, q.gy, q.sten, q.g->rowptr, q.g->nbr,
                       q.wpk, q.gxp, q.gwp, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <int R, int B, bool FACTORED>
static int launch_backward(const BwdPtrs& q, const BwdArgs& a, const BwdPlan& p, hipStream_t stream) {
    const int need = (p.ngw + kWaves - 1) / kWaves;
    if (need <= 2) return launch_backward_t<R, B, 2, FACTORED>(q, a, p, stream);
    if (need <= 4) return launch_backward_t<R, B, 4, FACTORED>(q, a, p, stream);
    if (need <= kMaxGwTiles) return launch_backward_t<R, B, kMaxGwTiles, FACTORED>(q, a, p, stream);
    return FC_ERR_UNSUPPORTED;
}

int backward_impl(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk,
                  void* ws, size_t ws_bytes, const fc_dims* d, bool factored, hipStream_t stream) {
    const BwdPlan p = plan_backward(d);
    if (!(factored ? p.ok_factored : p.ok)) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.gxp_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    BwdArgs a;
    BwdPtrs q;
    q.x = reinterpret_cast<const float2*>(x);
    q.gy = reinterpret_cast<const float2*>(gy);
    q.sten = sten;
    q.g = g;
    q.wpk = wpk;
    q.gxp = reinterpret_cast<float2*>(ws);
    q.gwp = reinterpret_cast<float2*>(static_cast<char*>(ws) + p.gxp_bytes);
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.g = p.g;
    a.ntiles = p.ntiles;
    a.ngw = p.ngw;
    { const char* e = getenv("FC_DEBUG_BWD"); a.dbg = e ? atoi(e) : 0; }
    int rc = FC_ERR_UNSUPPORTED;
#define FC_CASE(RR, BB)                                                              \
    if (d->R == RR && d->B == BB)                                                    \
        rc = factored ? launch_backward<RR, BB, true>(q, a, p, stream) : launch_backward<RR, BB, false>(q, a, p, stream);
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    return rc;
}

// Second stage of the backward pass: fixed-order sums of the per-frequency gx terms and the
// per-workgroup gW partials left in the workspace by backward_impl.
int backward_finish_impl(float* gx, float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BwdPlan p = plan_backward(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.gxp_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float2* gxp = reinterpret_cast<const float2*>(ws);
    const float2* gwp = reinterpret_cast<const float2*>(static_cast<char*>(ws) + p.gxp_bytes);
    const size_t NI = (size_t)d->N * d->I;
    hipLaunchKernelGGL(fc_reduce_gx_kernel, dim3((unsigned)((NI + 255) / 256)), dim3(256), 0, stream, gxp,
                       reinterpret_cast<float2*>(gx), NI, p.F);
    const int total = p.F * d->R * d->O * d->I;
    hipLaunchKernelGGL(fc_reduce_gw_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gwp,
                       reinterpret_cast<float2*>(gw_eff), p.P, p.F, d->R, d->O, d->I, p.KP, p.IP);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc
