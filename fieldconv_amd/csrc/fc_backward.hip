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
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kMaxGwTiles = 8;   // 16x16 complex gW tiles a wavefront can own

// (pointers are separate __restrict__ kernel parameters, see fc_forward.hip)
struct BwdArgs {
    int N, I, O;
    int IP, KP, KS;      // IP = ceil16(I), KP = ceil16(R*O)
    int NIT, NKP, KST;   // input-channel tiles, k partitions, 16-wide k blocks
    int ntiles;
    int ngw;             // KST * NIT gW tiles per frequency
};

// T = 16x16 complex gW tiles owned by each wavefront (ceil(ngw / 16)); 8*T accumulator VGPRs.
template <int R, int B, int T>
__global__ __launch_bounds__(kThreads) void fc_backward_kernel(
    const float2* __restrict__ gx_, const float2* __restrict__ ggy, const float* __restrict__ gsten,
    const int32_t* __restrict__ growptr, const int32_t* __restrict__ gnbr,
    const float* __restrict__ gwpk, float2* __restrict__ ggxp /* [F][N][I] */,
    float2* __restrict__ ggwp /* [P][F][KP][IP] */, const BwdArgs a) {
    constexpr int F = 2 * B + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = a.KS, KP = a.KP, IP = a.IP, I = a.I, O = a.O;
    float* const hre = reinterpret_cast<float*>(smem);     // [16][KS]
    float* const him = hre + kTile * KS;                    // [16][KS]
    float* const xtr = him + kTile * KS;                    // [IP][16]
    float* const xti = xtr + IP * kTile;                    // [IP][16]
    float* const part = xti + IP * kTile;                   // [NKP][16][IP][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int m = f - B;

    for (int idx = tid; idx < 2 * kTile * KS + 2 * IP * kTile; idx += kThreads) hre[idx] = 0.f;
    __syncthreads();

    const int it = wave % a.NIT;
    const int kp = wave / a.NIT;
    const bool mma_active = kp < a.NKP;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const bool has_o = lane < O;
    const int ol = has_o ? lane : 0;

    // my gW tiles: u = wave + 16 n  ->  (row tile rt over k = (r,o), column tile ct over i)
    int gw_h[T], gw_x[T];       // wave-uniform LDS offsets, -1 when the slot is unused
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int u = wave + kWaves * n;
        const int rt = u / a.NIT, ct = u - rt * a.NIT;
        gw_h[n] = (u < a.ngw) ? rt * 16 : -1;
        gw_x[n] = ct * 16 * kTile;
    }
    const int h_lane = (4 * fq) * KS + fr;        // A fragment: H[vertex 4fq+s][k = rt*16 + fr]
    const int x_lane = fr * kTile + 4 * fq;       // B fragment: xt[i = ct*16 + fr][vertex 4fq..4fq+3]

    f32x4 gre[T], gim[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        // ------------------------------------------------------------ gather H for my source
        float hr[R], hi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { hr[r] = 0.f; hi[r] = 0.f; }
        const int j = tile * kTile + wave;
        int beg = 0, end = 0;
        if (j < a.N) { beg = growptr[j]; end = growptr[j + 1]; }

        int d_next = 0, row_next = 0;
        float2 g_next = make_float2(0.f, 0.f);
        if (beg < end) {
            d_next = gnbr[beg];
            row_next = beg;
            g_next = ggy[(size_t)d_next * O + ol];
        }
        for (int e = beg; e < end; ++e) {
            const int row = __builtin_amdgcn_readfirstlane(row_next);
            float2 g = g_next;
            if (e + 1 < end) {
                d_next = gnbr[e + 1];
                row_next = e + 1;
                g_next = ggy[(size_t)d_next * O + ol];
            }
            if (!has_o) g = make_float2(0.f, 0.f);
            const float* __restrict__ S = gsten + (size_t)row * (2 * R * F) + 2 * f;   // wave-uniform
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float sr = S[2 * r * F];
                const float si = S[2 * r * F + 1];
                hr[r] = fmaf(g.x, sr, hr[r]);
                hr[r] = fmaf(g.y, si, hr[r]);
                hi[r] = fmaf(g.y, sr, hi[r]);
                hi[r] = fmaf(-g.x, si, hi[r]);
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
                hre[wave * KS + r * O + lane] = hr[r];
                him[wave * KS + r * O + lane] = hi[r];
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ (a) gxt = H . conj(W)/F
        if (mma_active) {
            f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;
            const float* wre = gwpk + ((size_t)(f * 2 + 0) * IP + it * 16 + fr) * KP + 4 * fq;
            const float* wim = gwpk + ((size_t)(f * 2 + 1) * IP + it * 16 + fr) * KP + 4 * fq;
            const float* bre = hre + fr * KS + 4 * fq;
            const float* bim = him + fr * KS + 4 * fq;
            for (int kb = kp; kb < a.KST; kb += a.NKP) {
                const float4 wr = *reinterpret_cast<const float4*>(wre + 16 * kb);
                const float4 wi = *reinterpret_cast<const float4*>(wim + 16 * kb);
                const float4 br = *reinterpret_cast<const float4*>(bre + 16 * kb);
                const float4 bi = *reinterpret_cast<const float4*>(bim + 16 * kb);
                acc_re = mfma16(wr.x, br.x, acc_re); acc_im = mfma16(wi.x, br.x, acc_im);
                acc_re = mfma16(-wi.x, bi.x, acc_re); acc_im = mfma16(wr.x, bi.x, acc_im);
                acc_re = mfma16(wr.y, br.y, acc_re); acc_im = mfma16(wi.y, br.y, acc_im);
                acc_re = mfma16(-wi.y, bi.y, acc_re); acc_im = mfma16(wr.y, bi.y, acc_im);
                acc_re = mfma16(wr.z, br.z, acc_re); acc_im = mfma16(wi.z, br.z, acc_im);
                acc_re = mfma16(-wi.z, bi.z, acc_re); acc_im = mfma16(wr.z, bi.z, acc_im);
                acc_re = mfma16(wr.w, br.w, acc_re); acc_im = mfma16(wi.w, br.w, acc_im);
                acc_re = mfma16(-wi.w, bi.w, acc_re); acc_im = mfma16(wr.w, bi.w, acc_im);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int i = it * 16 + 4 * fq + jj;
                float* p = part + ((size_t)(kp * kTile + fr) * IP + i) * 2;
                p[0] = acc_re[jj];
                p[1] = acc_im[jj];
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ------------------------------------------------------------ (b) gW += H^T . conj(xt)
        // re += Hre*Xre + Him*Xim ; im += Him*Xre - Hre*Xim
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
        __syncthreads();

        // ------------------------------------------------------------ gx term of this frequency
        for (int idx = tid; idx < kTile * I; idx += kThreads) {
            const int v = idx / I, i = idx - v * I;
            const int jn = tile * kTile + v;
            if (jn >= a.N) continue;
            float zr = 0.f, zi = 0.f;
            for (int q = 0; q < a.NKP; ++q) {
                const float* p = part + ((size_t)(q * kTile + v) * IP + i) * 2;
                zr += p[0];
                zi += p[1];
            }
            const float2 z = make_float2(zr, zi);
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
    int IP, KP, KS, NIT, NKP, KST, ntiles, ngw, P, F;
    size_t lds, gxp_bytes, gwp_bytes;
    bool ok;
};

static BwdPlan plan_backward(const fc_dims* d) {
    BwdPlan p;
    p.F = 2 * d->B + 1;
    p.IP = round_up(d->I, 16);
    p.KP = round_up(d->R * d->O, 16);
    p.KS = slab_stride(p.KP);
    p.NIT = p.IP / 16;
    p.KST = p.KP / 16;
    p.NKP = kWaves / p.NIT;
    if (p.NKP > p.KST) p.NKP = p.KST;
    p.ntiles = (d->N + kTile - 1) / kTile;
    p.ngw = p.KST * p.NIT;
    int P = 256 / p.F;                       // one workgroup per CU across the F frequency slices
    if (P < 1) P = 1;
    if (P > p.ntiles) P = p.ntiles;
    p.P = P;
    p.lds = (size_t)(2 * kTile * p.KS + 2 * p.IP * kTile + p.NKP * kTile * p.IP * 2) * sizeof(float);
    p.gxp_bytes = (((size_t)p.F * d->N * d->I * sizeof(float2) + 255) / 256) * 256;
    p.gwp_bytes = (size_t)p.P * p.F * p.KP * p.IP * sizeof(float2);
    p.ok = p.lds <= kMaxLds && p.ngw <= kMaxGwTiles * kWaves && p.NIT <= kWaves;
    return p;
}

size_t backward_workspace_bytes(const fc_dims* d) {
    const BwdPlan p = plan_backward(d);
    return p.gxp_bytes + p.gwp_bytes + 256;
}

struct BwdPtrs {
    const float2* x; const float2* gy; const float* sten; const fc_csr* g; const float* wpk; float2* gxp; float2* gwp;
};

template <int R, int B, int T>
static int launch_backward_t(const BwdPtrs& q, const BwdArgs& a, const BwdPlan& p, hipStream_t stream) {
    auto kern = fc_backward_kernel<R, B, T>;
    if (p.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)p.lds) != hipSuccess)
            return FC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.P, p.F), dim3(kThreads), p.lds, stream, q.x, q.gy, q.sten, q.g->rowptr, q.g->nbr,
                       q.wpk, q.gxp, q.gwp, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <int R, int B>
static int launch_backward(const BwdPtrs& q, const BwdArgs& a, const BwdPlan& p, hipStream_t stream) {
    const int need = (p.ngw + kWaves - 1) / kWaves;
    if (need <= 2) return launch_backward_t<R, B, 2>(q, a, p, stream);
    if (need <= 4) return launch_backward_t<R, B, 4>(q, a, p, stream);
    if (need <= kMaxGwTiles) return launch_backward_t<R, B, kMaxGwTiles>(q, a, p, stream);
    return FC_ERR_UNSUPPORTED;
}

int backward_impl(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk,
                  void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BwdPlan p = plan_backward(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
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
    a.IP = p.IP; a.KP = p.KP; a.KS = p.KS;
    a.NIT = p.NIT; a.NKP = p.NKP; a.KST = p.KST;
    a.ntiles = p.ntiles;
    a.ngw = p.ngw;
    int rc = FC_ERR_UNSUPPORTED;
#define FC_CASE(RR, BB) if (d->R == RR && d->B == BB) rc = launch_backward<RR, BB>(q, a, p, stream);
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
