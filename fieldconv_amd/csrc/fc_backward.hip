// FieldConv backward: filter-gradient kernel, the reductions, the fp32-MFMA instantiation of the data
// kernel and the mode dispatch (data kernel: fc_backward_kernels.hpp).
#include "fc_backward_kernels.hpp"

namespace fc {

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
    // fixed summation order, eight independent loads in flight
    float2 s = make_float2(0.f, 0.f);
    const size_t stride = (size_t)F * KP * IP;
    const float2* src = gwp + ((size_t)f * KP + k) * IP + i;
    int p = 0;
    for (; p + 8 <= P; p += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; }
    }
    for (; p < P; ++p) {
        const float2 v = src[(size_t)p * stride];
        s.x += v.x;
        s.y += v.y;
    }
    const float sc = 1.f / (float)F;
    const int r = k / O, o = k - r * O;
    gw[(((size_t)o * I + i) * R + r) * F + f] = make_float2(s.x * sc, s.y * sc);
}

size_t backward_workspace_bytes(const fc_dims* d) {
    const BwdPlan p = plan_backward(d, split_mode());
    return p.hdump_bytes + p.gwp_bytes + 256;
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
    const BwdPlan p = plan_backward(d, split_mode());
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
    const BwdPlan p = plan_backward(d, split_mode());
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float2* gwp = reinterpret_cast<const float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const int total = p.F * d->R * d->O * d->I;
    hipLaunchKernelGGL(fc_reduce_gw_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gwp,
                       reinterpret_cast<float2*>(gw_eff), p.P, p.F, d->R, d->O, d->I, p.KP, p.IP);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

#define FC_BWD_DATA_ARGS const float*, const float*, const float*, const fc_csr*, const float*, float*, void*, size_t, const fc_dims*, bool, hipStream_t
template int backward_data_impl_mode<false>(FC_BWD_DATA_ARGS);
extern template int backward_data_impl_mode<true>(FC_BWD_DATA_ARGS);

int backward_data_impl(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk, float* gx,
                       void* ws, size_t ws_bytes, const fc_dims* d, bool factored, hipStream_t stream) {
    return split_mode() ? backward_data_impl_mode<true>(x, gy, sten, g, wpk, gx, ws, ws_bytes, d, factored, stream)
                        : backward_data_impl_mode<false>(x, gy, sten, g, wpk, gx, ws, ws_bytes, d, factored, stream);
}

}  // namespace fc
