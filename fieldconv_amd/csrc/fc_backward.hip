// FieldConv backward: filter-gradient kernel, the reductions, the fp32-MFMA instantiation of the data
// kernel and the mode dispatch (data kernel: fc_backward_kernels.hpp).
#include "fc_backward_kernels.hpp"

namespace fc {

// ---------------------------------------------------------------------------------- filter gradient
// T = 16x16 complex gW tiles owned by each wavefront (ceil(ngw / 16)).
// The complex product H conj(X), H = a + ib, X = c + id, uses three real products instead of four
// (Gauss): k1 = (a+b) c, k2 = a (c+d), k3 = b (c-d); re = k1 - k3 = ac + bd, im = k1 - k2 = bc - ad.
// The X combinations are formed once per tile when the rotated features go to LDS, a+b costs one add
// per fragment element: 12 MFMAs per 16 vertices and tile instead of 16, 12*T accumulator VGPRs.
template <int T>
__global__ __launch_bounds__(kThreads) void fc_backward_filter_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ hdump, float2* __restrict__ ggwp /* [P][F][KP][IP] */,
    const BwdArgs a, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.g;
    const int KD = a.KD, KP = mg.KP, IP = mg.MP, I = a.I;
    float* const slab0 = reinterpret_cast<float*>(smem);               // [2][slab_stride]: H of a tile, [16 vertices][KD] interleaved (re, im)
    // rotated features of the tile, rows of kXtStride = 20 floats (16 vertices + pad): the 16-byte B-fragment
    // reads of 16 consecutive rows then fall into 16 distinct bank groups
    float* const xtr = slab0 + 2 * a.slab_stride;                      // [IP][20]  c
    float* const xts = xtr + IP * kXtStride;                           // [IP][20]  c + d
    float* const xtd = xts + IP * kXtStride;                           // [IP][20]  c - d

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int m = f - B;
    const int fr = lane & 15, fq = lane >> 4;

    for (int idx = tid; idx < 3 * IP * kXtStride; idx += kThreads) xtr[idx] = 0.f;

    // my gW tiles: u = wave + 16 n  ->  (row tile rt over k = (r,o), column tile ct over i)
    int gw_h[T], gw_x[T];       // wave-uniform LDS offsets, -1 when the slot is unused
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int u = wave + kWaves * n;
        const int rt = u / mg.NMT, ct = u - rt * mg.NMT;
        gw_h[n] = (u < a.ngw) ? rt * 32 : -1;            // floats: 16 complex entries per row tile
        gw_x[n] = ct * 16 * kXtStride;
    }
    const int h_lane = (4 * fq) * KD + 2 * fr;    // A fragment: H[vertex 4fq+s][k = rt*16 + fr], one 8-byte (re, im) read
    const int x_lane = fr * kXtStride + 4 * fq;   // B fragment: xt[i = ct*16 + fr][vertex 4fq..4fq+3]

    f32x4 k1[T], k2[T], k3[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { k1[n] = f32x4{0.f, 0.f, 0.f, 0.f}; k2[n] = k1[n]; k3[n] = k1[n]; }

    const int npieces = a.slab_stride / 256;      // 1 KiB DMA pieces per slab
    auto dma_slab = [&](const int tile, const int buf) {
        const float* src = hdump + ((size_t)tile * F + f) * a.slab_stride;
        if (!(a.dbg & 16))
        for (int p = wave; p < npieces; p += kWaves)
            lds_dma16_untracked(src + p * 256 + lane * 4, slab0 + buf * a.slab_stride + p * 256);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of slab `buf` have landed
        __syncthreads();            // everyone's have; the previous tile's reads are done
        if (lane < IP) {
            xtr[lane * kXtStride + wave] = xt.x;
            xts[lane * kXtStride + wave] = xt.x + xt.y;
            xtd[lane * kXtStride + wave] = xt.x - xt.y;
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

        const float* hsl = slab0 + buf * a.slab_stride;
        // gW += H^T . conj(xt)
        if (!(a.dbg & 4)) {
#pragma unroll
            for (int n = 0; n < T; ++n) {
                if (gw_h[n] >= 0) {
                    const float4 c4 = *reinterpret_cast<const float4*>(xtr + gw_x[n] + x_lane);
                    const float4 s4 = *reinterpret_cast<const float4*>(xts + gw_x[n] + x_lane);
                    const float4 d4 = *reinterpret_cast<const float4*>(xtd + gw_x[n] + x_lane);
                    const float* ha = hsl + gw_h[n] + h_lane;
                    const float2 h0 = *reinterpret_cast<const float2*>(ha), h1 = *reinterpret_cast<const float2*>(ha + KD);
                    const float2 h2 = *reinterpret_cast<const float2*>(ha + 2 * KD), h3 = *reinterpret_cast<const float2*>(ha + 3 * KD);
                    const float a0 = h0.x, a1 = h1.x, a2 = h2.x, a3 = h3.x;
                    const float b0 = h0.y, b1 = h1.y, b2 = h2.y, b3 = h3.y;
                    k1[n] = mfma16(a0 + b0, c4.x, k1[n]); k2[n] = mfma16(a0, s4.x, k2[n]); k3[n] = mfma16(b0, d4.x, k3[n]);
                    k1[n] = mfma16(a1 + b1, c4.y, k1[n]); k2[n] = mfma16(a1, s4.y, k2[n]); k3[n] = mfma16(b1, d4.y, k3[n]);
                    k1[n] = mfma16(a2 + b2, c4.z, k1[n]); k2[n] = mfma16(a2, s4.z, k2[n]); k3[n] = mfma16(b2, d4.z, k3[n]);
                    k1[n] = mfma16(a3 + b3, c4.w, k1[n]); k2[n] = mfma16(a3, s4.w, k2[n]); k3[n] = mfma16(b3, d4.w, k3[n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // flush my gW partial
#pragma unroll
    for (int n = 0; n < T; ++n) {
        if (gw_h[n] >= 0) {
            const int ct16 = gw_x[n] / kXtStride;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = gw_h[n] / 2 + 4 * fq + jj;
                const int i = ct16 + fr;
                ggwp[(((size_t)blockIdx.x * F + f) * KP + k) * IP + i] = make_float2(k1[n][jj] - k3[n][jj], k1[n][jj] - k2[n][jj]);
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
