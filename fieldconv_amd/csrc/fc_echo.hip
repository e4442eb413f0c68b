// ECHO descriptors for gfx950 (reference nn/echo.py:94-148): for every vertex n and channel c
// a histogram over the dS cells of a rasterised disk, hist[n,c,b] = sum over in-edges e of the bilinear votes of
// the point ln_e * exp(-i angle(x[src_e,c])) carrying the value x[src_e,c] * wxp_e; the descriptor is |hist|.
// The reference does this with `nonzero` compaction (host sync), four index_adds over E*C*4 votes and autograd;
// here: one target-centric kernel (a wavefront per vertex, lane = channel, lane-private histogram rows in LDS, no
// global atomics) and one source-centric kernel for the input gradient.
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kEchoWaves = 4;                 // vertices per workgroup
constexpr int kEchoAhead = 4;                 // edges whose gathers are in flight together
constexpr int kEchoMaxBins = 8;               // n_bins <= 8: (2*8+1)^2 = 289 raster cells, 213 of them inside the disk
constexpr int kEchoMaxCells = (2 * kEchoMaxBins + 1) * (2 * kEchoMaxBins + 1);

// Cells of the (2n+1)^2 raster inside the disk of radius n + 0.25, numbered in row-major order; cells outside alias
// bin 0 (reference nn/echo.py:11-27).  Returns the number of bins.
__host__ __device__ inline int echo_hist_dim(int n) {
    int d = 0;
    for (int i = -n; i <= n; ++i)
        for (int j = -n; j <= n; ++j) d += (16 * (i * i + j * j) <= (4 * n + 1) * (4 * n + 1)) ? 1 : 0;       // r^2 <= (n + 1/4)^2
    return d;
}
__device__ inline void echo_build_dmap(int* dmap, int n) {     // one thread
    const int w = 2 * n + 1;
    int d = 0;
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) {
            const bool in = 16 * ((i - n) * (i - n) + (j - n) * (j - n)) <= (4 * n + 1) * (4 * n + 1);
            dmap[i * w + j] = in ? d : 0;
            d += in ? 1 : 0;
        }
}

// Bilinear vote of the point p (complex, scaled to raster units q = p * n): four weights and four cells
// (reference nn/echo.py:30-61).  dq0[k], dq1[k]: derivatives of weight k with respect to Re q, Im q.
struct EchoVote {
    float w[4];
    int cell[4];
    float dq0[4], dq1[4];
};
__device__ __forceinline__ EchoVote echo_rasterize(float2 p, int n) {
    const float nf = (float)n;
    const float q0 = p.x * nf, q1 = p.y * nf;
    const float c0 = fminf(fmaxf(ceilf(q0), -nf), nf), c1 = fminf(fmaxf(ceilf(q1), -nf), nf);
    const float f0 = fminf(fmaxf(floorf(q0), -nf), nf), f1 = fminf(fmaxf(floorf(q1), -nf), nf);
    const float up0 = c0 - q0, up1 = c1 - q1, dn0 = q0 - f0, dn1 = q1 - f1;
    const int w = 2 * n + 1;
    const int ic0 = (int)c0 + n, ic1 = (int)c1 + n, if0 = (int)f0 + n, if1 = (int)f1 + n;
    EchoVote v;
    v.w[0] = up0 * up1; v.cell[0] = w * if0 + if1; v.dq0[0] = -up1; v.dq1[0] = -up0;
    v.w[1] = dn0 * dn1; v.cell[1] = w * ic0 + ic1; v.dq0[1] = dn1;  v.dq1[1] = dn0;
    v.w[2] = dn0 * up1; v.cell[2] = w * ic0 + if1; v.dq0[2] = up1;  v.dq1[2] = -dn0;
    v.w[3] = up0 * dn1; v.cell[3] = w * if0 + ic1; v.dq0[3] = -dn1; v.dq1[3] = up0;
    return v;
}

// frame = exp(-i softAngle(x)) (1 inside the origin box), reference nn/echo.py:113-118 / utils/field.py:40-48
__device__ __forceinline__ float2 echo_frame(float2 x, bool& live) {
    live = !is_origin(x);
    return unit_conj(x);
}

// ------------------------------------------------------------------------------------------ forward
// `wpv` (1, 2 or 4) wavefronts share one vertex: each takes every wpv-th in-edge into its own histogram, and the
// histograms are added in wavefront order at the end.  Meshes with few vertices and large supports (the reference's
// segmentation meshes: ~1k vertices, ~128 neighbours) would otherwise leave most of the chip idle.
__global__ __launch_bounds__(kEchoWaves * kWave) void echo_forward_kernel(
    const float2* __restrict__ x, const float2* __restrict__ ln_t, const float2* __restrict__ wxp_t,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr, float2* __restrict__ hist, float* __restrict__ desc,
    int N, int C, int n, int dS, int wpv, int ldc, int c0) {
    // (C channels [c0, c0 + C) of rows with ldc channels: wider inputs run as channel blocks of one entry-point call)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const dmap = reinterpret_cast<int*>(smem);                                  // [kEchoMaxCells]
    float* const hl = reinterpret_cast<float*>(smem) + kEchoMaxCells + 3;           // [waves][C][dS][2]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) echo_build_dmap(dmap, n);
    const int CS = C * dS;
    float* const mine = hl + (size_t)wave * CS * 2;
    for (int idx = lane; idx < CS * 2; idx += kWave) mine[idx] = 0.f;
    __syncthreads();
    const int v = (blockIdx.x * kEchoWaves + wave) / wpv;
    const int sub = wave % wpv;
    const bool active = v < N;
    const int beg = active ? rowptr[v] : 0, end = active ? rowptr[v + 1] : 0;
    const int cl = lane < C ? lane : 0;
    float* const row = mine + (size_t)cl * dS * 2;
    // The per-edge chain (edge -> source row -> vote) is latency bound: the source rows of kEchoAhead edges are
    // requested together, then consumed.
    for (int e0 = beg + sub; e0 < end; e0 += kEchoAhead * wpv) {
        float2 xs[kEchoAhead], les[kEchoAhead], wes[kEchoAhead];
#pragma unroll
        for (int u = 0; u < kEchoAhead; ++u) {
            const int e = min(e0 + u * wpv, end - 1);
            les[u] = ln_t[e];
            wes[u] = wxp_t[e];
            xs[u] = x[(size_t)nbr[e] * ldc + c0 + cl];
        }
#pragma unroll
        for (int u = 0; u < kEchoAhead; ++u) {
            if (e0 + u * wpv >= end) break;
            const float2 xv = xs[u];
            bool live;
            const float2 fr = echo_frame(xv, live);
            const EchoVote vt = echo_rasterize(cmul(les[u], fr), n);
            const float2 xw = live ? cmul(xv, wes[u]) : make_float2(0.f, 0.f);
            if (lane < C) {
                // lane-private row, plain read-modify-write.  The four bins are read together, updated and written
                // together (one LDS round trip instead of four dependent ones).  Votes that fall into the same bin (a
                // coordinate on a raster line or clamped at the rim) first pool their weights, so every write of a
                // shared bin carries the full sum.
                int b[4];
                float2 h[4];
                float w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) b[k] = dmap[vt.cell[k]];
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = *reinterpret_cast<const float2*>(row + 2 * b[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w[k] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[k] += (b[j] == b[k]) ? vt.w[j] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float2*>(row + 2 * b[k]) = make_float2(h[k].x + xw.x * w[k], h[k].y + xw.y * w[k]);
            }
        }
    }
    if (wpv > 1) __syncthreads();
    if (!active) return;
    // the block [C][dS] of this vertex, contiguous in LDS and in the outputs; the vertex's wavefronts share the rows
    float2* const hout = hist + ((size_t)v * ldc + c0) * dS;
    float* const dout = desc + ((size_t)v * ldc + c0) * dS;
    const float* const first = hl + (size_t)(wave - sub) * CS * 2;
    for (int idx = sub * kWave + lane; idx < CS; idx += wpv * kWave) {
        float2 h = make_float2(0.f, 0.f);
        for (int s = 0; s < wpv; ++s) {
            h.x += first[(size_t)s * CS * 2 + 2 * idx];
            h.y += first[(size_t)s * CS * 2 + 2 * idx + 1];
        }
        hout[idx] = h;
        dout[idx] = is_origin(h) ? 0.f : sqrtf(h.x * h.x + h.y * h.y);        // softAbs, reference utils/field.py:29-37
    }
}

// ------------------------------------------------------------------------------------------ backward
// gh[n,c,b] = g_desc * hist / |hist| (0 where hist is inside the origin box): the gradient arriving at the histogram
__global__ void echo_hist_grad_kernel(const float2* __restrict__ hist, const float* __restrict__ g_desc, float2* __restrict__ gh,
                                      size_t count) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const float2 h = hist[idx];
    float2 out = make_float2(0.f, 0.f);
    if (!is_origin(h)) {
        const float s = g_desc[idx] * __frsqrt_rn(h.x * h.x + h.y * h.y);
        out = make_float2(h.x * s, h.y * s);
    }
    gh[idx] = out;
}

// gx[j,c] = sum over out-edges e of j: conj(wxp_e) sum_k w_k gh_k      (through the vote's value)
//         + the angle term: the votes' weights depend on q = n * ln_e * frame, frame = exp(-i angle(x[j,c]))
// `wpv` wavefronts share a source vertex as in the forward kernel; their sums are added in wavefront order.
__global__ __launch_bounds__(kEchoWaves * kWave) void echo_backward_kernel(
    const float2* __restrict__ x, const float2* __restrict__ ln_s, const float2* __restrict__ wxp_s,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr, const float2* __restrict__ gh_all,
    float2* __restrict__ gx, int N, int C, int n, int dS, int wpv, int ldc, int c0) {
    __shared__ int dmap[kEchoMaxCells];
    __shared__ float4 s_part[kEchoWaves][kWave];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) echo_build_dmap(dmap, n);
    __syncthreads();
    const int j = (blockIdx.x * kEchoWaves + wave) / wpv;
    const int sub = wave % wpv;
    const bool active = j < N;
    const int beg = active ? rowptr[j] : 0, end = active ? rowptr[j + 1] : 0;
    const int cl = lane < C ? lane : 0;
    const float2 xv = active ? x[(size_t)j * ldc + c0 + cl] : make_float2(1.f, 0.f);
    bool live;
    const float2 fr = echo_frame(xv, live);
    float2 gval = make_float2(0.f, 0.f);        // gradient through the vote values
    float2 gframe = make_float2(0.f, 0.f);      // gradient with respect to frame
    // (bound by the L1 line rate: every edge touches four scattered entries per channel of the target's block)
    for (int e = beg + sub; e < end; e += wpv) {
        const int dst = nbr[e];
        const float2 le = ln_s[e], we = wxp_s[e];
        const EchoVote vt = echo_rasterize(cmul(le, fr), n);
        const float2 xw = cmul(xv, we);
        const float2* const ghrow = gh_all + ((size_t)dst * ldc + c0 + cl) * dS;
        float2 acc = make_float2(0.f, 0.f);
        float gq0 = 0.f, gq1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 gh = ghrow[dmap[vt.cell[k]]];
            acc.x += vt.w[k] * gh.x;
            acc.y += vt.w[k] * gh.y;
            const float t = gh.x * xw.x + gh.y * xw.y;          // Re(conj(gh) xw)
            gq0 += t * vt.dq0[k];
            gq1 += t * vt.dq1[k];
        }
        const float2 gv = cmul_conj(acc, we);                   // acc * conj(wxp)
        gval.x += gv.x;
        gval.y += gv.y;
        const float2 ga = make_float2((float)n * gq0, (float)n * gq1);      // gradient with respect to aligned = ln * frame
        const float2 gf = cmul_conj(ga, le);                    // ga * conj(ln)
        gframe.x += gf.x;
        gframe.y += gf.y;
    }
    if (wpv > 1) {
        s_part[wave][lane] = make_float4(gval.x, gval.y, gframe.x, gframe.y);
        __syncthreads();
        if (sub != 0) return;
        gval = make_float2(0.f, 0.f);
        gframe = gval;
        for (int s = 0; s < wpv; ++s) {
            const float4 p = s_part[wave + s][lane];
            gval.x += p.x; gval.y += p.y; gframe.x += p.z; gframe.y += p.w;
        }
    }
    if (!active) return;
    float2 out = make_float2(0.f, 0.f);
    if (live) {
        // frame = exp(-i theta): dL/dtheta = Im(conj(gframe) frame); theta = angle(x): gx += dL/dtheta * i x / |x|^2
        const float gth = gframe.x * fr.y - gframe.y * fr.x;
        const float inv2 = 1.f / (xv.x * xv.x + xv.y * xv.y);
        out = make_float2(gval.x - xv.y * gth * inv2, gval.y + xv.x * gth * inv2);
    }
    if (lane < C) gx[(size_t)j * ldc + c0 + lane] = out;
}

// wavefronts per vertex: spread large supports over the workgroup when the mesh alone cannot fill the chip
static int echo_waves_per_vertex(int N, int E) {
    static const int forced = [] { const char* e = dev_env("FC_ECHO_WPV"); return e ? atoi(e) : 0; }();       // development, read once
    if (forced) return forced;
    const long deg = N > 0 ? (long)E / N : 0;
    if (deg >= 64 && N < 65536) return 4;
    if (deg >= 32 && N < 131072) return 2;
    return 1;
}

}  // namespace fc

extern "C" {

int fc_echo_hist_dim(int32_t n_bins) { return (n_bins >= 1 && n_bins <= fc::kEchoMaxBins) ? fc::echo_hist_dim(n_bins) : 0; }

// Channels one launch can take: the forward kernel keeps a workgroup's histograms [vertices][C][dS] in LDS.
int fc_echo_channel_block(int32_t n_bins) {
    const int dS = fc_echo_hist_dim(n_bins);
    if (dS == 0) return 0;
    const long floats = (long)(fc::kMaxLds / sizeof(float)) - fc::kEchoMaxCells - 3;
    const long c = floats / ((long)fc::kEchoWaves * dS * 2);
    return (int)(c > fc::kWave ? fc::kWave : c);
}

int fc_echo_forward(const float* x, const float* ln_t, const float* wxp_t, const fc_csr* by_target, float* hist, float* desc,
                    int32_t N, int32_t E, int32_t C, int32_t n_bins, void* stream) {
    if (!x || !by_target || !by_target->rowptr || !hist || !desc || N <= 0 || E < 0 || C <= 0) return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!ln_t || !wxp_t || !by_target->nbr)) return FC_ERR_BAD_ARGUMENT;
    if (n_bins < 1 || n_bins > fc::kEchoMaxBins) return FC_ERR_UNSUPPORTED;
    const int dS = fc::echo_hist_dim(n_bins);
    const int blk = fc_echo_channel_block(n_bins);       // channels per launch: one per lane, a workgroup's histograms in LDS
    auto kern = fc::echo_forward_kernel;
    static bool lds_ok[fc::kMaxDevices] = {};
    const int wpv = fc::echo_waves_per_vertex(N, E);
    const int per_wg = fc::kEchoWaves / wpv;
    for (int c0 = 0; c0 < C; c0 += blk) {               // the channels are independent: blocks of `blk`
        const int cb = C - c0 < blk ? C - c0 : blk;
        const size_t lds = (size_t)(fc::kEchoMaxCells + 3 + fc::kEchoWaves * cb * dS * 2) * sizeof(float);
        if (lds > fc::kMaxLds) return FC_ERR_UNSUPPORTED;
        if (!fc::allow_full_lds(reinterpret_cast<const void*>(kern), lds, lds_ok)) return FC_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3((N + per_wg - 1) / per_wg), dim3(fc::kEchoWaves * fc::kWave), lds,
                           static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(ln_t),
                           reinterpret_cast<const float2*>(wxp_t), by_target->rowptr, by_target->nbr, reinterpret_cast<float2*>(hist),
                           desc, N, cb, n_bins, dS, wpv, C, c0);
    }
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_echo_backward(const float* x, const float* ln_s, const float* wxp_s, const fc_csr* by_source, const float* hist,
                     const float* g_desc, float* gx, float* hist_grad_workspace, int32_t N, int32_t E, int32_t C, int32_t n_bins,
                     void* stream) {
    if (!x || !by_source || !by_source->rowptr || !hist || !g_desc || !gx || !hist_grad_workspace || N <= 0 || E < 0 || C <= 0)
        return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!ln_s || !wxp_s || !by_source->nbr)) return FC_ERR_BAD_ARGUMENT;
    if (n_bins < 1 || n_bins > fc::kEchoMaxBins) return FC_ERR_UNSUPPORTED;
    const int dS = fc::echo_hist_dim(n_bins);
    const size_t count = (size_t)N * C * dS;
    float2* const gh = reinterpret_cast<float2*>(hist_grad_workspace);
    hipLaunchKernelGGL(fc::echo_hist_grad_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(hist), g_desc, gh, count);
    const int wpv = fc::echo_waves_per_vertex(N, E);
    const int per_wg = fc::kEchoWaves / wpv;
    for (int c0 = 0; c0 < C; c0 += fc::kWave) {         // one channel per lane: blocks of 64
        const int cb = C - c0 < fc::kWave ? C - c0 : fc::kWave;
        hipLaunchKernelGGL(fc::echo_backward_kernel, dim3((N + per_wg - 1) / per_wg), dim3(fc::kEchoWaves * fc::kWave), 0,
                           static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(ln_s),
                           reinterpret_cast<const float2*>(wxp_s), by_source->rowptr, by_source->nbr, gh, reinterpret_cast<float2*>(gx), N, cb,
                           n_bins, dS, wpv, C, c0);
    }
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
