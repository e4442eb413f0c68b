// Per-vertex tangent-feature operators for gfx950:
//   TangentLin    (reference nn/tangent_lin.py:27-29)   dense complex channel mix on MFMA
//   TangentNonLin (reference nn/tangent_nonlin.py:24-35) modReLU with the origin-box rule
// plus their adjoints (the reference leaves those to torch autograd).
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

// ------------------------------------------------------------------------------------------
// out[n, m] = sum_k in[n, k] * Wc(m, k)          (complex, no bias)
//   TRANSPOSED = false : Wc(m,k) = Re[m*ldw + k] + i Im[m*ldw + k]            (forward, m = o, k = i)
//   TRANSPOSED = true  : Wc(m,k) = Re[k*ldw + m] - i Im[k*ldw + m]            (input grad, m = i, k = o)
// Real-expanded on v_mfma_f32_16x16x4_f32: the interleaved (re,im) input row is the B operand as it
// lies in memory (k_real = 2k + c), the A operand rows are [Wre, -Wim] for the real part of the
// output and [Wim, Wre] for the imaginary part.  One wavefront = 16 vertices x all output tiles.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void tangent_lin_kernel(const float2* __restrict__ in, const float* __restrict__ wre,
                                                          const float* __restrict__ wim, float2* __restrict__ out,
                                                          int N, int K, int M, int ldw) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = wave * 16;
    if (n0 >= N) return;
    const int n = n0 + fr;
    const bool nvalid = n < N;
    const int MT = (M + 15) / 16;
    const int KB = (2 * K + 15) / 16;
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + fr;
        f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;
        for (int kb = 0; kb < KB; ++kb) {
            const int k0 = 8 * kb + 2 * fq;        // two complex k per lane: k0, k0 + 1
            float2 b0 = make_float2(0.f, 0.f), b1 = b0;
            if (nvalid && k0 < K) b0 = in[(size_t)n * K + k0];
            if (nvalid && k0 + 1 < K) b1 = in[(size_t)n * K + k0 + 1];
            float r0 = 0.f, i0 = 0.f, r1 = 0.f, i1 = 0.f;
            if (m < M) {
                if (k0 < K) {
                    const size_t w = TRANSPOSED ? (size_t)k0 * ldw + m : (size_t)m * ldw + k0;
                    r0 = wre[w];
                    i0 = TRANSPOSED ? -wim[w] : wim[w];
                }
                if (k0 + 1 < K) {
                    const size_t w = TRANSPOSED ? (size_t)(k0 + 1) * ldw + m : (size_t)m * ldw + k0 + 1;
                    r1 = wre[w];
                    i1 = TRANSPOSED ? -wim[w] : wim[w];
                }
            }
            acc_re = mfma16(r0, b0.x, acc_re);  acc_im = mfma16(i0, b0.x, acc_im);
            acc_re = mfma16(-i0, b0.y, acc_re); acc_im = mfma16(r0, b0.y, acc_im);
            acc_re = mfma16(r1, b1.x, acc_re);  acc_im = mfma16(i1, b1.x, acc_im);
            acc_re = mfma16(-i1, b1.y, acc_re); acc_im = mfma16(r1, b1.y, acc_im);
        }
        // D layout: column = vertex fr, rows = outputs mt*16 + 4*fq + j
        if (nvalid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mo = mt * 16 + 4 * fq + j;
                if (mo < M) out[(size_t)n * M + mo] = make_float2(acc_re[j], acc_im[j]);
            }
        }
    }
}

// Weight gradient gW[o,i] = sum_n gy[n,o] conj(x[n,i]) on MFMA with the vertices as the k dimension.
// Each wavefront walks every `stride`-th block of 16 vertices and keeps all (o-tile, i-tile)
// accumulators; partial[wave][o][i] is reduced by tangent_lin_gw_reduce_kernel in a fixed order.
template <int MAXT>   // max tiles per side (channels <= 16*MAXT)
__global__ __launch_bounds__(256) void tangent_lin_gw_kernel(const float2* __restrict__ x, const float2* __restrict__ gy,
                                                             float2* __restrict__ partial, int N, int I, int O,
                                                             int nwaves_total) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int OT = (O + 15) / 16, IT = (I + 15) / 16;
    f32x4 are[MAXT][MAXT], aim[MAXT][MAXT];
#pragma unroll
    for (int a = 0; a < MAXT; ++a)
#pragma unroll
        for (int b = 0; b < MAXT; ++b) { are[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; aim[a][b] = are[a][b]; }

    const int nblocks = (N + 15) / 16;
    for (int blk = wave; blk < nblocks; blk += nwaves_total) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int n = blk * 16 + 4 * fq + s;
            float2 g[MAXT], v[MAXT];
#pragma unroll
            for (int a = 0; a < MAXT; ++a) {
                const int o = a * 16 + fr, i = a * 16 + fr;
                g[a] = (n < N && a < OT && o < O) ? gy[(size_t)n * O + o] : make_float2(0.f, 0.f);
                v[a] = (n < N && a < IT && i < I) ? x[(size_t)n * I + i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int a = 0; a < MAXT; ++a)
#pragma unroll
                for (int b = 0; b < MAXT; ++b) {
                    if (a < OT && b < IT) {
                        // re += g.re x.re + g.im x.im ; im += g.im x.re - g.re x.im
                        are[a][b] = mfma16(g[a].x, v[b].x, are[a][b]); aim[a][b] = mfma16(g[a].y, v[b].x, aim[a][b]);
                        are[a][b] = mfma16(g[a].y, v[b].y, are[a][b]); aim[a][b] = mfma16(-g[a].x, v[b].y, aim[a][b]);
                    }
                }
        }
    }
#pragma unroll
    for (int a = 0; a < MAXT; ++a)
#pragma unroll
        for (int b = 0; b < MAXT; ++b) {
            if (a < OT && b < IT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = a * 16 + 4 * fq + j, i = b * 16 + fr;
                    if (o < O && i < I) partial[((size_t)wave * O + o) * I + i] = make_float2(are[a][b][j], aim[a][b][j]);
                }
            }
        }
}

__global__ void tangent_lin_gw_reduce_kernel(const float2* __restrict__ partial, float* __restrict__ g_re,
                                             float* __restrict__ g_im, int nparts, int OI) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= OI) return;
    float re = 0.f, im = 0.f;
    for (int p = 0; p < nparts; ++p) {
        const float2 v = partial[(size_t)p * OI + idx];
        re += v.x;
        im += v.y;
    }
    g_re[idx] = re;
    g_im[idx] = im;
}

// ------------------------------------------------------------------------------------------
// modReLU.  Outside the origin box: y = relu(|x| + b) * x/|x| (reference computes polar(relu(r+b),
// angle(x)), tangent_nonlin.py:30-33); inside it the entry passes through (:26 clone).
__global__ void tangent_nonlin_fwd_kernel(const float2* __restrict__ x, const float* __restrict__ bias,
                                          float2* __restrict__ y, size_t total, int C) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float2 v = x[idx];
    float2 o = v;
    if (!is_origin(v)) {
        const float r = sqrtf(v.x * v.x + v.y * v.y);
        const float f = fmaxf(r + bias[idx % C], 0.f);
        const float s = f / r;
        o = make_float2(v.x * s, v.y * s);
    }
    y[idx] = o;
}

// gx = e (f'(r) g_r + i f(r)/r g_t) with e = x/|x|, g_r + i g_t = gy conj(e); origin entries: gx = gy.
// gbias[c] = sum_n [r + b > 0] g_r, accumulated per block in LDS then written as a partial.
constexpr int kNonlinRows = 64;    // rows per block
__global__ __launch_bounds__(256) void tangent_nonlin_bwd_kernel(const float2* __restrict__ x,
                                                                 const float* __restrict__ bias,
                                                                 const float2* __restrict__ gy,
                                                                 float2* __restrict__ gx, float* __restrict__ partial,
                                                                 int N, int C) {
    extern __shared__ float sh[];      // C floats
    for (int c = threadIdx.x; c < C; c += blockDim.x) sh[c] = 0.f;
    __syncthreads();
    const int r0 = blockIdx.x * kNonlinRows;
    const int rows = min(kNonlinRows, N - r0);
    // thread t owns channel slots c = t, t + 256, ... and walks the rows: fixed order -> deterministic
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float b = bias[c];
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) {
            const size_t idx = (size_t)(r0 + r) * C + c;
            const float2 v = x[idx];
            const float2 g = gy[idx];
            float2 o = g;
            if (!is_origin(v)) {
                const float rad = sqrtf(v.x * v.x + v.y * v.y);
                const float inv = 1.f / rad;
                const float ex = v.x * inv, ey = v.y * inv;
                const float gr = g.x * ex + g.y * ey;      // Re(g conj(e))
                const float gt = g.y * ex - g.x * ey;      // Im(g conj(e))
                const bool act = (rad + b) > 0.f;
                const float fr_ = act ? gr : 0.f;
                const float ft = (act ? (rad + b) : 0.f) * inv * gt;
                o = make_float2(ex * fr_ - ey * ft, ey * fr_ + ex * ft);
                acc += fr_;
            }
            gx[idx] = o;
        }
        sh[c] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) partial[(size_t)blockIdx.x * C + c] = sh[c];
}

__global__ void tangent_nonlin_gb_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gbias,
                                                int nparts, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * C + c];
    gbias[c] = s;
}

constexpr int kLinGwWaves = 256;

}  // namespace fc

extern "C" {

int fc_tangent_lin_forward(const float* x, const float* re_w, const float* im_w, float* y, int32_t N, int32_t I,
                           int32_t O, void* stream) {
    if (!x || !re_w || !im_w || !y || N <= 0 || I <= 0 || O <= 0) return FC_ERR_BAD_ARGUMENT;
    const int waves = (N + 15) / 16;
    hipLaunchKernelGGL(fc::tangent_lin_kernel<false>, dim3((waves + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(x), re_w, im_w, reinterpret_cast<float2*>(y), N, I, O, I);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

size_t fc_tangent_lin_backward_workspace_bytes(int32_t N, int32_t I, int32_t O) {
    (void)N;
    return (size_t)fc::kLinGwWaves * O * I * sizeof(float2);
}

int fc_tangent_lin_backward(const float* x, const float* gy, const float* re_w, const float* im_w, float* gx,
                            float* g_re, float* g_im, void* workspace, size_t workspace_bytes, int32_t N, int32_t I,
                            int32_t O, void* stream) {
    if (!x || !gy || !re_w || !im_w || !gx || !g_re || !g_im || N <= 0 || I <= 0 || O <= 0) return FC_ERR_BAD_ARGUMENT;
    if (I > 64 || O > 64) return FC_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < fc_tangent_lin_backward_workspace_bytes(N, I, O)) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int waves = (N + 15) / 16;
    hipLaunchKernelGGL(fc::tangent_lin_kernel<true>, dim3((waves + 3) / 4), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(gy), re_w, im_w, reinterpret_cast<float2*>(gx), N, O, I, I);
    float2* part = reinterpret_cast<float2*>(workspace);
    const int nblocks = (N + 15) / 16;
    int nw = fc::kLinGwWaves;
    if (nw > nblocks) nw = (nblocks + 3) / 4 * 4;
    if (I <= 32 && O <= 32)
        hipLaunchKernelGGL(fc::tangent_lin_gw_kernel<2>, dim3(nw / 4), dim3(256), 0, s, reinterpret_cast<const float2*>(x),
                           reinterpret_cast<const float2*>(gy), part, N, I, O, nw);
    else
        hipLaunchKernelGGL(fc::tangent_lin_gw_kernel<4>, dim3(nw / 4), dim3(256), 0, s, reinterpret_cast<const float2*>(x),
                           reinterpret_cast<const float2*>(gy), part, N, I, O, nw);
    hipLaunchKernelGGL(fc::tangent_lin_gw_reduce_kernel, dim3((O * I + 255) / 256), dim3(256), 0, s, part, g_re, g_im, nw,
                       O * I);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_tangent_nonlin_forward(const float* x, const float* bias, float* y, int32_t N, int32_t C, void* stream) {
    if (!x || !bias || !y || N <= 0 || C <= 0) return FC_ERR_BAD_ARGUMENT;
    const size_t total = (size_t)N * C;
    hipLaunchKernelGGL(fc::tangent_nonlin_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(x), bias,
                       reinterpret_cast<float2*>(y), total, C);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

size_t fc_tangent_nonlin_backward_workspace_bytes(int32_t N, int32_t C) {
    const int nblk = (N + fc::kNonlinRows - 1) / fc::kNonlinRows;
    return (size_t)nblk * C * sizeof(float);
}

int fc_tangent_nonlin_backward(const float* x, const float* bias, const float* gy, float* gx, float* gbias,
                               void* workspace, size_t workspace_bytes, int32_t N, int32_t C, void* stream) {
    if (!x || !bias || !gy || !gx || !gbias || N <= 0 || C <= 0) return FC_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < fc_tangent_nonlin_backward_workspace_bytes(N, C)) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (N + fc::kNonlinRows - 1) / fc::kNonlinRows;
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(fc::tangent_nonlin_bwd_kernel, dim3(nblk), dim3(256), C * sizeof(float), s,
                       reinterpret_cast<const float2*>(x), bias, reinterpret_cast<const float2*>(gy),
                       reinterpret_cast<float2*>(gx), part, N, C);
    hipLaunchKernelGGL(fc::tangent_nonlin_gb_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, gbias, nblk, C);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
