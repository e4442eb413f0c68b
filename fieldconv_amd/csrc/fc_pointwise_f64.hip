// TangentNonLin (modReLU, reference nn/tangent_nonlin.py:24-35) and its adjoint in double precision: the reference's modules
// run under .double() (its fp64 fixtures pin that), and the fp32 kernels of fc_pointwise.hip are tuned for interleaved float
// pairs.  TangentLin in double precision is a plain complex GEMM (fc_cgemm).  A correctness path: one thread per entry, the
// bias gradient through per-group partials summed in a fixed order.
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

struct cd { double x, y; };
__device__ __forceinline__ bool is_origin_d(cd z) { return (fabs(z.x) < 1e-7) && (fabs(z.y) < 1e-7); }      // reference utils/field.py:10-16

__global__ void tangent_nonlin_fwd_f64_kernel(const cd* __restrict__ x, const double* __restrict__ bias, cd* __restrict__ y, size_t total,
                                              int C) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const cd v = x[idx];
    cd o = v;
    if (!is_origin_d(v)) {
        const double r = sqrt(v.x * v.x + v.y * v.y);
        const double s = fmax(r + bias[idx % C], 0.0) / r;
        o = cd{v.x * s, v.y * s};
    }
    y[idx] = o;
}

constexpr int kNl64Rows = 64;       // rows per workgroup of the backward kernel

// gx = e (f'(r) g_r + i f(r)/r g_t) with e = x/|x|, g_r + i g_t = gy conj(e); origin entries: gx = gy.  partial[g][c] = the
// group's share of gbias[c] = sum_n [r + b > 0] g_r (rows walked in order by the channel's thread).
__global__ __launch_bounds__(256) void tangent_nonlin_bwd_f64_kernel(const cd* __restrict__ x, const double* __restrict__ bias,
                                                                     const cd* __restrict__ gy, cd* __restrict__ gx,
                                                                     double* __restrict__ partial, int N, int C) {
    const int r0 = blockIdx.x * kNl64Rows, r1 = min(N, r0 + kNl64Rows);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double b = bias[c];
        double acc = 0.0;
        for (int r = r0; r < r1; ++r) {
            const size_t idx = (size_t)r * C + c;
            const cd v = x[idx], g = gy[idx];
            cd o = g;
            if (!is_origin_d(v)) {
                const double rad = sqrt(v.x * v.x + v.y * v.y), inv = 1.0 / rad;
                const double ex = v.x * inv, ey = v.y * inv;
                const double gr = g.x * ex + g.y * ey, gt = g.y * ex - g.x * ey;
                const bool act = (rad + b) > 0.0;
                const double fr = act ? gr : 0.0;
                const double ft = (act ? (rad + b) : 0.0) * inv * gt;
                o = cd{ex * fr - ey * ft, ey * fr + ex * ft};
                acc += fr;
            }
            gx[idx] = o;
        }
        partial[(size_t)blockIdx.x * C + c] = acc;
    }
}

__global__ void tangent_nonlin_gb_reduce_f64_kernel(const double* __restrict__ partial, double* __restrict__ gbias, int nparts, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * C + c];
    gbias[c] = s;
}

}  // namespace fc

extern "C" {

int fc_tangent_nonlin_forward_f64(const double* x, const double* bias, double* y, int32_t N, int32_t C, void* stream) {
    if (!x || !bias || !y || N <= 0 || C <= 0) return FC_ERR_BAD_ARGUMENT;
    const size_t total = (size_t)N * C;
    hipLaunchKernelGGL(fc::tangent_nonlin_fwd_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const fc::cd*>(x), bias, reinterpret_cast<fc::cd*>(y), total, C);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

size_t fc_tangent_nonlin_backward_workspace_bytes_f64(int32_t N, int32_t C) {
    return (size_t)((N + fc::kNl64Rows - 1) / fc::kNl64Rows) * C * sizeof(double);
}

int fc_tangent_nonlin_backward_f64(const double* x, const double* bias, const double* gy, double* gx, double* gbias, void* workspace,
                                   size_t workspace_bytes, int32_t N, int32_t C, void* stream) {
    if (!x || !bias || !gy || !gx || !gbias || N <= 0 || C <= 0) return FC_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < fc_tangent_nonlin_backward_workspace_bytes_f64(N, C)) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ngroups = (N + fc::kNl64Rows - 1) / fc::kNl64Rows;
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(fc::tangent_nonlin_bwd_f64_kernel, dim3(ngroups), dim3(256), 0, s, reinterpret_cast<const fc::cd*>(x), bias,
                       reinterpret_cast<const fc::cd*>(gy), reinterpret_cast<fc::cd*>(gx), part, N, C);
    hipLaunchKernelGGL(fc::tangent_nonlin_gb_reduce_f64_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, gbias, ngroups, C);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
