// FieldConv for (n_rings, band_limit) pairs outside the compiled set (n_rings > 8, band_limit > 3 or 0): run-time loops,
// dense stencil rows, no specialisation -- a correctness path so that the module takes every shape the reference does
// (reference nn/field_conv.py:62-98 accepts any band_limit >= 0 and n_rings >= 1).  The two edge-sized steps are kernels;
// the dense contractions with the filter are plain complex GEMMs on (N, I*R*F) matrices and are left to the caller's BLAS
// (fieldconv_amd/functional.py: rocBLAS through torch.matmul).
//
//   fc_generic_gather   contrib[n,i,r,f] = sum_{e: dst_e = n} x[src_e,i] e^{-i (f-B) phi[src_e,i]} S[e,r,f]     (:128-134)
//   fc_generic_scatter  gxt[j,i,f] = sum_{e: src_e = j} sum_r gC[dst_e,i,r,f] conj(S[e,r,f]),  then the chain rule through
//                       the rotation: gx = sum_f gxt_f conj(u_f) + [x != 0] (i x/|x|^2) sum_f m_f Im(conj(gxt_f) xt_f)
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kGenThreads = 256;

// one workgroup per target vertex; thread-owned accumulators in LDS (entry (i, r, f) belongs to the thread of (i, f))
__global__ __launch_bounds__(kGenThreads) void fc_generic_gather_kernel(
    const float2* __restrict__ x, const float2* __restrict__ sten_t, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr,
    float2* __restrict__ contrib, const int I, const int R, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* const acc = reinterpret_cast<float2*>(smem);          // [I][R][F]
    const int n = blockIdx.x;
    const int nif = I * F, RF = R * F;
    for (int idx = threadIdx.x; idx < I * RF; idx += kGenThreads) acc[idx] = make_float2(0.f, 0.f);
    __syncthreads();
    const int b = rowptr[n], e = rowptr[n + 1];
    for (int idx = threadIdx.x; idx < nif; idx += kGenThreads) {
        const int i = idx / F, f = idx - i * F;
        float2* const mine = acc + (size_t)i * RF + f;
        for (int s = b; s < e; ++s) {
            const float2 xv = x[(size_t)nbr[s] * I + i];
            const float2 xt = cmul(xv, unit_power(unit_conj(xv), f - B));
            const float2* S = sten_t + (size_t)s * RF + f;
            for (int r = 0; r < R; ++r) {
                const float2 v = cmul(xt, S[r * F]);
                mine[r * F].x += v.x;
                mine[r * F].y += v.y;
            }
        }
    }
    __syncthreads();
    float2* out = contrib + (size_t)n * I * RF;
    for (int idx = threadIdx.x; idx < I * RF; idx += kGenThreads) out[idx] = acc[idx];
}

// one workgroup per source vertex
__global__ __launch_bounds__(kGenThreads) void fc_generic_scatter_kernel(
    const float2* __restrict__ x, const float2* __restrict__ gc, const float2* __restrict__ sten_s, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ nbr, float2* __restrict__ gx, const int I, const int R, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* const gxt = reinterpret_cast<float2*>(smem);          // [I][F]
    const int j = blockIdx.x;
    const int nif = I * F, RF = R * F;
    const int b = rowptr[j], e = rowptr[j + 1];
    for (int idx = threadIdx.x; idx < nif; idx += kGenThreads) {
        const int i = idx / F, f = idx - i * F;
        float2 a = make_float2(0.f, 0.f);
        for (int s = b; s < e; ++s) {
            const float2* g = gc + ((size_t)nbr[s] * I + i) * RF + f;
            const float2* S = sten_s + (size_t)s * RF + f;
            for (int r = 0; r < R; ++r) {
                const float2 v = cmul_conj(g[r * F], S[r * F]);
                a.x += v.x;
                a.y += v.y;
            }
        }
        gxt[idx] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < I; i += kGenThreads) {
        const float2 xv = x[(size_t)j * I + i];
        const float2 u = unit_conj(xv);
        const float inv2 = is_origin(xv) ? 0.f : 1.f / (xv.x * xv.x + xv.y * xv.y);
        float2 acc = make_float2(0.f, 0.f);
        float eq = 0.f;
        for (int f = 0; f < F; ++f) {
            const int m = f - B;
            const float2 c = unit_power(u, m);
            const float2 z = gxt[i * F + f];
            const float2 xtv = cmul(xv, c);
            const float2 out = cmul_conj(z, c);
            acc.x += out.x;
            acc.y += out.y;
            eq += (float)m * (z.x * xtv.y - z.y * xtv.x);
        }
        const float q = eq * inv2;
        acc.x += -xv.y * q;
        acc.y += xv.x * q;
        gx[(size_t)j * I + i] = acc;
    }
}

}  // namespace fc

extern "C" {

int fc_shape_compiled(int32_t n_rings, int32_t band_limit) { return fc::shape_compiled(n_rings, band_limit) ? 1 : 0; }

int fc_generic_gather(const float* x, const float* sten_t, const fc_csr* by_target, float* contrib, int32_t n_targets, int32_t I, int32_t R,
                      int32_t B, void* stream) {
    if (!x || !contrib || !by_target || !by_target->rowptr || n_targets < 0 || I <= 0 || R <= 0 || B < 0) return FC_ERR_BAD_ARGUMENT;
    if (n_targets == 0) return FC_OK;
    const int F = 2 * B + 1;
    const size_t lds = (size_t)I * R * F * sizeof(float2);
    if (lds > fc::kMaxLds) return FC_ERR_UNSUPPORTED;
    static bool lds_ok[fc::kMaxDevices] = {};
    if (!fc::allow_full_lds(reinterpret_cast<const void*>(fc::fc_generic_gather_kernel), lds, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(fc::fc_generic_gather_kernel, dim3(n_targets), dim3(fc::kGenThreads), lds, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(sten_t), by_target->rowptr, by_target->nbr,
                       reinterpret_cast<float2*>(contrib), I, R, F, B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_generic_scatter(const float* x, const float* g_contrib, const float* sten_s, const fc_csr* by_source, float* gx, int32_t N, int32_t I,
                       int32_t R, int32_t B, void* stream) {
    if (!x || !g_contrib || !gx || !by_source || !by_source->rowptr || N < 0 || I <= 0 || R <= 0 || B < 0) return FC_ERR_BAD_ARGUMENT;
    if (N == 0) return FC_OK;
    const int F = 2 * B + 1;
    const size_t lds = (size_t)I * F * sizeof(float2);
    if (lds > fc::kMaxLds) return FC_ERR_UNSUPPORTED;
    static bool lds_ok[fc::kMaxDevices] = {};
    if (!fc::allow_full_lds(reinterpret_cast<const void*>(fc::fc_generic_scatter_kernel), lds, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(fc::fc_generic_scatter_kernel, dim3(N), dim3(fc::kGenThreads), lds, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(g_contrib), reinterpret_cast<const float2*>(sten_s),
                       by_source->rowptr, by_source->nbr, reinterpret_cast<float2*>(gx), I, R, F, B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
