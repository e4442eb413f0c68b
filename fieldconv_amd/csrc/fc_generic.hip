// FieldConv for (n_rings, band_limit) pairs outside the compiled set (n_rings > 8, band_limit > 3 or 0): run-time loops,
// dense stencil rows, no specialisation -- a correctness path so that the module takes every shape the reference does
// (reference nn/field_conv.py:62-98 accepts any band_limit >= 0 and n_rings >= 1) and every dtype it runs in: the kernels
// are templates over float / double (the reference's modules work under .double(); its fp64 fixtures pin that path).  The
// two edge-sized steps are the kernels below; the dense contractions with the filter are complex GEMMs on (N, I*R*F)
// matrices, fc_cgemm (csrc/fc_cgemm.hip).
//
//   fc_generic_gather   contrib[n,i,r,f] = sum_{e: dst_e = n} x[src_e,i] e^{-i (f-B) phi[src_e,i]} S[e,r,f]     (:128-134)
//   fc_generic_scatter  gxt[j,i,f] = sum_{e: src_e = j} sum_r gC[dst_e,i,r,f] conj(S[e,r,f]),  then the chain rule through
//                       the rotation: gx = sum_f gxt_f conj(u_f) + [x != 0] (i x/|x|^2) sum_f m_f Im(conj(gxt_f) xt_f)
#include <algorithm>
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kGenThreads = 256;

template <typename T> struct Cx { T x, y; };
template <typename T> __device__ __forceinline__ Cx<T> gmul(Cx<T> a, Cx<T> b) { return Cx<T>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T> __device__ __forceinline__ Cx<T> gmul_conj(Cx<T> a, Cx<T> b) { return Cx<T>{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// reference utils/field.py:10-16, 40-48 (the box test uses the same 1e-7 in either precision, as the reference does)
template <typename T> __device__ __forceinline__ bool g_is_origin(Cx<T> z) { return (fabs(z.x) < (T)1e-7) && (fabs(z.y) < (T)1e-7); }
template <typename T> __device__ __forceinline__ Cx<T> g_unit_conj(Cx<T> z) {
    if (g_is_origin(z)) return Cx<T>{(T)1, (T)0};
    const T inv = (T)1 / sqrt(z.x * z.x + z.y * z.y);
    return Cx<T>{z.x * inv, -z.y * inv};
}
template <typename T> __device__ __forceinline__ Cx<T> g_unit_power(Cx<T> u, int m) {
    Cx<T> p{(T)1, (T)0};
    const int am = m < 0 ? -m : m;
    for (int k = 0; k < am; ++k) p = (m > 0) ? gmul(p, u) : gmul_conj(p, u);
    return p;
}

// one workgroup per (target vertex, block of Ib input channels: blockIdx.y); thread-owned accumulators in LDS (entry (i, r, f) belongs
// to the thread of (i, f)).  The channels are independent, so any I runs as ceil(I / Ib) blocks whose accumulators fit the CU's LDS.
template <typename T>
__global__ __launch_bounds__(kGenThreads) void fc_generic_gather_kernel(
    const Cx<T>* __restrict__ x, const Cx<T>* __restrict__ sten_t, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr,
    Cx<T>* __restrict__ contrib, const int Ifull, const int Ib, const int R, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Cx<T>* const acc = reinterpret_cast<Cx<T>*>(smem);          // [I][R][F]
    const int n = blockIdx.x;
    const int i0 = blockIdx.y * Ib;
    const int I = min(Ib, Ifull - i0);
    const int nif = I * F, RF = R * F;
    for (int idx = threadIdx.x; idx < I * RF; idx += kGenThreads) acc[idx] = Cx<T>{(T)0, (T)0};
    __syncthreads();
    const int b = rowptr[n], e = rowptr[n + 1];
    for (int idx = threadIdx.x; idx < nif; idx += kGenThreads) {
        const int i = idx / F, f = idx - i * F;
        Cx<T>* const mine = acc + (size_t)i * RF + f;
        for (int s = b; s < e; ++s) {
            const Cx<T> xv = x[(size_t)nbr[s] * Ifull + i0 + i];
            const Cx<T> xt = gmul(xv, g_unit_power(g_unit_conj(xv), f - B));
            const Cx<T>* S = sten_t + (size_t)s * RF + f;
            for (int r = 0; r < R; ++r) {
                const Cx<T> v = gmul(xt, S[r * F]);
                mine[r * F].x += v.x;
                mine[r * F].y += v.y;
            }
        }
    }
    __syncthreads();
    Cx<T>* out = contrib + ((size_t)n * Ifull + i0) * RF;
    for (int idx = threadIdx.x; idx < I * RF; idx += kGenThreads) out[idx] = acc[idx];
}

// one workgroup per (source vertex, block of Ib input channels)
template <typename T>
__global__ __launch_bounds__(kGenThreads) void fc_generic_scatter_kernel(
    const Cx<T>* __restrict__ x, const Cx<T>* __restrict__ gc, const Cx<T>* __restrict__ sten_s, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ nbr, Cx<T>* __restrict__ gx, const int Ifull, const int Ib, const int R, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Cx<T>* const gxt = reinterpret_cast<Cx<T>*>(smem);          // [I][F]
    const int j = blockIdx.x;
    const int i0 = blockIdx.y * Ib;
    const int I = min(Ib, Ifull - i0);
    const int nif = I * F, RF = R * F;
    const int b = rowptr[j], e = rowptr[j + 1];
    for (int idx = threadIdx.x; idx < nif; idx += kGenThreads) {
        const int i = idx / F, f = idx - i * F;
        Cx<T> a{(T)0, (T)0};
        for (int s = b; s < e; ++s) {
            const Cx<T>* g = gc + ((size_t)nbr[s] * Ifull + i0 + i) * RF + f;
            const Cx<T>* S = sten_s + (size_t)s * RF + f;
            for (int r = 0; r < R; ++r) {
                const Cx<T> v = gmul_conj(g[r * F], S[r * F]);
                a.x += v.x;
                a.y += v.y;
            }
        }
        gxt[idx] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < I; i += kGenThreads) {
        const Cx<T> xv = x[(size_t)j * Ifull + i0 + i];
        const Cx<T> u = g_unit_conj(xv);
        const T inv2 = g_is_origin(xv) ? (T)0 : (T)1 / (xv.x * xv.x + xv.y * xv.y);
        Cx<T> acc{(T)0, (T)0};
        T eq = 0;
        for (int f = 0; f < F; ++f) {
            const int m = f - B;
            const Cx<T> c = g_unit_power(u, m);
            const Cx<T> z = gxt[i * F + f];
            const Cx<T> xtv = gmul(xv, c);
            const Cx<T> out = gmul_conj(z, c);
            acc.x += out.x;
            acc.y += out.y;
            eq += (T)m * (z.x * xtv.y - z.y * xtv.x);
        }
        const T q = eq * inv2;
        acc.x += -xv.y * q;
        acc.y += xv.x * q;
        gx[(size_t)j * Ifull + i0 + i] = acc;
    }
}

template <typename T>
static int launch_gather(const void* x, const void* sten_t, const fc_csr* by_target, void* contrib, int n_targets, int I, int R, int B,
                         hipStream_t stream) {
    const int F = 2 * B + 1;
    const size_t per_channel = (size_t)R * F * sizeof(Cx<T>);
    if (per_channel > kMaxLds) return FC_ERR_UNSUPPORTED;                      // (more than 20 480 float / 10 240 double (ring, frequency) pairs)
    const int Ib = (int)std::min<size_t>((size_t)I, kMaxLds / per_channel);    // input channels per workgroup: its accumulators in LDS
    const size_t lds = (size_t)Ib * per_channel;
    const int nblk = (I + Ib - 1) / Ib;
    if (nblk > 65535) return FC_ERR_UNSUPPORTED;
    static bool lds_ok[kMaxDevices] = {};
    if (!allow_full_lds(reinterpret_cast<const void*>(fc_generic_gather_kernel<T>), lds, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(fc_generic_gather_kernel<T>, dim3(n_targets, nblk), dim3(kGenThreads), lds, stream, static_cast<const Cx<T>*>(x),
                       static_cast<const Cx<T>*>(sten_t), by_target->rowptr, by_target->nbr, static_cast<Cx<T>*>(contrib), I, Ib, R, F, B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <typename T>
static int launch_scatter(const void* x, const void* g_contrib, const void* sten_s, const fc_csr* by_source, void* gx, int N, int I, int R,
                          int B, hipStream_t stream) {
    const int F = 2 * B + 1;
    const size_t per_channel = (size_t)F * sizeof(Cx<T>);
    if (per_channel > kMaxLds) return FC_ERR_UNSUPPORTED;
    const int Ib = (int)std::min<size_t>((size_t)I, kMaxLds / per_channel);
    const size_t lds = (size_t)Ib * per_channel;
    const int nblk = (I + Ib - 1) / Ib;
    if (nblk > 65535) return FC_ERR_UNSUPPORTED;
    static bool lds_ok[kMaxDevices] = {};
    if (!allow_full_lds(reinterpret_cast<const void*>(fc_generic_scatter_kernel<T>), lds, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(fc_generic_scatter_kernel<T>, dim3(N, nblk), dim3(kGenThreads), lds, stream, static_cast<const Cx<T>*>(x),
                       static_cast<const Cx<T>*>(g_contrib), static_cast<const Cx<T>*>(sten_s), by_source->rowptr, by_source->nbr,
                       static_cast<Cx<T>*>(gx), I, Ib, R, F, B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc

extern "C" {

int fc_shape_compiled(int32_t n_rings, int32_t band_limit) { return fc::shape_compiled(n_rings, band_limit) ? 1 : 0; }

int fc_generic_gather(const void* x, const void* sten_t, const fc_csr* by_target, void* contrib, int32_t n_targets, int32_t I, int32_t R,
                      int32_t B, int32_t dtype, void* stream) {
    if (!x || !contrib || !by_target || !by_target->rowptr || n_targets < 0 || I <= 0 || R <= 0 || B < 0) return FC_ERR_BAD_ARGUMENT;
    if (dtype != FC_F32 && dtype != FC_F64) return FC_ERR_BAD_ARGUMENT;
    if (n_targets == 0) return FC_OK;
    return dtype == FC_F64 ? fc::launch_gather<double>(x, sten_t, by_target, contrib, n_targets, I, R, B, static_cast<hipStream_t>(stream))
                           : fc::launch_gather<float>(x, sten_t, by_target, contrib, n_targets, I, R, B, static_cast<hipStream_t>(stream));
}

int fc_generic_scatter(const void* x, const void* g_contrib, const void* sten_s, const fc_csr* by_source, void* gx, int32_t N, int32_t I,
                       int32_t R, int32_t B, int32_t dtype, void* stream) {
    if (!x || !g_contrib || !gx || !by_source || !by_source->rowptr || N < 0 || I <= 0 || R <= 0 || B < 0) return FC_ERR_BAD_ARGUMENT;
    if (dtype != FC_F32 && dtype != FC_F64) return FC_ERR_BAD_ARGUMENT;
    if (N == 0) return FC_OK;
    return dtype == FC_F64 ? fc::launch_scatter<double>(x, g_contrib, sten_s, by_source, gx, N, I, R, B, static_cast<hipStream_t>(stream))
                           : fc::launch_scatter<float>(x, g_contrib, sten_s, by_source, gx, N, I, R, B, static_cast<hipStream_t>(stream));
}

}  // extern "C"
