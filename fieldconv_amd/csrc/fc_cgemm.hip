// Complex GEMM on the matrix pipe for the run-time FieldConv path (any n_rings / band_limit / channel count, fp32 or fp64;
// reference nn/field_conv.py:10-33 is the forward product, its two autograd twins the other two):
//
//     C[m, n] = alpha * sum_k A[m*sam + k*sak] * op(B[k*sbk + n*sbn]),      op = identity or complex conjugate,
//
// C row-major and contiguous, A and B complex with arbitrary element strides, so that one kernel serves
//     y         = contrib . W^T            (A = contrib (n, K), B = W (O, K): sbk = 1, sbn = K)
//     g_contrib = gy . conj(W)             (A = gy (n, O),      B = W (O, K): sbk = K, sbn = 1, conjugated)
//     gW        = gy^T . conj(contrib)     (A = gy read transposed: sam = 1, sak = O; B = contrib (n, K), conjugated)
// and TangentLin in double precision.  A complex product is four real ones on v_mfma_f32_16x16x4_f32 /
// v_mfma_f64_16x16x4_f64 (fp64 runs at the same matrix rate as fp32 on gfx950).  A workgroup of four wavefronts owns a
// 64 x 64 tile of C, each wavefront a 32 x 32 quarter (2 x 2 MFMA tiles, re and im accumulators); A and B pass through LDS in
// k chunks of 16 as separate re / im planes.  Products whose output is small and whose contraction is long -- the two weight-gradient
// products (gW = gy^T . conj(contrib), TangentLin's gW = gy^T . conj(x)): M x N = a few 64 x 64 tiles, K = the vertex count -- are split
// along k over gridDim.z workgroups (fc_cgemm_workspace_bytes > 0): per-slice partials in the caller's workspace, summed in slice order by
// a second launch (deterministic).  No double buffering: a correctness path.
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kGemmTile = 64;
constexpr int kGemmK = 16;
constexpr int kGemmThreads = 256;
constexpr int kGemmPad = 1;       // LDS row padding (elements)

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_real(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x4 mfma_real(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };
template <> struct Vec4<double> { typedef f64x4 type; };

struct GemmArgs {
    int M, N, K;
    long sam, sak, sbk, sbn;
    int conj_b;
    double alpha;
    int kchunk;         // k range of one workgroup along gridDim.z (a multiple of kGemmK; >= K: no split)
};

// how many k slices a product is cut into: only when the output tiles cannot occupy the CUs and the contraction is long
static int cgemm_ksplit(int M, int N, int K, int* kchunk) {
    const long tiles = (long)((M + kGemmTile - 1) / kGemmTile) * ((N + kGemmTile - 1) / kGemmTile);
    const int cus = num_cus();
    int slices = 1;
    if (tiles * 2 <= cus && K >= 2048) {
        long want = (2L * cus) / tiles;                      // about two workgroups per CU
        const long by_len = (K + 511) / 512;                 // at least 512 k entries per slice
        slices = (int)(want < by_len ? want : by_len);
        if (slices > 256) slices = 256;
        if (slices < 1) slices = 1;
    }
    int chunk = ((K + slices - 1) / slices + kGemmK - 1) / kGemmK * kGemmK;
    if (chunk < kGemmK) chunk = kGemmK;
    *kchunk = chunk;
    return (K + chunk - 1) / chunk > 0 ? (K + chunk - 1) / chunk : 1;
}

// C: the output, or with gridDim.z > 1 the per-slice partials [z][M][N] (alpha applied by the reduction)
template <typename T>
__global__ __launch_bounds__(kGemmThreads) void fc_cgemm_kernel(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                                                 const GemmArgs g) {
    typedef typename Vec4<T>::type V4;
    // planes: A as [m][k], B as [k][n]
    __shared__ T a_re[kGemmTile][kGemmK + kGemmPad], a_im[kGemmTile][kGemmK + kGemmPad];
    __shared__ T b_re[kGemmK][kGemmTile + kGemmPad], b_im[kGemmK][kGemmTile + kGemmPad];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * kGemmTile, n0 = blockIdx.y * kGemmTile;      // (M, the vertex count in two of the three uses, on the unbounded grid axis)
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;       // this wavefront's quarter of the tile
    const int fr = lane & 15, fq = lane >> 4;
    V4 cre[2][2], cim[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { cre[i][j] = V4{0, 0, 0, 0}; cim[i][j] = V4{0, 0, 0, 0}; }
    const bool a_k_fast = g.sak == 1, b_n_fast = g.sbn == 1;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    for (int k0 = kbeg; k0 < kend; k0 += kGemmK) {
        // 64 x 16 elements of A and 16 x 64 of B, four per thread each; the thread index runs along the contiguous direction
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + j * kGemmThreads;
            const int am = a_k_fast ? idx / kGemmK : idx % kGemmTile, ak = a_k_fast ? idx % kGemmK : idx / kGemmTile;
            T re = 0, im = 0;
            if (m0 + am < g.M && k0 + ak < kend) {
                const T* p = A + 2 * ((long)(m0 + am) * g.sam + (long)(k0 + ak) * g.sak);
                re = p[0]; im = p[1];
            }
            a_re[am][ak] = re; a_im[am][ak] = im;
            const int bn = b_n_fast ? idx % kGemmTile : idx / kGemmK, bk = b_n_fast ? idx / kGemmTile : idx % kGemmK;
            re = 0; im = 0;
            if (n0 + bn < g.N && k0 + bk < kend) {
                const T* p = B + 2 * ((long)(k0 + bk) * g.sbk + (long)(n0 + bn) * g.sbn);
                re = p[0]; im = g.conj_b ? -p[1] : p[1];
            }
            b_re[bk][bn] = re; b_im[bk][bn] = im;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kGemmK; ks += 4) {
            T are[2], aim[2], bre[2], bim[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                are[i] = a_re[wm + 16 * i + fr][ks + fq];
                aim[i] = a_im[wm + 16 * i + fr][ks + fq];
                bre[i] = b_re[ks + fq][wn + 16 * i + fr];
                bim[i] = b_im[ks + fq][wn + 16 * i + fr];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    cre[i][j] = mfma_real(are[i], bre[j], cre[i][j]);
                    cre[i][j] = mfma_real(-aim[i], bim[j], cre[i][j]);
                    cim[i][j] = mfma_real(are[i], bim[j], cim[i][j]);
                    cim[i][j] = mfma_real(aim[i], bre[j], cim[i][j]);
                }
        }
        __syncthreads();
    }
    // D layout of the 16x16x4 instructions: lane l holds column l & 15 and, in its register t, row 4 (l >> 4) + t (fp32) or
    // row 4 t + (l >> 4) (fp64: measured on MI355X -- the f64 instruction interleaves the lane groups' rows; test_cgemm_three_layouts)
    constexpr bool kF64 = std::is_same<T, double>::value;
    const T alpha = gridDim.z > 1 ? (T)1 : (T)g.alpha;
    C += 2 * (long)blockIdx.z * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + wm + 16 * i + (kF64 ? 4 * t + fq : 4 * fq + t), n = n0 + wn + 16 * j + fr;
                if (m < g.M && n < g.N) {
                    T* p = C + 2 * ((long)m * g.N + n);
                    p[0] = alpha * cre[i][j][t];
                    p[1] = alpha * cim[i][j][t];
                }
            }
}

// C[idx] = alpha * sum over the slices, in slice order
template <typename T>
__global__ void fc_cgemm_reduce_kernel(const T* __restrict__ part, T* __restrict__ C, long count, int slices, T alpha) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // real numbers (2 per complex entry)
    if (idx >= count) return;
    T s = 0;
    for (int z = 0; z < slices; ++z) s += part[(long)z * count + idx];
    C[idx] = alpha * s;
}

template <typename T>
static int launch_cgemm(const void* A, const void* B, void* C, GemmArgs g, void* ws, size_t ws_bytes, hipStream_t stream) {
    int kchunk = 0;
    int slices = cgemm_ksplit(g.M, g.N, g.K, &kchunk);
    const size_t need = (size_t)slices * g.M * g.N * 2 * sizeof(T);
    if (slices > 1 && (!ws || ws_bytes < need)) {           // no workspace: the unsplit product (same result up to summation order)
        slices = 1;
        kchunk = (g.K + kGemmK - 1) / kGemmK * kGemmK;
    }
    g.kchunk = kchunk > 0 ? kchunk : kGemmK;
    const dim3 grid((g.M + kGemmTile - 1) / kGemmTile, (g.N + kGemmTile - 1) / kGemmTile, slices);
    if (grid.y > 65535u) return FC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fc_cgemm_kernel<T>, grid, dim3(kGemmThreads), 0, stream, static_cast<const T*>(A), static_cast<const T*>(B),
                       slices > 1 ? static_cast<T*>(ws) : static_cast<T*>(C), g);
    if (slices > 1) {
        const long count = 2L * g.M * g.N;
        hipLaunchKernelGGL(fc_cgemm_reduce_kernel<T>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, static_cast<const T*>(ws),
                           static_cast<T*>(C), count, slices, (T)g.alpha);
    }
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc

extern "C" size_t fc_cgemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t dtype) {
    if (M <= 0 || N <= 0 || K <= 0 || (dtype != FC_F32 && dtype != FC_F64)) return 0;
    int kchunk = 0;
    const int slices = fc::cgemm_ksplit(M, N, K, &kchunk);
    return slices > 1 ? (size_t)slices * M * N * 2 * (dtype == FC_F64 ? sizeof(double) : sizeof(float)) : 0;
}

extern "C" int fc_cgemm(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t sam, int64_t sak, int64_t sbk,
                        int64_t sbn, int32_t conj_b, double alpha, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0 || (dtype != FC_F32 && dtype != FC_F64)) return FC_ERR_BAD_ARGUMENT;
    if (M == 0 || N == 0) return FC_OK;
    fc::GemmArgs g;
    g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn;
    g.conj_b = conj_b ? 1 : 0;
    g.alpha = alpha;
    g.kchunk = 0;
    return dtype == FC_F64 ? fc::launch_cgemm<double>(A, B, C, g, workspace, workspace_bytes, static_cast<hipStream_t>(stream))
                           : fc::launch_cgemm<float>(A, B, C, g, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
