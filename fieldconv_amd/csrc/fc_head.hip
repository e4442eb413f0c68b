// ECHOBlock behind its descriptors (reference nn/echo_block.py:95-103):
//
//     y = lin3(relu(lin2(relu(lin1(d))))) + res(softAbs(x))
//
// d (N, D) the flattened ECHO descriptors, x (N, C) the block's complex input, lin1: D -> H1 (128 in the reference), lin2: H1 -> H2
// (64), lin3: H2 -> Q, res: C -> Q (torch.nn.Linear layouts: weight (out, in), bias (out)).  Composed of ATen GEMMs this tail is ~30
// launches per step (12 GEMMs -- the four weight gradients contract over the vertices into a few tiles: 42 us each at 4 999 vertices --
// five bias sums, ReLU masks, adds); here it is three launches per pass, all products on the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32: the arithmetic of an fp32 GEMM, sums in another order):
//
//   forward   fc_rgemm_kernel      h1 = d W1^T (k-split over the workgroups when the output tiles alone cannot fill the chip)
//             fc_rgemm_finish      h1 = relu(sum of the slices + b1)                        (only with a k split)
//             fc_head_fwd_kernel   per 16 rows: h2 = relu(h1 W2^T + b2), y = h2 W3^T + b3 + softAbs(x) Wr^T + br
//   backward  fc_head_bwd_kernel   per 16 rows: g_h2 = (g W3) [h2 > 0], g_h1 = (g_h2 W2) [h1 > 0], gx = (g Wr) x / |x|, and the rows'
//                                  column sums of g, g_h2, g_h1 (bias-gradient partials)
//             fc_rgemm_kernel      ONE grouped launch: g_d = g_h1 W1, and the four weight gradients X^T Y over the vertices
//                                  (g_h1^T d, g_h2^T h1, g^T h2, g^T softAbs(x)) k-split into per-slice partials
//             fc_rgemm_finish      the partials summed in slice order, the bias-gradient partials in tile order
// Everything lives in caller-owned buffers; sums run in a fixed order (deterministic).
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

// ------------------------------------------------------------------------------------------------------------ grouped real GEMM
// C[m, n] = sum_k A(m, k) B(k, n),  A(m, k) = A[m * sam + k * sak],  B(k, n) = B[k * sbk + n * sbn]  (element strides, so that one kernel
// serves X W^T, X W and X^T Y).  A workgroup of four wavefronts owns a 64 x 64 tile of C (each wavefront a 32 x 32 quarter) for ONE k
// slice; operands pass through LDS in k chunks of 32, the next chunk's loads in flight during the products.  Several problems share a
// launch (blockIdx.x -> problem by its first block).
constexpr int kRgTile = 64;
constexpr int kRgChunk = 32;
constexpr int kRgThreads = 256;
constexpr int kRgStride = kRgTile + 16;      // LDS row stride: the four k rows of a fragment read fall into distinct bank groups
// column swizzle of the staged operands, uniform over the four k rows of a fragment read (so those stay conflict-free): an operand whose
// contiguous direction is k is stored by 32 lanes along k -- without it they would share four banks
__device__ __forceinline__ int rg_swz(int k) { return (k >> 2) & 7; }
constexpr int kRgMaxProblems = 6;
constexpr int kRgPer = kRgTile * kRgChunk / kRgThreads;      // elements per thread, operand and chunk

struct RgProblem {
    const float* A;
    const float* B;
    float* C;               // the output (slices == 1) or the per-slice partials [slice][M][N]
    const float* bias;      // slices == 1 only: C = relu?(sum + bias[n]); may be null
    int M, N, K;
    long sam, sak, sbk, sbn;
    int b_abs;              // B is complex64 (element strides in complex numbers): B(k, n) = softAbs of it
    int relu;
    int tiles_m, tiles_n, slices, kchunk;       // kchunk: k range of one slice, a multiple of kRgChunk
    int block0;             // first workgroup of this problem
};
struct RgArgs {
    RgProblem p[kRgMaxProblems];
    int count;
};

__device__ __forceinline__ float soft_abs_of(float2 v) { return is_origin(v) ? 0.f : sqrtf(v.x * v.x + v.y * v.y); }

__global__ __launch_bounds__(kRgThreads) void fc_rgemm_kernel(const RgArgs args) {
    __shared__ float As[2][kRgChunk][kRgStride], Bs[2][kRgChunk][kRgStride];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kRgMaxProblems; ++q)
        if (q < args.count && (int)blockIdx.x >= args.p[q].block0) pi = q;
    const RgProblem& g = args.p[pi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x - g.block0;
    const int tm = b % g.tiles_m;
    b /= g.tiles_m;
    const int tn = b % g.tiles_n, slice = b / g.tiles_n;
    const int m0 = tm * kRgTile, n0 = tn * kRgTile;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int fr = lane & 15, fq = lane >> 4;
    const int kbeg = slice * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    // the thread index runs along each operand's contiguous direction
    const bool a_k_fast = g.sak == 1, b_n_fast = g.sbn == 1;
    float ra[kRgPer], rb[kRgPer];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int j = 0; j < kRgPer; ++j) {
            const int idx = tid + j * kRgThreads;
            const int am = a_k_fast ? idx / kRgChunk : idx % kRgTile, ak = a_k_fast ? idx % kRgChunk : idx / kRgTile;
            ra[j] = (m0 + am < g.M && k0 + ak < kend) ? g.A[(long)(m0 + am) * g.sam + (long)(k0 + ak) * g.sak] : 0.f;
            const int bn = b_n_fast ? idx % kRgTile : idx / kRgChunk, bk = b_n_fast ? idx / kRgTile : idx % kRgChunk;
            float v = 0.f;
            if (n0 + bn < g.N && k0 + bk < kend) {
                const long at = (long)(k0 + bk) * g.sbk + (long)(n0 + bn) * g.sbn;
                v = g.b_abs ? soft_abs_of(reinterpret_cast<const float2*>(g.B)[at]) : g.B[at];
            }
            rb[j] = v;
        }
    };
    auto stage = [&](const int buf) {
#pragma unroll
        for (int j = 0; j < kRgPer; ++j) {
            const int idx = tid + j * kRgThreads;
            const int am = a_k_fast ? idx / kRgChunk : idx % kRgTile, ak = a_k_fast ? idx % kRgChunk : idx / kRgTile;
            As[buf][ak][am ^ rg_swz(ak)] = ra[j];
            const int bn = b_n_fast ? idx % kRgTile : idx / kRgChunk, bk = b_n_fast ? idx / kRgTile : idx % kRgChunk;
            Bs[buf][bk][bn ^ rg_swz(bk)] = rb[j];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kbeg < kend) {
        fetch(kbeg);
        int buf = 0;
        for (int k0 = kbeg; k0 < kend; k0 += kRgChunk, buf ^= 1) {
            stage(buf);
            __syncthreads();                                   // chunk k0 is in LDS; the products of chunk k0 - 32 (other buffer) are done
            if (k0 + kRgChunk < kend) fetch(k0 + kRgChunk);    // in flight during the products below
#pragma unroll
            for (int ks = 0; ks < kRgChunk; ks += 4) {
                float a[2], bb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = As[buf][ks + fq][(wm + 16 * i + fr) ^ rg_swz(ks)];
                    bb[i] = Bs[buf][ks + fq][(wn + 16 * i + fr) ^ rg_swz(ks)];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], bb[j], acc[i][j]);
            }
        }
    }
    // D layout of v_mfma_f32_16x16x4_f32: lane l holds column l & 15, register t row 4 (l >> 4) + t
    float* C = g.C + (g.slices > 1 ? (size_t)slice * g.M * g.N : 0);
    const bool epi = g.slices == 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + wm + 16 * i + 4 * fq + t, n = n0 + wn + 16 * j + fr;
                if (m < g.M && n < g.N) {
                    float v = acc[i][j][t];
                    if (epi && g.bias) v += g.bias[n];
                    if (epi && g.relu) v = fmaxf(v, 0.f);
                    C[(size_t)m * g.N + n] = v;
                }
            }
}

// The second launch of a k-split product and of the bias gradients, grouped like the first: task t sums `parts` arrays of `count`
// floats, `stride` apart, in order -- out = relu?(sum + bias[idx % N]) -- or, for a bias gradient, the per-tile column sums.
constexpr int kRfMaxTasks = 10;
struct RfTask {
    const float* part;
    float* out;
    float* out2;            // a second destination of the same sums (lin3.bias and res.bias see the same cotangent); may be null
    const float* bias;
    long count, stride;
    int parts, N, relu;
    int block0;
};
struct RfArgs {
    RfTask t[kRfMaxTasks];
    int count;
};
constexpr int kRfThreads = 256;

__global__ __launch_bounds__(kRfThreads) void fc_rgemm_finish_kernel(const RfArgs args) {
    int ti = 0;
#pragma unroll
    for (int q = 1; q < kRfMaxTasks; ++q)
        if (q < args.count && (int)blockIdx.x >= args.t[q].block0) ti = q;
    const RfTask& t = args.t[ti];
    const long idx = (long)((int)blockIdx.x - t.block0) * kRfThreads + threadIdx.x;
    if (idx >= t.count) return;
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= t.parts; p += 8) {          // fixed order, eight loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t.part[(long)(p + u) * t.stride + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < t.parts; ++p) s += t.part[(long)p * t.stride + idx];
    if (t.bias) s += t.bias[idx % t.N];
    if (t.relu) s = fmaxf(s, 0.f);
    t.out[idx] = s;
    if (t.out2) t.out2[idx] = s;
}

// k slices of a product: only when its output tiles cannot occupy the chip and the contraction is long enough to cut
static void rg_plan(RgProblem& g) {
    g.tiles_m = (g.M + kRgTile - 1) / kRgTile;
    g.tiles_n = (g.N + kRgTile - 1) / kRgTile;
    const long tiles = (long)g.tiles_m * g.tiles_n;
    const int cus = num_cus();
    int slices = 1;
    if (tiles < cus && g.K >= 4 * kRgChunk) {
        const long want = (2L * cus + tiles - 1) / tiles;                  // about two workgroups per CU
        const long by_len = (g.K + 4 * kRgChunk - 1) / (4 * kRgChunk);     // at least 128 k entries per slice
        slices = (int)(want < by_len ? want : by_len);
        if (slices < 1) slices = 1;
    }
    int chunk = ((g.K + slices - 1) / slices + kRgChunk - 1) / kRgChunk * kRgChunk;
    if (chunk < kRgChunk) chunk = kRgChunk;
    g.kchunk = chunk;
    g.slices = (g.K + chunk - 1) / chunk > 0 ? (g.K + chunk - 1) / chunk : 1;
}

static RgProblem rg_problem(const float* A, const float* B, int M, int N, int K, long sam, long sak, long sbk, long sbn, bool b_abs = false) {
    RgProblem g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn;
    g.b_abs = b_abs ? 1 : 0;
    rg_plan(g);
    return g;
}

static int rg_launch(RgArgs& a, hipStream_t stream) {
    int blocks = 0;
    for (int q = 0; q < a.count; ++q) {
        a.p[q].block0 = blocks;
        blocks += a.p[q].tiles_m * a.p[q].tiles_n * a.p[q].slices;
    }
    if (blocks == 0) return FC_OK;
    hipLaunchKernelGGL(fc_rgemm_kernel, dim3(blocks), dim3(kRgThreads), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

static int rf_launch(RfArgs& a, hipStream_t stream) {
    int blocks = 0;
    for (int q = 0; q < a.count; ++q) {
        a.t[q].block0 = blocks;
        blocks += (int)((a.t[q].count + kRfThreads - 1) / kRfThreads);
    }
    if (blocks == 0) return FC_OK;
    hipLaunchKernelGGL(fc_rgemm_finish_kernel, dim3(blocks), dim3(kRfThreads), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------------------ the 16-row kernels
// One workgroup of four wavefronts per 16 rows.  A product of the row tile with a weight matrix is one 16 x 16 output tile per
// wavefront and instruction chain: lane l supplies A[row l & 15][k] and B[k][column l & 15]; the k entries of lane group l >> 4 are the
// group's own contiguous quarter of the contraction (both operands agree, the order of a sum is free), so a lane reads its operand
// values as consecutive floats.
constexpr int kHdThreads = 256;
constexpr int kHdWaves = 4;
constexpr int kHdPad = 4;
constexpr int kHdMaxH1 = 128, kHdMaxH2 = 64, kHdMaxQ = 64, kHdMaxC = 64;

struct HeadDims { int N, D, H1, H2, C, Q; };

// acc += A[16 rows][K] (LDS, row stride lda) . W^T, W (n, k) row-major with row stride ldw (a torch Linear weight): column tile n0
__device__ __forceinline__ f32x4 tile_times_wt(const float* a_lds, int lda, const float* __restrict__ w, int ldw, int n0, int nmax, int K, f32x4 acc,
                                               int fr, int fq) {
    const int kq = (K + 3) / 4;                    // k entries per lane group
    const int n = n0 + fr;
    const float* wrow = w + (size_t)min(n, nmax - 1) * ldw;
    const float keep = n < nmax ? 1.f : 0.f;
    if ((K & 15) == 0 && (ldw & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {      // four k entries per load
        for (int j = 0; j < kq; j += 4) {
            const int k = fq * kq + j;
            const float4 a4 = *reinterpret_cast<const float4*>(a_lds + fr * lda + k);
            const float4 b4 = *reinterpret_cast<const float4*>(wrow + k);
            acc = mfma16(a4.x, b4.x * keep, acc);
            acc = mfma16(a4.y, b4.y * keep, acc);
            acc = mfma16(a4.z, b4.z * keep, acc);
            acc = mfma16(a4.w, b4.w * keep, acc);
        }
        return acc;
    }
    for (int j = 0; j < kq; ++j) {
        const int k = fq * kq + j;
        const float av = k < K ? a_lds[fr * lda + k] : 0.f;
        const float bv = k < K ? wrow[k] * keep : 0.f;
        acc = mfma16(av, bv, acc);
    }
    return acc;
}
// acc += A[16 rows][K] (LDS) . W, W (k, n) row-major with row stride ldw (the same weight seen from its output side): column tile n0
__device__ __forceinline__ f32x4 tile_times_w(const float* a_lds, int lda, const float* __restrict__ w, int ldw, int n0, int nmax, int K, f32x4 acc,
                                              int fr, int fq) {
    const int kq = (K + 3) / 4;
    const int n = n0 + fr;
    const float keep = n < nmax ? 1.f : 0.f;
    const int nc = min(n, nmax - 1);
    for (int j = 0; j < kq; ++j) {
        const int k = fq * kq + j;
        const float av = k < K ? a_lds[fr * lda + k] : 0.f;
        const float bv = k < K ? w[(size_t)k * ldw + nc] * keep : 0.f;
        acc = mfma16(av, bv, acc);
    }
    return acc;
}

__global__ __launch_bounds__(kHdThreads) void fc_head_fwd_kernel(float* __restrict__ h1 /* in: lin1's output before its ReLU (unless relu_done); out: after */,
                                                                 const float2* __restrict__ x, const float* __restrict__ w2, const float* __restrict__ b2,
                                                                 const float* __restrict__ w3, const float* __restrict__ b3, const float* __restrict__ wr,
                                                                 const float* __restrict__ br, float* __restrict__ h2, float* __restrict__ y,
                                                                 const HeadDims d, const int relu_done) {
    __shared__ float h1s[16][kHdMaxH1 + kHdPad], h2s[16][kHdMaxH2 + kHdPad], as_[16][kHdMaxC + kHdPad];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int r0 = blockIdx.x * 16;
    for (int idx = tid; idx < 16 * d.H1; idx += kHdThreads) {
        const int r = idx / d.H1, c = idx - r * d.H1;
        float v = 0.f;
        if (r0 + r < d.N) {
            v = h1[(size_t)(r0 + r) * d.H1 + c];
            if (!relu_done) {
                v = fmaxf(v, 0.f);
                h1[(size_t)(r0 + r) * d.H1 + c] = v;
            }
        }
        h1s[r][c] = v;
    }
    for (int idx = tid; idx < 16 * d.C; idx += kHdThreads) {
        const int r = idx / d.C, c = idx - r * d.C;
        as_[r][c] = r0 + r < d.N ? soft_abs_of(x[(size_t)(r0 + r) * d.C + c]) : 0.f;
    }
    __syncthreads();
    for (int nt = wave; nt * 16 < d.H2; nt += kHdWaves) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_times_wt(&h1s[0][0], kHdMaxH1 + kHdPad, w2, d.H1, nt * 16, d.H2, d.H1, acc, fr, fq);
        const int n = nt * 16 + fr;
        const float bias = n < d.H2 ? b2[n] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * fq + t;
            const float v = fmaxf(acc[t] + bias, 0.f);
            if (n < d.H2) {
                h2s[r][n] = v;
                if (r0 + r < d.N) h2[(size_t)(r0 + r) * d.H2 + n] = v;
            }
        }
    }
    __syncthreads();
    for (int nt = wave; nt * 16 < d.Q; nt += kHdWaves) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_times_wt(&h2s[0][0], kHdMaxH2 + kHdPad, w3, d.H2, nt * 16, d.Q, d.H2, acc, fr, fq);
        acc = tile_times_wt(&as_[0][0], kHdMaxC + kHdPad, wr, d.C, nt * 16, d.Q, d.C, acc, fr, fq);
        const int n = nt * 16 + fr;
        const float bias = n < d.Q ? b3[n] + br[n] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * fq + t;
            if (n < d.Q && r0 + r < d.N) y[(size_t)(r0 + r) * d.Q + n] = acc[t] + bias;
        }
    }
}

// sum over the 16 rows of a D-layout accumulator's column (lane l: column l & 15, rows 4 (l >> 4) + t): valid in lanes 0..15
__device__ __forceinline__ float column_sum(f32x4 v) {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}

__global__ __launch_bounds__(kHdThreads) void fc_head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h1, const float* __restrict__ h2,
                                                                 const float2* __restrict__ x, const float* __restrict__ w2, const float* __restrict__ w3,
                                                                 const float* __restrict__ wr, float* __restrict__ g_h1, float* __restrict__ g_h2,
                                                                 float2* __restrict__ gx, float* __restrict__ bias_part /* [tiles][H1 + H2 + Q] */,
                                                                 const HeadDims d) {
    __shared__ float gs[16][kHdMaxQ + kHdPad], g2s[16][kHdMaxH2 + kHdPad];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int r0 = blockIdx.x * 16;
    float* const bp = bias_part + (size_t)blockIdx.x * (d.H1 + d.H2 + d.Q);
    for (int idx = tid; idx < 16 * d.Q; idx += kHdThreads) {
        const int r = idx / d.Q, c = idx - r * d.Q;
        gs[r][c] = r0 + r < d.N ? g[(size_t)(r0 + r) * d.Q + c] : 0.f;
    }
    __syncthreads();
    if (tid < d.Q) {        // lin3.bias / res.bias: the rows' sum of g, in row order
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += gs[r][tid];
        bp[d.H1 + d.H2 + tid] = s;
    }
    // g_h2 = (g W3) [h2 > 0]: W3 (Q, H2) seen as (k, n)
    for (int nt = wave; nt * 16 < d.H2; nt += kHdWaves) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_times_w(&gs[0][0], kHdMaxQ + kHdPad, w3, d.H2, nt * 16, d.H2, d.Q, acc, fr, fq);
        const int n = nt * 16 + fr;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * fq + t;
            const bool on = n < d.H2 && r0 + r < d.N && h2[(size_t)(r0 + r) * d.H2 + n] > 0.f;
            acc[t] = on ? acc[t] : 0.f;
            if (n < d.H2) {
                g2s[r][n] = acc[t];
                if (r0 + r < d.N) g_h2[(size_t)(r0 + r) * d.H2 + n] = acc[t];
            }
        }
        const float cs = column_sum(acc);
        if (fq == 0 && n < d.H2) bp[d.H1 + n] = cs;
    }
    // gx = (g Wr) x / |x|: Wr (Q, C) seen as (k, n)
    for (int nt = wave; nt * 16 < d.C; nt += kHdWaves) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_times_w(&gs[0][0], kHdMaxQ + kHdPad, wr, d.C, nt * 16, d.C, d.Q, acc, fr, fq);
        const int n = nt * 16 + fr;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * fq + t;
            if (n < d.C && r0 + r < d.N) {
                const float2 v = x[(size_t)(r0 + r) * d.C + n];
                float2 o = make_float2(0.f, 0.f);
                if (!is_origin(v)) {
                    const float s = acc[t] / sqrtf(v.x * v.x + v.y * v.y);
                    o = make_float2(v.x * s, v.y * s);
                }
                gx[(size_t)(r0 + r) * d.C + n] = o;
            }
        }
    }
    __syncthreads();
    // g_h1 = (g_h2 W2) [h1 > 0]: W2 (H2, H1) seen as (k, n)
    for (int nt = wave; nt * 16 < d.H1; nt += kHdWaves) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_times_w(&g2s[0][0], kHdMaxH2 + kHdPad, w2, d.H1, nt * 16, d.H1, d.H2, acc, fr, fq);
        const int n = nt * 16 + fr;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * fq + t;
            const bool on = n < d.H1 && r0 + r < d.N && h1[(size_t)(r0 + r) * d.H1 + n] > 0.f;
            acc[t] = on ? acc[t] : 0.f;
            if (n < d.H1 && r0 + r < d.N) g_h1[(size_t)(r0 + r) * d.H1 + n] = acc[t];
        }
        const float cs = column_sum(acc);
        if (fq == 0 && n < d.H1) bp[n] = cs;
    }
}

static bool head_dims_ok(const fc_echo_head_params* p, int N) {
    return p && N > 0 && p->D > 0 && p->H1 > 0 && p->H1 <= kHdMaxH1 && p->H2 > 0 && p->H2 <= kHdMaxH2 && p->C_in > 0 && p->C_in <= kHdMaxC &&
           p->C_out > 0 && p->C_out <= kHdMaxQ;
}
static HeadDims head_dims(const fc_echo_head_params* p, int N) { return HeadDims{N, p->D, p->H1, p->H2, p->C_in, p->C_out}; }
static size_t al256(size_t b) { return (b + 255) / 256 * 256; }

// the backward pass's products and where their partials lie in the workspace
struct HeadBwdPlan {
    RgProblem gd, gw1, gw2, gw3, gwr;
    size_t off_gh2, off_bias, off_p1, off_p2, off_p3, off_pr, total;
    int tiles;
};
static HeadBwdPlan head_bwd_plan(const fc_echo_head_params* p, int N) {
    HeadBwdPlan pl{};
    const int D = p->D, H1 = p->H1, H2 = p->H2, C = p->C_in, Q = p->C_out;
    pl.gd = rg_problem(nullptr, nullptr, N, D, H1, H1, 1, D, 1);            // g_d = g_h1 W1: A (N, H1), B(k, n) = W1[k D + n]
    pl.gd.slices = 1;                                                        // (never cut: its contraction is H1)
    pl.gd.kchunk = round_up(H1, kRgChunk);
    pl.gw1 = rg_problem(nullptr, nullptr, H1, D, N, 1, H1, D, 1);           // g_h1^T d
    pl.gw2 = rg_problem(nullptr, nullptr, H2, H1, N, 1, H2, H1, 1);         // g_h2^T h1
    pl.gw3 = rg_problem(nullptr, nullptr, Q, H2, N, 1, Q, H2, 1);           // g^T h2
    pl.gwr = rg_problem(nullptr, nullptr, Q, C, N, 1, Q, C, 1, true);       // g^T softAbs(x)
    pl.tiles = (N + 15) / 16;
    size_t off = 0;
    pl.off_gh2 = off;  off += al256((size_t)N * H2 * 4);
    pl.off_bias = off; off += al256((size_t)pl.tiles * (H1 + H2 + Q) * 4);
    pl.off_p1 = off;   off += al256((size_t)pl.gw1.slices * H1 * D * 4);
    pl.off_p2 = off;   off += al256((size_t)pl.gw2.slices * H2 * H1 * 4);
    pl.off_p3 = off;   off += al256((size_t)pl.gw3.slices * Q * H2 * 4);
    pl.off_pr = off;   off += al256((size_t)pl.gwr.slices * Q * C * 4);
    pl.total = off;
    return pl;
}

}  // namespace fc

extern "C" {

size_t fc_echo_head_forward_workspace_bytes(int32_t N, const fc_echo_head_params* p) {
    if (!fc::head_dims_ok(p, N)) return 0;
    const fc::RgProblem g = fc::rg_problem(nullptr, nullptr, N, p->H1, p->D, p->D, 1, 1, p->D);
    return g.slices > 1 ? fc::al256((size_t)g.slices * N * p->H1 * 4) : 0;
}

int fc_echo_head_forward(const float* d, const float* x, const fc_echo_head_params* p, float* h1, float* h2, float* y, void* workspace,
                         size_t workspace_bytes, int32_t N, void* stream) {
    if (!p || N <= 0 || !d || !x || !h1 || !h2 || !y || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->w3 || !p->b3 || !p->wr || !p->br)
        return FC_ERR_BAD_ARGUMENT;
    if (!fc::head_dims_ok(p, N)) return FC_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // h1 = relu(d W1^T + b1): A = d (N, D), B(k, n) = W1[n D + k]
    fc::RgArgs ra{};
    ra.count = 1;
    ra.p[0] = fc::rg_problem(d, p->w1, N, p->H1, p->D, p->D, 1, 1, p->D);
    const bool split = ra.p[0].slices > 1;
    if (split && (!workspace || workspace_bytes < fc_echo_head_forward_workspace_bytes(N, p))) return FC_ERR_WORKSPACE;
    ra.p[0].C = split ? static_cast<float*>(workspace) : h1;
    ra.p[0].bias = p->b1;
    ra.p[0].relu = 1;
    int rc = fc::rg_launch(ra, s);
    if (rc != FC_OK) return rc;
    if (split) {
        fc::RfArgs fa{};
        fa.count = 1;
        fa.t[0] = fc::RfTask{static_cast<const float*>(workspace), h1, nullptr, p->b1, (long)N * p->H1, (long)N * p->H1, ra.p[0].slices, p->H1, 1, 0};
        rc = fc::rf_launch(fa, s);
        if (rc != FC_OK) return rc;
    }
    const fc::HeadDims hd = fc::head_dims(p, N);
    hipLaunchKernelGGL(fc::fc_head_fwd_kernel, dim3((N + 15) / 16), dim3(fc::kHdThreads), 0, s, h1, reinterpret_cast<const float2*>(x), p->w2, p->b2,
                       p->w3, p->b3, p->wr, p->br, h2, y, hd, 1);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

size_t fc_echo_head_backward_workspace_bytes(int32_t N, const fc_echo_head_params* p) {
    if (!fc::head_dims_ok(p, N)) return 0;
    return fc::head_bwd_plan(p, N).total;
}

int fc_echo_head_backward(const float* d, const float* x, const float* h1, const float* h2, const float* g, const fc_echo_head_params* p,
                          float* g_d, float* gx, float* g_h1, void* workspace, size_t workspace_bytes, int32_t N, void* stream) {
    if (!p || N <= 0 || !d || !x || !h1 || !h2 || !g || !g_d || !gx || !g_h1 || !p->w1 || !p->w2 || !p->w3 || !p->wr || !p->g_w1 || !p->g_b1 ||
        !p->g_w2 || !p->g_b2 || !p->g_w3 || !p->g_b3 || !p->g_wr || !p->g_br)
        return FC_ERR_BAD_ARGUMENT;
    if (!fc::head_dims_ok(p, N)) return FC_ERR_UNSUPPORTED;
    fc::HeadBwdPlan pl = fc::head_bwd_plan(p, N);
    if (!workspace || workspace_bytes < pl.total) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    float* g_h2 = reinterpret_cast<float*>(ws + pl.off_gh2);
    float* bias_part = reinterpret_cast<float*>(ws + pl.off_bias);
    const fc::HeadDims hd = fc::head_dims(p, N);
    const int H1 = p->H1, H2 = p->H2, C = p->C_in, Q = p->C_out, D = p->D;
    hipLaunchKernelGGL(fc::fc_head_bwd_kernel, dim3(pl.tiles), dim3(fc::kHdThreads), 0, s, g, h1, h2, reinterpret_cast<const float2*>(x), p->w2, p->w3,
                       p->wr, g_h1, g_h2, reinterpret_cast<float2*>(gx), bias_part, hd);
    if (hipGetLastError() != hipSuccess) return FC_ERR_LAUNCH;
    // one grouped launch: g_d and the four weight gradients' slices
    fc::RgArgs ra{};
    ra.count = 5;
    ra.p[0] = pl.gd;  ra.p[0].A = g_h1; ra.p[0].B = p->w1; ra.p[0].C = g_d;
    struct Wg { fc::RgProblem* g; const float* A; const float* B; size_t off; float* out; };
    Wg wg[4] = {{&pl.gw1, g_h1, d, pl.off_p1, p->g_w1}, {&pl.gw2, g_h2, h1, pl.off_p2, p->g_w2}, {&pl.gw3, g, h2, pl.off_p3, p->g_w3},
                {&pl.gwr, g, x, pl.off_pr, p->g_wr}};
    fc::RfArgs fa{};
    fa.count = 0;
    for (int q = 0; q < 4; ++q) {
        fc::RgProblem& gp = *wg[q].g;
        gp.A = wg[q].A;
        gp.B = wg[q].B;
        const bool split = gp.slices > 1;
        gp.C = split ? reinterpret_cast<float*>(ws + wg[q].off) : wg[q].out;
        ra.p[1 + q] = gp;
        if (split)
            fa.t[fa.count++] = fc::RfTask{reinterpret_cast<const float*>(ws + wg[q].off), wg[q].out, nullptr, nullptr, (long)gp.M * gp.N,
                                          (long)gp.M * gp.N, gp.slices, gp.N, 0, 0};
    }
    int rc = fc::rg_launch(ra, s);
    if (rc != FC_OK) return rc;
    // the bias gradients: per-tile column sums in tile order (lin3.bias and res.bias see the same cotangent)
    const long bstride = H1 + H2 + Q;
    fa.t[fa.count++] = fc::RfTask{bias_part, p->g_b1, nullptr, nullptr, H1, bstride, pl.tiles, H1, 0, 0};
    fa.t[fa.count++] = fc::RfTask{bias_part + H1, p->g_b2, nullptr, nullptr, H2, bstride, pl.tiles, H2, 0, 0};
    fa.t[fa.count++] = fc::RfTask{bias_part + H1 + H2, p->g_b3, p->g_br, nullptr, Q, bstride, pl.tiles, Q, 0, 0};
    (void)C; (void)D;
    return fc::rf_launch(fa, s);
}

}  // extern "C"
