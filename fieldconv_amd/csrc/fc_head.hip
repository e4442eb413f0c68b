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
constexpr int kRgDepth = 2;             // chunks in flight per thread (registers; more would cost the occupancy that hides the rest)
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
    int dbg;                // development (FC_DEBUG_RG): 1 no products, 2 no loads, 4 no staging
};

__device__ __forceinline__ float soft_abs_of(float2 v) { return is_origin(v) ? 0.f : sqrtf(v.x * v.x + v.y * v.y); }

// One 64 x 64 tile of one problem for one k slice.  AK / BN: the operand's contiguous direction is k (A) / n (B) -- the thread index runs
// along it, so a wavefront's loads are whole segments.  A thread's eight elements per operand and chunk lie a constant step apart; rows
// and k entries beyond the problem are read clamped (always valid memory) and multiplied by zero: no branch per element.
template <bool AK, bool BN, bool BABS>
__device__ __forceinline__ void rgemm_tile(const RgProblem& g, const int tm, const int tn, const int slice, float (&As)[2][kRgChunk][kRgStride],
                                           float (&Bs)[2][kRgChunk][kRgStride], const int dbg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = tm * kRgTile, n0 = tn * kRgTile;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int fr = lane & 15, fq = lane >> 4;
    // the problem's fields, read once (scalar registers)
    const float* const gA = g.A;
    const float* const gB = g.B;
    const int gM = g.M, gN = g.N;
    const long sam = g.sam, sak = g.sak, sbk = g.sbk, sbn = g.sbn;
    const int kbeg = slice * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    // element j of this thread: A (am0 + AM_STEP j, ak0 + AK_STEP j), B (bk0 + BK_STEP j, bn0 + BN_STEP j)
    constexpr int AM_STEP = AK ? kRgThreads / kRgChunk : 0, AK_STEP = AK ? 0 : kRgThreads / kRgTile;
    constexpr int BN_STEP = BN ? 0 : kRgThreads / kRgChunk, BK_STEP = BN ? kRgThreads / kRgTile : 0;
    const int am0 = AK ? tid / kRgChunk : tid % kRgTile, ak0 = AK ? tid % kRgChunk : tid / kRgTile;
    const int bn0 = BN ? tid % kRgTile : tid / kRgChunk, bk0 = BN ? tid / kRgTile : tid % kRgChunk;
    // A thread's loads walk eight pointers per operand, advanced by one chunk per round (rows / columns beyond the problem are clamped to
    // the last valid one: every address is valid memory); the LDS slots are fixed per thread.  Raw loads only, all sixteen in flight -- a
    // select on a loaded value would make the compiler wait for it there and then: entries outside the problem are zeroed when the chunk
    // is staged, and only edge tiles and the last, partial chunk of a slice take that path at all.
    typedef typename std::conditional<BABS, float2, float>::type BT;
    const float* pa[kRgPer];
    const BT* pb[kRgPer];
    int sa[kRgPer], sb[kRgPer];
    unsigned amask = 0, bmask = 0;
#pragma unroll
    for (int j = 0; j < kRgPer; ++j) {
        const int am = am0 + AM_STEP * j, ak = ak0 + AK_STEP * j;
        pa[j] = gA + (long)min(m0 + am, gM - 1) * sam + (long)(kbeg + ak) * sak;
        sa[j] = ak * kRgStride + (am ^ rg_swz(ak));
        amask |= (m0 + am < gM ? 1u : 0u) << j;
        const int bn = bn0 + BN_STEP * j, bk = bk0 + BK_STEP * j;
        pb[j] = reinterpret_cast<const BT*>(gB) + (long)(kbeg + bk) * sbk + (long)min(n0 + bn, gN - 1) * sbn;
        sb[j] = bk * kRgStride + (bn ^ rg_swz(bk));
        bmask |= (n0 + bn < gN ? 1u : 0u) << j;
    }
    const bool interior = m0 + kRgTile <= gM && n0 + kRgTile <= gN;
    const long stepa = (long)kRgChunk * sak, stepb = (long)kRgChunk * sbk;
    // kRgDepth chunks are in flight per thread (registers)
    float ra[kRgDepth][kRgPer];
    BT rb[kRgDepth][kRgPer];
    auto fetch = [&](auto dc, const int k0) {                  // chunks are fetched in k order (the pointers walk)
        constexpr int d = decltype(dc)::value;
        if (dbg & 2) return;
        if (k0 + kRgChunk <= kend) {
#pragma unroll
            for (int j = 0; j < kRgPer; ++j) {
                ra[d][j] = *pa[j];
                rb[d][j] = *pb[j];
                pa[j] += stepa;
                pb[j] += stepb;
            }
        } else {                // the slice's last, partial chunk: k clamped to its last entry
#pragma unroll
            for (int j = 0; j < kRgPer; ++j) {
                const int ka = k0 + ak0 + AK_STEP * j, kb = k0 + bk0 + BK_STEP * j;
                ra[d][j] = pa[j][(long)(min(ka, kend - 1) - ka) * sak];
                rb[d][j] = pb[j][(long)(min(kb, kend - 1) - kb) * sbk];
            }
        }
    };
    float* const Asf = &As[0][0][0];
    float* const Bsf = &Bs[0][0][0];
    auto bval = [](const BT& v) {
        if constexpr (BABS) return soft_abs_of(v);
        else return v;
    };
    auto stage = [&](auto dc, const int kf, const int buf) {     // chunk kf, held by register set d, into LDS buffer buf
        constexpr int d = decltype(dc)::value;
        if (dbg & 4) return;
        float* const ad = Asf + buf * (kRgChunk * kRgStride);
        float* const bd = Bsf + buf * (kRgChunk * kRgStride);
        if (interior && kf + kRgChunk <= kend) {
#pragma unroll
            for (int j = 0; j < kRgPer; ++j) {
                ad[sa[j]] = ra[d][j];
                bd[sb[j]] = bval(rb[d][j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kRgPer; ++j) {
                const int ak = ak0 + AK_STEP * j, bk = bk0 + BK_STEP * j;
                ad[sa[j]] = ((amask >> j) & 1u) && kf + ak < kend ? ra[d][j] : 0.f;
                bd[sb[j]] = ((bmask >> j) & 1u) && kf + bk < kend ? bval(rb[d][j]) : 0.f;
            }
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kbeg < kend) {
        static_for<0, kRgDepth>([&](auto dc) {
            const int k0 = kbeg + decltype(dc)::value * kRgChunk;
            if (k0 < kend) fetch(dc, k0);
        });
        int buf = 0;
        for (int kg = kbeg; kg < kend; kg += kRgDepth * kRgChunk) {
            static_for<0, kRgDepth>([&](auto dc) {
                const int k0 = kg + decltype(dc)::value * kRgChunk;
                if (k0 < kend) {                                           // (uniform)
                    stage(dc, k0, buf);
                    __syncthreads();                                       // chunk k0 is in LDS; the products of the chunk before it (other buffer) are done
                    if (k0 + kRgDepth * kRgChunk < kend) fetch(dc, k0 + kRgDepth * kRgChunk);
                    if (!(dbg & 1))
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
                    buf ^= 1;
                }
            });
        }
    }
    // The tile leaves through LDS: the matrix pipe's result layout (lane l: column l & 15, register t: row 4 (l >> 4) + t) would store
    // 64-byte pieces; from LDS every row of the tile goes out as 256 contiguous bytes (11 MB of outputs and partials per backward pass).
    float* C = g.C + (g.slices > 1 ? (size_t)slice * gM * gN : 0);
    const bool epi = g.slices == 1;
    const float* const bias = epi ? g.bias : nullptr;
    const bool relu = epi && g.relu;
    __syncthreads();                                // everyone's products are done: the operand buffers are free
    float* const tile = &As[0][0][0];               // [64][64 + 4] floats (17 KB of the 20 KB buffer pair)
    constexpr int TS = kRgTile + 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) tile[(wm + 16 * i + 4 * fq + t) * TS + wn + 16 * j + fr] = acc[i][j][t];
    __syncthreads();
    if ((gN & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 16 + (tid >> 4), c = (tid & 15) * 4;
            const int m = m0 + r, n = n0 + c;
            if (m < gM && n < gN) {                 // (n < N and N a multiple of 4: the four columns exist)
                float4 v = *reinterpret_cast<const float4*>(tile + r * TS + c);
                if (bias) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(C + (size_t)m * gN + n) = v;
            }
        }
    } else {
        for (int idx = tid; idx < kRgTile * kRgTile; idx += kRgThreads) {
            const int r = idx / kRgTile, c = idx % kRgTile;
            const int m = m0 + r, n = n0 + c;
            if (m < gM && n < gN) {
                float v = tile[r * TS + c];
                if (bias) v += bias[n];
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)m * gN + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(kRgThreads) void fc_rgemm_kernel(const RgArgs args) {
    __shared__ float As[2][kRgChunk][kRgStride], Bs[2][kRgChunk][kRgStride];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kRgMaxProblems; ++q)
        if (q < args.count && (int)blockIdx.x >= args.p[q].block0) pi = q;
    pi = __builtin_amdgcn_readfirstlane(pi);           // (uniform: the problem's fields are scalar loads from the kernel arguments)
    const RgProblem& g = args.p[pi];
    int b = blockIdx.x - g.block0;
    const int tm = b % g.tiles_m;
    b /= g.tiles_m;
    const int tn = b % g.tiles_n, slice = b / g.tiles_n;
    const bool ak = g.sak == 1, bn = g.sbn == 1;
    // (B as softAbs of complex numbers: with n as its contiguous direction only -- rg_launch checks)
    if (g.b_abs) {
        if (ak) rgemm_tile<true, true, true>(g, tm, tn, slice, As, Bs, args.dbg);
        else rgemm_tile<false, true, true>(g, tm, tn, slice, As, Bs, args.dbg);
    } else if (ak && bn) rgemm_tile<true, true, false>(g, tm, tn, slice, As, Bs, args.dbg);
    else if (ak) rgemm_tile<true, false, false>(g, tm, tn, slice, As, Bs, args.dbg);
    else if (bn) rgemm_tile<false, true, false>(g, tm, tn, slice, As, Bs, args.dbg);
    else rgemm_tile<false, false, false>(g, tm, tn, slice, As, Bs, args.dbg);
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
    int wave;               // one wavefront per output entry (few entries, many parts)
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
    if (t.wave) {
        // few sums of many parts (the bias gradients: one number per channel from every 16-row tile): a wavefront per sum -- lanes take
        // every 64th part in order, then a fixed butterfly
        const int lane = threadIdx.x & 63;
        const long col = (long)((int)blockIdx.x - t.block0) * (kRfThreads / 64) + (threadIdx.x >> 6);
        if (col >= t.count) return;
        float s = 0.f;
        for (int p = lane; p < t.parts; p += 64) s += t.part[(long)p * t.stride + col];
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) s += __shfl_xor(s, dd, 64);
        if (lane == 0) {
            t.out[col] = s;
            if (t.out2) t.out2[col] = s;
        }
        return;
    }
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
    if (tiles < 2 * cus && g.K >= 8 * kRgChunk) {
        const long want = (3L * cus + tiles - 1) / tiles;                  // about three workgroups per CU: they hide each other's load latency
        const long by_len = (g.K + 4 * kRgChunk - 1) / (4 * kRgChunk);     // at least 128 k entries per slice (requested whole: kRgDepth)
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
        if (a.p[q].b_abs && a.p[q].sbn != 1) return FC_ERR_UNSUPPORTED;
        a.p[q].block0 = blocks;
        blocks += a.p[q].tiles_m * a.p[q].tiles_n * a.p[q].slices;
    }
    if (blocks == 0) return FC_OK;
    static const int dbg = [] { const char* e = dev_env("FC_DEBUG_RG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    hipLaunchKernelGGL(fc_rgemm_kernel, dim3(blocks), dim3(kRgThreads), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

static int rf_launch(RfArgs& a, hipStream_t stream) {
    int blocks = 0;
    for (int q = 0; q < a.count; ++q) {
        a.t[q].block0 = blocks;
        const long per = a.t[q].wave ? kRfThreads / 64 : kRfThreads;
        blocks += (int)((a.t[q].count + per - 1) / per);
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

// A lane first requests a batch of its weight entries -- raw, clamped loads: every one valid memory, none behind a condition, so all are in
// flight together (one round trip to L2 per 16 entries instead of one per entry) -- then runs the instruction chain, masking at the use.
constexpr int kHdBatch = 16;

// acc += A[16 rows][K] (LDS, row stride lda) . W^T, W (n, k) row-major with row stride ldw (a torch Linear weight): column tile n0
__device__ __forceinline__ f32x4 tile_times_wt(const float* a_lds, int lda, const float* __restrict__ w, int ldw, int n0, int nmax, int K, f32x4 acc,
                                               int fr, int fq) {
    const int kq = (K + 3) / 4;                    // k entries per lane group
    const int n = n0 + fr;
    const float* wrow = w + (size_t)min(n, nmax - 1) * ldw;
    const float keep = n < nmax ? 1.f : 0.f;
    const bool vec = (K & 15) == 0 && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;      // four k entries per load
    for (int j0 = 0; j0 < kq; j0 += kHdBatch) {
        float bv[kHdBatch];
        if (vec) {
#pragma unroll
            for (int j = 0; j < kHdBatch; j += 4) {
                const float4 b4 = *reinterpret_cast<const float4*>(wrow + min(fq * kq + j0 + j, K - 4));
                bv[j] = b4.x; bv[j + 1] = b4.y; bv[j + 2] = b4.z; bv[j + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kHdBatch; ++j) bv[j] = wrow[min(fq * kq + j0 + j, K - 1)];
        }
#pragma unroll
        for (int j = 0; j < kHdBatch; ++j) {
            const int k = fq * kq + j0 + j;
            const float av = a_lds[fr * lda + min(k, K - 1)];
            const float b = (j0 + j < kq && k < K) ? bv[j] * keep : 0.f;       // (entries past the contraction: a zero product)
            acc = mfma16(av, b, acc);
        }
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
    for (int j0 = 0; j0 < kq; j0 += kHdBatch) {
        float bv[kHdBatch];
#pragma unroll
        for (int j = 0; j < kHdBatch; ++j) bv[j] = w[(size_t)min(fq * kq + j0 + j, K - 1) * ldw + nc];
#pragma unroll
        for (int j = 0; j < kHdBatch; ++j) {
            const int k = fq * kq + j0 + j;
            const float av = a_lds[fr * lda + min(k, K - 1)];
            const float b = (j0 + j < kq && k < K) ? bv[j] * keep : 0.f;
            acc = mfma16(av, b, acc);
        }
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
        fa.t[0] = fc::RfTask{static_cast<const float*>(workspace), h1, nullptr, p->b1, (long)N * p->H1, (long)N * p->H1, ra.p[0].slices, p->H1, 1, 0, 0};
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
                                          (long)gp.M * gp.N, gp.slices, gp.N, 0, 0, 0};
    }
    int rc = fc::rg_launch(ra, s);
    if (rc != FC_OK) return rc;
    // the bias gradients: per-tile column sums in tile order (lin3.bias and res.bias see the same cotangent)
    const long bstride = H1 + H2 + Q;
    fa.t[fa.count++] = fc::RfTask{bias_part, p->g_b1, nullptr, nullptr, H1, bstride, pl.tiles, H1, 0, 0, 1};
    fa.t[fa.count++] = fc::RfTask{bias_part + H1, p->g_b2, nullptr, nullptr, H2, bstride, pl.tiles, H2, 0, 0, 1};
    fa.t[fa.count++] = fc::RfTask{bias_part + H1 + H2, p->g_b3, p->g_br, nullptr, Q, bstride, pl.tiles, Q, 0, 0, 1};
    (void)C; (void)D;
    return fc::rf_launch(fa, s);
}

}  // extern "C"
