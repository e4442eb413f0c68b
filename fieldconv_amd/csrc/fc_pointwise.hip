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
// Real-expanded on v_mfma_f32_16x16x4_f32 with k_real = 2k + c: the interleaved (re,im) input row is
// the A operand exactly as it lies in memory; the workgroup stages the two expanded filter planes
//   Wr[m][2k] = Re Wc, Wr[m][2k+1] = -Im Wc      (real part of the output)
//   Wi[m][2k] = Im Wc, Wi[m][2k+1] =  Re Wc      (imaginary part)
// in LDS once (row stride = slab_stride, conflict-free float4 fragments) while the first input
// fragments are already in flight, and every wavefront then walks blocks of 16 vertices with up to 64
// outputs accumulated together.  D rows = vertices, columns = outputs: each accumulator row is one
// 128-byte store.  Loads are unconditional from clamped addresses (no control flow around them).
// addend (optional, (N,M) complex64, may alias out): out = product + addend -- the block-level backward pass adds the residual
// branch's input gradient to the convolution's this way instead of a launch of its own.
constexpr int kLinThreads = 256;
constexpr int kLinWaves = kLinThreads / kWave;
constexpr int kLinChunk = 6;      // k blocks (8 complex inputs each) loaded ahead of their MFMAs
constexpr int kLinTiles = 4;      // output tiles accumulated together
constexpr int kLinStage = 16;     // filter entries in flight per thread while staging

__host__ __device__ inline int lin_plane_floats(int M, int K) { return round_up(M, 16) * slab_stride(round_up(2 * K, 16)); }

__device__ __forceinline__ void lin_load_fragments(float4 (&a)[kLinChunk], const float* row, int kc, int fq, int K, bool vec,
                                                   float keep) {
    // raw, clamped loads first -- all in flight together -- then the masks: a select right behind its load makes the compiler wait for
    // that load before it issues the next one (six round trips to L2 / HBM per fragment set instead of one)
    const int last = 2 * K - 1;
    if (vec) {
#pragma unroll
        for (int u = 0; u < kLinChunk; ++u) a[u] = *reinterpret_cast<const float4*>(row + min(16 * (kc + u) + 4 * fq, 2 * K - 4));
        return;                 // (masked where they are used: lin_mask_fragments)
    }
    // odd channel counts / unaligned rows: element by element (a rare path: its loads stay behind their conditions)
#pragma unroll
    for (int u = 0; u < kLinChunk; ++u) {
        const int kr0 = 16 * (kc + u) + 4 * fq;
        a[u].x = kr0 <= last ? row[min(kr0, last)] : 0.f;
        a[u].y = kr0 + 1 <= last ? row[min(kr0 + 1, last)] : 0.f;
        a[u].z = kr0 + 2 <= last ? row[min(kr0 + 2, last)] : 0.f;
        a[u].w = kr0 + 3 <= last ? row[min(kr0 + 3, last)] : 0.f;
        a[u].x *= keep; a[u].y *= keep; a[u].z *= keep; a[u].w *= keep;
    }
}

// the vector path's raw fragments: rows beyond the mesh and k entries beyond the row count as zero
__device__ __forceinline__ void lin_mask_fragments(float4 (&a)[kLinChunk], int kc, int fq, int K, bool vec, float keep) {
    if (!vec) return;
    const bool on = keep != 0.f;
#pragma unroll
    for (int u = 0; u < kLinChunk; ++u)
        if (!(on && 16 * (kc + u) + 4 * fq <= 2 * K - 4)) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool TRANSPOSED>
__global__ __launch_bounds__(kLinThreads) void tangent_lin_kernel(const float2* __restrict__ in, const float* __restrict__ wre,
                                                                  const float* __restrict__ wim, float2* out,
                                                                  const float2* addend, int N, int K, int M, int ldw, int split) {
    extern __shared__ float lds[];
    const int KR = round_up(2 * K, 16), KS = slab_stride(KR), MP = round_up(M, 16);
    float* Wr = lds;
    float* Wi = lds + MP * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int nblocks = (N + 15) / 16, MT = MP / 16, KB = KR / 16;
    const float* in_f = reinterpret_cast<const float*>(in);
    const bool vec = (K & 1) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;       // 16-byte aligned rows

    // split (meshes whose 16-vertex blocks cannot occupy the chip with four of them per workgroup): a workgroup takes ONE block at a time
    // and its four wavefronts share the block's output tiles -- a wavefront's serial chain of fp32 MFMAs (32 cycles each) is a quarter as
    // long, the input rows are read by all four (L1 hits).  Same sums in the same order: bit-identical to the unsplit walk.
    const int bstep = split ? gridDim.x : gridDim.x * kLinWaves;
    const int tcount = split ? 1 : kLinTiles, mstart = split ? wave : 0, mstride = split ? kLinWaves : kLinTiles;
    // first input fragments of this wavefront: in flight while the filter is staged
    const int blk0 = split ? blockIdx.x : blockIdx.x * kLinWaves + wave;
    float4 a[kLinChunk];
    {
        const int n = min(blk0, nblocks - 1) * 16 + fr;
        lin_load_fragments(a, in_f + (size_t)min(n, N - 1) * 2 * K, 0, fq, K, vec, n < N ? 1.f : 0.f);
    }

    for (int idx = threadIdx.x; idx < 2 * MP * KS; idx += kLinThreads) lds[idx] = 0.f;
    __syncthreads();
    // filter entries in memory order ([M][K] forward, [K][M] transposed), kLinStage loads in flight
    const int total = M * K, cols = TRANSPOSED ? M : K;
    for (int base = 0; base < total; base += kLinStage * kLinThreads) {
        float re[kLinStage], im[kLinStage];
        int at[kLinStage];
#pragma unroll
        for (int u = 0; u < kLinStage; ++u) {
            const int idx = min(base + u * kLinThreads + (int)threadIdx.x, total - 1);
            const int r = idx / cols, c = idx - r * cols;
            re[u] = wre[(size_t)r * ldw + c];
            im[u] = TRANSPOSED ? -wim[(size_t)r * ldw + c] : wim[(size_t)r * ldw + c];
            at[u] = TRANSPOSED ? c * KS + 2 * r : r * KS + 2 * c;
        }
#pragma unroll
        for (int u = 0; u < kLinStage; ++u) {
            if (base + u * kLinThreads + (int)threadIdx.x < total) {
                *reinterpret_cast<float2*>(Wr + at[u]) = make_float2(re[u], -im[u]);
                *reinterpret_cast<float2*>(Wi + at[u]) = make_float2(im[u], re[u]);
            }
        }
    }
    __syncthreads();

    bool loaded = true;
    for (int blk = blk0; blk < nblocks; blk += bstep) {
        const int n = blk * 16 + fr;
        const float* row = in_f + (size_t)min(n, N - 1) * 2 * K;
        const float keep = n < N ? 1.f : 0.f;
        for (int mg = mstart; mg < MT; mg += mstride) {
            f32x4 acc_re[kLinTiles], acc_im[kLinTiles];
#pragma unroll
            for (int t = 0; t < kLinTiles; ++t) { acc_re[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_im[t] = acc_re[t]; }
            for (int kc = 0; kc < KB; kc += kLinChunk) {
                if (!loaded) lin_load_fragments(a, row, kc, fq, K, vec, keep);
                loaded = false;
                lin_mask_fragments(a, kc, fq, K, vec, keep);
#pragma unroll
                for (int t = 0; t < kLinTiles; ++t) {
                    if (t < tcount && mg + t < MT) {
                        const float* pr = Wr + ((mg + t) * 16 + fr) * KS + 4 * fq;
                        const float* pi = Wi + ((mg + t) * 16 + fr) * KS + 4 * fq;
#pragma unroll
                        for (int u = 0; u < kLinChunk; ++u) {
                            if (kc + u < KB) {
                                const float4 br = *reinterpret_cast<const float4*>(pr + 16 * (kc + u));
                                const float4 bi = *reinterpret_cast<const float4*>(pi + 16 * (kc + u));
                                acc_re[t] = mfma16(a[u].x, br.x, acc_re[t]); acc_im[t] = mfma16(a[u].x, bi.x, acc_im[t]);
                                acc_re[t] = mfma16(a[u].y, br.y, acc_re[t]); acc_im[t] = mfma16(a[u].y, bi.y, acc_im[t]);
                                acc_re[t] = mfma16(a[u].z, br.z, acc_re[t]); acc_im[t] = mfma16(a[u].z, bi.z, acc_im[t]);
                                acc_re[t] = mfma16(a[u].w, br.w, acc_re[t]); acc_im[t] = mfma16(a[u].w, bi.w, acc_im[t]);
                            }
                        }
                    }
                }
            }
            // (the addend may be `out` itself: each entry is read and written by this thread only.  A tile's four entries are requested
            // together, from clamped -- always valid -- addresses, before the first is used: one round trip per tile instead of four)
#pragma unroll
            for (int t = 0; t < kLinTiles; ++t) {
                if (t < tcount && mg + t < MT) {
                    const int mo = (mg + t) * 16 + fr;
                    float2 add[4];
                    if (addend) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) add[j] = addend[(size_t)min(blk * 16 + 4 * fq + j, N - 1) * M + min(mo, M - 1)];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int no = blk * 16 + 4 * fq + j;
                        if (no < N && mo < M) {
                            float2 v = make_float2(acc_re[t][j], acc_im[t][j]);
                            if (addend) {
                                v.x += add[j].x;
                                v.y += add[j].y;
                            }
                            out[(size_t)no * M + mo] = v;
                        }
                    }
                }
            }
        }
    }
}

// The same product WITHOUT the filter's way through LDS (the default whenever its shape conditions hold: it is faster than the staged walk
// above at every mesh size measured, tools/time_tangent_lin.py: 4.8 against 7.2 us at 1 024 vertices and 48 channels, 12.3 / 18.6 at
// 20 000, 72 / 96 at 200 000 x 64): one 16-vertex block per workgroup at a time, its output tiles dealt to the wavefronts; a wavefront
// needs only its own 16 output channels of the filter -- requested once, with each block's input rows, as a batch of loads straight into
// the matrix pipe's operand layout (lane l: output channel l & 15, input channels 8 kb + 2 (l >> 4) and the next) -- so the kernel has no
// LDS, no barrier and one round trip to L2 before its instruction chain, where the staged walk zeroes 67 KB, stages the filter and
// synchronises twice.  The products and their order are the staged walk's: bit-identical.  K % 8 == 0, K <= 64, more than one output
// tile, 16-byte aligned rows; otherwise the staged walk.
constexpr int kLinDirectKB = 8;          // k blocks of 16 real entries: up to 64 complex input channels

template <bool TRANSPOSED>
__global__ __launch_bounds__(kLinThreads) void tangent_lin_direct_kernel(const float2* __restrict__ in, const float* __restrict__ wre,
                                                                         const float* __restrict__ wim, float2* out, const float2* addend, int N,
                                                                         int K, int M, int ldw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int nblocks = (N + 15) / 16, MT = (M + 15) / 16, KB = K / 8;
    const float* in_f = reinterpret_cast<const float*>(in);
    for (int mt = wave; mt < MT; mt += kLinWaves) {
        const int m = mt * 16 + fr, mc = min(m, M - 1);
        // my filter fragments: (re, -im | im, re) of two input channels per k block
        float2 w_re[kLinDirectKB], w_im[kLinDirectKB];
#pragma unroll
        for (int kb = 0; kb < kLinDirectKB; ++kb) {
            const int c0 = min(8 * kb + 2 * fq, K - 2);
            if (TRANSPOSED) {
                w_re[kb] = make_float2(wre[(size_t)c0 * ldw + mc], wre[(size_t)(c0 + 1) * ldw + mc]);
                w_im[kb] = make_float2(-wim[(size_t)c0 * ldw + mc], -wim[(size_t)(c0 + 1) * ldw + mc]);
            } else {
                w_re[kb] = *reinterpret_cast<const float2*>(wre + (size_t)mc * ldw + c0);
                w_im[kb] = *reinterpret_cast<const float2*>(wim + (size_t)mc * ldw + c0);
            }
        }
        for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
            const int n = blk * 16 + fr;
            const float* row = in_f + (size_t)min(n, N - 1) * 2 * K;
            float4 a[kLinDirectKB];
#pragma unroll
            for (int kb = 0; kb < kLinDirectKB; ++kb) a[kb] = *reinterpret_cast<const float4*>(row + min(16 * kb + 4 * fq, 2 * K - 4));
            float2 add[4];
            const int mo = mt * 16 + fr;
            if (addend) {
#pragma unroll
                for (int j = 0; j < 4; ++j) add[j] = addend[(size_t)min(blk * 16 + 4 * fq + j, N - 1) * M + min(mo, M - 1)];
            }
            f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;
            const bool row_on = n < N;
#pragma unroll
            for (int kb = 0; kb < kLinDirectKB; ++kb) {
                if (kb < KB) {
                    const float4 av = row_on ? a[kb] : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float2 r2 = w_re[kb], i2 = w_im[kb];
                    acc_re = mfma16(av.x, r2.x, acc_re);  acc_im = mfma16(av.x, i2.x, acc_im);
                    acc_re = mfma16(av.y, -i2.x, acc_re); acc_im = mfma16(av.y, r2.x, acc_im);
                    acc_re = mfma16(av.z, r2.y, acc_re);  acc_im = mfma16(av.z, i2.y, acc_im);
                    acc_re = mfma16(av.w, -i2.y, acc_re); acc_im = mfma16(av.w, r2.y, acc_im);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int no = blk * 16 + 4 * fq + j;
                if (no < N && mo < M) {
                    float2 v = make_float2(acc_re[j], acc_im[j]);
                    if (addend) {
                        v.x += add[j].x;
                        v.y += add[j].y;
                    }
                    out[(size_t)no * M + mo] = v;
                }
            }
        }
    }
}

// Weight gradient gW[o,i] = sum_n gy[n,o] conj(x[n,i]) on MFMA with the vertices as the k dimension.
// One wavefront per (o-tile, i-tile) pair (blockIdx.y picks the group of up to 16 pairs); workgroup g
// walks every gridDim.x-th block of 16 vertices, so a wavefront owns its 16x16 tile of the partial
// outright: no cross-wavefront combination, no LDS.  partial[g][o][i] is reduced by
// tangent_lin_gw_reduce_kernel in a fixed order.
constexpr int kLinGwGroups = 256;
constexpr int kLinGwPairs = 16;     // wavefronts per workgroup

__global__ __launch_bounds__(kLinGwPairs * kWave) void tangent_lin_gw_kernel(const float2* __restrict__ x,
                                                                             const float2* __restrict__ gy,
                                                                             float2* __restrict__ partial, int N, int I, int O) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int IT = (I + 15) / 16, OT = (O + 15) / 16;
    const int pair = blockIdx.y * kLinGwPairs + wave;
    if (pair >= OT * IT) return;
    const int a = pair / IT, b = pair - a * IT;
    const int co = a * 16 + fr, ci = b * 16 + fr;
    const size_t go = min(co, O - 1), xi = min(ci, I - 1);
    const float keep_o = co < O ? 1.f : 0.f, keep_i = ci < I ? 1.f : 0.f;
    f32x4 are = {0.f, 0.f, 0.f, 0.f}, aim = are;
    const int nblocks = (N + 15) / 16;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        float2 g[4], v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {      // unconditional loads from clamped rows, zeroed below
            const size_t n = min(blk * 16 + 4 * fq + s, N - 1);
            g[s] = gy[n * O + go];
            v[s] = x[n * I + xi];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float ko = (blk * 16 + 4 * fq + s < N) ? keep_o : 0.f;
            const float gr = g[s].x * ko, gi = g[s].y * ko, vr = v[s].x * keep_i, vi = v[s].y * keep_i;
            // re += g.re x.re + g.im x.im ; im += g.im x.re - g.re x.im
            are = mfma16(gr, vr, are); aim = mfma16(gi, vr, aim);
            are = mfma16(gi, vi, are); aim = mfma16(-gr, vi, aim);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = a * 16 + 4 * fq + j;
        if (o < O && ci < I) partial[((size_t)blockIdx.x * O + o) * I + ci] = make_float2(are[j], aim[j]);
    }
}

// Fixed-order sum of the workgroup partials: 64 entries per block, four wavefronts take every fourth
// partial each and combine through LDS.
__global__ __launch_bounds__(256) void tangent_lin_gw_reduce_kernel(const float2* __restrict__ partial, float* __restrict__ g_re,
                                                                    float* __restrict__ g_im, int nparts, int OI) {
    __shared__ float2 sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    float re = 0.f, im = 0.f;
    if (idx < OI) {
#pragma unroll 8
        for (int p = wave; p < nparts; p += 4) {
            const float2 v = partial[(size_t)p * OI + idx];
            re += v.x;
            im += v.y;
        }
    }
    sh[wave][lane] = make_float2(re, im);
    __syncthreads();
    if (wave == 0 && idx < OI) {
        const float2 a = sh[0][lane], b = sh[1][lane], c = sh[2][lane], d = sh[3][lane];
        g_re[idx] = (a.x + b.x) + (c.x + d.x);
        g_im[idx] = (a.y + b.y) + (c.y + d.y);
    }
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
// gbias[c] = sum_n [r + b > 0] g_r.  A workgroup owns a contiguous range of rows; its threads are laid
// out as (row lane, channel) so that consecutive threads touch consecutive entries, every thread
// walks its rows in order, and the row lanes are combined through LDS in a fixed order.
constexpr int kNonlinThreads = 256;
constexpr int kNonlinMaxGroups = 512;

__host__ __device__ inline int nonlin_rows_per_group(int N) {
    int rows = (N + kNonlinMaxGroups - 1) / kNonlinMaxGroups;
    return rows < 8 ? 8 : rows;
}

__global__ __launch_bounds__(kNonlinThreads) void tangent_nonlin_bwd_kernel(const float2* __restrict__ x,
                                                                            const float* __restrict__ bias,
                                                                            const float2* __restrict__ gy,
                                                                            float2* __restrict__ gx, float* __restrict__ partial,
                                                                            int N, int C, int rows_per_group) {
    extern __shared__ float sh[];      // [row lanes][C]
    const int L = C < kNonlinThreads ? kNonlinThreads / C : 1;       // row lanes
    const int r0 = blockIdx.x * rows_per_group;
    const int rows = min(rows_per_group, N - r0);
    const int rl = threadIdx.x / C;
    if (rl < L) {
        for (int c = threadIdx.x - rl * C; c < C; c += kNonlinThreads) {
            const float b = bias[c];
            float acc = 0.f;
            for (int r = rl; r < rows; r += L) {
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
            sh[rl * C + c] = acc;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kNonlinThreads) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += sh[l * C + c];
        partial[(size_t)blockIdx.x * C + c] = s;
    }
}

// One wavefront per channel: lanes take every 64th partial in order, then a fixed butterfly.
__global__ __launch_bounds__(64) void tangent_nonlin_gb_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gbias,
                                                                      int nparts, int C) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int p = lane; p < nparts; p += 64) s += partial[(size_t)p * C + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) gbias[c] = s;
}

// softAbs (reference utils/field.py:29-37: |z| outside the origin box, 0 inside; ECHOBlock's residual branch, nn/echo_block.py:103) and its
// VJP g z/|z| -- one launch each instead of a dozen elementwise torch kernels and their autograd nodes
__global__ void soft_abs_fwd_kernel(const float2* __restrict__ x, float* __restrict__ y, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float2 v = x[idx];
    y[idx] = is_origin(v) ? 0.f : sqrtf(v.x * v.x + v.y * v.y);
}
__global__ void soft_abs_bwd_kernel(const float2* __restrict__ x, const float* __restrict__ gy, float2* __restrict__ gx, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float2 v = x[idx];
    float2 o = make_float2(0.f, 0.f);
    if (!is_origin(v)) {
        const float s = gy[idx] / sqrtf(v.x * v.x + v.y * v.y);
        o = make_float2(v.x * s, v.y * s);
    }
    gx[idx] = o;
}

}  // namespace fc

namespace fc {
int bias_partials_reduce_impl(const float* partials, int nparts, int C, float* g_bias, hipStream_t stream) {
    hipLaunchKernelGGL(tangent_nonlin_gb_reduce_kernel, dim3(C), dim3(64), 0, stream, partials, g_bias, nparts, C);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}
}  // namespace fc

extern "C" size_t fc_tangent_lin_backward_workspace_bytes(int32_t N, int32_t I, int32_t O);

namespace fc {
static int lin_grid(int N) {
    const int groups = ((N + 15) / 16 + kLinWaves - 1) / kLinWaves;
    return groups < 4 * num_cus() ? groups : 4 * num_cus();
}
// one 16-vertex block per workgroup at a time, its output tiles dealt to the four wavefronts (tangent_lin_kernel: split), when there is
// more than one tile and the blocks alone cannot give every SIMD of the chip work
static bool lin_split(int N, int M) { return M > 16 && (N + 15) / 16 <= 8 * num_cus(); }
// the walk without LDS (tangent_lin_direct_kernel): K input channels, rows and filter 16-byte / 8-byte aligned
static bool lin_direct(const void* in, const float* wre, const float* wim, int K, int ldw, bool transposed) {
    static const bool off = [] { const char* e = dev_env("FC_LIN_DIRECT"); return e && atoi(e) == 0; }();       // development: the staged walk
    if (off) return false;
    if (K % 8 != 0 || K > 8 * kLinDirectKB || (reinterpret_cast<uintptr_t>(in) & 15)) return false;
    return transposed || ((ldw & 1) == 0 && (reinterpret_cast<uintptr_t>(wre) & 7) == 0 && (reinterpret_cast<uintptr_t>(wim) & 7) == 0);
}
static int lin_grid_split(int N) {
    const int nblocks = (N + 15) / 16;
    return nblocks < 4 * num_cus() ? nblocks : 4 * num_cus();
}
static int lin_gw_groups(int N) {
    const int nblocks = (N + 15) / 16;
    return nblocks < kLinGwGroups ? nblocks : kLinGwGroups;
}

int tangent_lin_backward_impl(const float* x, const float* gy, const float* re_w, const float* im_w, float* gx, const float* gx_addend,
                              float* g_re, float* g_im, void* workspace, size_t workspace_bytes, int N, int I, int O, hipStream_t s) {
    if (!x || !gy || !re_w || !im_w || !gx || !g_re || !g_im || N <= 0 || I <= 0 || O <= 0) return FC_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < fc_tangent_lin_backward_workspace_bytes(N, I, O)) return FC_ERR_WORKSPACE;
    const size_t lds = 2 * (size_t)lin_plane_floats(I, O) * sizeof(float);
    if (lds > kMaxLds) return FC_ERR_UNSUPPORTED;
    const bool split = lin_split(N, I);
    if (I > 16 && lin_direct(gy, re_w, im_w, O, I, true))
        hipLaunchKernelGGL(tangent_lin_direct_kernel<true>, dim3(lin_grid_split(N)), dim3(kLinThreads), 0, s, reinterpret_cast<const float2*>(gy),
                           re_w, im_w, reinterpret_cast<float2*>(gx), reinterpret_cast<const float2*>(gx_addend), N, O, I, I);
    else
        hipLaunchKernelGGL(tangent_lin_kernel<true>, dim3(split ? lin_grid_split(N) : lin_grid(N)), dim3(kLinThreads), lds, s,
                           reinterpret_cast<const float2*>(gy), re_w, im_w, reinterpret_cast<float2*>(gx),
                           reinterpret_cast<const float2*>(gx_addend), N, O, I, I, split ? 1 : 0);
    float2* part = reinterpret_cast<float2*>(workspace);
    const int ng = lin_gw_groups(N);
    const int pairs = ((O + 15) / 16) * ((I + 15) / 16);
    const int per_group = pairs < kLinGwPairs ? pairs : kLinGwPairs;
    hipLaunchKernelGGL(tangent_lin_gw_kernel, dim3(ng, (pairs + per_group - 1) / per_group), dim3(per_group * kWave), 0, s,
                       reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(gy), part, N, I, O);
    hipLaunchKernelGGL(tangent_lin_gw_reduce_kernel, dim3((O * I + 63) / 64), dim3(256), 0, s, part, g_re, g_im, ng,
                       O * I);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}
}  // namespace fc

extern "C" {

int fc_tangent_lin_forward(const float* x, const float* re_w, const float* im_w, float* y, int32_t N, int32_t I,
                           int32_t O, void* stream) {
    if (!x || !re_w || !im_w || !y || N <= 0 || I <= 0 || O <= 0) return FC_ERR_BAD_ARGUMENT;
    const size_t lds = 2 * (size_t)fc::lin_plane_floats(O, I) * sizeof(float);
    if (lds > fc::kMaxLds) return FC_ERR_UNSUPPORTED;
    const bool split = fc::lin_split(N, O);
    if (O > 16 && fc::lin_direct(x, re_w, im_w, I, I, false))
        hipLaunchKernelGGL(fc::tangent_lin_direct_kernel<false>, dim3(fc::lin_grid_split(N)), dim3(fc::kLinThreads), 0,
                           static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(x), re_w, im_w, reinterpret_cast<float2*>(y),
                           (const float2*)nullptr, N, I, O, I);
    else
        hipLaunchKernelGGL(fc::tangent_lin_kernel<false>, dim3(split ? fc::lin_grid_split(N) : fc::lin_grid(N)), dim3(fc::kLinThreads), lds,
                           static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(x), re_w, im_w,
                           reinterpret_cast<float2*>(y), (const float2*)nullptr, N, I, O, I, split ? 1 : 0);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

size_t fc_tangent_lin_backward_workspace_bytes(int32_t N, int32_t I, int32_t O) {
    (void)N;
    return (size_t)fc::kLinGwGroups * O * I * sizeof(float2);
}

int fc_tangent_lin_backward(const float* x, const float* gy, const float* re_w, const float* im_w, float* gx,
                            float* g_re, float* g_im, void* workspace, size_t workspace_bytes, int32_t N, int32_t I,
                            int32_t O, void* stream) {
    return fc::tangent_lin_backward_impl(x, gy, re_w, im_w, gx, nullptr, g_re, g_im, workspace, workspace_bytes, N, I, O,
                                         static_cast<hipStream_t>(stream));
}

int fc_soft_abs_forward(const float* x, float* y, size_t count, void* stream) {
    if (!x || !y) return FC_ERR_BAD_ARGUMENT;
    if (count == 0) return FC_OK;
    hipLaunchKernelGGL(fc::soft_abs_fwd_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(x), y, count);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_soft_abs_backward(const float* x, const float* gy, float* gx, size_t count, void* stream) {
    if (!x || !gy || !gx) return FC_ERR_BAD_ARGUMENT;
    if (count == 0) return FC_OK;
    hipLaunchKernelGGL(fc::soft_abs_bwd_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(x), gy, reinterpret_cast<float2*>(gx), count);
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
    const int rows = fc::nonlin_rows_per_group(N);
    const int ngroups = (N + rows - 1) / rows;
    return (size_t)ngroups * C * sizeof(float);
}

int32_t fc_tangent_nonlin_backward_groups(int32_t N) {
    if (N <= 0) return 0;
    const int rows = fc::nonlin_rows_per_group(N);
    return (N + rows - 1) / rows;
}

int fc_tangent_nonlin_backward_partial(const float* x, const float* bias, const float* gy, float* gx, void* workspace,
                                       size_t workspace_bytes, int32_t N, int32_t C, void* stream) {
    if (!x || !bias || !gy || !gx || N <= 0 || C <= 0) return FC_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < fc_tangent_nonlin_backward_workspace_bytes(N, C)) return FC_ERR_WORKSPACE;
    const int rows = fc::nonlin_rows_per_group(N);
    const int ngroups = (N + rows - 1) / rows;
    const int L = C < fc::kNonlinThreads ? fc::kNonlinThreads / C : 1;
    hipLaunchKernelGGL(fc::tangent_nonlin_bwd_kernel, dim3(ngroups), dim3(fc::kNonlinThreads), (size_t)L * C * sizeof(float),
                       static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(x), bias, reinterpret_cast<const float2*>(gy),
                       reinterpret_cast<float2*>(gx), reinterpret_cast<float*>(workspace), N, C, rows);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_tangent_nonlin_backward(const float* x, const float* bias, const float* gy, float* gx, float* gbias,
                               void* workspace, size_t workspace_bytes, int32_t N, int32_t C, void* stream) {
    if (!gbias) return FC_ERR_BAD_ARGUMENT;
    const int rc = fc_tangent_nonlin_backward_partial(x, bias, gy, gx, workspace, workspace_bytes, N, C, stream);
    if (rc != FC_OK) return rc;
    return fc::bias_partials_reduce_impl(reinterpret_cast<const float*>(workspace), fc_tangent_nonlin_backward_groups(N), C, gbias,
                                         static_cast<hipStream_t>(stream));
}

}  // extern "C"
