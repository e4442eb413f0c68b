// FieldConv backward on per-edge records, RING-MAJOR with 32-vertex tiles (reference: torch autograd through
// nn/field_conv.py:128-137; the adjoint's formulas are in fc_backward_kernels.hpp).
//
// What bounds the frequency-major data kernel (fc_backward_kernels.hpp): its phases add up -- every
// 16-vertex tile pulls the whole packed filter (553 KB at config 2) from L2, exchanges k-partials through LDS once per
// frequency, and sits at ten barriers.  Here:
//
//   * a wavefront owns TWO source vertices (rows w and w + 16 of a 32-row tile) and keeps two RINGS of their response in
//     registers (4F complex numbers per lane): the records of a source are sorted by their lower ring q, so after run q
//     ring q is final and goes to the LDS slab -- one slab per ring, k = f * KI + o;
//   * the slab of ring q is contracted against the conjugated filter of ring q.  The frequency is not summed over in
//     gxt[j,i,f] = sum_{o,r} H[j,o,r,f] conj(W[o,i,r,f]) / F, so a slab is F products of K = O: wavefront w owns the
//     (input-channel tile, frequency) pair w and keeps ITS accumulators in registers across the R slabs -- no k-partials,
//     no exchange until the tile is complete; a filter fragment fetched from L2 serves two row tiles: half the filter
//     traffic per vertex;
//   * the slab -- already split into halves -- is copied to HBM as it lies in LDS, with its row scales and the column
//     scales of the other operand; the filter-gradient kernel below streams it back with LDS-DMA and feeds the matrix
//     pipe straight from the image (transposed 16-bit reads): no conversion pass, no second copy.
//
// Precision: as everywhere in the default mode -- every operand two halves with power-of-two scales per slab row /
// filter row (fc_tile.hpp), products hi*hi + hi*lo + lo*hi accumulated in fp32.
#pragma once
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

constexpr int kBrMaxF = 7;
constexpr int kBrRows = 32;        // source vertices per tile
// Interleaving the slots of a wavefront's two streams (four source rows in flight instead of two) was measured slower: the
// second set of per-slot temporaries pushes the kernel from 46 to 115 spilled registers (contraction and conversion phases
// three times as long).
constexpr bool kBrInterleave = false;
constexpr int kBrXPad = 2;         // complex numbers of padding per row of the gxt exchange buffer (bank spread)

// Geometry shared by the packed filter image, the data kernel and the filter-gradient kernel.
struct BrGeom {
    int I, O, R, F;
    int IP, NMT;                   // ceil16(I) rows of gxt, 16-row tiles of it
    int KI;                        // ceil8(O): channel stride inside k
    int KP, KS, KST;               // k entries per slab row (ceil32(F*KI)), halves per LDS row (4*KP + 8), k blocks of 32
    int kb0[kBrMaxF];              // first k block that overlaps frequency f (k = f*KI .. f*KI + KI - 1)
    int nb[kBrMaxF];               // how many do (1..3; the kernels take shapes with at most 2)
    int boff[kBrMaxF];             // index of the first of them in the image's block list
    int BT;                        // blocks per ring in the image
    int OT;                        // 16-row tiles of (f, o) per frequency in the filter-gradient kernel: ceil(KI / 16)
};

__host__ __device__ inline BrGeom br_geom(int I, int O, int R, int F) {
    BrGeom g;
    g.I = I; g.O = O; g.R = R; g.F = F;
    g.IP = round_up(I, 16);
    g.NMT = g.IP / 16;
    g.KI = round_up(O, 8);
    g.KP = round_up(F * g.KI, 32);
    g.KS = 4 * g.KP + 8;
    g.KST = g.KP / 32;
    int off = 0;
    for (int f = 0; f < kBrMaxF; ++f) {
        g.kb0[f] = 0; g.nb[f] = 0; g.boff[f] = 0;
        if (f < F) {
            g.kb0[f] = (f * g.KI) >> 5;
            g.nb[f] = (((f + 1) * g.KI + 31) >> 5) - g.kb0[f];
            g.boff[f] = off;
            off += g.nb[f];
        }
    }
    g.BT = off;
    g.OT = (g.KI + 15) / 16;
    return g;
}

// Packed image: [IP] inverse row scales (floats), then [R][re_hi, re_lo, im_hi, im_lo][BT blocks][IP][32] halves.  Block
// boff[f] + s holds conj(W[o, i, r, f]) / F for the 32 slab entries k = 32 (kb0[f] + s) .. + 31 that belong to frequency
// f (k - f*KI = o < O), zeros elsewhere.
__host__ __device__ inline size_t br_image_floats(const BrGeom& g) {
    return (size_t)g.IP + (size_t)g.R * 4 * g.BT * g.IP * 32 / 2;
}

// Kept slab of (work item vt, ring q) in the workspace: [32 rows][4*KP halves] as the rows lie in LDS (without the row
// pad), then floats [32] s_j, [32] 1/s_j (row scales), [IP] t_i, [IP] 1/t_i (column scales of x~[j][i] / s_j).
__host__ __device__ inline size_t br_slab_bytes(const BrGeom& g) {
    return (size_t)round_up(kBrRows * 8 * g.KP + (2 * kBrRows + 2 * g.IP) * 4, 256);
}

// Work items: [0, nv_full) are whole 32-vertex tiles; the last, partly filled round of the persistent grid is cut into
// HALF tiles (16 vertices, the wavefronts' second stream idle) so that every workgroup gets a share of it.
struct BrItems {
    int nv_full, nv_total, grid;
};
inline BrItems br_items(int N, int max_grid) {
    BrItems it;
    const int nt = (N + kBrRows - 1) / kBrRows;
    it.grid = nt < max_grid ? nt : max_grid;
    if (it.grid < 1) it.grid = 1;
    it.nv_full = nt;
    it.nv_total = nt;
    const int rem = nt % it.grid;
    if (rem > 0 && 2 * rem <= it.grid) {
        it.nv_full = nt - rem;
        it.nv_total = it.nv_full + 2 * rem;
    }
    return it;
}
// vertex of row `row` (0..31) of work item vt; rows without a vertex return a value >= N
__host__ __device__ inline int br_vertex(int vt, int row, int nv_full, int N) {
    if (vt < nv_full) return vt * kBrRows + row;
    const int h = vt - nv_full;
    return row < 16 ? (nv_full + (h >> 1)) * kBrRows + (h & 1) * 16 + row : N;
}

struct BrArgs {
    int N, I, O;
    BrGeom g;
    int nv_full, nv_total;
    uint32_t wpk_bytes;
    uint32_t ring_bytes_w;      // bytes of one ring's planes in the packed image: 4 * BT * IP * 64
    uint32_t hs_bytes;          // br_slab_bytes
    int nr;                     // 1 KiB record chunks per stream in the LDS ring (2 or 4)
    int xs;                     // complex numbers per row of the gxt exchange buffer: F * IP + kBrXPad
    uint32_t region_bytes;      // LDS bytes of the slab / exchange region
    int dbg;                    // development only (FC_DEBUG_BWD): bit0 skip gather, bit1 skip MFMA, bit3 skip the slab copy
    unsigned long long* stamps; // development only (fc_debug_stamp_buffer): s_memtime stamps of workgroup 0, [16 waves][256]
};

struct BrLds {
    char* slab;         // [32][KS halves]; after a tile's last contraction: gxt exchange [32][xs] complex
    float* vinv;        // [32] inverse row scales of the current slab
    float* vsc;         // [32] row scales
    uint32_t* cmax;     // [2 slab parities][64] column maxima of |x[j][i]| / s_j (bit patterns of non-negative floats)
    int* runs;          // [16 wavefronts][2 streams][2 tile parities][8]
    float* ring;        // [16 wavefronts][2 streams][nr][256]
};
__host__ __device__ inline uint32_t br_region_bytes(const BrGeom& g) {
    const uint32_t slab = (uint32_t)kBrRows * g.KS * 2;
    const uint32_t xch = (uint32_t)kBrRows * (g.F * g.IP + kBrXPad) * 8;
    return round_up((int)(slab > xch ? slab : xch), 16);
}
__host__ __device__ inline size_t br_lds_bytes(const BrGeom& g, int nr) {
    return (size_t)br_region_bytes(g) + (2 * kBrRows + 2 * 64) * 4 + kWaves * 32 * 4 + (size_t)kWaves * 2 * nr * 1024;
}
__device__ __forceinline__ BrLds br_lds(char* smem, const BrArgs& a) {
    BrLds l;
    l.slab = smem;
    l.vinv = reinterpret_cast<float*>(smem + a.region_bytes);
    l.vsc = l.vinv + kBrRows;
    l.cmax = reinterpret_cast<uint32_t*>(l.vsc + kBrRows);
    l.runs = reinterpret_cast<int*>(l.cmax + 2 * 64);
    l.ring = reinterpret_cast<float*>(l.runs + kWaves * 32);
    return l;
}

// ------------------------------------------------------------------------------------ data gradient
template <int R, int B>
__global__ __launch_bounds__(kThreads) void fc_backward_ring_data_kernel(
    const float2* __restrict__ gx_, const float2* __restrict__ ggy, const float* __restrict__ grec,
    const int32_t* __restrict__ growptr, const int32_t* __restrict__ gruns, const float* __restrict__ gwpk,
    float2* __restrict__ ggx, char* __restrict__ hdump, const BrArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int RECF = factored_record_floats(B);
    constexpr int LOG_CR = factored_log_chunk_records(B);
    constexpr int CR = 1 << LOG_CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BrGeom& g = a.g;
    const BrLds l = br_lds(smem, a);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nr = a.nr;
    const int I = a.I, O = a.O;
    const int KS = g.KS;

    // zero the slab once: the k padding of a row is re-zeroed after every use of the region as exchange buffer
    for (int idx = tid; idx < (int)(a.region_bytes / 16); idx += kThreads)
        reinterpret_cast<f32x4*>(l.slab)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < kBrRows) { l.vinv[tid] = 0.f; l.vsc[tid] = 1.f; }
    if (tid < 2 * 64) l.cmax[tid] = 0u;
    __syncthreads();

    const int ol = lane < O ? lane : 0;          // lanes >= O gather channel 0; lanes >= KI are never stored
    Stamper stamp{(a.stamps && blockIdx.x == 0) ? a.stamps + wave * 256 : nullptr, 0};
    // The SIMDs arbitrate by priority, then age: the younger wavefronts of a workgroup (the later ones) otherwise lose every
    // arbitration to their older partners and everybody waits for them at the slab barriers
    {
        static_assert(kWaves == 16, "priorities below assume four wavefronts per SIMD");
        const int mode = (a.dbg >> 4) & 3;           // development: 0 = two levels (default), 1 = none, 2 = four levels
        if (mode == 0 && wave >= 8) __builtin_amdgcn_s_setprio(1);
        if (mode == 2) {
            if (wave >= 12) __builtin_amdgcn_s_setprio(3);
            else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
            else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        }
    }
    // my (input-channel tile, frequency) pair of the contraction
    const bool mma_active = wave < g.NMT * F;
    const int it = mma_active ? wave % g.NMT : 0;
    const int mf = mma_active ? wave / g.NMT : 0;
    const int my_kb0 = g.kb0[mf], my_nb = g.nb[mf], my_boff = g.boff[mf];
    const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
    const int planes0 = g.IP * 4;                                          // bytes: the planes follow the IP row scales
    const int wplane = g.BT * g.IP * 64;                                   // bytes per plane of a ring
    const int wblk = g.IP * 64;                                            // bytes per block
    const int wv = ((it * 16 + (lane & 15)) * 32 + 8 * (lane >> 4)) * 2;   // per-lane byte offset inside an [IP][32] block

    // gx epilogue: entries e = tid + 1024 n of the tile's [32][I] block
    const int nent = kBrRows * I;
    int e_id[2];                       // (row << 8 | channel), -1: no entry
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int e = tid + kThreads * n;
        e_id[n] = e < nent ? ((e / I) << 8) | (e - (e / I) * I) : -1;
    }
    auto load_entry = [&](const int vt, const int n) {
        float2 v = make_float2(0.f, 0.f);
        if (e_id[n] >= 0) {
            const int vtx = br_vertex(vt, e_id[n] >> 8, a.nv_full, a.N);
            if (vtx < a.N) v = gx_[(size_t)vtx * I + (e_id[n] & 255)];
        }
        return v;
    };

    auto ring_of = [&](const int j) { return l.ring + (wave * 2 + j) * nr * 256; };
    auto dma_chunk = [&](const int j, const int first, const int ch) {
        const float* src = grec + ((size_t)first + (size_t)ch * CR) * RECF + lane * 4;
        lds_dma16_untracked(src, ring_of(j) + (ch & (nr - 1)) * 256);
    };
    // slots [b, e) of stream j's source in work item vt; its ring-run offsets go to LDS
    auto slot_range = [&](const int vt, const int j, const int par, int& b, int& e) {
        b = 0;
        e = 0;
        int run[R];
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        if (vt < a.nv_total) {
            const int t = br_vertex(vt, wave + 16 * j, a.nv_full, a.N);
            if (t < a.N) {
                b = growptr[t];
                e = growptr[t + 1];
#pragma unroll
                for (int q = 0; q < R; ++q) run[q] = gruns[(size_t)t * kRunStride + q];
            }
        }
        if (lane == 0) {
            int* lro = l.runs + (wave * 2 + j) * 16 + par * 8;
#pragma unroll
            for (int q = 0; q < R; ++q) lro[q] = run[q];
        }
    };
    auto rec_ptr = [&](const float* ring, const int s) {
        if constexpr (CR * RECF == 256) return ring + ((s * RECF) & (nr * 256 - 1));
        else return ring + ((s >> LOG_CR) & (nr - 1)) * 256 + (s & (CR - 1)) * RECF;
    };
    auto first_rows = [&](const int j, const int nslots, float2& r0, float2& r1) {
        r0 = make_float2(0.f, 0.f);
        r1 = r0;
        if (nslots > 0) {
            const float* ring = ring_of(j);
            const int n0 = __float_as_int(ring[3]);
            const int n1 = __float_as_int(ring[min(1, nslots - 1) * RECF + 3]);
            r0 = gather_row(ggy, n0, 8u * O, 8u * ol);
            r1 = gather_row(ggy, n1, 8u * O, 8u * ol);
        }
    };

    int beg[2], end[2], par = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        slot_range(first_tile_of_block(), j, 0, beg[j], end[j]);
        const int nch = (end[j] - beg[j] + CR - 1) >> LOG_CR;
        for (int ch = 0; ch < min(nch, nr); ++ch) dma_chunk(j, beg[j], ch);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float2 px[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) first_rows(j, end[j] - beg[j], px[j][0], px[j][1]);

    // running gxt of my pair for the two row tiles (D layout: column = vertex = lane & 15, rows i = 16 it + 4 (lane >> 4) + jj)
    f32x4 tot_re[2], tot_im[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { tot_re[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; tot_im[rt] = tot_re[rt]; }

    u32x4 wf0[4], wf1[4];
    auto load_w = [&](const int rq, const int blk, u32x4 (&dst)[4]) {
        const int base = planes0 + rq * (int)a.ring_bytes_w + wblk * (my_boff + blk);
#pragma unroll
        for (int pnum = 0; pnum < 4; ++pnum) dst[pnum] = buffer_load16(wimg, wv, base + pnum * wplane);
    };
    auto prefetch_w = [&](const int rq) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        if (mma_active) {
            load_w(rq, 0, wf0);
            load_w(rq, my_nb > 1 ? 1 : 0, wf1);
        } else {
#pragma unroll
            for (int pnum = 0; pnum < 4; ++pnum) { wf0[pnum] = zero; wf1[pnum] = zero; }
        }
    };
    // one k block against one row tile: 12 MFMAs (lo*hi + hi*lo + hi*hi per real product)
    auto mma_block = [&](const int rt, const int kb, const u32x4 (&w)[4], f32x4& are, f32x4& aim) {
        const int fr = lane & 15, fq = lane >> 4;
        const u32x4 sign = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
        const lds_f16* s0 = (const lds_f16*)l.slab + (16 * rt + fr) * KS + 32 * fq + 128 * kb;
        {
            const u32x4 srh = *reinterpret_cast<lds_u32x4*>(s0), srl = *reinterpret_cast<lds_u32x4*>(s0 + 8);
            are = mfma32h(w[1], srh, are); aim = mfma32h(w[3], srh, aim);
            are = mfma32h(w[0], srl, are); aim = mfma32h(w[2], srl, aim);
            are = mfma32h(w[0], srh, are); aim = mfma32h(w[2], srh, aim);
        }
        {
            u32x4 sih = *reinterpret_cast<lds_u32x4*>(s0 + 16), sil = *reinterpret_cast<lds_u32x4*>(s0 + 24);
            aim = mfma32h(w[1], sih, aim);
            aim = mfma32h(w[0], sil, aim);
            aim = mfma32h(w[0], sih, aim);
            sih ^= sign;
            sil ^= sign;
            are = mfma32h(w[3], sih, are);
            are = mfma32h(w[2], sil, are);
            are = mfma32h(w[2], sih, are);
        }
    };
    // my pair's share of slab rq (fragments of my first two blocks are in wf0 / wf1); `half`: rows 16..31 are empty
    auto contract = [&](const int rq, const bool half) {
        (void)rq;
        if (mma_active && !(a.dbg & 2)) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (rt == 0 || !half) {
                    // (a slab's product starts from zero: its rows carry their own scales)
                    f32x4 tre = {0.f, 0.f, 0.f, 0.f}, tim = tre;
                    mma_block(rt, my_kb0, wf0, tre, tim);
                    if (my_nb > 1) mma_block(rt, my_kb0 + 1, wf1, tre, tim);
                    const float inv = l.vinv[16 * rt + (lane & 15)];
                    tot_re[rt] += tre * inv;
                    tot_im[rt] += tim * inv;
                }
            }
        }
    };
    // ring values c[f] of my stream-j source -> row wave + 16 j of the slab
    // xmag: |x[row][lane]| of the stream's source (for the column scales of the filter kernel's second operand); sp: slab parity
    auto flush_row = [&](const f32x2 (&c)[F], const int j, const float xmag, const int sp) {
        const int row_i = wave + 16 * j;
        float mx = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) mx = fmaxf(mx, fmaxf(fabsf(c[f].x), fabsf(c[f].y)));
        mx = wave_max_nonneg(mx);
        float scale, inv;
        split_scale(mx, scale, inv);
        // An all-zero row (a source without edges in this ring) contributes nothing; its inverse scale is kept as 0 so that
        // the row drops out of the filter kernel's second operand x~[j][i] / s_j and of that operand's column scales
        // (with 1 it would dominate them whenever the other rows' scales are large, i.e. the cotangent is small)
        if (mx == 0.f) inv = 0.f;
        if (lane == 0) { l.vinv[row_i] = inv; l.vsc[row_i] = scale; }
        // a bound of the modulus of x~[j][i] / s_j is |x[j][i]| / s_j: its column maximum over the tile's rows (non-negative
        // floats order like their bit patterns; a maximum does not depend on the order of the updates)
        if (lane < g.IP) atomicMax(l.cmax + sp * 64 + lane, __float_as_uint(xmag * inv));
        if (lane < g.KI) {
            lds_u32* const row = (lds_u32*)l.slab + row_i * (KS / 2);
            int o0 = split_pair_offset(lane, 2);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                f16x2 hi, lo;
                split_halves2(c[f], scale, hi, lo);
                split_pair_store(row, o0, hi, lo, lane, 2);
                o0 += 2 * g.KI;
                asm volatile("" : "+v"(o0));
            }
        }
    };
    // after the slab barrier: copy the slab (as it lies in LDS, without the row pads) and its scales to the workspace
    auto keep_slab = [&](const int vt, const int rq, const int sp) {
        char* const dst = hdump + ((size_t)vt * R + rq) * a.hs_bytes;
        if (!(a.dbg & 8)) {
            // wavefront w copies rows w and w + 16: 8*KP bytes each, 16 bytes per lane and instruction
            const int row_pieces = g.KP / 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wave + 16 * j;
                const __attribute__((address_space(3))) char* src = (const __attribute__((address_space(3))) char*)l.slab + row * KS * 2;
                char* const drow = dst + (uint32_t)row * (uint32_t)(8 * g.KP);
                for (int pc = lane; pc < row_pieces; pc += kWave) {
                    const u32x4 v = *reinterpret_cast<lds_u32x4*>(src + pc * 16);
                    *reinterpret_cast<u32x4*>(drow + (uint32_t)pc * 16u) = v;
                }
            }
        }
        if (wave == kWaves - 1) {
            // row scales, and the power-of-two column scales of the filter kernel's second operand x~[j][i] / s_j
            float* const tail = reinterpret_cast<float*>(dst + (size_t)kBrRows * 8 * g.KP);
            if (lane < kBrRows) { tail[lane] = l.vsc[lane]; tail[kBrRows + lane] = l.vinv[lane]; }
            if (lane < g.IP) {
                const float cm = __uint_as_float(l.cmax[sp * 64 + lane]);
                float t, inv_t;
                split_scale(cm * 1.0000002f, t, inv_t);
                tail[2 * kBrRows + lane] = t;
                tail[2 * kBrRows + g.IP + lane] = inv_t;
                l.cmax[(sp ^ 1) * 64 + lane] = 0u;          // for the next slab (its updates come behind this slab's second barrier)
            }
        }
    };

    stamp.realtime(29);
    stamp(28);
    for (int vt = first_tile_of_block(); vt < a.nv_total; vt += gridDim.x) {
        const bool half = vt >= a.nv_full;
        stamp(10);
        int nbeg[2], nend[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) slot_range(vt + gridDim.x, j, par ^ 1, nbeg[j], nend[j]);

        // |x| of my two sources' rows, lane = input channel (used at every flush)
        float xmag[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int vtx = br_vertex(vt, wave + 16 * j, a.nv_full, a.N);
            float2 xv = make_float2(0.f, 0.f);
            if (vtx < a.N && lane < I) xv = gx_[(size_t)vtx * I + lane];
            xmag[j] = sqrtf(xv.x * xv.x + xv.y * xv.y);
        }
        f32x2 clo[2][F], chi[2][F];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int f = 0; f < F; ++f) { clo[j][f] = f32x2{0.f, 0.f}; chi[j][f] = clo[j][f]; }
        float2 gq[2][2] = {{px[0][0], px[0][1]}, {px[1][0], px[1][1]}};

        // one slot of stream J: lo += w0 z, hi += w1 z with z_f = gy conj(ph_f)
        auto slot = [&](auto jc, const int s_, float2& gcur) {
            constexpr int J = decltype(jc)::value;
            const float* const ring = ring_of(J);
            const int nslots = end[J] - beg[J];
            const int nch = (nslots + CR - 1) >> LOG_CR;
            if ((s_ & (CR - 1)) == 0 && s_ > 0) {
                const int ch = s_ >> LOG_CR;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (ch - 1 + nr < nch) dma_chunk(J, beg[J], ch - 1 + nr);
            }
            if (nr == 2 && ((s_ + 2) & (CR - 1)) < 1)       // the look-ahead below enters a chunk issued at the last chunk entry
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float* rp = rec_ptr(ring, s_);
            const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
            const int d2 = __float_as_int(rec_ptr(ring, min(s_ + 2, nslots - 1))[3]);
            const f32x2 gv = f32x2{gcur.x, gcur.y};
            gcur = gather_row(ggy, d2, 8u * O, 8u * ol);
            const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
            f32x2 ph[F], z[F];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                z[f] = cmul_conj_pk_step1(gv, ph[f]);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) z[f] = cmul_conj_pk_step2(gv, ph[f], z[f]);
#pragma unroll
            for (int f = 0; f < F; ++f) clo[J][f] = __builtin_elementwise_fma(w0v, z[f], clo[J][f]);
#pragma unroll
            for (int f = 0; f < F; ++f) chi[J][f] = __builtin_elementwise_fma(w1v, z[f], chi[J][f]);
        };
        // the rest of one stream's run
        auto gather_rest = [&](auto jc, int s, const int run_end) {
            constexpr int J = decltype(jc)::value;
            for (; s + 1 < run_end; s += 2) {
                slot(jc, s, gq[J][0]);
                slot(jc, s + 1, gq[J][1]);
            }
            if (s < run_end) {      // odd tail: rotate the two prefetch registers
                slot(jc, s, gq[J][0]);
                const float2 t = gq[J][0]; gq[J][0] = gq[J][1]; gq[J][1] = t;
            }
        };
        // ring run q of BOTH streams, their slots interleaved: four source rows in flight per wavefront instead of two (the
        // gather waits for L2, not for the vector pipes)
        auto gather_runs = [&](int s0, const int e0, int s1, const int e1) {
            constexpr std::integral_constant<int, 0> j0{};
            constexpr std::integral_constant<int, 1> j1{};
            for (; s0 + 1 < e0 && s1 + 1 < e1; s0 += 2, s1 += 2) {
                slot(j0, s0, gq[0][0]);
                __builtin_amdgcn_sched_barrier(0);       // (keeps the slots' temporaries from overlapping: 128 registers)
                slot(j1, s1, gq[1][0]);
                __builtin_amdgcn_sched_barrier(0);
                slot(j0, s0 + 1, gq[0][1]);
                __builtin_amdgcn_sched_barrier(0);
                slot(j1, s1 + 1, gq[1][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            gather_rest(j0, s0, e0);
            gather_rest(j1, s1, e1);
        };

        for (int q = 0; q < R - 1; ++q) {
            if (!(a.dbg & 1)) {
                int rs[2], re[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int* lro = l.runs + (wave * 2 + j) * 16 + par * 8;
                    rs[j] = __builtin_amdgcn_readfirstlane(lro[q]);
                    re[j] = (q + 1 < R - 1) ? __builtin_amdgcn_readfirstlane(lro[q + 1]) : end[j] - beg[j];
                }
                if constexpr (kBrInterleave) {
                    gather_runs(rs[0], re[0], rs[1], re[1]);
                } else {
                    gather_rest(std::integral_constant<int, 0>{}, rs[0], re[0]);
                    gather_rest(std::integral_constant<int, 1>{}, rs[1], re[1]);
                }
            }
            stamp(0);
            if (q == R - 2) {
                // my sources are done: start streaming the first record chunks of my next tile's sources
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nnch = (nend[j] - nbeg[j] + CR - 1) >> LOG_CR;
                    for (int ch = 0; ch < min(nnch, nr); ++ch) dma_chunk(j, nbeg[j], ch);
                }
            }
            flush_row(clo[0], 0, xmag[0], q & 1);
            if (!half) flush_row(clo[1], 1, xmag[1], q & 1);
            prefetch_w(q);                                   // ring q's filter fragments fly during the barrier and the slab copy
            stamp(1);
            __syncthreads();
            stamp(2);
            keep_slab(vt, q, q & 1);
            stamp(7);
            contract(q, half);
            stamp(3);
            __syncthreads();
            stamp(4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < F; ++f) { clo[j][f] = chi[j][f]; chi[j][f] = f32x2{0.f, 0.f}; }
        }
        flush_row(clo[0], 0, xmag[0], (R - 1) & 1);
        if (!half) flush_row(clo[1], 1, xmag[1], (R - 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the next tile's first record chunks have landed)
        prefetch_w(R - 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            beg[j] = nbeg[j];
            end[j] = nend[j];
            first_rows(j, end[j] - beg[j], px[j][0], px[j][1]);
        }
        stamp(2);
        keep_slab(vt, R - 1, (R - 1) & 1);
        stamp(7);
        contract(R - 1, half);
        stamp(3);
        __syncthreads();                                     // every read of the slab is done: the region becomes the exchange buffer
        stamp(4);
        float2 exl[2];                                       // my entries of x (issued here, used behind the next barrier)
#pragma unroll
        for (int n = 0; n < 2; ++n) exl[n] = load_entry(vt, n);
        if (mma_active) {
            float2* const xb = reinterpret_cast<float2*>(l.slab);
            const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float2* p = xb + (size_t)(16 * rt + fr) * a.xs + mf * g.IP + it * 16 + 4 * fq;
                *reinterpret_cast<f32x4*>(p) = f32x4{tot_re[rt][0], tot_im[rt][0], tot_re[rt][1], tot_im[rt][1]};
                *reinterpret_cast<f32x4*>(p + 2) = f32x4{tot_re[rt][2], tot_im[rt][2], tot_re[rt][3], tot_im[rt][3]};
                tot_re[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
                tot_im[rt] = tot_re[rt];
            }
        }
        __syncthreads();
        stamp(5);
        if (tid < 2 * 64) l.cmax[tid] = 0u;                  // (both parities, for the next tile: R may be odd; ordered by the barrier below)
        // gx[j,i] = sum_f gxt_f conj(u^m) + [x != 0] (i x / |x|^2) sum_f m Im(conj(gxt_f) x u^m),  m = f - B,  u = exp(-i angle(x))
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (e_id[n] >= 0) {
                const int e_row = e_id[n] >> 8, e_i = e_id[n] & 255;
                const int vtx = br_vertex(vt, e_row, a.nv_full, a.N);
                const float2 xv = exl[n];
                const float2 u1 = unit_conj(xv);
                const float2 u2 = cmul(u1, u1);
                const float2 u3 = cmul(u2, u1);
                const float inv2 = is_origin(xv) ? 0.f : 1.f / (xv.x * xv.x + xv.y * xv.y);
                const float wk = gwpk[e_i];                       // the filter row's scale (a power of two)
                const float2* xr = reinterpret_cast<const float2*>(l.slab) + (size_t)e_row * a.xs + e_i;
                float2 acc = make_float2(0.f, 0.f);
                float eq = 0.f;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int m = f - B;
                    const int am = m < 0 ? -m : m;
                    float2 z = xr[f * g.IP];
                    z.x *= wk;
                    z.y *= wk;
                    float2 c = am == 0 ? make_float2(1.f, 0.f) : (am == 1 ? u1 : (am == 2 ? u2 : u3));
                    if (m < 0) c.y = -c.y;
                    const float2 xtv = cmul(xv, c);
                    const float2 out = cmul_conj(z, c);
                    acc.x += out.x;
                    acc.y += out.y;
                    eq += (float)m * (z.x * xtv.y - z.y * xtv.x);
                }
                const float qv = eq * inv2;
                acc.x += -xv.y * qv;
                acc.y += xv.x * qv;
                if (vtx < a.N) ggx[(size_t)vtx * I + e_i] = acc;
            }
        }
        stamp(6);
        __syncthreads();                                     // the exchange buffer is consumed
        stamp(8);
        {
            // the region is a slab again: zero what the gathers never write -- the k padding of every row, and for a half
            // tile to come the rows 16..31 (their inverse scales are 0)
            const bool next_half = vt + (int)gridDim.x >= a.nv_full;
            const int kpad0 = (F * g.KI) >> 3, kpadn = (g.KP >> 3) - kpad0;          // 8-k fragments of 64 bytes
            for (int idx = tid; idx < kBrRows * kpadn * 4; idx += kThreads) {
                const int row = idx / (kpadn * 4), pc = idx - row * (kpadn * 4);
                *reinterpret_cast<f32x4*>(l.slab + (size_t)row * KS * 2 + kpad0 * 64 + pc * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (next_half) {
                const int row16 = KS * 2 / 16;                    // 16-byte pieces per row (KS * 2 bytes, a multiple of 16)
                for (int idx = tid; idx < 16 * row16; idx += kThreads)
                    *reinterpret_cast<f32x4*>(l.slab + (size_t)16 * KS * 2 + (size_t)idx * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                if (tid >= 16 && tid < 32) { l.vinv[tid] = 0.f; l.vsc[tid] = 1.f; }
            }
        }
        stamp(9);
        par ^= 1;
        // (no barrier here: the next tile's flushes write k < F*KI of rows that are not being zeroed, and the first read of
        //  the zeroed bytes is behind the next slab barrier)
    }
    stamp(30);
    stamp.realtime(31);
}

}  // namespace fc
