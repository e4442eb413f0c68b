// FieldConv forward on per-edge records, RING-MAJOR, two workgroups per CU (reference nn/field_conv.py:128-137; records:
// fc_forward_kernels.hpp).
//
// What bounds the frequency-major kernel (fc_forward_kernels.hpp) on MI355X (rocprofv3 SQ counters): every
// vector instruction costs its SIMD four cycles, the layer needs ~43 M of them, and the SIMDs are busy half of the launch:
// one 16-wavefront workgroup owns a CU, and while it streams the 553 KB filter image through its MFMA phase (or waits at
// one of ten barriers per tile) nothing else is resident to use the vector pipes.
//
// Here a workgroup is EIGHT wavefronts with two target vertices each (still a 16-vertex MFMA tile) and at most 80 KB of
// LDS, so that TWO independent workgroups share a CU: while one contracts, waits for filter fragments or sits at a barrier,
// the other gathers.  Two vertices per wavefront fit because the response goes to the contraction one RING at a time:
// FCPrecomp's stencil touches two adjacent rings per edge (q, q+1) and a target's records are sorted by q, so after run q
// ring q is final.  A wavefront keeps two rings x two vertices in registers (4F complex numbers per lane, 40 VGPRs at
// config 2, against R*F = 60 for ONE vertex frequency-major), drops ring q of both into the LDS slab when run q ends, and
// the workgroup contracts that slab with the filter of ring q:
//
//     out^T[o, vertex] += W[o, k = f*KI + i; q] * slab_q[vertex, k],        K = F*KI per slab, R slabs per tile.
//
// Precision: every slab row (vertex, ring) carries its own power-of-two scale; a slab's product is accumulated from zero on
// the matrix pipe and added to the fp32 running output with that scale divided out (fc_tile.hpp, split mode).  These
// kernels exist for the default two-halves mode.
#pragma once
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

constexpr int kDuoWaves = 8;                     // wavefronts per workgroup
constexpr int kDuoThreads = kDuoWaves * kWave;
constexpr size_t kDuoMaxLds = 80 * 1024;         // two workgroups per CU

struct RingArgs {
    int N, I, O;
    MmaGeom g;              // M = O, k = f*KI + i (ring_geom)
    int ntiles;
    int parts_log2;         // edge split for small meshes, as in FwdArgs
    uint32_t part_stride;
    int nr;                 // record chunks per stream in the LDS ring (2 or 4)
    int alias_part;         // 1: the k-partials of the tile epilogue live in the slab's first bytes (plans that would not fit half a
                            // CU's LDS otherwise: 64 channels at band limit 3); costs two more barriers and a re-zeroing per tile
    int nv_full;            // work items [0, nv_full) are whole 16-vertex tiles; items beyond are HALF tiles (8 vertices, one per
                            // wavefront): the last, partly filled round of a persistent grid is cut in two so that every
                            // workgroup gets a share of it (1250 tiles on 512 workgroups: 2.6 tile times instead of 3)
    int nv_total;           // work items in all
    int spread_cus;         // > 0: ONE item per workgroup on a grid of 2 * spread_cus workgroups (a mesh between one and two rounds of
                            // whole tiles, e.g. a FAUST-sized one: 313 tiles on 256 CUs) -- the nv_full whole tiles are dealt evenly to
                            // the first spread_cus workgroups, every other workgroup takes a half tile, so that a CU holds a whole and a
                            // half tile or two half tiles instead of two whole tiles on some CUs and one on the others (ring_item)
    uint32_t wpk_bytes;
    uint32_t slab_bytes_w;  // bytes of one ring's planes in the packed image: 2 * halves * MP * KP * 2
    int dbg;                // development only (FC_DEBUG): bit0 skip gather, bit1 skip MFMA
    FwdEpi epi;             // residual / modReLU applied to the output tile (when parts_log2 == 0; otherwise after the parts' sum)
    unsigned long long* stamps;   // development only (fc_debug_stamp_buffer): s_memtime stamps of workgroup 0, [8 waves][256]
};

// Contraction geometry of a ring slab for an eight-wavefront workgroup: wavefront w owns output tile w % NMT and the k
// blocks kp, kp + NKP, ... with kp = w / NMT.
__host__ __device__ inline MmaGeom ring_geom(int M, int F, int channels, int halves) {
    MmaGeom g = make_mma_geom(M, F, channels, halves);
    g.NKP = kDuoWaves / g.NMT;
    if (g.NKP > g.KST) g.NKP = g.KST;
    if (g.NKP < 1) g.NKP = 1;
    return g;
}

// Work item of workgroup `b`'s first (with spread_cus: only) pass.
__device__ __forceinline__ int ring_item(const RingArgs& a) {
    if (a.spread_cus == 0) return first_tile_of_block();
    const int b = blockIdx.x, C = a.spread_cus, W = a.nv_full;
    if (b >= C) return a.nv_full + (C - W) + (b - C);          // the second workgroup of a CU: always a half tile
    const int w0 = (b * W) / C, w1 = ((b + 1) * W) / C;        // whole tiles before / up to this workgroup, evenly spaced
    return w1 > w0 ? w0 : a.nv_full + (b - w0);
}

struct RingLds {
    float* slab;        // [16][KS halves]
    float* part;        // [NKP][MP][kPartStride]
    float* vinv;        // [16] inverse slab scales of the sixteen rows
    int* runs;          // [8 wavefronts][2 streams][2 tile parities][8] ring-run offsets
    float* ring;        // [8 wavefronts][2 streams][nr][256]
};
// chunk_floats: 256 (1 KiB record chunks) or 128 (half-size chunks); alias: the partials share the slab's memory
__host__ __device__ inline size_t ring_lds_floats(const MmaGeom& g, int nr, int chunk_floats = 256, bool alias = false) {
    const size_t slab = slab_floats(g), part = partial_floats(g.NKP, g.MP);
    return (alias ? (slab > part ? slab : part) : slab + part) + kTile + kDuoWaves * 32 + (size_t)kDuoWaves * 2 * nr * chunk_floats;
}
__device__ __forceinline__ RingLds ring_lds(char* smem, const MmaGeom& g, bool alias) {
    RingLds l;
    l.slab = reinterpret_cast<float*>(smem);
    l.part = alias ? l.slab : l.slab + slab_floats(g);
    const size_t slab = slab_floats(g), part = partial_floats(g.NKP, g.MP);
    l.vinv = l.slab + (alias ? (slab > part ? slab : part) : slab + part);
    l.runs = reinterpret_cast<int*>(l.vinv + kTile);
    l.ring = reinterpret_cast<float*>(l.runs + kDuoWaves * 32);
    return l;
}

// LOGH = 1: half-size record chunks (512 bytes, 32 DMA lanes) -- a stream's ring is then 1 KiB at nr = 2
template <int R, int B, bool GEO, int LOGH = 0>
__global__ __launch_bounds__(kDuoThreads, 4) void fc_forward_ring_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ grec, const int32_t* __restrict__ growptr,
    const int32_t* __restrict__ gruns, const float* __restrict__ gwpk, float2* __restrict__ gy_, const RingArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int RECF = GEO ? kGeoRecordFloats : factored_record_floats(B);
    constexpr int LOG_CR = (GEO ? kGeoLogChunkRecords : factored_log_chunk_records(B)) - LOGH;
    constexpr int CR = 1 << LOG_CR;
    constexpr int CHUNK = 256 >> LOGH;          // floats per ring chunk
    static_assert(LOG_CR >= 2, "a record chunk holds at least four records");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& g = a.g;
    const RingLds l = ring_lds(smem, g, a.alias_part != 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nr = a.nr;
    const int I = a.I;
    const int sfl = slab_floats(g);

    // zero the slab once (the k padding of a row is never written)
    for (int idx = tid; idx < sfl; idx += kDuoThreads) l.slab[idx] = 0.f;
    __syncthreads();

    // The second wavefront of a workgroup on each SIMD (waves 4..7) is the younger one and loses the issue arbitration to
    // its older partner all the way: it gathers ~30 % slower and the older half waits for it at every barrier.  A static
    // priority for the younger half evens that out (MI355X_MICROARCH.md, "Two waves per SIMD", item 4).
    if (wave >= kDuoWaves / 2 && !(a.dbg & 4)) __builtin_amdgcn_s_setprio(1);
    Stamper stamp{(a.stamps && blockIdx.x == 0) ? a.stamps + wave * 256 : nullptr, 0};
    const int cl = lane < I ? lane : 0;      // lanes >= I gather channel 0; lanes >= KI are never stored
    const int mt = wave % g.NMT, kp = wave / g.NMT;
    const bool mma_active = kp < g.NKP;
    const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
    const int planes0 = g.MP * 4;            // bytes: the planes follow the MP row scales
    const int pl = a.parts_log2;
    const int nvt = a.ntiles << pl;

    // stream j in {0, 1}: my target in row wave + 8 j of the tile
    auto ring_of = [&](const int j) { return l.ring + (wave * 2 + j) * nr * CHUNK; };
    auto dma_chunk = [&](const int j, const int first, const int ch) {
        const float* src = grec + ((size_t)first + (size_t)ch * CR) * RECF + lane * 4;
        if constexpr (LOGH == 0) lds_dma16_untracked(src, ring_of(j) + (ch & (nr - 1)) * CHUNK);
        else lds_dma16_untracked_lanes(src, ring_of(j) + (ch & (nr - 1)) * CHUNK, 0xffffffffull);        // 32 lanes x 16 bytes
    };
    // slots [b, e) of stream j's target in virtual tile vt; its ring-run offsets (relative to b, clipped to the part) go to LDS
    auto slot_range = [&](const int vt, const int j, const int par, int& b, int& e) {
        b = 0;
        e = 0;
        int run[R];
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        int t = (vt >> pl) * kTile + wave + kDuoWaves * j;
        bool live = vt < nvt;
        if (vt >= a.nv_full) {                 // half tile: stream 0 only
            const int h = vt - a.nv_full;
            t = (a.nv_full + (h >> 1)) * kTile + (h & 1) * kDuoWaves + wave;
            live = vt < a.nv_total && j == 0;
        }
        if (live && t < a.N) {
            const int rb = growptr[t];
            const int n = growptr[t + 1] - rb;
            const int part = vt & ((1 << pl) - 1);
            const int s0 = (n * part) >> pl, s1 = (n * (part + 1)) >> pl;
            b = rb + s0;
            e = rb + s1;
#pragma unroll
            for (int q = 0; q < R; ++q) run[q] = min(max(gruns[(size_t)t * kRunStride + q], s0), s1) - s0;
        }
        if (lane == 0) {
            int* lro = l.runs + (wave * 2 + j) * 16 + par * 8;
#pragma unroll
            for (int q = 0; q < R; ++q) lro[q] = run[q];
        }
    };
    auto rec_ptr = [&](const float* ring, const int s) {
        if constexpr (CR * RECF == CHUNK) return ring + ((s * RECF) & (nr * CHUNK - 1));
        else return ring + ((s >> LOG_CR) & (nr - 1)) * CHUNK + (s & (CR - 1)) * RECF;
    };
    // source rows of the first two slots of a stream whose first record chunk is on its way to the ring
    auto first_rows = [&](const int j, const int nslots, float2& r0, float2& r1) {
        r0 = make_float2(0.f, 0.f);
        r1 = r0;
        if (nslots > 0) {
            const float* ring = ring_of(j);
            const int n0 = __float_as_int(ring[3]);
            const int n1 = __float_as_int(ring[min(1, nslots - 1) * RECF + 3]);
            r0 = gather_row(gx_, n0, 8u * I, 8u * cl);
            r1 = gather_row(gx_, n1, 8u * I, 8u * cl);
        }
    };

    int beg[2], end[2], par = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        slot_range(ring_item(a), j, 0, beg[j], end[j]);
        const int nch = (end[j] - beg[j] + CR - 1) >> LOG_CR;
        for (int ch = 0; ch < min(nch, nr); ++ch) dma_chunk(j, beg[j], ch);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
    float2 px[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) first_rows(j, end[j] - beg[j], px[j][0], px[j][1]);

    f32x4 tot_re = {0.f, 0.f, 0.f, 0.f}, tot_im = tot_re;

    // ---- filter fragments: two k blocks in registers (wf0, wf1), fetched from L2 before the conversions of the flush that
    // precedes their use; further blocks of the slab take their place as soon as the MFMAs that read them are issued
    const int wv = ((mt * 16 + (lane & 15)) * 32 + 8 * (lane >> 4)) * 2;       // per-lane byte offset inside a [MP][32] k block
    const int wplane = g.MP * g.KP * 2;                                        // bytes per plane
    const int wkb = g.MP * 64;                                                 // bytes per k block
    u32x4 wf0[4], wf1[4];
    auto load_w = [&](const int rq, const int kb, u32x4 (&dst)[4]) {
        const int kc = min(kb, g.KST - 1);                                     // (blocks past the end: a harmless re-read)
        const int base = planes0 + rq * (int)a.slab_bytes_w + wkb * kc;
#pragma unroll
        for (int pnum = 0; pnum < 4; ++pnum) dst[pnum] = buffer_load16(wimg, wv, base + pnum * wplane);
    };
    // (every path defines all eight registers: a conditional definition would make them loop-carried values)
    auto prefetch_w = [&](const int rq) {
        if (mma_active) {
            load_w(rq, kp, wf0);
            load_w(rq, kp + g.NKP, wf1);
        } else {
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int pnum = 0; pnum < 4; ++pnum) { wf0[pnum] = zero; wf1[pnum] = zero; }
        }
    };
    // one k block: 12 MFMAs (lo*hi + hi*lo + hi*hi per real product)
    auto mma_block = [&](const lds_f16* sp, const int kb, const u32x4 (&w)[4], f32x4& are, f32x4& aim) {
        const int fr = lane & 15, fq = lane >> 4;
        const u32x4 sign = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
        const lds_f16* s0 = sp + fr * g.KS + 32 * fq + 128 * kb;
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
    // my MFMA share on the slab (ring rq; fragments of my first two k blocks are in wf0 / wf1)
    auto contract = [&](const int rq) {
        if (mma_active && !(a.dbg & 2)) {
            f32x4 tre = {0.f, 0.f, 0.f, 0.f}, tim = tre;
            const lds_f16* sp = (const lds_f16*)l.slab;
            for (int kb = kp; kb < g.KST; kb += 2 * g.NKP) {
                mma_block(sp, kb, wf0, tre, tim);
                if (kb + 2 * g.NKP < g.KST) load_w(rq, kb + 2 * g.NKP, wf0);
                if (kb + g.NKP < g.KST) {
                    mma_block(sp, kb + g.NKP, wf1, tre, tim);
                    if (kb + 3 * g.NKP < g.KST) load_w(rq, kb + 3 * g.NKP, wf1);
                }
            }
            const float inv = l.vinv[lane & 15];
            tot_re += tre * inv;
            tot_im += tim * inv;
        }
    };
    // ring values c[f] of my stream-j target -> row wave + 8 j of the slab
    auto flush_row = [&](const f32x2 (&c)[F], const int j) {
        const int row_i = wave + kDuoWaves * j;
        float mx = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) mx = fmaxf(mx, fmaxf(fabsf(c[f].x), fabsf(c[f].y)));
        mx = wave_max_nonneg(mx);
        float scale, inv;
        split_scale(mx, scale, inv);
        if (lane == 0) l.vinv[row_i] = inv;
        if (lane < g.KI) {
            lds_u32* const row = (lds_u32*)l.slab + row_i * (g.KS / 2);
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

    stamp.realtime(29);
    stamp(28);
    for (int vt = ring_item(a); vt < a.nv_total; vt += gridDim.x) {
        int nbeg[2], nend[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) slot_range(vt + gridDim.x, j, par ^ 1, nbeg[j], nend[j]);

        f32x2 clo[2][F], chi[2][F];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int f = 0; f < F; ++f) { clo[j][f] = f32x2{0.f, 0.f}; chi[j][f] = clo[j][f]; }
        float2 xq[2][2] = {{px[0][0], px[0][1]}, {px[1][0], px[1][1]}};

        // one run of one stream: lo += w0 z, hi += w1 z with z_f = ph_f * x~_f over the slots [s, run_end).  xe / xo hold the source
        // rows of the next even / odd slot (a slot requests the row of slot + 2 into its own registers: no rotation at odd run
        // ends).  The record ring's events -- a chunk is entered, the look-ahead is about to leave the chunk -- are handled
        // between SEGMENTS of the run, so that the slots themselves carry no checks.
        auto gather_run = [&](const int j, int s, const int run_end, f32x2 (&lo)[F], f32x2 (&hi)[F], float2& xe, float2& xo) {
            const float* const ring = ring_of(j);
            const int nslots = end[j] - beg[j];
            const int nch = (nslots + CR - 1) >> LOG_CR;
            auto slot = [&](const int s_, float2& xcur) {
                const float* rp = rec_ptr(ring, s_);
                const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                const int n2 = __float_as_int(rec_ptr(ring, min(s_ + 2, nslots - 1))[3]);
                const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
                f32x2 z[F];
                if constexpr (GEO) {
                    const f32x4 cg = *reinterpret_cast<const f32x4*>(rp + 4);
                    rotate_geometric<B>(f32x2{xcur.x, xcur.y}, f32x2{cg.x, cg.y}, f32x2{cg.z, cg.w}, z);
                    xcur = gather_row(gx_, n2, 8u * I, 8u * cl);
                } else {
                    f32x2 xt[F], ph[F];
                    rotate_all<B>(f32x2{xcur.x, xcur.y}, xt);
                    xcur = gather_row(gx_, n2, 8u * I, 8u * cl);
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                        z[f] = cmul_pk_step1(ph[f], xt[f]);
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) z[f] = cmul_pk_step2(ph[f], xt[f], z[f]);
                }
#pragma unroll
                for (int f = 0; f < F; ++f) lo[f] = __builtin_elementwise_fma(w0v, z[f], lo[f]);
#pragma unroll
                for (int f = 0; f < F; ++f) hi[f] = __builtin_elementwise_fma(w1v, z[f], hi[f]);
            };
            while (s < run_end) {
                const int m = s & (CR - 1);
                if (m == 0 && s > 0) {
                    // entering a chunk: the chunk before it is consumed, its ring slot is refilled nr - 1 chunks ahead
                    const int ch = s >> LOG_CR;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (ch - 1 + nr < nch) dma_chunk(j, beg[j], ch - 1 + nr);
                }
                int stop = min(run_end, s - m + CR);                  // the end of this chunk
                if (nr == 2) {
                    // the look-ahead of position CR - 2 enters a chunk requested at the last chunk entry
                    if (m == CR - 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (m < CR - 2) stop = min(stop, s - m + CR - 2);
                }
                if ((s & 1) && s < stop) {
                    slot(s, xo);
                    ++s;
                }
                for (; s + 1 < stop; s += 2) {
                    slot(s, xe);
                    slot(s + 1, xo);
                }
                if (s < stop) {
                    slot(s, xe);
                    ++s;
                }
            }
        };

        stamp(10);
        for (int q = 0; q < R - 1; ++q) {
            if (!(a.dbg & 1)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int* lro = l.runs + (wave * 2 + j) * 16 + par * 8;
                    const int s = __builtin_amdgcn_readfirstlane(lro[q]);
                    const int run_end = (q + 1 < R - 1) ? __builtin_amdgcn_readfirstlane(lro[q + 1]) : end[j] - beg[j];
                    gather_run(j, s, run_end, clo[j], chi[j], xq[j][0], xq[j][1]);
                }
            }
            stamp(0);
            if (q == R - 2) {
                // my targets are done: start streaming the first record chunks of my next tile's targets
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nnch = (nend[j] - nbeg[j] + CR - 1) >> LOG_CR;
                    for (int ch = 0; ch < min(nnch, nr); ++ch) dma_chunk(j, nbeg[j], ch);
                }
            }
            prefetch_w(q);                               // ring q's first filter fragments fly during the conversions
            flush_row(clo[0], 0);                        // ring q is final for both of my targets
            if (vt < a.nv_full) flush_row(clo[1], 1);     // (half tile: rows 8..15 keep whatever finite values they hold)
            stamp(1);
            __syncthreads();
            stamp(2);
            contract(q);
            stamp(3);
            __syncthreads();                             // the slab is free again
            stamp(4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < F; ++f) { clo[j][f] = chi[j][f]; chi[j][f] = f32x2{0.f, 0.f}; }
        }
        prefetch_w(R - 1);
        flush_row(clo[0], 0);                            // the outermost ring
        if (vt < a.nv_full) flush_row(clo[1], 1);
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the next tile's first record chunks have landed)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            beg[j] = nbeg[j];
            end[j] = nend[j];
            first_rows(j, end[j] - beg[j], px[j][0], px[j][1]);       // the next tile's first source rows fly during the epilogue
        }
        contract(R - 1);
        if (a.alias_part) __syncthreads();               // every wavefront's last reads of the slab are done: its memory takes the partials
        if (mma_active) store_partial(l.part, g, mt, kp, lane, tot_re, tot_im);
        tot_re = f32x4{0.f, 0.f, 0.f, 0.f};
        tot_im = tot_re;
        stamp(5);
        __syncthreads();                                 // partials complete; every MFMA read of the slab is done
        {
            const bool half = vt >= a.nv_full;
            const int hh = vt - a.nv_full;
            const int row0 = half ? (a.nv_full + (hh >> 1)) * kTile + (hh & 1) * kDuoWaves : (vt >> pl) * kTile;
            const int nrows = half ? kDuoWaves : kTile;
            float2* const yout = gy_ + (size_t)(half ? 0 : (vt & ((1 << pl) - 1))) * a.part_stride;
            for (int idx = tid; idx < nrows * a.O; idx += kDuoThreads) {
                const int v = idx / a.O, o = idx - v * a.O;
                const int n = row0 + v;
                float2 sm = sum_partials(l.part, g, v, o);
                const float k = gwpk[o];             // the filter row's scale (a power of two)
                sm.x *= k;
                sm.y *= k;
                if (n < a.N) {
                    if (pl == 0) sm = apply_epilogue(sm, (size_t)n * a.O + o, o, a.epi);
                    yout[(size_t)n * a.O + o] = sm;
                }
            }
        }
        if (a.alias_part) {
            // the slab's memory held the partials: zero them (the k padding of a slab row is never written by flush_row and meets
            // zero filter entries in the contraction -- it must not hold the bit patterns of partial sums)
            __syncthreads();
            for (int idx = tid; idx < partial_floats(g.NKP, g.MP); idx += kDuoThreads) l.slab[idx] = 0.f;
            __syncthreads();
        }
        stamp(6);
        par ^= 1;
    }
    stamp(30);
    stamp.realtime(31);
}

// LDS plan: record chunks per stream (4, else 2) as fit beside slab and partials in half a CU's LDS.
struct RingPlan {
    MmaGeom g;
    int nr;
    int logh;               // 1: half-size record chunks
    int alias;              // 1: partials aliased onto the slab
    size_t lds;
    bool ok;
};
// In order of preference: 4 or 2 full chunks per stream; then the partials aliased onto the slab; then half-size chunks as well
// (64 channels at band limit 3: 107 KB -> 90 KB -> 74 KB).  Half-size chunks exist for band limits >= 2 (kernel instantiations).
inline RingPlan plan_ring(int M, int F, int channels, int halves) {
    RingPlan p;
    p.g = ring_geom(M, F, channels, halves);
    p.ok = false;
    p.nr = 0; p.lds = 0; p.logh = 0; p.alias = 0;
    if (halves != 2 || p.g.NMT > kDuoWaves) return p;
    static const bool compact = !(dev_env("FC_RING_COMPACT") && atoi(dev_env("FC_RING_COMPACT")) == 0);      // 0: round-3 plans only
    const int tries[4][3] = {{4, 0, 0}, {2, 0, 0}, {2, 0, 1}, {2, 1, 1}};       // nr, logh, alias
    for (int t = 0; t < (compact ? 4 : 2); ++t) {
        if (tries[t][1] && F < 5) continue;
        const size_t lds = ring_lds_floats(p.g, tries[t][0], 256 >> tries[t][1], tries[t][2] != 0) * sizeof(float);
        if (lds <= kDuoMaxLds) {
            p.nr = tries[t][0]; p.logh = tries[t][1]; p.alias = tries[t][2]; p.lds = lds; p.ok = true;
            break;
        }
    }
    return p;
}

}  // namespace fc
