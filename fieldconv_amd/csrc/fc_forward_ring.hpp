// FieldConv forward on per-edge records, RING-MAJOR: the response of a target is handed to the contraction one ring
// at a time instead of one frequency at a time (reference nn/field_conv.py:128-137; records: fc_forward_kernels.hpp).
//
// FCPrecomp's stencil touches two adjacent rings per edge (q, q+1) and the records of a target are sorted by q, so the
// walk over a target's in-edges is R-1 runs, and after run q ring q never changes again.  A wavefront therefore keeps
// only TWO rings of its target's response in registers (2F complex numbers per lane instead of R*F: 20 VGPRs at
// config 2 instead of 60) and drops ring q into an LDS slab the moment run q ends; the workgroup contracts that slab
// with the filter of ring q,
//
//     out^T[o, vertex] += W[o, k = f*KI + i; q] * slab_q[vertex, k],        K = F*KI per slab, R slabs per tile,
//
// while the wavefronts are already gathering run q+1.  What that buys over the frequency-major kernels:
//   * the gather (packed-fp32 VALU) and the contraction (MFMA + filter fragments from L2) of one tile overlap: every
//     wavefront does a short MFMA share per ring between two runs of gather work, instead of all sixteen sitting in one
//     long MFMA phase fed from L2 after all of them finished gathering;
//   * 40 more registers per lane for prefetch depth;
//   * frequency groups disappear (2F <= 14 complex accumulators for every compiled band limit).
// Precision: every slab row (vertex, ring) carries its own power-of-two scale; a slab's product is accumulated from zero
// on the matrix pipe and added to the fp32 running output with that scale divided out (fc_tile.hpp, split mode).
//
// Synchronisation: slabs live in a ring of NS buffers.  Slab s (counted per workgroup since the start of the kernel) is
// complete once sixteen rows arrived (counter full[s % NS]) and free again once every wavefront has done its MFMA share
// on it (counter done[s % NS]).  Counters are monotonic LDS words, bumped by one lane per wavefront after its LDS
// traffic drained, and polled with s_sleep in between.  A wavefront contracts slab s-1 only after it has dropped its
// row of slab s ("deferred by one"), so it waits for the slowest gatherer of run s-1 while itself one run ahead:
// per-run differences in edge counts between the sixteen targets of a tile cost 3 % instead of 20 % (a barrier per run).
#pragma once
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

struct RingArgs {
    int N, I, O;
    MmaGeom g;              // M = O, k = f*KI + i: make_mma_geom(O, F, I, halves) with NKP balanced (ring_geom)
    int ntiles;
    int parts_log2;         // edge split for small meshes, as in FwdArgs
    uint32_t part_stride;
    int NS;                 // slab buffers (2..4)
    int nr;                 // 1 KiB record chunks per wavefront in the LDS ring (2 or 4)
    uint32_t wpk_bytes;
    uint32_t slab_bytes_w;  // bytes of one ring's planes in the packed image: 2 * halves * MP * KP * 2
    int dbg;                // development only (FC_DEBUG): bit0 skip gather, bit1 skip MFMA
    unsigned long long* stamps;   // development only (fc_debug_stamp_buffer): s_memtime stamps of workgroup 0, [16 waves][256]
};

// Contraction geometry of a ring slab.  The k blocks of a slab are dealt to NKP wavefronts per output tile; among the
// partition counts that give the smallest number of blocks per wavefront the smallest one is taken (fewer k-partials).
__host__ __device__ inline MmaGeom ring_geom(int M, int F, int channels, int halves) {
    MmaGeom g = make_mma_geom(M, F, channels, halves);
    const int per = (g.KST + g.NKP - 1) / g.NKP;
    while (g.NKP > 1 && (g.KST + g.NKP - 2) / (g.NKP - 1) == per) --g.NKP;
    return g;
}

constexpr int kRingPrefetch = 4;    // source rows a wavefront keeps in flight during the gather

struct RingLds {
    float* slab;        // [NS][slab_floats]
    float* part;        // [NKP][MP][kPartStride]
    float* vinv;        // [NS][16] inverse slab scales of the sixteen rows
    uint32_t* cnt;      // [0..NS) full, [NS..2NS) done, [2NS] tile epilogue arrive, [2NS+1] tile epilogue done; 16 words
    int* runs;          // [16 wavefronts][2 tile parities][8] ring-run offsets of the wavefront's target
    float* ring;        // [16 wavefronts][nr][256]
};
__host__ __device__ inline size_t ring_lds_floats(const MmaGeom& g, int NS, int nr) {
    return (size_t)NS * slab_floats(g) + partial_floats(g.NKP, g.MP) + (size_t)NS * kTile + 16 + kWaves * 16 + (size_t)kWaves * nr * 256;
}
__device__ __forceinline__ RingLds ring_lds(char* smem, const MmaGeom& g, int NS) {
    RingLds l;
    l.slab = reinterpret_cast<float*>(smem);
    l.part = l.slab + NS * slab_floats(g);
    l.vinv = l.part + partial_floats(g.NKP, g.MP);
    l.cnt = reinterpret_cast<uint32_t*>(l.vinv + NS * kTile);
    l.runs = reinterpret_cast<int*>(l.cnt + 16);
    l.ring = reinterpret_cast<float*>(l.runs + kWaves * 16);
    return l;
}

// ---- LDS counters ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) volatile uint32_t lds_cnt;
// One lane bumps the counter after every LDS access of this wavefront has completed (writes visible, reads returned).
__device__ __forceinline__ void lds_arrive(uint32_t* c, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
        const uint32_t addr = (uint32_t)(uintptr_t)c;       // LDS byte address
        asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
    }
}
// Wait until the counter has reached `target` (wrap-safe).  Wave-uniform: every lane reads the same word.
__device__ __forceinline__ void lds_wait(uint32_t* c, uint32_t target) {
    lds_cnt* const p = (lds_cnt*)c;
    while (true) {
        const uint32_t v = *p;
        if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// In-kernel time stamps (development): lane 0 of every wavefront of workgroup 0 appends (label << 56 | s_memtime).
struct Stamper {
    unsigned long long* p;      // wave-uniform: this wavefront's 256 slots, or nullptr
    int n;
    __device__ __forceinline__ void operator()(int label) {
#ifndef FC_NO_STAMPS
        if (p) {
            if (n < 256 && (threadIdx.x & 63) == 0)
                p[n] = ((unsigned long long)label << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);
            ++n;
        }
#endif
    }
};

template <int R, int B, bool GEO>
__global__ __launch_bounds__(kThreads) void fc_forward_ring_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ grec, const int32_t* __restrict__ growptr,
    const int32_t* __restrict__ gruns, const float* __restrict__ gwpk, float2* __restrict__ gy_, const RingArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int RECF = GEO ? kGeoRecordFloats : factored_record_floats(B);
    constexpr int LOG_CR = GEO ? kGeoLogChunkRecords : factored_log_chunk_records(B);
    constexpr int CR = 1 << LOG_CR;
    constexpr int D = kRingPrefetch;         // source rows in flight per wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& g = a.g;
    const RingLds l = ring_lds(smem, g, a.NS);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = a.NS, nr = a.nr;
    float* const ring = l.ring + wave * nr * 256;
    int* const lro = l.runs + wave * 16;     // ring-run offsets of my target: [tile parity][8]
    const int I = a.I;
    const int sfl = slab_floats(g);

    // zero the slabs once (the k padding of a row is never written) and the counters
    for (int idx = tid; idx < NS * sfl; idx += kThreads) l.slab[idx] = 0.f;
    if (tid < 16) l.cnt[tid] = 0u;
    __syncthreads();

    Stamper stamp{(a.stamps && blockIdx.x == 0) ? a.stamps + wave * 256 : nullptr, 0};
    const int cl = lane < I ? lane : 0;      // lanes >= I gather channel 0; lanes >= KI are never stored
    const int mt = wave % g.NMT, kp = wave / g.NMT;
    const bool mma_active = kp < g.NKP;
    const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
    const int planes0 = g.MP * 4;            // bytes: the planes follow the MP row scales

    auto dma_chunk = [&](const int first, const int ch) {
        const float* src = grec + ((size_t)first + (size_t)ch * CR) * RECF + lane * 4;
        lds_dma16_untracked(src, ring + (ch & (nr - 1)) * 256);
    };
    const int pl = a.parts_log2;
    const int nvt = a.ntiles << pl;
    // slots [b, e) of my target in virtual tile vt; its ring-run offsets (relative to b, clipped to the part) go to lro[par]
    auto slot_range = [&](const int vt, const int par, int& b, int& e) {
        b = 0;
        e = 0;
        int run[R];
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        const int t = (vt >> pl) * kTile + wave;
        if (vt < nvt && t < a.N) {
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
#pragma unroll
            for (int q = 0; q < R; ++q) lro[par * 8 + q] = run[q];
        }
    };
    auto rec_ptr = [&](const int s) {
        if constexpr (CR * RECF == 256) return ring + ((s * RECF) & (nr * 256 - 1));
        else return ring + ((s >> LOG_CR) & (nr - 1)) * 256 + (s & (CR - 1)) * RECF;
    };
    // source rows of the first D slots of a target whose first record chunk is on its way to the ring
    auto first_rows = [&](const int nslots, float2 (&r)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) r[d] = make_float2(0.f, 0.f);
        if (nslots > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int n0 = __float_as_int(ring[min(d, nslots - 1) * RECF + 3]);
                r[d] = gather_row(gx_, n0, 8u * I, 8u * cl);
            }
        }
    };

    // Stagger: the upper half of the wavefronts (two of the four on every SIMD) starts late, so that its MFMA shares fall
    // into the gather phases of the lower half and vice versa (the slab ring gives a run of slack, the skew persists).
    if (wave >= kWaves / 2) {
        for (int i = 0; i < ((a.dbg >> 8) & 0xff); ++i) __builtin_amdgcn_s_sleep(16);
    }
    int beg = 0, end = 0, par = 0;
    {
        slot_range(first_tile_of_block(), 0, beg, end);
        const int nch = (end - beg + CR - 1) >> LOG_CR;
        for (int ch = 0; ch < min(nch, nr); ++ch) dma_chunk(beg, ch);
    }
    float2 px[D];
    first_rows(end - beg, px);

    uint32_t sseq = 0;          // slabs this workgroup has started (the same in every wavefront)
    uint32_t tseq = 0;          // tiles whose output rows have been written
    f32x4 tot_re = {0.f, 0.f, 0.f, 0.f}, tot_im = tot_re;

    // ---- filter fragments: two k blocks in registers (wf0: block kp, wf1: block kp + NKP of the ring to be contracted
    // next), fetched from L2 when the flush that precedes the contraction starts: a fetch takes 1500-2000 cycles under
    // load, about what the conversions of a flush and the wait for the slab's last row take.
    const int wv = ((mt * 16 + (lane & 15)) * 32 + 8 * (lane >> 4)) * 2;       // per-lane byte offset inside a [MP][32] k block
    const int wplane = g.MP * g.KP * 2;                                        // bytes per plane
    const int wkb = g.MP * 64;                                                 // bytes per k block
    u32x4 wf0[4], wf1[4];
    auto load_w = [&](const int rq, const int kb, u32x4 (&dst)[4]) {
        const int base = planes0 + rq * (int)a.slab_bytes_w + wkb * kb;
#pragma unroll
        for (int pnum = 0; pnum < 4; ++pnum) dst[pnum] = buffer_load16(wimg, wv, base + pnum * wplane);
    };
    // (every path defines all eight registers: a conditional definition would make them loop-carried values that stay
    // live -- and get spilled -- across the whole gather)
    auto prefetch_w = [&](const int rq) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        if (mma_active) {
            load_w(rq, kp, wf0);
            if (kp + g.NKP < g.KST) load_w(rq, kp + g.NKP, wf1);
            else { wf1[0] = zero; wf1[1] = zero; wf1[2] = zero; wf1[3] = zero; }
        } else {
            wf0[0] = zero; wf0[1] = zero; wf0[2] = zero; wf0[3] = zero;
            wf1[0] = zero; wf1[1] = zero; wf1[2] = zero; wf1[3] = zero;
        }
    };
    // one k block: 12 MFMAs (lo*hi + hi*lo + hi*hi per real product; these kernels run in the default two-halves mode only)
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
    // my MFMA share on slab number s (ring rq of its tile; fragments of its first two k blocks are in wf0 / wf1): wait
    // until its sixteen rows are in, multiply, add to the running output with the rows' scales divided out, release the
    // buffer.  rq_next >= 0: another contraction follows at once, fetch its fragments.
    auto contract = [&](const uint32_t s, const int rq, const int rq_next) {
        const int b = (int)(s % (uint32_t)NS);
        lds_wait(l.cnt + b, kWaves * (s / (uint32_t)NS + 1));
        stamp(3);
        if (mma_active && !(a.dbg & 2)) {
            f32x4 tre = {0.f, 0.f, 0.f, 0.f}, tim = tre;
            const lds_f16* sp = (const lds_f16*)(l.slab + b * sfl);
            mma_block(sp, kp, wf0, tre, tim);
            if (kp + g.NKP < g.KST) mma_block(sp, kp + g.NKP, wf1, tre, tim);
            for (int kb = kp + 2 * g.NKP; kb < g.KST; kb += g.NKP) {      // wide layers only: fetched here, latency exposed
                u32x4 w[4];
                load_w(rq, kb, w);
                mma_block(sp, kb, w, tre, tim);
            }
            const float inv = l.vinv[b * kTile + (lane & 15)];
            tot_re += tre * inv;
            tot_im += tim * inv;
        }
        stamp(4);
        lds_arrive(l.cnt + NS + b, lane);
        if (rq_next >= 0) prefetch_w(rq_next);
    };
    // The k-partials of a tile are combined one run late: a wavefront parks its partial sums when it has contracted the
    // tile's last slab and moves on; the sums are read after the first run of the next tile, when the slowest
    // wavefront has long arrived.
    int ep_tile = -1;            // virtual tile whose partials wait (or are about to be parked) in LDS
    bool tail_pending = false;   // the previous tile's two outermost slabs are not contracted yet
    auto epilogue_store = [&]() {
        if (tseq > 0) lds_wait(l.cnt + 2 * NS + 1, kWaves * tseq);      // the previous tile's sums have been read
        if (mma_active) store_partial(l.part, g, mt, kp, lane, tot_re, tot_im);
        tot_re = f32x4{0.f, 0.f, 0.f, 0.f};
        tot_im = tot_re;
        lds_arrive(l.cnt + 2 * NS, lane);
        stamp(6);
    };
    auto epilogue_sum = [&]() {
        lds_wait(l.cnt + 2 * NS, kWaves * (tseq + 1));
        stamp(7);
        const int tile = ep_tile >> pl;
        float2* const yout = gy_ + (size_t)(ep_tile & ((1 << pl) - 1)) * a.part_stride;
        for (int idx = wave * kWave + lane; idx < kTile * a.O; idx += kThreads) {
            const int v = idx / a.O, o = idx - v * a.O;
            const int n = tile * kTile + v;
            float2 sm = sum_partials(l.part, g, v, o);
            const float k = gwpk[o];             // the filter row's scale (a power of two)
            sm.x *= k;
            sm.y *= k;
            if (n < a.N) yout[(size_t)n * a.O + o] = sm;
        }
        stamp(8);
        lds_arrive(l.cnt + 2 * NS + 1, lane);
        ++tseq;
        ep_tile = -1;
    };
    // drop ring values c[f] of my target into slab number s; w_ring >= 0: the contraction of that ring follows, fetch
    // its filter fragments first
    auto flush = [&](const f32x2 (&c)[F], const uint32_t s, const int w_ring) {
        if (w_ring >= 0) prefetch_w(w_ring);
        const int b = (int)(s % (uint32_t)NS);
        float mx = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) mx = fmaxf(mx, fmaxf(fabsf(c[f].x), fabsf(c[f].y)));
        mx = wave_max_nonneg(mx);
        float scale, inv;
        split_scale(mx, scale, inv);
        if (s >= (uint32_t)NS) lds_wait(l.cnt + NS + b, kWaves * (s / (uint32_t)NS));     // every MFMA on the buffer's previous slab is done
        stamp(1);
        if (lane == 0) l.vinv[b * kTile + wave] = inv;
        if (lane < g.KI) {
            lds_u32* const row = (lds_u32*)(l.slab + b * sfl) + wave * (g.KS / 2);
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
        lds_arrive(l.cnt + b, lane);
        stamp(2);
    };

    for (int vt = first_tile_of_block(); vt < nvt; vt += gridDim.x) {
        const int nslots = end - beg;
        const int nch = (nslots + CR - 1) >> LOG_CR;
        int nbeg = 0, nend = 0;
        slot_range(vt + gridDim.x, par ^ 1, nbeg, nend);

        f32x2 clo[F], chi[F];
#pragma unroll
        for (int f = 0; f < F; ++f) { clo[f] = f32x2{0.f, 0.f}; chi[f] = clo[f]; }
        float2 x0 = px[0], x1 = px[1], x2 = px[2], x3 = px[3];

        // one slot of a run: clo += w0 z, chi += w1 z with z_f = ph_f * x~_f.  `xcur` holds the slot's source row on entry
        // and the row of slot s + D on exit.
        auto slot = [&](const int s, float2& xcur) {
            if ((s & (CR - 1)) == 0 && s > 0) {
                // entering a chunk: the chunk before it is consumed, its ring slot is refilled nr - 1 chunks ahead
                const int ch = s >> LOG_CR;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (ch - 1 + nr < nch) dma_chunk(beg, ch - 1 + nr);
            }
            if (nr == 2 && ((s + D) & (CR - 1)) < 1)          // the look-ahead below enters a chunk issued at the last chunk entry
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float* rp = rec_ptr(s);
            const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
            const int nd = __float_as_int(rec_ptr(min(s + D, nslots - 1))[3]);
            const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
            f32x2 z[F];
            if constexpr (GEO) {
                const f32x4 cg = *reinterpret_cast<const f32x4*>(rp + 4);
                rotate_geometric<B>(f32x2{xcur.x, xcur.y}, f32x2{cg.x, cg.y}, f32x2{cg.z, cg.w}, z);
                xcur = gather_row(gx_, nd, 8u * I, 8u * cl);
            } else {
                f32x2 xt[F], ph[F];
                rotate_all<B>(f32x2{xcur.x, xcur.y}, xt);
                xcur = gather_row(gx_, nd, 8u * I, 8u * cl);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                    z[f] = cmul_pk_step1(ph[f], xt[f]);
                }
#pragma unroll
                for (int f = 0; f < F; ++f) z[f] = cmul_pk_step2(ph[f], xt[f], z[f]);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) clo[f] = __builtin_elementwise_fma(w0v, z[f], clo[f]);
#pragma unroll
            for (int f = 0; f < F; ++f) chi[f] = __builtin_elementwise_fma(w1v, z[f], chi[f]);
        };

        const uint32_t s0 = sseq;           // slab number of ring 0 of this tile
        stamp(10);
        for (int q = 0; q < R - 1; ++q) {
            if (!(a.dbg & 1)) {
                int s = __builtin_amdgcn_readfirstlane(lro[par * 8 + q]);
                const int run_end = (q + 1 < R - 1) ? __builtin_amdgcn_readfirstlane(lro[par * 8 + q + 1]) : nslots;
                for (; s + 3 < run_end; s += 4) {
                    slot(s, x0);
                    slot(s + 1, x1);
                    slot(s + 2, x2);
                    slot(s + 3, x3);
                }
                // tail of the run: up to three slots, then the prefetch registers are rotated so that x0 is the next slot's row
                const int rem = run_end - s;
                if (rem == 1) {
                    slot(s, x0);
                    const float2 t = x0; x0 = x1; x1 = x2; x2 = x3; x3 = t;
                } else if (rem == 2) {
                    slot(s, x0);
                    slot(s + 1, x1);
                    float2 t = x0; x0 = x2; x2 = t;
                    t = x1; x1 = x3; x3 = t;
                } else if (rem == 3) {
                    slot(s, x0);
                    slot(s + 1, x1);
                    slot(s + 2, x2);
                    const float2 t = x3; x3 = x2; x2 = x1; x1 = x0; x0 = t;
                }
            }
            stamp(0);
            if (q == R - 2) {
                // my target is done: start streaming the first record chunks of my next tile's target
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int nnch = (nend - nbeg + CR - 1) >> LOG_CR;
                for (int ch = 0; ch < min(nnch, nr); ++ch) dma_chunk(nbeg, ch);
            }
            if (q == 0 && tail_pending) {
                // the previous tile's two outermost rings (contracted a run late, see below), then its k-partials
                flush(clo, s0, R - 2);
                contract(s0 - 2, R - 2, R - 1);
                contract(s0 - 1, R - 1, -1);
                epilogue_store();
                tail_pending = false;
            } else {
                flush(clo, s0 + q, q - 1);                   // ring q is final
                if (q > 0) contract(s0 + q - 1, q - 1, -1);   // the slab before it: its slowest row had a whole run of slack
            }
            if (q == 1 && ep_tile >= 0) epilogue_sum();      // the previous tile's output rows
#pragma unroll
            for (int f = 0; f < F; ++f) { clo[f] = chi[f]; chi[f] = f32x2{0.f, 0.f}; }
        }
        // The outermost two rings become final together, after the last run: contracting them here would make every
        // wavefront wait for the slowest one's whole tile.  With three slab buffers they wait, as slabs, for the first flush
        // of the next tile.
        if (R == 2 && ep_tile >= 0) epilogue_sum();     // (no second run to hide it under)
        first_rows(nend - nbeg, px);                     // the next tile's first source rows
        stamp(5);
        if (NS >= 3 && R >= 3 && vt + (int)gridDim.x < nvt) {
            flush(clo, s0 + R - 1, -1);
            tail_pending = true;
        } else {
            flush(clo, s0 + R - 1, R - 2);
            contract(s0 + R - 2, R - 2, R - 1);
            contract(s0 + R - 1, R - 1, -1);
            epilogue_store();
        }
        sseq = s0 + R;
        ep_tile = vt;
        beg = nbeg;
        end = nend;
        par ^= 1;
    }
    if (ep_tile >= 0) epilogue_sum();
}

// LDS plan: as many slab buffers (up to 3) and record chunks (4, else 2) as fit.
struct RingPlan {
    MmaGeom g;
    int NS, nr;
    size_t lds;
    bool ok;
};
inline RingPlan plan_ring(int M, int F, int channels, int halves) {
    RingPlan p;
    p.g = ring_geom(M, F, channels, halves);
    p.ok = false;
    p.NS = 0; p.nr = 0; p.lds = 0;
    if (!halves) return p;
    const int tries[4][2] = {{3, 4}, {3, 2}, {2, 4}, {2, 2}};
    for (const auto& t : tries) {
        const size_t lds = ring_lds_floats(p.g, t[0], t[1]) * sizeof(float);
        if (lds <= kMaxLds) {
            p.NS = t[0]; p.nr = t[1]; p.lds = lds; p.ok = true;
            break;
        }
    }
    return p;
}

}  // namespace fc
