// FieldConv backward for gfx950 (the reference has no backward code: it relies on torch autograd
// through nn/field_conv.py:128-137, i.e. the saved (E,C,R,F) product and index_select/scatter
// twins).  Here the adjoint is evaluated source-centrically and never touches an edge-sized
// temporary or an atomic:
//
//   H[j,o,r,f]  = sum_{e: src_e = j} gy[dst_e,o] conj(S[e,r,f])            (gather, CSR by source)
//   gxt[j,i,f]  = 1/F sum_{o,r} H[j,o,r,f] conj(W[o,i,r,f])                (MFMA, K = R*O)
//   gx[j,i]     = sum_f gxt_f conj(u_f) + [x != 0] (i x/|x|^2) sum_f m_f Im(conj(gxt_f) xt_f)
//   gW[o,i,r,f] = 1/F sum_j    H[j,o,r,f] conj(xt[j,i,f])                  (MFMA, K = vertices)
//
// Two kernels.
//  fc_backward_data_kernel is the forward kernel transposed: one wavefront per SOURCE vertex gathers
//  H[j,:,:,:] (lane = output channel o, all frequencies, dense or factored stencil exactly as in
//  fc_forward.hip), drops one LDS slab per frequency, the workgroup contracts it with the packed
//  conjugated filter on MFMA, and every thread folds its (vertex, channel) entry of gxt_f into a
//  running gx.  Each slab is also copied to HBM (`hdump`, 2*16*KS floats per tile and frequency).
//  The filter-gradient kernels (fc_backward.hip) need every tile's H for one frequency and 553 KB of
//  accumulators in total -- more than a CU's register file -- so blockIdx.y = f: a persistent workgroup
//  keeps gW[:,:,:,f] (KP x IP complex, spread over its wavefronts' MFMA accumulators) in registers,
//  streams the dumped slabs of its frequency back (no gather at all: the default kernel loads every
//  wavefront's vertex row into registers one slab ahead, the fp32 one uses double-buffered LDS-DMA)
//  and writes one partial at the end; fc_backward_finish sums the partials in a fixed order.
//  The 2 x 237 MB of slab traffic at config 2 replace a five-fold repetition of the gather.
#pragma once
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {


constexpr int kXbStride = 24;    // halves per row of the half-precision filter kernel's second-operand planes (16 vertices + pad)
constexpr int kXtStride = 20;    // floats per row of the filter kernel's rotated-feature tiles in LDS
constexpr int kMaxGwTiles = 8;   // 16x16 complex gW tiles a wavefront can own

// (pointers are separate __restrict__ kernel parameters, see fc_forward.hip)
struct BwdArgs {
    int N, I, O;
    MmaGeom g;           // M = I (rows of gxt), K = R*O in fp32 blocking: layout of the H slabs kept for the filter kernel
    MmaGeom gd;          // the same contraction in the data kernel's MFMA mode (fp32 or split)
    uint32_t wpk_bytes;  // size of the packed backward filter image
    int ntiles;          // work items (tile_items): (N/16 vertex tiles) << parts_log2, or whole tiles followed by the last round's
                         // tiles as half tiles; one set of H slabs each
    int nv_full;         // items below are whole tiles
    int parts_log2;      // 2^parts_log2 workgroups share a vertex tile, each with that share of every source's edges (see
                         // FwdArgs); virtual tile vt covers the vertices of tile vt >> parts_log2
    uint32_t part_stride;   // complex numbers between the parts' partial gx arrays
    int gsplit;          // 1: the two frequency GROUPS of a tile (shapes with F*R > 32: band limit 3) are separate work items -- item
                         // vt < ntiles runs group 0 of tile vt, item ntiles + t group 1 of tile t -- each with its own walk and its own
                         // partial gx (sum_parts adds the two): a mesh of 1.2 x the CUs' tiles takes 3 rounds of half items instead of 2
                         // of whole ones; on a mesh small enough for an edge split the groups replace its last doubling (half the partial
                         // H slabs, no contraction done twice).  The partial gx of (group g, edge part p) is array (g << parts_log2) + p.
                         // Not with half tiles (plan_backward)
    int ngw;             // KST * NMT 16x16 gW tiles per frequency
    int KD;              // row stride (floats) of the H slabs kept for the filter kernel, one row of interleaved
                         // (re, im) pairs per vertex: 2*KP + 8, so that the filter kernel's 8-byte A-fragment reads
                         // (rows 4 apart per lane group) hit distinct banks
    int slab_floats;     // 16 * KD
    int slab_stride;     // floats between consecutive (tile, f) slabs in hdump (multiple of 256)
    int tails;           // 1: every slab is followed by the scales the half-precision filter kernel needs:
                         // [16] s_v, [16] 1/s_v (the data kernel's vertex scales), [IP] t[i], [IP] 1/t[i] (column scales
                         // of x~[v][i] / s_v over the tile's vertices)
    int dump_halves;     // 1: a kept slab holds, per complex entry, the data kernel's own split -- (hi.re, hi.im | lo.re, lo.im) halves,
                         // 8 bytes like the fp32 pair -- so that the half2 filter-gradient kernel permutes instead of converting;
                         // 0: fp32 pairs (fp32 and single-half modes, the LDS-staged filter kernel)
    int nt_dump;         // 1: the H slabs go out with non-temporal stores (dumps that do not fit the Infinity Cache beside the rest:
                         // 244 MB at config 2 would sweep the cotangent rows and the filter out of L2 on their way; a dump that
                         // fits -- 80 MB on a FAUST-sized mesh -- is better left cached for the filter-gradient kernel)
    unsigned long long* stamps;   // development only (fc_debug_stamp_buffer): s_memtime stamps of workgroup 0 of ONE of the two kernels
    int stamp_who;                // 1: the data kernel stamps, 2: the half2 filter-gradient kernel (FC_STAMP_KERNEL=data|filter)
    int dbg;             // development only: bit0 skip gather, bit1 skip gxt MFMA, bit2 skip gW MFMA, bit3 skip the slab dump,
                         // bit4 the half2 filter kernel re-reads its first tile's rows (from L2) instead of walking the dump,
                         // bit5 the half2 filter kernel also runs a cost prototype of the H-streaming contraction (fc_backward.hip),
                         // bit6 the data kernel as a cost prototype of a gather-only kernel (walk, convert, store; nothing else)
};

// Frequency groups of the gather: NG walks of the edges with MG frequencies (R * MG complex accumulators per lane) each.  NGX forces a
// group count -- the (6, 2) layer has 30 accumulators and one group; its two-group variant exists for BwdArgs::gsplit (bwd_forced_groups)
template <int R, int B, int NGX = 0>
struct BwdShape {
    static constexpr int F = 2 * B + 1;
    static constexpr int NG = NGX ? NGX : (F * R + 31) / 32;
    static constexpr int MG = (F + NG - 1) / NG;
};
// shapes with ONE native group whose two-group kernel is instantiated (the reference's default layer)
__host__ __device__ constexpr bool bwd_forced_groups(int R, int B) { return R == 6 && B == 2; }

// ------------------------------------------------------------------------------------ data gradient
template <int R, int B, bool FACTORED, bool SPLIT, int NGX = 0>
__global__ __launch_bounds__(kThreads) void fc_backward_data_kernel(
    const float2* __restrict__ gx_, const float2* __restrict__ ggy, const float* __restrict__ gsten,
    const int32_t* __restrict__ growptr, const int32_t* __restrict__ gnbr, const float* __restrict__ gwpk,
    float2* __restrict__ ggx, float* __restrict__ hdump, const BwdArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int NG = BwdShape<R, B, NGX>::NG;
    constexpr int MG = BwdShape<R, B, NGX>::MG;
    constexpr int ROWF = 2 * R * F;
    constexpr int RECF = factored_record_floats(B);
    constexpr int LOG_CR = factored_log_chunk_records(B);
    constexpr int CR = 1 << LOG_CR;
    constexpr int NR = kRingChunks;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.gd;
    const int KS = mg.KS, IP = mg.MP, I = a.I, O = a.O;
    const int KP = a.g.KP;                                  // k entries per row of the kept H slabs (fp32 blocking)
    float* const hre = reinterpret_cast<float*>(smem);     // fp32: [16][KS] floats (re), split: [16][KS] halves, planes interleaved
    float* const him = hre + kTile * KS;                    // fp32: [16][KS] floats (im)
    float* const part = hre + slab_floats(mg);              // [NKP][16][IP][2]
    float* const vscale = part + partial_floats(mg.NKP, IP);   // split: 2 buffers of [16] vertex scales + [16] inverses

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const colmag = vscale + 4 * kTile;                    // tails: [16 vertices][64] magnitudes |x[v][i]| / s_v
    float* const ring = colmag + (a.tails ? kTile * 64 : 0) + wave * NR * 256;   // factored: [NR][256] floats per wavefront
    float* vs = vscale;                                          // scale buffer of the current frequency group

    for (int idx = tid; idx < slab_floats(mg); idx += kThreads) hre[idx] = 0.f;
    __syncthreads();

    const int it = wave % mg.NMT;
    const int kp = wave / mg.NMT;
    const bool mma_active = kp < mg.NKP;
    const int ol = lane < O ? lane : 0;       // lanes >= O gather channel 0 and are never stored
    // gx epilogue: thread -> (vertex v, channel i) of the tile, fixed for the whole kernel
    const int ev = tid / I, ei = tid - ev * I;
    const bool e_active = tid < kTile * I;

    auto dma_chunk = [&](const int first, const int ch) {
        const float* src = gsten + ((size_t)first + ((size_t)ch << LOG_CR)) * RECF + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (ch & (NR - 1)) * 256), 16, 0, 0);
    };

    // slots of my source in virtual tile vt: part p of 2^pl takes [n p / parts, n (p+1) / parts) of the source's n slots,
    // with the ring-run offsets clipped to that range (see fc_forward_kernels.hpp)
    const int pl = a.parts_log2;
    const int nitems = a.gsplit ? 2 * a.ntiles : a.ntiles;
    auto tile_of = [&](const int vt) { return (a.gsplit && vt >= a.ntiles) ? vt - a.ntiles : vt; };
    auto slot_range = [&](const int vt, int& b, int& e, int (&run)[R]) {
        b = 0;
        e = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        const int tp = tile_of(vt);                       // (tile << pl) + edge part
        const int j = item_vertex(tp, wave, a.nv_full, pl, a.N);
        if (vt < nitems && j < a.N) {
            const int rb = growptr[j];
            const int n = growptr[j + 1] - rb;
            const int part = tp & ((1 << pl) - 1);
            const int s0 = (n * part) >> pl, s1 = (n * (part + 1)) >> pl;
            b = rb + s0;
            e = rb + s1;
            if (FACTORED) {
#pragma unroll
                for (int q = 0; q < R; ++q) run[q] = min(max(gnbr[(size_t)j * kRunStride + q], s0), s1) - s0;
            }
        }
    };
    // Item of this workgroup's round k (nitems: none).  Rounds deal the items in order; with the frequency groups as work items (gsplit:
    // the items of group 0 -- one frequency heavier -- come first) the LAST, partly filled round is dealt from the TOP of the grid: those
    // workgroups took the lighter group-1 items in the round before, so the heaviest workgroup carries (group 0 + group 1 + group 1)
    // instead of (group 0 + group 0 + group 1) -- a FAUST-sized mesh (313 tiles, band limit 3): 10 frequency-walks instead of 11.
    const int grid = gridDim.x, first = first_tile_of_block();
    const int full = nitems / grid * grid, rem = nitems - full;
    auto item_of = [&](const int k) {
        const int vt = first + k * grid;
        if (!a.gsplit || vt < full) return vt < nitems ? vt : nitems;
        const int pos = first - (grid - rem);
        return (k == full / grid && pos >= 0) ? full + pos : nitems;
    };
    int beg = 0, end = 0, ro[R];      // ro: ring-run offsets of my source (factored), see fc_forward.hip
    {
        slot_range(item_of(0), beg, end, ro);
        if (FACTORED) {
            const int nch = (end - beg + CR - 1) >> LOG_CR;
            for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
        }
    }

    Stamper stamp{(a.stamps && a.stamp_who == 1 && blockIdx.x == 0) ? a.stamps + wave * 256 : nullptr, 0};       // development: in-kernel timeline
    stamp.realtime(29);
    stamp(28);
    for (int rk = 0, vt = item_of(0); vt < nitems; vt = item_of(++rk)) {
        stamp(10);
        const int tile = tile_of(vt);
        const int gsel = (a.gsplit && vt >= a.ntiles) ? 1 : 0;      // gsplit: the one frequency group this item runs
        int nbeg = 0, nend = 0, nro[R];      // my source in the next item
        slot_range(item_of(rk + 1), nbeg, nend, nro);
        const int nslots = end - beg;
        const int nch = (nslots + CR - 1) >> LOG_CR;
        // my (vertex, channel) entry of x for the gx epilogue: issued now, consumed after the first slab
        const int ejn = e_active ? item_vertex(tile, ev, a.nv_full, pl, a.N) : a.N;
        float2 exs = make_float2(0.f, 0.f);
        if (e_active && ejn < a.N) exs = gx_[(size_t)ejn * I + ei];
        float2 gxacc = make_float2(0.f, 0.f);
        float eq = 0.f;                                   // sum_f m Im(conj(gxt_f) xt_f) of my entry
        float2 eu1 = make_float2(1.f, 0.f), eu2 = eu1, eu3 = eu1;       // u, u^2, u^3 with u = exp(-i angle(x)) (1 inside the origin box)
        float einv2 = 0.f;                                // 1 / |x|^2, 0 inside the origin box
        bool e_ready = false;

#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (NG > 1 && a.gsplit && g != gsel) continue;
            constexpr int MGc = MG;
            const int f0 = g * MGc;
            f32x2 h[R][MG];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int ff = 0; ff < MG; ++ff) h[r][ff] = f32x2{0.f, 0.f};

            // ---------------------------------------------------------------- gather H for my source
            if constexpr (FACTORED) {
                if (g > 0 && !a.gsplit) for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);   // walk the slots again
                auto rec_ptr = [&](const int s) {
                // when the records fill their 1 KiB chunks exactly the ring is one contiguous array of NR*CR records
                if constexpr (CR * RECF == 256) return ring + ((s * RECF) & (NR * 256 - 1));
                else return ring + ((s >> LOG_CR) & (NR - 1)) * 256 + (s & (CR - 1)) * RECF;
            };
                float2 ga = make_float2(0.f, 0.f), gb = ga;
                if (nslots > 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
                    const int d0 = __float_as_int(rec_ptr(0)[3]);
                    const int d1 = __float_as_int(rec_ptr(min(1, nslots - 1))[3]);
                    ga = gather_row(ggy, d0, 8u * O, 8u * ol);
                    gb = gather_row(ggy, d1, 8u * O, 8u * ol);
                }
                // one slot with compile-time lower ring Q: z_f = g conj(ph_f); h[Q] += w0 z; h[Q+1] += w1 z
                // (ga / gb hold the cotangent rows of the next even / odd slot: a slot requests the row of slot + 2 into its own
                //  registers; the record ring's chunk entries are handled between segments of a run, not in the slots)
                auto slot = [&](auto qc, const int s, float2& gcur) {
                    constexpr int Q = decltype(qc)::value;
                    const float* rp = rec_ptr(s);
                    const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                    const int d2 = __float_as_int(rec_ptr(min(s + 2, nslots - 1))[3]);
                    const f32x2 gv = f32x2{gcur.x, gcur.y};
                    gcur = gather_row(ggy, d2, 8u * O, 8u * ol);
                    const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
                    // z_f = g conj(ph_f) in two passes, then the ring updates (see fc_forward_kernels.hpp)
                    f32x2 ph[MG], z[MG];
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) {
                            ph[ff] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * (f0 + ff));
                            z[ff] = cmul_conj_pk_step1(gv, ph[ff]);
                        }
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) z[ff] = cmul_conj_pk_step2(gv, ph[ff], z[ff]);
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) h[Q][ff] = __builtin_elementwise_fma(w0v, z[ff], h[Q][ff]);
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) h[Q + 1][ff] = __builtin_elementwise_fma(w1v, z[ff], h[Q + 1][ff]);
                };
                if (!(a.dbg & 1)) {
                    static_for<0, R - 1>([&](auto qc) {
                        constexpr int Q = decltype(qc)::value;
                        int s = ro[Q];
                        const int run_end = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                        while (s < run_end) {
                            const int m = s & (CR - 1);
                            if (m == 0 && s > 0) {       // entering a chunk: the one before it is consumed, its ring slot refilled
                                const int ch = s >> LOG_CR;
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                if (ch - 1 + NR < nch) dma_chunk(beg, ch - 1 + NR);
                            }
                            const int stop = min(run_end, s - m + CR);
                            if ((s & 1) && s < stop) {
                                slot(qc, s, gb);
                                ++s;
                            }
                            for (; s + 1 < stop; s += 2) {
                                slot(qc, s, ga);
                                slot(qc, s + 1, gb);
                            }
                            if (s < stop) {
                                slot(qc, s, ga);
                                ++s;
                            }
                        }
                    });
                }
                if (g + 1 == NG || a.gsplit) {
                    // my source is done: stream the first record chunks of my next item's source
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    const int nnch = (nend - nbeg + CR - 1) >> LOG_CR;
                    for (int ch = 0; ch < min(nnch, NR); ++ch) dma_chunk(nbeg, ch);
                }
            } else {
                const int last = end - 1;
                int nx = 0;
                float2 ga = make_float2(0.f, 0.f), gb = ga;
                if (beg < end) {
                    const int d0 = gnbr[beg];
                    const int d1 = gnbr[min(beg + 1, last)];
                    nx = gnbr[min(beg + 2, last)];
                    ga = ggy[(size_t)d0 * O + ol];
                    gb = ggy[(size_t)d1 * O + ol];
                }
                auto slot = [&](const int e, float2& gcur) {
                    const f32x2* __restrict__ Se = reinterpret_cast<const f32x2*>(gsten + (size_t)e * ROWF);   // wave-uniform
                    const int n3 = gnbr[min(e + 3, last)];
                    const f32x2 gv = f32x2{gcur.x, gcur.y}, gs = f32x2{gcur.y, -gcur.x};
                    gcur = ggy[(size_t)nx * O + ol];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int ff = 0; ff < MG; ++ff)
                            if (f0 + ff < F) cmac_gconjs(h[r][ff], Se[r * F + f0 + ff], gv, gs);
                    nx = n3;
                };
                if (!(a.dbg & 1))
                    for (int e = beg; e < end; e += 2) {
                        slot(e, ga);
                        if (e + 1 < end) slot(e + 1, gb);
                    }
            }

            stamp(0);
            // ---------------------------------------------------------------- slabs -> gxt_f -> gx
            // Development only (FC_DEBUG_BWD bit 6, split mode; never in the product library): a cost prototype of a GATHER-ONLY data kernel --
            // every wavefront walks, converts and stores its own rows of H and nothing else: no LDS slab, no workgroup barrier, no
            // contraction, no gx (what the H-streaming restructure of the backward pass would need this kernel to become)
            const bool gather_only = kDevSwitches && SPLIT && (a.dbg & 64);
            float scale = 1.f, inv_scale = 1.f;
            if constexpr (SPLIT) {
                // one power-of-two scale for this wavefront's vertex and frequency group (fc_tile.hpp, split mode)
                float mx = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) mx = fmaxf(mx, fmaxf(fabsf(h[r][ff].x), fabsf(h[r][ff].y)));
                mx = wave_max_nonneg(mx);
                split_scale(mx, scale, inv_scale);
                // an all-zero row (a source without out-edges, or with a zero cotangent on all of them) drops out of the filter
                // kernel's second operand x~ / s_v and of its column scales: inverse scale 0 instead of 1
                if (mx == 0.f) inv_scale = 0.f;
                if (lane == 0) { vs[wave] = scale; vs[kTile + wave] = inv_scale; }     // read after the slab barrier below
            }
#pragma unroll
            for (int ff = 0; ff < MG; ++ff) {
                const int f = f0 + ff;
                if (f < F) {
                    float* const dst = hdump + ((size_t)tile * F + f) * a.slab_stride;     // this slab, kept for the filter kernel
                    if constexpr (!SPLIT) {
                        if (lane < O) {
                            int o0 = wave * KS + lane;       // running LDS offset (opaque: see fc_forward_kernels.hpp)
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                hre[o0] = h[r][ff].x;
                                him[o0] = h[r][ff].y;
                                o0 += O;
                                asm volatile("" : "+v"(o0));
                            }
                        }
                        __syncthreads();
                        {   // 16 B per thread: two (re, im) pairs of a vertex row, rows re-strided to KD
                            const int k2n = KP / 2;
                            for (int idx = tid; idx < kTile * k2n; idx += kThreads) {
                                const int row = idx / k2n, k2 = idx - row * k2n;
                                const float2 re = *reinterpret_cast<const float2*>(hre + row * KS + 2 * k2);
                                const float2 im = *reinterpret_cast<const float2*>(him + row * KS + 2 * k2);
                                *reinterpret_cast<float4*>(dst + row * a.KD + 4 * k2) = make_float4(re.x, im.x, re.y, im.y);
                            }
                        }
                    } else {
                        if (lane < mg.KI) {
                            lds_u32* const row = (lds_u32*)hre + wave * (KS / 2);    // LDS row of my vertex, see fc_forward_kernels.hpp
                            int o0 = split_pair_offset(lane, mg.split);
                            // kept slab: my vertex's row, the rings in PAIRS (dump_k): entries (2p, lane) and (2p + 1, lane) are 16
                            // consecutive bytes, one store -- three per frequency instead of six (a store costs the CU's memory
                            // path by the instruction, not by the byte: data kernel 165 -> 161 us); the last ring of an odd count alone
                            f32x2* const dst2 = reinterpret_cast<f32x2*>(dst) + wave * (a.KD / 2);
                            // (the launch's constants as compile-time flags for the two default combinations: the selects and
                            //  branches they remove sit between the loop's 4-instruction conversions)
                            auto rows = [&](auto known, auto halves_c, auto nt_c) {
                                constexpr bool kKnown = decltype(known)::value;
                                const bool keep_halves = kKnown ? decltype(halves_c)::value : (a.dump_halves != 0);
                                const bool nt = kKnown ? decltype(nt_c)::value : (a.nt_dump != 0);
                                const int halves = kKnown ? 2 : mg.split;
                                int oo = o0, d0 = 2 * lane;
                                f32x2 even = f32x2{0.f, 0.f};
#pragma unroll
                                for (int r = 0; r < R; ++r) {
                                    f16x2 hi, lo;
                                    split_halves2(h[r][ff], scale, hi, lo);
                                    f32x2 kept = h[r][ff];
                                    if (keep_halves) kept = f32x2{__builtin_bit_cast(float, hi), __builtin_bit_cast(float, lo)};
                                    if (lane < O && (kKnown || !(a.dbg & 8))) {
                                        if (r & 1) {
                                            const f32x4 two = f32x4{even.x, even.y, kept.x, kept.y};
                                            if (nt) __builtin_nontemporal_store(two, reinterpret_cast<f32x4*>(dst2 + d0));
                                            else *reinterpret_cast<f32x4*>(dst2 + d0) = two;
                                        } else if (r == R - 1) {
                                            if (nt) __builtin_nontemporal_store(kept, dst2 + (R - 1) * O + lane);
                                            else dst2[(R - 1) * O + lane] = kept;
                                        }
                                    }
                                    even = kept;
                                    if (!gather_only) split_pair_store(row, oo, hi, lo, lane, halves);
                                    oo += halves * mg.KI;
                                    if (r & 1) d0 += 2 * O;
                                    asm volatile("" : "+v"(oo), "+v"(d0));
                                }
                            };
                            using yes = std::true_type;
                            using no = std::false_type;
                            if (a.dump_halves && mg.split == 2 && !(a.dbg & 8)) {
                                if (a.nt_dump) rows(yes{}, yes{}, yes{});
                                else rows(yes{}, yes{}, no{});
                            } else {
                                rows(no{}, no{}, no{});
                            }
                        }
                        if (a.tails && lane == 0) {
                            dst[a.slab_floats + wave] = scale;
                            dst[a.slab_floats + kTile + wave] = inv_scale;
                        }
                        if (gather_only) continue;              // (development: no slab, no barrier, no contraction, no gx -- see above)
                        stamp(1);
                        __syncthreads();
                        stamp(2);
                        if (a.tails && ff == 0 && e_active)     // bound of the filter kernel's second operand: |x[v][i]| / s_v
                            colmag[ev * 64 + ei] = sqrtf(exs.x * exs.x + exs.y * exs.y) * vs[kTile + ev] * 1.0000002f;
                    }
                    if (mma_active) {
                        f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;
                        if (!(a.dbg & 2)) {
                            const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
                            if constexpr (SPLIT)
                                mma_slab_split(wimg, IP * 4 + f * (2 * mg.split * IP * mg.KP * 2), (const lds_f16*)hre, mg, it, kp, lane, acc_re, acc_im);
                            else
                                mma_slab(wimg, f * (2 * IP * mg.KP * 4), hre, him, mg, it, kp, lane, acc_re, acc_im);
                        }
                        store_partial(part, mg, it, kp, lane, acc_re, acc_im);
                    }
                    stamp(3);
                    __syncthreads();
                    stamp(4);
                    if constexpr (SPLIT) {
                        if (a.tails && ff == 0 && tid < IP) {
                            // power-of-two column scales, the same for every frequency of this group
                            float cm = 0.f;
                            if (tid < I)
                                for (int v = 0; v < kTile; ++v) cm = fmaxf(cm, colmag[v * 64 + tid]);
                            float t, inv_t;
                            split_scale(cm, t, inv_t);
#pragma unroll
                            for (int f2 = 0; f2 < MG; ++f2)
                                if (f0 + f2 < F) {
                                    float* tail = hdump + ((size_t)tile * F + f0 + f2) * a.slab_stride + a.slab_floats + 2 * kTile;
                                    tail[tid] = t;
                                    tail[IP + tid] = inv_t;
                                }
                        }
                    }
                    if (e_active && !e_ready) {         // (after the first slab: the load of exs has long landed)
                        eu1 = unit_conj(exs);
                        eu2 = cmul(eu1, eu1);
                        if (B >= 3) eu3 = cmul(eu2, eu1);
                        einv2 = is_origin(exs) ? 0.f : 1.f / (exs.x * exs.x + exs.y * exs.y);
                        e_ready = true;
                    }
                    if (e_active) {
                        const int m = f - B;
                        float2 z = sum_partials(part, mg, ev, ei);
                        if constexpr (SPLIT) {      // undo the vertex scale and the filter-row scale (powers of two)
                            const float k = vs[kTile + ev] * gwpk[ei];
                            z.x *= k;
                            z.y *= k;
                        }
                        // gx += gxt_f conj(u^m) + [x != 0] (i x / |x|^2) m Im(conj(gxt_f) x u^m); u, u^2, u^3 and 1/|x|^2 are
                        // formed once per tile (eu1..eu3, einv2), the second term is collected as a scalar (eq)
                        const int am = m < 0 ? -m : m;
                        float2 c = am == 0 ? make_float2(1.f, 0.f) : (am == 1 ? eu1 : (am == 2 ? eu2 : eu3));
                        if (m < 0) c.y = -c.y;
                        const float2 xtv = cmul(exs, c);
                        const float2 out = cmul_conj(z, c);
                        gxacc.x += out.x;
                        gxacc.y += out.y;
                        eq += (float)m * (z.x * xtv.y - z.y * xtv.x);
                    }
                    // (`part` is next written after the following slab's first barrier)
                    stamp(5);
                }
            }
            if constexpr (SPLIT) vs = (vs == vscale) ? vscale + 2 * kTile : vscale;   // the next group writes the other buffer
        }
        if (e_active && ejn < a.N) {
            const float q = eq * einv2;                   // (i x / |x|^2) sum_f m Im(conj(gxt_f) xt_f)
            gxacc.x += -exs.y * q;
            gxacc.y += exs.x * q;
            ggx[(size_t)((gsel << pl) + (tile & ((1 << pl) - 1))) * a.part_stride + (size_t)ejn * I + ei] = gxacc;
        }
        beg = nbeg;
        end = nend;
#pragma unroll
        for (int q = 0; q < R; ++q) ro[q] = nro[q];
    }
    stamp(30);
    stamp.realtime(31);
}

// Row stride (halves) of the half-precision filter kernel's image of a tile, [16 vertices][4 planes][KP]: a multiple of
// 128 plus 16, so that the 8-byte transposed reads of eight consecutive vertices fall into distinct banks.
__host__ __device__ inline int filter_image_stride(int KP) { return round_up(4 * KP, 128) + 16; }

struct BwdPlan {
    MmaGeom g, gd;
    int IP, KP, KD, ntiles, nv_full, parts_log2, ngw, P, F, slab_floats, slab_stride, fhalf;
    size_t lds_data, lds_data_factored, lds_filter, hdump_bytes, gwp_bytes, gxp_bytes, gx_part_stride;
    bool ok, ok_factored;
    int gsplit;      // BwdArgs::gsplit
    int ngx;         // 2: the forced two-group instantiation (bwd_forced_groups) runs
};

inline BwdPlan plan_backward(const fc_dims* d, int halves) {
    BwdPlan p;
    p.F = 2 * d->B + 1;
    p.g = make_mma_geom(d->I, d->R, d->O);
    p.gd = make_mma_geom(d->I, d->R, d->O, halves);
    p.IP = p.g.MP;
    p.KP = p.g.KP;
    p.parts_log2 = edge_parts_log2(d);
    // frequency groups as work items (BwdArgs::gsplit).  FC_GROUP_SPLIT=0 switches them off (2, development: wherever legal).
    static const int gsw = [] { const char* e = dev_env("FC_GROUP_SPLIT"); return e ? atoi(e) : 1; }();
    const int native_groups = (p.F * d->R + 31) / 32;
    const bool groups = native_groups == 2 || (native_groups == 1 && bwd_forced_groups(d->R, d->B) && halves != 0);
    p.gsplit = 0;
    if (groups && gsw && p.parts_log2 > 0) {      // small mesh: the groups instead of the edge split's last doubling
        p.gsplit = 1;
        --p.parts_log2;
    }
    {
        const int nt = (d->N + kTile - 1) / kTile;
        // Half tiles in the last round (tile_items) pay in the forward kernel only: at 4 999 vertices, C = 64, B = 3 the data
        // kernel gains 3.6 us and the filter kernel loses 4.9 (57 more, half-empty slabs to stream): FC_HALF_TILES=2 turns them
        // on here as well.
        static const bool bwd_halves = dev_env("FC_HALF_TILES") && atoi(dev_env("FC_HALF_TILES")) == 2;
        const TileItems items = tile_items(nt, bwd_halves ? num_cus() : 0, p.parts_log2);
        p.ntiles = items.nv_total;                                             // work items: one set of H slabs each
        p.nv_full = items.nv_full;
    }
    p.gx_part_stride = part_stride((size_t)d->N * d->I);
    if (groups && gsw && !p.gsplit && p.parts_log2 == 0 && p.nv_full == p.ntiles) {
        // larger meshes: when half items fill the CUs' rounds better than whole tiles (a FAUST-sized mesh at band limit 3: 313 tiles on
        // 256 CUs are two rounds, 626 half items three half rounds).  Measured at 64 channels, band limit 3 (tools/time_kernels.py,
        // 263 ... 875 tiles): the split wins whenever the half items save half a round (-15 % at two rounds, -8 % at three, -5 % at four)
        // and loses 2-6 % when they do not
        const int cus = num_cus();
        const int r1 = (p.ntiles + cus - 1) / cus, r2 = (2 * p.ntiles + cus - 1) / cus;
        p.gsplit = (gsw == 2 || (r2 < 2 * r1 && r1 <= 6)) ? 1 : 0;
    }
    p.ngx = (p.gsplit && native_groups == 1) ? 2 : 0;
    const int gx_parts = (1 << p.parts_log2) << p.gsplit;
    p.gxp_bytes = gx_parts > 1 ? p.gx_part_stride * gx_parts * sizeof(float2) : 0;
    p.ngw = p.g.KST * p.g.NMT;
    int P = num_cus() / p.F;                   // filter kernel: one workgroup per CU across the F frequency slices
    if (P < 1) P = 1;
    if (P > p.ntiles) P = p.ntiles;
    p.P = P;
    p.KD = 2 * p.g.KP + 8;
    p.slab_floats = kTile * p.KD;
    // filter kernel on half-precision operands (fc_backward.hip) when its image fits beside the two fp32 slabs
    const int stride_tails = round_up(p.slab_floats + 2 * kTile + 2 * p.IP, 256);
    const size_t lds_half = (size_t)(2 * stride_tails + (kTile * filter_image_stride(p.KP) + 6 * p.IP * kXbStride) / 2 + 4) * sizeof(float);
    // (the register-fed half2 kernel needs two images and no fp32 slab in LDS; FC_FILTER2=0 selects the LDS-staged one)
    static const bool staged = [] { const char* e = dev_env("FC_FILTER2"); return e && atoi(e) == 0; }();
    const size_t lds_half2 = (size_t)(2 * kTile * filter_image_stride(p.KP) + 2 * 6 * p.IP * kXbStride + 8) * sizeof(_Float16) + 16;
    p.fhalf = (halves != 0 && p.IP <= 64 && p.KP <= 512 && (staged ? lds_half : lds_half2) <= kMaxLds) ? 1 : 0;
    p.slab_stride = p.fhalf ? stride_tails : round_up(p.slab_floats, 256);
    p.lds_data = (size_t)(slab_floats(p.gd) + partial_floats(p.gd.NKP, p.IP) + 4 * kTile + (p.fhalf ? kTile * 64 : 0)) * sizeof(float);
    p.lds_data_factored = p.lds_data + (size_t)kWaves * kRingChunks * 1024;
    p.lds_filter = p.fhalf ? (staged ? lds_half : lds_half2) : (size_t)(2 * p.slab_stride + 3 * p.IP * kXtStride) * sizeof(float);
    p.hdump_bytes = ((size_t)p.ntiles * p.F * p.slab_stride + 256) * sizeof(float);
    p.gwp_bytes = (size_t)p.P * p.F * p.KP * p.IP * sizeof(float2);
    p.ok = p.lds_data <= kMaxLds && p.lds_filter <= kMaxLds && p.ngw <= kMaxGwTiles * kWaves && p.g.NMT <= kWaves &&
           kTile * d->I <= kThreads;
    p.ok_factored = p.ok && p.lds_data_factored <= kMaxLds;
    return p;
}

inline BwdArgs make_args(const fc_dims* d, const BwdPlan& p) {
    BwdArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.g = p.g;
    a.gd = p.gd;
    a.wpk_bytes = (uint32_t)(packed_image_floats(d->I, d->R, d->O, p.F, p.gd.split) * sizeof(float));
    a.ntiles = p.ntiles;
    a.nv_full = p.nv_full;
    a.parts_log2 = p.parts_log2;
    a.part_stride = (uint32_t)p.gx_part_stride;
    a.gsplit = p.gsplit;
    a.ngw = p.ngw;
    a.KD = p.KD;
    a.slab_floats = p.slab_floats;
    a.slab_stride = p.slab_stride;
    a.tails = p.fhalf;
    static const bool staged = [] { const char* e = dev_env("FC_FILTER2"); return e && atoi(e) == 0; }();
    a.dump_halves = (p.fhalf && p.gd.split == 2 && !staged) ? 1 : 0;
    a.nt_dump = p.hdump_bytes > ((size_t)192 << 20) ? 1 : 0;
    static const int dbg = [] { const char* e = dev_env("FC_DEBUG_BWD"); return e ? atoi(e) : 0; }();       // read once per process
    a.dbg = dbg;
    a.stamps = debug_stamp_buffer();
    static const int who = [] { const char* e = dev_env("FC_STAMP_KERNEL"); return (e && e[0] == 'd') ? 1 : 2; }();
    a.stamp_who = who;
    return a;
}

template <int R, int B, bool FACTORED, bool SPLIT, int NGX = 0>
static int launch_backward_data(const float2* x, const float2* gy, const float* sten, const fc_csr* g, const float* wpk,
                                float2* gx, float* hdump, const BwdArgs& a, const BwdPlan& p, hipStream_t stream) {
    auto kern = fc_backward_data_kernel<R, B, FACTORED, SPLIT, NGX>;
    const size_t lds = FACTORED ? p.lds_data_factored : p.lds_data;
    static bool lds_ok[kMaxDevices] = {};        // per kernel instantiation (this function is a template)
    if (!allow_full_lds(reinterpret_cast<const void*>(kern), lds, lds_ok)) return FC_ERR_LAUNCH;
    const int nitems = p.ntiles << p.gsplit;
    const int grid = FACTORED ? (nitems < num_cus() ? nitems : num_cus()) : nitems;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, x, gy, sten, g->rowptr, FACTORED ? g->runs : g->nbr, wpk, gx,
                       hdump, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <bool SPLIT>
int backward_data_impl_mode(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk, float* gx,
                            void* ws, size_t ws_bytes, const fc_dims* d, bool factored, hipStream_t stream, bool defer_gx_sum) {
    const BwdPlan p = plan_backward(d, SPLIT ? halves_of(d) : 0);
    if (!(factored ? p.ok_factored : p.ok)) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes + p.gxp_bytes) return FC_ERR_WORKSPACE;
    const BwdArgs a = make_args(d, p);
    float* hdump = reinterpret_cast<float*>(ws);
    // with an edge split the parts write partial gx arrays behind the slabs and the filter partials
    float* gxp = reinterpret_cast<float*>(static_cast<char*>(ws) + p.hdump_bytes + p.gwp_bytes);
    const int gx_parts = (1 << p.parts_log2) << p.gsplit;
    float2* gx2 = reinterpret_cast<float2*>(gx_parts > 1 ? gxp : gx);
    int rc = FC_ERR_UNSUPPORTED;
    if constexpr (SPLIT) {
        if (p.ngx == 2 && d->R == 6 && d->B == 2) {      // (bwd_forced_groups)
            rc = factored ? launch_backward_data<6, 2, true, SPLIT, 2>(reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(gy), sten,
                                                                       g, wpk, gx2, hdump, a, p, stream)
                          : launch_backward_data<6, 2, false, SPLIT, 2>(reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(gy), sten,
                                                                        g, wpk, gx2, hdump, a, p, stream);
            if (rc != FC_OK || defer_gx_sum) return rc;
            return sum_parts(gxp, gx, (size_t)d->N * d->I, p.gx_part_stride, gx_parts, stream);
        }
    }
#define FC_CASE(RR, BB)                                                                                                \
    if (d->R == RR && d->B == BB)                                                                                      \
        rc = factored ? launch_backward_data<RR, BB, true, SPLIT>(reinterpret_cast<const float2*>(x),                  \
                                                           reinterpret_cast<const float2*>(gy), sten, g, wpk, gx2,     \
                                                           hdump, a, p, stream)                                        \
                      : launch_backward_data<RR, BB, false, SPLIT>(reinterpret_cast<const float2*>(x),                 \
                                                            reinterpret_cast<const float2*>(gy), sten, g, wpk, gx2,    \
                                                            hdump, a, p, stream);
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    if (rc != FC_OK || gx_parts == 1 || defer_gx_sum) return rc;
    return sum_parts(gxp, gx, (size_t)d->N * d->I, p.gx_part_stride, gx_parts, stream);
}


}  // namespace fc
