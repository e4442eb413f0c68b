// FieldConv forward for gfx950: gather -> rotate -> stencil-multiply -> segmented reduce ->
// filter contraction, fused in one kernel (replaces reference nn/field_conv.py:128-137).
//
// A 16-wavefront workgroup owns a tile of 16 target vertices, one wavefront per target.
//
//  Phase A (VALU).  Lane c of the wavefront owns input channel c.  The wavefront walks the
//    target's in-edges (CSR by target: no atomics, fixed summation order).  The source row
//    x[src,:] is one coalesced 8-byte-per-lane load, prefetched two slots ahead; the response
//    contrib[c, r, f] (R*F complex per lane) is accumulated in registers with packed fp32 FMAs.
//    Two variants differ in how the wave-uniform stencil of a slot reaches the FMAs:
//      dense     the R*F complex entries come through the scalar cache as SGPR pairs
//                (any stencil; 8*R*F bytes per edge)
//      factored  the stencil is rank-1 and 2-sparse in the ring index, S[r,f] = w_r * ph_f with
//                w_q, w_{q+1} the only non-zeros -- which is what FCPrecomp produces (reference
//                transforms/fc_precomp.py:24-25,95).  One small record (q, w_q, w_{q+1}, ph_f) per
//                edge is DMA'd (global_load_lds) into a per-wavefront LDS ring ahead of use and
//                broadcast-read: about 4F packed FMAs per edge instead of 2*R*F, 64 B instead of 240 B.
//  Phase B (MFMA).  For each angular frequency f the wavefronts drop their contrib[:, :, f]
//    slab into LDS ([vertex][k = r*I + c], re and im planes) and the workgroup multiplies it by
//    the packed filter on v_mfma_f32_16x16x4_f32 (fc_tile.hpp): out^T[o, vertex] += W[o,k] slab[vertex,k].
//    Wavefront w takes output tile (w % NOT) and every NKP-th 16-wide k block; k-partials are
//    combined through LDS in a fixed order (bitwise reproducible).
#pragma once
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

// Pointers travel as separate __restrict__ kernel parameters (not inside this struct) so that the
// compiler may treat the index and stencil streams as read-only and fetch the wave-uniform ones
// through the scalar cache (s_load) instead of per-lane vector loads.
struct FwdArgs {
    int N, I, O;
    MmaGeom g;          // M = O, K = R*I
    int ntiles;
    int nv_full, nv_total;   // work items (tile_items): whole tiles, then the last round's tiles as half tiles
    int parts_log2;     // factored kernels: every tile is processed by 2^parts_log2 workgroups, each taking that share of
                        // every target's slots and writing its own partial output (y + part * part_stride); 0 = whole tiles.
                        // Meshes with few vertices and wide supports would otherwise occupy ntiles of the 256 CUs.
    uint32_t part_stride;   // complex numbers between the partial outputs: N*O rounded up to a multiple of 2
    int ring_chunks;    // factored: 1 KiB chunks per wavefront in the LDS record ring
    int slabs;          // slab buffers in LDS: 2 = consecutive frequencies alternate buffers and need one barrier each;
                        // the k-partials of the epilogue then live in whichever buffer is idle
    uint32_t wpk_bytes; // size of the packed filter image
    int dbg;            // development only (FC_DEBUG env): bit0 skip gather loop, bit1 skip MFMA loop
    FwdEpi epi;         // residual / modReLU applied to the output tile (only when parts_log2 == 0; otherwise after the parts' sum)
};

// Frequencies are processed in NG groups of at most MG so that the per-lane response
// (R*MG complex numbers) stays within the 128-VGPR budget of a 16-wavefront workgroup.
template <int R, int B>
struct FwdShape {
    static constexpr int F = 2 * B + 1;
    static constexpr int NG = (F * R + 31) / 32;
    static constexpr int MG = (F + NG - 1) / NG;
};

// LDS carve-up shared by both kernels: slab (128*KS bytes in either mode), k-partials, per-vertex scales.
struct FwdLds {
    float* slab;     // fp32: [2][16][KS] floats; split: [4][16][KS] halves
    float* part;     // [NKP][MP][kPartStride]
    float* vscale;   // split: two buffers (tile parity) of [16] vertex slab scales + [16] inverses
    float* end;
};
__device__ __forceinline__ FwdLds forward_lds(char* smem, const MmaGeom& g, int slabs) {
    FwdLds l;
    l.slab = reinterpret_cast<float*>(smem);
    l.part = l.slab + slabs * slab_floats(g);
    l.vscale = l.part + (slabs == 2 ? 0 : partial_floats(g.NKP, g.MP));
    l.end = l.vscale + 4 * kTile;
    return l;
}
__host__ inline size_t forward_lds_floats(const MmaGeom& g, int slabs) {
    return (size_t)slabs * slab_floats(g) + (slabs == 2 ? 0 : partial_floats(g.NKP, g.MP)) + 4 * kTile;
}

// Phase B for one frequency group: slabs -> MFMA accumulate.  c[r][ff] holds this lane's response.
// first_group: no earlier group has contributed to acc in this tile.  vs: this tile's scale buffer
// (alternates between consecutive tiles of the workgroup, so a wavefront that runs ahead into the
// next tile never overwrites scales the epilogue of the previous one still reads).
template <int R, int B, int MG, bool MULTI_GROUP, bool SPLIT>
__device__ __forceinline__ void forward_phase_b(const f32x2 (&c)[R][MG], int f0, const FwdLds& l, float* vs, int& buf,
                                                const float* __restrict__ gwpk, const FwdArgs& a, int wave, int lane,
                                                bool first_group, f32x4& acc_re, f32x4& acc_im) {
    constexpr int F = 2 * B + 1;
    const MmaGeom& g = a.g;
    const int mt = wave % g.NMT, kp = wave / g.NMT;
    const bool mma_active = kp < g.NKP;
    if constexpr (!SPLIT) {
        float* const cre = l.slab;
        float* const cim = l.slab + kTile * g.KS;
        const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
#pragma unroll
        for (int ff = 0; ff < MG; ++ff) {
            const int f = f0 + ff;
            if (f < F) {
                if (lane < a.I) {
                    int o0 = wave * g.KS + lane;       // running LDS offset, see the split branch
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        cre[o0] = c[r][ff].x;
                        cim[o0] = c[r][ff].y;
                        o0 += a.I;
                        asm volatile("" : "+v"(o0));
                    }
                }
                __syncthreads();
                if (mma_active && !(a.dbg & 2))
                    mma_slab(wimg, f * (2 * g.MP * g.KP * 4), cre, cim, g, mt, kp, lane, acc_re, acc_im);
                __syncthreads();
            }
        }
    } else {
    // ---- split mode: this wavefront's vertex gets one power-of-two scale for the whole group
    float mx = 0.f;
    if (!(a.dbg & 16))
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int ff = 0; ff < MG; ++ff)
            if (f0 + ff < F) mx = fmaxf(mx, fmaxf(fabsf(c[r][ff].x), fabsf(c[r][ff].y)));
    mx = wave_max_nonneg(mx);
    float scale, inv;
    split_scale(mx, scale, inv);
    if (MULTI_GROUP && !first_group) {
        // the accumulators are in units of the previous group's scales: bring them to this group's
        const float old_inv = vs[kTile + (lane & 15)];
        __syncthreads();
        if (lane == 0) { vs[wave] = scale; vs[kTile + wave] = inv; }
        __syncthreads();
        const float ratio = old_inv * vs[lane & 15];
        acc_re *= ratio;
        acc_im *= ratio;
    } else if (lane == 0) {
        vs[wave] = scale;          // read after the slab barriers below
        vs[kTile + wave] = inv;
    }
    const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
    const int planes0 = g.MP * 4;                                  // bytes: the planes follow the MP row scales
#pragma unroll
    for (int ff = 0; ff < MG; ++ff) {
        const int f = f0 + ff;
        if (f < F) {
            // With two slab buffers consecutive frequencies alternate: a wavefront that has finished its MFMAs
            // on slab t writes slab t+1 into the other buffer at once, and that buffer is free because every
            // wavefront passed barrier t only after its MFMAs on slab t-1.  One barrier per slab instead of two,
            // and the conversions of one wavefront overlap the MFMAs of the others.
            lds_f16* const sp = (lds_f16*)(l.slab + buf * slab_floats(g));
            if (lane < g.KI && !(a.dbg & 4)) {
                // element k = r*KI + lane of my vertex's row (fc_tile.hpp: split_pair_store); one running offset
                // advanced by a ring per step -- the empty asm keeps hipcc from materialising all R addresses for
                // the whole kernel.  Lanes in [I, KI) hold copies of channel 0 and land in padding the filter zeroes.
                lds_u32* const row = (lds_u32*)sp + wave * (g.KS / 2);
                // (the default mode's two halves as a compile-time constant: no branch between the conversions)
                auto rows = [&](auto two) {
                    const int halves = decltype(two)::value ? 2 : g.split;
                    int o0 = split_pair_offset(lane, halves);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f16x2 hi, lo;
                        split_halves2(c[r][ff], scale, hi, lo);
                        split_pair_store(row, o0, hi, lo, lane, halves);
                        o0 += halves * g.KI;
                        asm volatile("" : "+v"(o0));
                    }
                };
                if (g.split == 2) rows(std::true_type{});
                else rows(std::false_type{});
            }
            __syncthreads();
            if (mma_active && !(a.dbg & 2))
                mma_slab_split(wimg, planes0 + f * (2 * g.split * g.MP * g.KP * 2), sp, g, mt, kp, lane, acc_re, acc_im);
            if (a.slabs == 2) buf ^= 1;
            else __syncthreads();
        }
    }
    }
}

// `part`: the k-partial buffer -- its own LDS region, or with two slab buffers the idle one (the last slab's
// MFMAs ran on the other; every wavefront finished the MFMAs on this one before that slab's barrier, and the
// next tile writes its first slab into the other buffer again).
template <bool SPLIT>
__device__ __forceinline__ void forward_epilogue(const FwdLds& l, float* part, const float* vs, const float* __restrict__ gwpk, const FwdArgs& a, int tile,
                                                 int wave, int lane, const f32x4& acc_re, const f32x4& acc_im,
                                                 float2* __restrict__ gy_) {
    const MmaGeom& g = a.g;
    const int mt = wave % g.NMT, kp = wave / g.NMT;
    if (kp < g.NKP) store_partial(part, g, mt, kp, lane, acc_re, acc_im);
    __syncthreads();
    if (!(a.dbg & 8))
    for (int idx = wave * kWave + lane; idx < kTile * a.O; idx += kThreads) {
        const int v = idx / a.O, o = idx - v * a.O;
        const int n = item_vertex(tile, v, a.nv_full, a.parts_log2, a.N);        // (`tile`: the work item)
        float2 s = sum_partials(part, g, v, o);
        if constexpr (SPLIT) {      // undo the slab scale of vertex v and the filter scale of row o (both powers of two)
            const float k = vs[kTile + v] * gwpk[o];
            s.x *= k;
            s.y *= k;
        }
        if (n < a.N) {
            if (a.parts_log2 == 0) s = apply_epilogue(s, (size_t)n * a.O + o, o, a.epi);
            gy_[(size_t)n * a.O + o] = s;
        }
    }
    // `part` is rewritten only after the next tile's slab barriers; the next tile uses the other scale buffer
}

// ------------------------------------------------------------------------------------------ dense
template <int R, int B, bool SPLIT>
__global__ __launch_bounds__(kThreads) void fc_forward_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ gsten, const int32_t* __restrict__ growptr,
    const int32_t* __restrict__ gnbr, const float* __restrict__ gwpk,
    float2* __restrict__ gy_, const FwdArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int NG = FwdShape<R, B>::NG;
    constexpr int MG = FwdShape<R, B>::MG;
    constexpr int ROWF = 2 * R * F;                              // floats per stencil row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FwdLds l = forward_lds(smem, a.g, a.slabs);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = a.I;

    // zero the slab once: the k padding [R*I, KP) is never written again and must not hold NaNs
    for (int idx = tid; idx < a.slabs * slab_floats(a.g); idx += kThreads) l.slab[idx] = 0.f;
    __syncthreads();

    const int cl = lane < I ? lane : 0;      // lanes >= I gather channel 0 and are never stored

    float* vs = l.vscale;
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int t = tile * kTile + wave;
        int beg = 0, end = 0;
        if (t < a.N) { beg = growptr[t]; end = growptr[t + 1]; }
        f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;

#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int MGc = MG;
            const int f0 = g * MGc;
            f32x2 c[R][MG];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int ff = 0; ff < MG; ++ff) c[r][ff] = f32x2{0.f, 0.f};

            const int last = end - 1;
            int nx = 0;                                       // source of the slot two ahead (wave-uniform)
            float2 xa = make_float2(0.f, 0.f), xb = xa;       // x rows of the even / odd slot in flight
            if (beg < end) {
                const int n0 = gnbr[beg];
                const int n1 = gnbr[min(beg + 1, last)];
                nx = gnbr[min(beg + 2, last)];
                xa = gx_[(size_t)n0 * I + cl];
                xb = gx_[(size_t)n1 * I + cl];
            }
            // one slot: `xcur` holds its source row on entry and the row of slot e+2 on exit
            auto slot = [&](const int e, float2& xcur) {
                const f32x2* __restrict__ Se = reinterpret_cast<const f32x2*>(gsten + (size_t)e * ROWF);   // uniform
                const int n3 = gnbr[min(e + 3, last)];
                f32x2 xt[F];
                rotate_all<B>(f32x2{xcur.x, xcur.y}, xt);
                xcur = gx_[(size_t)nx * I + cl];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff) {
                        const int f = f0 + ff;
                        if (f < F) {
                            cmac_sx(c[r][ff], Se[r * F + f], xt[f], f32x2{-xt[f].y, xt[f].x});
                        }
                    }
                nx = n3;
            };
            if (!(a.dbg & 1))
                for (int e = beg; e < end; e += 2) {
                    slot(e, xa);
                    if (e + 1 < end) slot(e + 1, xb);
                }
            forward_phase_b<R, B, MG, (NG > 1), SPLIT>(c, f0, l, vs, buf, gwpk, a, wave, lane, g == 0, acc_re, acc_im);
        }
        forward_epilogue<SPLIT>(l, a.slabs == 2 ? l.slab + buf * slab_floats(a.g) : l.part, vs, gwpk, a, tile, wave, lane, acc_re,
                                acc_im, gy_);
        buf ^= (a.slabs == 2);
        vs = (vs == l.vscale) ? l.vscale + 2 * kTile : l.vscale;
    }
}

// --------------------------------------------------------------------------------------- factored
// Record of one edge, RECF = ceil4(4 + 2F) floats: [0] lower ring q (int bits), [1] w_q, [2] w_{q+1},
// [3] source vertex (int bits), [4 + 2f], [5 + 2f] = ph_f.  Records are stored in target-slot order; a wavefront streams the
// records of its target in chunks of CR = 256 / RECF (one 1 KiB global_load_lds per chunk) through
// a private LDS ring of NR chunks.
// GEO: records in the geometric-phase form (fc_common.hpp: rotate_geometric), 8 floats per edge.
template <int R, int B, bool SPLIT, bool GEO>
__global__ __launch_bounds__(kThreads) void fc_forward_factored_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ grec, const int32_t* __restrict__ growptr,
    const int32_t* __restrict__ gruns, const float* __restrict__ gwpk,
    float2* __restrict__ gy_, const FwdArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int NG = FwdShape<R, B>::NG;
    constexpr int MG = FwdShape<R, B>::MG;
    constexpr int RECF = GEO ? kGeoRecordFloats : factored_record_floats(B);
    constexpr int LOG_CR = GEO ? kGeoLogChunkRecords : factored_log_chunk_records(B);   // records per chunk: a power of two, CR*RECF*4 <= 1 KiB
    constexpr int CR = 1 << LOG_CR;
    constexpr int NR = kRingChunks;                              // ring slots per wavefront (power of two)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FwdLds l = forward_lds(smem, a.g, a.slabs);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ring = l.end + wave * NR * 256;                 // [NR][256] floats, this wavefront's
    const int I = a.I;

    for (int idx = tid; idx < a.slabs * slab_floats(a.g); idx += kThreads) l.slab[idx] = 0.f;
    __syncthreads();

    const int cl = lane < I ? lane : 0;

    // chunk ch of the record run that starts at slot `first` -> ring slot ch % NR (1 KiB, asynchronous)
    auto dma_chunk = [&](const int first, const int ch) {
        const float* src = grec + ((size_t)first + (size_t)ch * CR) * RECF + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (ch & (NR - 1)) * 256), 16, 0, 0);
    };

    // ro[q] = first slot (relative to beg) whose ring index is >= q: the slots of ring q are [ro[q], ro[q+1])
    // A workgroup walks VIRTUAL tiles vt = (tile << parts_log2) + part: part p of 2^parts_log2 takes the slots
    // [n p / parts, n (p+1) / parts) of each of the tile's targets, with the ring-run offsets clipped to that range.
    const int pl = a.parts_log2;
    auto slot_range = [&](const int vt, int& b, int& e, int (&run)[R]) {
        b = 0;
        e = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        const int t = item_vertex(vt, wave, a.nv_full, pl, a.N);
        if (vt < a.nv_total && t < a.N) {
            const int rb = growptr[t];
            const int n = growptr[t + 1] - rb;
            const int part = vt & ((1 << pl) - 1);
            const int s0 = (n * part) >> pl, s1 = (n * (part + 1)) >> pl;
            b = rb + s0;
            e = rb + s1;
#pragma unroll
            for (int q = 0; q < R; ++q) run[q] = min(max(gruns[(size_t)t * kRunStride + q], s0), s1) - s0;
        }
    };
    int beg = 0, end = 0, ro[R];
    {
        slot_range(first_tile_of_block(), beg, end, ro);
        const int nch = (end - beg + CR - 1) >> LOG_CR;
        for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
    }
    // Source rows of the first two slots of a target whose first record chunk is in (or on its way to) the
    // ring: issued one tile ahead, before the epilogue of the previous tile, so their latency is hidden.
    auto first_rows = [&](const int nslots, float2& r0, float2& r1) {
        r0 = make_float2(0.f, 0.f);
        r1 = r0;
        if (nslots > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
            const int n0 = __float_as_int(ring[3]);
            const int n1 = __float_as_int(ring[min(1, nslots - 1) * RECF + 3]);
            r0 = gather_row(gx_, n0, 8u * I, 8u * cl);
            r1 = gather_row(gx_, n1, 8u * I, 8u * cl);
        }
    };
    float2 pxa, pxb;
    first_rows(end - beg, pxa, pxb);

    float* vs = l.vscale;
    int buf = 0;
    for (int vt = first_tile_of_block(); vt < a.nv_total; vt += gridDim.x) {
        const int nch = (end - beg + CR - 1) >> LOG_CR;
        f32x4 acc_re = {0.f, 0.f, 0.f, 0.f}, acc_im = acc_re;

        // next tile's slot range for this wavefront (its first chunks are DMA'd during phase B)
        int nbeg = 0, nend = 0, nro[R];
        slot_range(vt + gridDim.x, nbeg, nend, nro);

#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int MGc = MG;
            const int f0 = g * MGc;
            f32x2 c[R][MG];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int ff = 0; ff < MG; ++ff) c[r][ff] = f32x2{0.f, 0.f};

            if (g > 0) {   // later frequency groups walk the same slots again: restart the ring
                for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
            }
            const int nslots = end - beg;
            // record s of this target (s relative to beg) lives at ring[((s >> LOG_CR) & (NR-1)) * 256 + (s & (CR-1)) * RECF]
            auto rec_ptr = [&](const int s) {
                // when the records fill their 1 KiB chunks exactly the ring is one contiguous array of NR*CR records
                if constexpr (CR * RECF == 256) return ring + ((s * RECF) & (NR * 256 - 1));
                else return ring + ((s >> LOG_CR) & (NR - 1)) * 256 + (s & (CR - 1)) * RECF;
            };
            float2 xa = pxa, xb = pxb;
            if (g > 0) first_rows(nslots, xa, xb);
            // One slot whose lower ring is the compile-time constant Q: contrib[Q] += w0 z, contrib[Q+1] += w1 z
            // with z_f = ph_f * xt_f.  `xcur` holds the slot's source row on entry and the row of slot s+2 on
            // exit.  No scalar memory loads in here (they share lgkmcnt with the LDS reads and return out of
            // order) and only shifts/masks in the address arithmetic: the CU's single scalar ALU serves all
            // 16 wavefronts.
            // (xa / xb hold the source rows of the next even / odd slot; chunk entries are handled between segments of a run)
            auto slot = [&](auto qc, const int s, float2& xcur) {
                constexpr int Q = decltype(qc)::value;
                const float* rp = rec_ptr(s);                                     // wave-uniform -> broadcast reads
                const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                const float w0 = head.y, w1 = head.z;
                const int n2 = __float_as_int(rec_ptr(min(s + 2, nslots - 1))[3]);   // source two slots ahead
                const f32x2 w0v = f32x2{w0, w0}, w1v = f32x2{w1, w1};
                f32x2 z[MG];
                if constexpr (GEO) {
                    const f32x4 cg = *reinterpret_cast<const f32x4*>(rp + 4);
                    f32x2 zf[F];
                    rotate_geometric<B>(f32x2{xcur.x, xcur.y}, f32x2{cg.x, cg.y}, f32x2{cg.z, cg.w}, zf);
                    xcur = gather_row(gx_, n2, 8u * I, 8u * cl);
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) z[ff] = zf[f0 + ff];
                } else {
                    f32x2 xt[F];
                    rotate_all<B>(f32x2{xcur.x, xcur.y}, xt);
                    xcur = gather_row(gx_, n2, 8u * I, 8u * cl);
                    // z_f = ph_f * xt_f in two passes over the frequencies, then the ring updates: no packed op
                    // directly follows the one it depends on
                    f32x2 ph[MG];
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) {
                            ph[ff] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * (f0 + ff));
                            z[ff] = cmul_pk_step1(ph[ff], xt[f0 + ff]);
                        }
#pragma unroll
                    for (int ff = 0; ff < MG; ++ff)
                        if (f0 + ff < F) z[ff] = cmul_pk_step2(ph[ff], xt[f0 + ff], z[ff]);
                }
#pragma unroll
                for (int ff = 0; ff < MG; ++ff)
                    if (f0 + ff < F) c[Q][ff] = __builtin_elementwise_fma(w0v, z[ff], c[Q][ff]);
#pragma unroll
                for (int ff = 0; ff < MG; ++ff)
                    if (f0 + ff < F) c[Q + 1][ff] = __builtin_elementwise_fma(w1v, z[ff], c[Q + 1][ff]);
            };
            // The records of a target are sorted by ring index, so the walk is R-1 consecutive runs, each
            // with statically indexed accumulators (no data-dependent register indexing, no switch).
            if (!(a.dbg & 1)) {
                static_for<0, R - 1>([&](auto qc) {
                    constexpr int Q = decltype(qc)::value;
                    int s = ro[Q];
                    const int run_end = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                    while (s < run_end) {
                        const int m = s & (CR - 1);
                        if (m == 0 && s > 0) {
                            // entering a chunk: its DMA (and every older one) must have landed; the ring slot that
                            // just became free is refilled NR - 1 chunks ahead
                            const int ch = s >> LOG_CR;
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (ch - 1 + NR < nch) dma_chunk(beg, ch - 1 + NR);
                        }
                        const int stop = min(run_end, s - m + CR);
                        if ((s & 1) && s < stop) {
                            slot(qc, s, xb);
                            ++s;
                        }
                        for (; s + 1 < stop; s += 2) {
                            slot(qc, s, xa);
                            slot(qc, s + 1, xb);
                        }
                        if (s < stop) {
                            slot(qc, s, xa);
                            ++s;
                        }
                    }
                });
            }
            if (g + 1 == NG) {
                // this target is done: start streaming the next tile's first chunks; they land while
                // the MFMAs below run
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const int nnch = (nend - nbeg + CR - 1) >> LOG_CR;
                for (int ch = 0; ch < min(nnch, NR); ++ch) dma_chunk(nbeg, ch);
            }
            forward_phase_b<R, B, MG, (NG > 1), SPLIT>(c, f0, l, vs, buf, gwpk, a, wave, lane, g == 0, acc_re, acc_im);
        }
        first_rows(nend - nbeg, pxa, pxb);       // the next tile's first source rows fly during the epilogue
        forward_epilogue<SPLIT>(l, a.slabs == 2 ? l.slab + buf * slab_floats(a.g) : l.part, vs, gwpk, a, vt, wave, lane, acc_re,
                                acc_im, gy_ + (size_t)(vt & ((1 << pl) - 1)) * a.part_stride);
        buf ^= (a.slabs == 2);
        vs = (vs == l.vscale) ? l.vscale + 2 * kTile : l.vscale;
        beg = nbeg;
        end = nend;
#pragma unroll
        for (int q = 0; q < R; ++q) ro[q] = nro[q];
    }
}

template <int R, int B, int KIND, bool SPLIT>      // KIND: 0 dense stencil, 1 factored records, 2 geometric records
static int launch_forward(const float2* x, const float* sten, const fc_csr* g, const float* wpk, float2* y,
                          const FwdArgs& a, size_t lds_bytes, int grid, hipStream_t stream) {
    auto kern = KIND == 2 ? fc_forward_factored_kernel<R, B, SPLIT, true>
                : KIND == 1 ? fc_forward_factored_kernel<R, B, SPLIT, false> : fc_forward_kernel<R, B, SPLIT>;
    static bool lds_ok[kMaxDevices] = {};        // per kernel instantiation (this function is a template)
    if (!allow_full_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds_bytes, stream, x, sten, g->rowptr, KIND ? g->runs : g->nbr, wpk, y, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

// Edge split for small meshes: how many workgroups share a tile (log2).  Only when the tiles alone leave CUs idle and
// every part still gets a few slots per target.
inline int forward_parts_log2(const fc_dims* d, int kind) { return kind == 0 ? 0 : edge_parts_log2(d); }
inline size_t forward_part_stride(const fc_dims* d) { return part_stride((size_t)d->N * d->O); }
inline size_t forward_workspace_bytes_impl(const fc_dims* d, int kind) {
    const int pl = forward_parts_log2(d, kind);
    return pl ? (forward_part_stride(d) << pl) * sizeof(float2) : 0;
}

template <bool SPLIT>
int forward_impl_mode(const float* x, const float* sten, const fc_csr* g, const float* wpk, float* y,
                      const fc_dims* d, int kind, void* ws, size_t ws_bytes, const fc_epilogue* epi, hipStream_t stream) {
    const bool factored = kind != 0;
    FwdArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.g = make_mma_geom(d->O, d->R, d->I, SPLIT ? halves_of(d) : 0);
    a.ntiles = (d->N + kTile - 1) / kTile;
    // without a workspace the tiles are not split (same result, fewer workgroups)
    a.parts_log2 = (ws && ws_bytes >= forward_workspace_bytes_impl(d, kind)) ? forward_parts_log2(d, kind) : 0;
    a.part_stride = (uint32_t)forward_part_stride(d);
    a.epi = make_epi(epi);
    a.wpk_bytes = (uint32_t)(packed_image_floats(d->O, d->R, d->I, 2 * d->B + 1, a.g.split) * sizeof(float));
    static const int dbg = [] { const char* e = dev_env("FC_DEBUG"); return e ? atoi(e) : 0; }();       // read once per process
    a.dbg = dbg;
    a.ring_chunks = factored ? kRingChunks : 0;
    const size_t ring = (size_t)kWaves * a.ring_chunks * 1024;
    // dense: one tile per workgroup; factored: persistent (the record ring is primed one tile ahead)
    const int nvt = a.ntiles << a.parts_log2;
    const int grid = factored ? (nvt < num_cus() ? nvt : num_cus()) : a.ntiles;
    const TileItems items = tile_items(a.ntiles, factored ? grid : 0, a.parts_log2);
    a.nv_full = items.nv_full;
    a.nv_total = items.nv_total;
    // With two slab buffers the epilogue parks its fp32 k-partials in the idle one.  A workgroup that walks several
    // tiles would then read those bits back as halves in the k padding [R*KI, KP) of the next tile's slab rows (the
    // padding is zeroed once, before the tile loop), and 0 x NaN poisons the accumulators: shapes with k padding
    // keep the partials in their own region whenever a workgroup sees more than one tile.
    const bool aliasing_hazard = a.g.KP > d->R * a.g.KI && grid < a.nv_total;
    a.slabs = (SPLIT && !aliasing_hazard && partial_floats(a.g.NKP, a.g.MP) <= slab_floats(a.g) &&
               forward_lds_floats(a.g, 2) * sizeof(float) + ring <= kMaxLds) ? 2 : 1;
    const size_t lds = forward_lds_floats(a.g, a.slabs) * sizeof(float) + ring;
    if (lds > kMaxLds) return FC_ERR_UNSUPPORTED;
    int rc = FC_ERR_UNSUPPORTED;
#define FC_CASE(RR, BB)                                                                                              \
    if (d->R == RR && d->B == BB) {                                                                                  \
        const float2* x2 = reinterpret_cast<const float2*>(x);                                                       \
        float2* y2 = reinterpret_cast<float2*>(a.parts_log2 ? static_cast<float*>(ws) : y);                          \
        if (kind == 2) rc = launch_forward<RR, BB, 2, SPLIT>(x2, sten, g, wpk, y2, a, lds, grid, stream);            \
        else if (kind == 1) rc = launch_forward<RR, BB, 1, SPLIT>(x2, sten, g, wpk, y2, a, lds, grid, stream);       \
        else rc = launch_forward<RR, BB, 0, SPLIT>(x2, sten, g, wpk, y2, a, lds, grid, stream);                      \
    }
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    if (rc != FC_OK || a.parts_log2 == 0) return rc;
    return sum_parts_epilogue(static_cast<const float*>(ws), y, (size_t)d->N * d->O, a.part_stride, 1 << a.parts_log2, d->O, a.epi, stream);
}

}  // namespace fc
