// FieldConv backward, the H-STREAMING arrangement for large meshes (split-half mode, record-driven; the maths of
// fc_backward_kernels.hpp -- reference: torch autograd through nn/field_conv.py:21,130,134).
//
//   fc_backward_gather_kernel   every wavefront walks the out-edges of ONE source vertex (lane = output channel), converts its
//                               30 ring x frequency sums H[j,o,r,f] to split halves and stores them; no LDS slab, no workgroup
//                               barrier, no contraction.  The entries of a vertex lie O-MAJOR, k = o*R + r, so that a lane's R
//                               rings are R consecutive halves of a plane row: one 2R-byte store per plane and frequency.  A
//                               (tile, frequency) record in HBM IS the image the next kernel wants in LDS:
//                                   [16 vertices][re_hi | re_lo | im_hi | im_lo][KP halves] (+16 halves of row pad), then the
//                                   vertices' power-of-two scales.
//   fc_backward_stream_kernel   blockIdx.y = frequency f.  A persistent workgroup brings its records in by LDS-DMA (no register
//                               staging, no conversion), double-buffered, ONE barrier per record, and takes BOTH products from one pass
//                               over H:
//                                   gW[k,i]  += sum_v H[v,k] conj(xt[v,i])      (9 of 16 wavefronts; transposing LDS reads)
//                                   gxt[v,i]  = sum_k H[v,k] conj(W_f[k,i])     (7 wavefronts; W_f lives in their REGISTERS for the
//                                                                                whole launch: 4 (row tile, k block) units each)
//                               The k-partials of gxt cross the wavefronts through a small LDS buffer and leave as fp32 (f, v, i).
//   fc_backward_gx_kernel       gx from the F gxt slices and x (phase-derivative term included).
//
// Against the two-kernel arrangement (data kernel with its own contraction + filter-gradient kernel) the filter no longer
// travels from L2 through every CU once per 16 vertices, the gather kernel has no slab / barrier skeleton, and H is
// regrouped by nobody.
#pragma once
#include <stdlib.h>
#include "fc_backward_kernels.hpp"

namespace fc {

constexpr int kStreamUnits = 4;      // (row tile, k block) units of the gxt product per gxt wavefront
constexpr int kStreamPartStride = 2 * kTile + 4;     // floats per row of a gxt partial tile: 16-byte aligned entry pairs, lane groups on distinct banks

struct StreamPlan {
    bool ok;
    int F, KP, IP, NMT, KST, KSI, NU, G, NW, T, ntiles, P, nslots;
    int KS;              // slices per frequency: the o-major k range in one piece or in two halves (lanes [0, O/2) | [O/2, O) of the gather)
    int FS, KPS;         // F * KS slices of KPS = KP / KS entries: what one workgroup of the streaming kernel takes (KST, KSI, NU, G, NW, T: per slice)
    int NG;              // walks per vertex of the gather kernel (at most 32 ring x frequency sums per lane and walk)
    int img_bytes;       // 16 * KSI halves
    int rec_bytes;       // image + [16] s_v + [16] 1/s_v, rounded up to whole KiB (DMA pieces)
    size_t lds;
    size_t hrec_bytes, gwp_bytes, gxt_bytes, wst_bytes;
    int nt_dump;
};

struct StreamArgs {
    int N, I, O, R, F, B;
    int KP, IP, NMT, KST, KSI, G, NW, ntiles, P, nslots;      // F, KP, KST, KSI: of a SLICE (StreamPlan::FS, KPS)
    int KS;                  // slices per frequency
    int img_bytes, rec_bytes;
    int wKI, wKP;            // channel stride and k entries per row of the packed backward image (k = r*wKI + o)
    int nt_dump;
    uint32_t* wst;           // [F][G][kStreamUnits][re_hi, re_lo, im_hi, im_lo][64 lanes][4 dwords]: written by the gather launch, read by the streaming one
    unsigned long long* stamps;   // development only (fc_debug_stamp_buffer, FC_STAMP_KERNEL=stream): s_memtime stamps of workgroup (0, 0)
    int dbg;                 // development only (FC_DEBUG_BWD): bit0 no walk, bit1 no gxt product, bit2 no gW product, bit3 no H stores,
                             // bit4 the stream kernel re-reads its first record (L2), bit5 H stored with the default cache policy, bit6 records last-produced first
};

// The arrangement serves meshes that fill the machine for several rounds, in the default arithmetic mode, on shapes whose
// registers and LDS it fits (the reference's default layer -- 48 channels, 6 rings, band limit 2 -- among them); everything
// else keeps the data / filter kernel pair.
inline StreamPlan plan_stream(const fc_dims* d, int halves, bool factored) {
    StreamPlan p = {};
    p.ok = false;
    static const bool off = [] { const char* e = dev_env("FC_BWD_STREAM"); return e && atoi(e) == 0; }();
    if (off || !factored || halves != 2 || d->N <= 0) return p;
    const int R = d->R, O = d->O, I = d->I;
    p.F = 2 * d->B + 1;
    if (R < 2 || R > 8 || (R & 1) || (R * O) % 32 != 0 || I > 64 || O > 64 || d->B < 1 || d->B > 3) return p;
    p.NG = (p.F * R + 31) / 32;                               // (42 sums at 6 rings and band limit 3: two walks, 4 + 3 frequencies)
    if (p.NG > 2) return p;
    p.KP = R * O;
    p.IP = round_up(I, 16);
    p.NMT = p.IP / 16;
    // the k range whole, else in two halves: the first that gives the two roles their wavefronts (64 -> 64 channels on 6 rings: 48 units
    // = 12 gxt wavefronts whole, 24 = 6 in halves)
    bool fits = false;
    for (p.KS = 1; p.KS <= 2 && !fits; ++p.KS) {
        if (p.KS == 2 && ((O & 1) || (p.KP / 2) % 32 != 0)) break;
        p.KPS = p.KP / p.KS;
        p.KST = p.KPS / 32;
        if (p.KST < 3) break;
        p.NU = p.NMT * p.KST;
        p.G = (p.NU + kStreamUnits - 1) / kStreamUnits;
        p.NW = kWaves - p.G;
        if (p.NW < p.NMT || p.NW < 8 || p.G < 1) continue;   // (eight gW wavefronts build the row pairs of the second operand)
        const int row_tiles = p.KPS / 16, min_ct_waves = p.NW / p.NMT;
        p.T = (row_tiles + min_ct_waves - 1) / min_ct_waves;
        fits = p.T <= 6;
        if (fits) break;
    }
    if (!fits) return p;
    p.FS = p.F * p.KS;
    p.KSI = filter_image_stride(p.KPS);
    p.nslots = p.G + p.NMT - 1;
    p.ntiles = (d->N + kTile - 1) / kTile;
    const int cus = num_cus();
    // From three quarters of a tile per CU (3 072 vertices on 256 CUs): measured against the kernel pair at 48 channels, 16 ... 96
    // neighbours (tools/time_kernels.py, data + filter launches): 1 536 / 2 048 vertices a tie (-1 ... -3 us of 50 ... 70), 3 072 /
    // 4 096 vertices -10 us of 62 ... 92, 6 000 / 8 000 vertices -25 us of 96 ... 107; 1 024 vertices with 128 neighbours (config 3) +9:
    // below, the kernel pair's edge-split / frequency-group / half-tile plans give the small mesh its parallelism.
    if (4 * p.ntiles < 3 * cus) return p;
    p.P = cus / p.FS;
    if (p.P < 1) p.P = 1;
    p.img_bytes = kTile * p.KSI * 2;
    p.rec_bytes = round_up(p.img_bytes + 2 * kTile * 4, 1024);
    if (I & 1) return p;                                     // (x rows travel by 16-byte DMA lanes)
    p.lds = (size_t)2 * p.rec_bytes + (size_t)2 * 4 * p.IP * kXbStride * 2 + (size_t)3 * kTile * I * 8 + 4 * 64 * 4 + 2 * 8 * 64 * 4 +
            2 * 64 * 4 + (size_t)2 * p.nslots * kTile * kStreamPartStride * 4;
    if (kTile * I * 8 > kWaves * 1024 - 1024) return p;
    if ((size_t)d->N * I * 8 >= ((size_t)1 << 32)) return p;            // (x rows and gxt slices are addressed with 32-bit offsets; the records with 64-bit ones)
    if (p.lds > kMaxLds) return p;
    p.hrec_bytes = (size_t)p.ntiles * p.FS * p.rec_bytes + 1024;
    p.gwp_bytes = (size_t)p.P * p.F * p.KP * p.IP * sizeof(float2);       // [P][F][KP][IP] = [P][slice][KPS][IP]
    p.gxt_bytes = (size_t)p.FS * d->N * I * sizeof(float2);
    p.wst_bytes = (size_t)p.FS * p.G * 4 * kStreamUnits * kWave * 16;     // the gxt wavefronts' filter fragments, in the order they load them
    p.nt_dump = p.hrec_bytes > ((size_t)192 << 20) ? 1 : 0;
    p.ok = true;
    return p;
}

inline StreamArgs make_stream_args(const fc_dims* d, const StreamPlan& p) {
    StreamArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O; a.R = d->R; a.F = p.FS; a.B = d->B; a.KS = p.KS;
    a.KP = p.KPS; a.IP = p.IP; a.NMT = p.NMT; a.KST = p.KST; a.KSI = p.KSI; a.G = p.G; a.NW = p.NW; a.ntiles = p.ntiles; a.P = p.P;
    a.nslots = p.nslots;
    a.img_bytes = p.img_bytes; a.rec_bytes = p.rec_bytes;
    const MmaGeom gw = make_mma_geom(d->I, d->R, d->O, 2);
    a.wKI = gw.KI; a.wKP = gw.KP;
    a.nt_dump = p.nt_dump;
    a.wst = nullptr;         // (the caller's workspace: backward_stream_impl)
    static const int dbg = [] { const char* e = dev_env("FC_DEBUG_BWD"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    static const bool stamp_me = [] { const char* e = dev_env("FC_STAMP_KERNEL"); return e && e[0] == 's'; }();
    a.stamps = stamp_me ? debug_stamp_buffer() : nullptr;
    return a;
}

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// R halves (one per ring) of one plane as R/2 dwords, stored with one instruction
template <int R, bool NT>
__device__ __forceinline__ void store_plane_row(uint32_t* dst, const uint32_t (&w)[R / 2]) {
    if constexpr (R == 2) {
        if (NT) __builtin_nontemporal_store(w[0], dst); else *dst = w[0];
    } else if constexpr (R == 4) {
        typedef u32x2 __attribute__((aligned(4))) u2;
        const u32x2 v = {w[0], w[1]};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u2*>(dst)); else *reinterpret_cast<u2*>(dst) = v;
    } else if constexpr (R == 6) {
        typedef u32x3 __attribute__((aligned(4))) u3;
        const u32x3 v = {w[0], w[1], w[2]};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u3*>(dst)); else *reinterpret_cast<u3*>(dst) = v;
    } else {
        typedef u32x4 __attribute__((aligned(4))) u4;
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst)); else *reinterpret_cast<u4*>(dst) = v;
    }
}

// ------------------------------------------------------------------------------------------------ gather
template <int R, int B>
__global__ __launch_bounds__(kThreads) void fc_backward_gather_kernel(
    const float2* __restrict__ ggy, const float* __restrict__ gsten, const int32_t* __restrict__ growptr,
    const int32_t* __restrict__ gruns, const float* __restrict__ gwpk, char* __restrict__ hrec, const StreamArgs a) {
    constexpr int F = 2 * B + 1;
    static_assert(F * R <= 64 && (R & 1) == 0, "at most two walks per vertex, an even ring count");
    constexpr int NG = (F * R + 31) / 32;             // walks per vertex
    constexpr int FG = (F + NG - 1) / NG;             // frequencies per walk (7 at 6 rings: 4 + 3)
    constexpr int RECF = factored_record_floats(B);
    constexpr int LOG_CR = factored_log_chunk_records(B);
    constexpr int CR = 1 << LOG_CR;
    constexpr int NR = kRingChunks;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ring = reinterpret_cast<float*>(smem) + wave * NR * 256;      // [NR][256] floats per wavefront
    const int O = a.O;
    const int ol = lane < O ? lane : 0;       // lanes >= O gather channel 0 and are never stored

    auto dma_chunk = [&](const int first, const int ch) {
        const float* src = gsten + ((size_t)first + ((size_t)ch << LOG_CR)) * RECF + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (ch & (NR - 1)) * 256), 16, 0, 0);
    };
    auto slot_range = [&](const int tile, int& b, int& e, int (&run)[R]) {
        b = 0;
        e = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        const int j = tile * kTile + wave;
        if (tile < a.ntiles && j < a.N) {
            b = growptr[j];
            e = growptr[j + 1];
#pragma unroll
            for (int q = 0; q < R; ++q) run[q] = gruns[(size_t)j * kRunStride + q];
        }
    };
    auto rec_ptr = [&](const int s) {
        if constexpr (CR * RECF == 256) return ring + ((s * RECF) & (NR * 256 - 1));
        else return ring + ((s >> LOG_CR) & (NR - 1)) * 256 + (s & (CR - 1)) * RECF;
    };

    const int grid = gridDim.x;
    int beg = 0, end = 0, ro[R];
    {
        slot_range(first_tile_of_block(), beg, end, ro);
        const int nch = (end - beg + CR - 1) >> LOG_CR;
        for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
    }
    // lanes of a vertex's row: output channel o = lane; with two slices per frequency the lanes [0, O/2) and [O/2, O) store into the
    // records of the first and of the second half of the o-major k range
    const int OH = O / a.KS;
    const int half = (lane >= OH && a.KS == 2) ? 1 : 0;
    const int lane_in_half = lane - half * OH;
    for (int tile = first_tile_of_block(); tile < a.ntiles; tile += grid) {
        int nbeg = 0, nend = 0, nro[R];
        slot_range(tile + grid, nbeg, nend, nro);
        const int nslots = end - beg;
        const int nch = (nslots + CR - 1) >> LOG_CR;
        char* const rec0 = hrec + (size_t)tile * a.F * a.rec_bytes;          // (a.F: slices per tile)
        const int row_off = wave * a.KSI * 2 + lane_in_half * R * 2;         // bytes: my vertex's row, my R entries of a plane

        // one walk per group of NF frequencies [F0, F0 + NF): at most 32 ring x frequency sums per lane and walk
        static_for<0, NG>([&](auto gc) {
            constexpr int GI = decltype(gc)::value;
            constexpr int F0 = GI * FG;
            constexpr int NF = (F - F0) < FG ? (F - F0) : FG;
            f32x2 h[R][NF];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int f = 0; f < NF; ++f) h[r][f] = f32x2{0.f, 0.f};

            // ---------------------------------------------------------------- the walk (fc_backward_data_kernel's, one group)
            float2 ga = make_float2(0.f, 0.f), gb = ga;
            if (nslots > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first chunks have landed
                const int d0 = __float_as_int(rec_ptr(0)[3]);
                const int d1 = __float_as_int(rec_ptr(min(1, nslots - 1))[3]);
                ga = gather_row(ggy, d0, 8u * O, 8u * ol);
                gb = gather_row(ggy, d1, 8u * O, 8u * ol);
            }
            auto slot = [&](auto qc, const int s, float2& gcur) {
                constexpr int Q = decltype(qc)::value;
                const float* rp = rec_ptr(s);
                const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                const int d2 = __float_as_int(rec_ptr(min(s + 2, nslots - 1))[3]);
                const f32x2 gv = f32x2{gcur.x, gcur.y};
                gcur = gather_row(ggy, d2, 8u * O, 8u * ol);
                const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
                f32x2 ph[NF], z[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * (F0 + f));
                    z[f] = cmul_conj_pk_step1(gv, ph[f]);
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) z[f] = cmul_conj_pk_step2(gv, ph[f], z[f]);
#pragma unroll
                for (int f = 0; f < NF; ++f) h[Q][f] = __builtin_elementwise_fma(w0v, z[f], h[Q][f]);
#pragma unroll
                for (int f = 0; f < NF; ++f) h[Q + 1][f] = __builtin_elementwise_fma(w1v, z[f], h[Q + 1][f]);
            };
            if (!(kDevSwitches && (a.dbg & 1))) {
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
            {   // this walk is done: stream the first record chunks of the next one -- my source again, or my next tile's source
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (GI + 1 < NG) {
                    for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(beg, ch);
                } else {
                    const int nnch = (nend - nbeg + CR - 1) >> LOG_CR;
                    for (int ch = 0; ch < min(nnch, NR); ++ch) dma_chunk(nbeg, ch);
                }
            }

            // ---------------------------------------------------------------- scale, split, store (this group's frequencies)
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int f = 0; f < NF; ++f) mx = fmaxf(mx, fmaxf(fabsf(h[r][f].x), fabsf(h[r][f].y)));
            mx = wave_max_nonneg(mx);
            float scale, inv_scale;
            split_scale(mx, scale, inv_scale);
            if (mx == 0.f) inv_scale = 0.f;           // an all-zero row drops out of the second operand x~ / s_v and its column scales
            auto rows = [&](auto nt_c) {
                constexpr bool NT = decltype(nt_c)::value;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    uint32_t hi[R], lo[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f16x2 h2, l2;
                        split_halves2(h[r][f], scale, h2, l2);
                        hi[r] = __builtin_bit_cast(uint32_t, h2);
                        lo[r] = __builtin_bit_cast(uint32_t, l2);
                    }
                    constexpr uint32_t kLow = 0x05040100u, kHigh = 0x07060302u;      // (b.lo16, a.lo16) / (b.hi16, a.hi16) of perm(a, b)
                    uint32_t p0[R / 2], p1[R / 2], p2[R / 2], p3[R / 2];
#pragma unroll
                    for (int dd = 0; dd < R / 2; ++dd) {
                        p0[dd] = __builtin_amdgcn_perm(hi[2 * dd + 1], hi[2 * dd], kLow);       // re_hi
                        p1[dd] = __builtin_amdgcn_perm(lo[2 * dd + 1], lo[2 * dd], kLow);       // re_lo
                        p2[dd] = __builtin_amdgcn_perm(hi[2 * dd + 1], hi[2 * dd], kHigh);      // im_hi
                        p3[dd] = __builtin_amdgcn_perm(lo[2 * dd + 1], lo[2 * dd], kHigh);      // im_lo
                    }
                    char* const recf = rec0 + (size_t)((F0 + f) * a.KS + half) * a.rec_bytes;     // the record of slice (frequency, my half)
                    if (lane < O && !(kDevSwitches && (a.dbg & 8))) {
                        char* const dst = recf + row_off;
                        store_plane_row<R, NT>(reinterpret_cast<uint32_t*>(dst), p0);
                        store_plane_row<R, NT>(reinterpret_cast<uint32_t*>(dst + a.KP * 2), p1);
                        store_plane_row<R, NT>(reinterpret_cast<uint32_t*>(dst + a.KP * 4), p2);
                        store_plane_row<R, NT>(reinterpret_cast<uint32_t*>(dst + a.KP * 6), p3);
                    }
                    if (lane_in_half == 0 && lane < O) {
                        float* tail = reinterpret_cast<float*>(recf + a.img_bytes);
                        tail[wave] = scale;
                        tail[kTile + wave] = inv_scale;
                    }
                }
            };
            if (a.nt_dump && !(kDevSwitches && (a.dbg & 32))) rows(std::true_type{}); else rows(std::false_type{});
        });
        beg = nbeg;
        end = nend;
#pragma unroll
        for (int q = 0; q < R; ++q) ro[q] = nro[q];
    }

    // ---- a rider for the launch that follows: the filter fragments of the streaming kernel's gxt wavefronts, copied out of the packed
    // backward image in the order those wavefronts load them -- entry ((f*G + g)*kStreamUnits + ui)*4 + plane, 64 lanes x 16 bytes:
    // W_f[i = mt*16 + fr][k' = kb*32 + 8*fq + j], j = 0..7, of unit u = 4g + ui = (row tile mt, k block kb), mt-major.  The packed image is
    // ring-major (k = r*KI + o) where the records are o-major (k' = o*R + r): picked in the streaming kernel itself -- 128 two-byte
    // reads per lane on seven wavefronts of every one of its workgroups, behind a staging pass through LDS and two barriers -- this was
    // 30 000 of that launch's 43 000 prologue cycles (in-kernel stamps); here one 16-byte entry per thread of the first few workgroups.
    {
        const int total = a.F * a.G * kStreamUnits * 4 * kWave;
        const int NU = a.NMT * a.KST;
        const size_t plane_sz = (size_t)a.IP * a.wKP;                               // halves per plane
        const uint16_t* const img0 = reinterpret_cast<const uint16_t*>(gwpk + a.IP);
        for (int e = blockIdx.x * kThreads + tid; e < total; e += gridDim.x * kThreads) {
            const int ln = e & 63, pl = (e >> 6) & 3, ui = (e >> 8) % kStreamUnits, fg = e / (kStreamUnits * 4 * kWave);
            const int g_ = fg % a.G, sl = fg / a.G;              // slice = (frequency, half of the k range)
            const int f_ = sl / a.KS, k0 = (sl - f_ * a.KS) * a.KP;
            const int fr_ = ln & 15, fq_ = ln >> 4;
            const int u = min(g_ * kStreamUnits + ui, NU - 1);
            const int mt = u / a.KST, kb = u - mt * a.KST;
            const int row = (mt * 16 + fr_) * 32;
            const uint16_t* const img = img0 + ((size_t)f_ * 4 + pl) * plane_sz;
            int o = (k0 + kb * 32 + 8 * fq_) / R, r = k0 + kb * 32 + 8 * fq_ - o * R;
            uint32_t v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                uint32_t pr[2];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int kk = r * a.wKI + o;
                    pr[e2] = img[(size_t)(kk >> 5) * a.IP * 32 + row + (kk & 31)];
                    if (++r == R) { r = 0; ++o; }
                }
                v[jj] = pr[0] | (pr[1] << 16);
            }
            *reinterpret_cast<u32x4*>(a.wst + (size_t)e * 4) = u32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

// ------------------------------------------------------------------------------------------------ stream
// T = 16x16 complex gW tiles a gW wavefront owns.
//
// Timeline of record k (ONE workgroup barrier per record).  Up to the barrier a wavefront only waits for its pieces of
// record k; everything else happens BEHIND the barrier, where it overlaps with the other wavefronts' matrix work:
//   gxt role: barrier(k) | DMA of record k+1 starts (its buffer was last read before the barrier), x rows of record k+3 requested |
//             gxt of record k-1 leaves (fixed-order sum of the k-partials) | gxt products of record k | wait for my DMA pieces | barrier(k+1)
//   gW role:  barrier(k) | gW products of record k | second operand of record k+1 (x~ / s_v * t[i] in halves, planes [i][vertex]) from
//             the column magnitudes written one record earlier, column magnitudes of record k+2 | barrier(k+1)
// The two roles of a SIMD take its matrix pipe in turns: one starts the interval with requests, LDS sums and stores and ends it with
// matrix instructions, the other starts with matrix instructions and ends with vector work (both roles vector-first, as first built:
// 112 us for the launch + gx on one box; this order: 101.5).
// Every LDS buffer is written in one barrier interval and read in the next, two copies each.
// KPT / IT: k entries per row and channel count as compile-time constants (0: from the arguments) -- the fragment reads of the
// default layer then carry their plane / row offsets as immediates instead of a vector add each.
template <int T, int KPT = 0, int IT = 0>
__global__ __launch_bounds__(kThreads) void fc_backward_stream_kernel(
    const float2* __restrict__ gx_, const char* __restrict__ hrec, const float* __restrict__ gwpk,
    float2* __restrict__ ggwp /* [P][F][KP][IP], k = o*R + r */, float2* __restrict__ ggxt /* [F][N][I] */, const StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KP = KPT ? KPT : a.KP, I = IT ? IT : a.I, IP = IT ? round_up(IT, 16) : a.IP, KSI = KPT ? filter_image_stride(KPT) : a.KSI;
    const int F = a.F, N = a.N;
    const int xplane = IP * kXbStride;
    const int xrs = I * 8;                                              // bytes of an x row (a multiple of 16)
    const int xtile = kTile * xrs;                                      // a tile's rows: contiguous in x
    lds_f16* const xb0 = (lds_f16*)(smem + 2 * a.rec_bytes);          // [2][c_hi, c_lo, d_hi, d_lo][IP][kXbStride] halves: second operand of gW
    char* const xrow0 = smem + 2 * a.rec_bytes + 2 * 4 * xplane * 2;    // [3][16 vertices][I] complex: the x rows of a tile (DMA)
    float* const tail0 = reinterpret_cast<float*>(xrow0 + 3 * xtile);   // [4][64]: [16] s_v, [16] 1/s_v of a record (DMA)
    float* const pmax0 = tail0 + 4 * 64;                                // [2][8 row pairs][64]  max over the pair of |x[v][i]|^2 / s_v^2
    float* const tinv0 = pmax0 + 2 * 8 * 64;                            // [2][64]  1 / t[i]
    float* const part0 = tinv0 + 2 * 64;                                // [2][nslots][16][kStreamPartStride]  k-partials of gxt
    const int part_floats = a.nslots * kTile * kStreamPartStride;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;                    // my slice: frequency f / KS, half f % KS of the k range (records, fragments, gxt, partials: per slice)
    const int m = f / a.KS - a.B;
    const int fr = lane & 15, fq = lane >> 4;
    const bool is_gxt = wave < a.G;
    // The launch is bound by the instructions its wavefronts issue (every vector instruction holds a SIMD for four cycles; a SIMD runs
    // wavefronts w, w+4, w+8, w+12) and by the memory pipe's issue rate, so the per-record chores are dealt by what the matrix roles leave
    // (in-kernel stamps, cycles per record: gxt products 1 800, gW products 2 600; the interval's 45 requests 1 300 per wavefront when
    // seven issue them, a row pair of the second operand 1 400, the gxt sum 800):
    //   gxt wavefronts: the requests, and (the first six) the fixed-order sum of the gxt partials;
    //   gW wavefronts:  one row pair of the second operand each (the first eight)
    const int op_pair = (wave >= a.G && wave - a.G < 8) ? wave - a.G : -1;
    const int red_idx = wave < 6 && wave < a.G ? wave : -1;
    const int n_red = a.G < 6 ? a.G : 6;

    // ---- my records: tiles blockIdx.x + k*P.  Everything a record needs arrives by LDS-DMA issued with instructions the compiler does
    // not track (fc_common.hpp) and is waited for once, in front of the record's barrier: between a record's requests and that wait the
    // wavefront executes NO vector-memory wait (no load whose value is used, no spill) -- vmcnt counts in issue order, and one early
    // wait behind a freshly requested record is that record's whole HBM latency.
    const int nrec = ((int)blockIdx.x < a.ntiles) ? (a.ntiles - (int)blockIdx.x + a.P - 1) / a.P : 0;
    const int npieces = a.rec_bytes >> 10;
    const int nxp = (xtile + 1023) >> 10;               // KiB pieces of a tile's x rows
    // (development: bit 4 -- always my first record, from L2; bit 6 -- my records last-produced first)
    const bool rev = kDevSwitches && (a.dbg & 64);
    auto tile_of = [&](const int k) { return (int)blockIdx.x + (rev ? nrec - 1 - k : k) * a.P; };
    const uint32_t lane16 = lane * 16;
    // The requests of a barrier interval -- the image of record k + 1 (whole KiB pieces), the x rows of tile k + 3 (contiguous in x; a
    // tile past the end of the mesh re-reads the last bytes: finite values, their 1 / s_v is 0) and that record's scales -- are dealt to
    // the first `nd` wavefronts (the gxt role: it has the shorter interval, and a wavefront that issues six pieces into a busy memory
    // pipe stands there for a thousand cycles)
    auto dma_piece = [&](const int q, const int krec, const int kx) {
        if (q < npieces) {
            if (krec < nrec) {
                const char* src = hrec + ((size_t)tile_of((kDevSwitches && (a.dbg & 16)) ? 0 : krec) * F + f) * a.rec_bytes;
                lds_dma16_saddr(src + q * 1024, lane16, smem + (krec & 1) * a.rec_bytes + q * 1024);
            }
        } else if (kx < nrec) {
            const int tile = tile_of(kx);
            const int xq = q - npieces;
            if (xq < nxp) {
                // (per-lane offsets, clamped into x: N * I * 8 < 4 GiB, plan_stream)
                const uint32_t voff = min((uint32_t)tile * (uint32_t)xtile + (uint32_t)xq * 1024u + lane16, (uint32_t)N * (uint32_t)xrs - 16u);
                const int left = xtile - xq * 1024;
                if (left >= 1024) lds_dma16_saddr(gx_, voff, xrow0 + (kx % 3) * xtile + xq * 1024);
                else lds_dma16_saddr_lanes(gx_, voff, xrow0 + (kx % 3) * xtile + xq * 1024, (1ull << (left / 16)) - 1ull);
            } else {
                lds_dma4_saddr(hrec + ((size_t)tile * F + f) * a.rec_bytes + a.img_bytes, lane * 4, tail0 + (kx & 3) * 64);
            }
        }
    };
    auto requests = [&](const int krec, const int kx, const int nd) {
        if (wave < nd)
            for (int q = wave; q < npieces + nxp + 1; q += nd) dma_piece(q, krec, kx);
    };
    // Second operand of gW for record k: x~[v][i] / s_v * t[i] in halves, planes [i][vertex], t[i] = power-of-two scale of column i over
    // the tile's vertices -- in two steps, one record apart, so that nobody forms a 16-row maximum alone:
    //   pair_max(k): my two rows' max of |x|^2 / s_v^2 per column -> pmax
    //   pair_rows(k): t[i] from the eight pair maxima; my two rows rotated, scaled, split and stored as (v, v+1) dwords
    auto pair_max = [&](const int k) {
        if (k >= nrec || op_pair < 0) return;
        const char* rows = xrow0 + (k % 3) * xtile + (2 * op_pair) * xrs;
        const float* tl = tail0 + (k & 3) * 64 + kTile + 2 * op_pair;
        const int lc = lane < I ? lane : 0;
        const float2 x0 = *reinterpret_cast<const float2*>(rows + lc * 8), x1 = *reinterpret_cast<const float2*>(rows + xrs + lc * 8);
        const float i0_ = tl[0], i1_ = tl[1];
        const float m0 = (x0.x * x0.x + x0.y * x0.y) * (i0_ * i0_), m1 = (x1.x * x1.x + x1.y * x1.y) * (i1_ * i1_);
        pmax0[((k & 1) * 8 + op_pair) * 64 + lane] = lane < I ? fmaxf(m0, m1) : 0.f;
    };
    const int pm_ = m < 0 ? -m : m;
    auto pair_rows = [&](const int k) {
        if (k >= nrec || op_pair < 0) return;
        lds_f16* const xb = xb0 + (k & 1) * 4 * xplane;
        const char* rows = xrow0 + (k % 3) * xtile + (2 * op_pair) * xrs;
        const float* tl = tail0 + (k & 3) * 64 + kTile + 2 * op_pair;
        const float* pm = pmax0 + (k & 1) * 8 * 64 + lane;
        const int lc = lane < I ? lane : 0;
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = pm[j * 64];
        float2 xv[2];
        float inv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            xv[e] = *reinterpret_cast<const float2*>(rows + e * xrs + lc * 8);
            inv[e] = tl[e];
        }
        float cm2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) cm2 = fmaxf(cm2, pv[j]);
        const float cm = __builtin_amdgcn_sqrtf(cm2) * 1.000001f;      // (never below the true value: v_sqrt_f32 is good to an ulp)
        float t, inv_t;
        split_scale(cm, t, inv_t);
        uint32_t hh[2], ll[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float2 x = lane < I ? xv[e] : make_float2(0.f, 0.f);
            const float2 u = unit_conj(x);
            float2 pw = pm_ == 0 ? make_float2(1.f, 0.f) : u;              // u^|m| (wave-uniform selection), conjugated for m < 0
            if (pm_ >= 2) pw = cmul(pw, u);
            if (pm_ >= 3) pw = cmul(pw, u);
            if (m < 0) pw.y = -pw.y;
            const float2 xt = cmul(x, pw);                                   // rotated feature of the source row
            f16x2 hi, lo;
            split_halves2(f32x2{xt.x, xt.y}, inv[e] * t, hi, lo);
            hh[e] = __builtin_bit_cast(uint32_t, hi);
            ll[e] = __builtin_bit_cast(uint32_t, lo);
        }
        if (lane < IP) {
            constexpr uint32_t kLow = 0x05040100u, kHigh = 0x07060302u;      // (b.lo16, a.lo16) / (b.hi16, a.hi16) of perm(a, b)
            lds_u32* p = (lds_u32*)xb + (lane * kXbStride) / 2 + op_pair;      // entries (v, v + 1) of row i = lane
            p[0] = __builtin_amdgcn_perm(hh[1], hh[0], kLow);                  // c_hi
            p[xplane / 2] = __builtin_amdgcn_perm(ll[1], ll[0], kLow);         // c_lo
            p[2 * (xplane / 2)] = __builtin_amdgcn_perm(hh[1], hh[0], kHigh);  // d_hi
            p[3 * (xplane / 2)] = __builtin_amdgcn_perm(ll[1], ll[0], kHigh);  // d_lo
        }
        if (op_pair == 0) tinv0[(k & 1) * 64 + lane] = inv_t;
    };
    // gxt of record k leaves: entries (v, v+1; i), fixed-order sum of the partials of row tile i / 16 (the vertex scales were divided out
    // when the partials were stored), filter-row scale divided out.  Six wavefronts, two complex entries per thread and round.
    const rsrc_t gxt_rs = make_rsrc(ggxt + (size_t)f * N * I, (uint32_t)((size_t)N * I * 8));
    // my entry of the first round (the only one up to 48 channels): filter-row scale, first partial's LDS offset (floats) | slots << 16 |
    // vertex pair << 20 | channel << 24
    float rsc = 0.f;
    int rin = -1;
    auto red_entry = [&](const int e, float& sc, int& in) {
        sc = 0.f;
        in = -1;
        if (e < 8 * IP) {
            const int vp = e / IP, ri = e - vp * IP;           // vertices 2 vp, 2 vp + 1; channel ri
            const int r_mt = ri >> 4;
            const int r_lo = (r_mt * a.KST) / kStreamUnits, r_n = ((r_mt + 1) * a.KST - 1) / kStreamUnits - r_lo + 1;
            if (ri < I) {
                sc = gwpk[ri];
                in = (((r_lo + r_mt) * kTile + (ri & 15)) * kStreamPartStride + 4 * vp) | (r_n << 16) | (vp << 20) | (ri << 24);
            }
        }
    };
    if (red_idx >= 0) red_entry(red_idx * 64 + lane, rsc, rin);
    const int red_round = n_red * 64;
    auto reduce_gxt = [&](const int k) {
        if (red_idx < 0) return;
        const float* pp0 = part0 + (k & 1) * part_floats;
        const int j0 = tile_of(k) * kTile;
        auto one = [&](const float sc, const int in) {
            if (in < 0) return;
            const float* pp = pp0 + (in & 0xffff);
            const int r_n = (in >> 16) & 15, vp2 = 2 * ((in >> 20) & 15), ri = in >> 24;
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < r_n; ++g) s4 += *reinterpret_cast<const f32x4*>(pp + g * kTile * kStreamPartStride);
            const int ob = ((j0 + vp2) * I + ri) * 8;
            if (j0 + vp2 < N) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{s4.x * sc, s4.y * sc}), gxt_rs, ob, 0, 0);
            if (j0 + vp2 + 1 < N)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{s4.z * sc, s4.w * sc}), gxt_rs, ob + I * 8, 0, 0);
        };
        one(rsc, rin);
        for (int e = red_idx * 64 + lane + red_round; e < 8 * IP; e += red_round) {       // further rounds (64 channels, fewer gxt wavefronts): their mapping is formed on the spot
            float sc2;
            int in2;
            red_entry(e, sc2, in2);
            one(sc2, in2);
        }
    };

    Stamper stamp{(kDevSwitches && a.stamps && blockIdx.x == 0 && blockIdx.y == 0) ? a.stamps + wave * 256 : nullptr, 0};
    stamp.realtime(29);
    stamp(28);
    // ---- the filter fragments of the gxt role: units u = 4*wave .. +3 of the NMT*KST (row tile mt over i, k block kb) grid, mt-major;
    // W_f[i = mt*16 + fr][k' = kb*32 + 8*fq + j] (conjugated, 1/F folded in, row-scaled halves) stay in registers for the whole launch.
    // They come as sixteen 16-byte loads per lane from the copy the gather launch left in fragment order (StreamArgs::wst).
    // (the gxt wavefronts request their fragments; then every wavefront:)
    auto first_requests = [&]() {
        __syncthreads();
        requests(0, 0, kWaves);
        requests(nrec, 1, kWaves);          // (no image: the x rows and scales of records 1 and 2)
        requests(nrec, 2, kWaves);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        pair_max(0);
        pair_max(1);
        __syncthreads();
        pair_rows(0);
    };

    // up to the barrier of record k, and what follows it for every wavefront alike
    auto head = [&](const int k) {
        stamp(10);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // my pieces of record k (and of the rows behind it) have landed; my LDS stores are done
        stamp(0);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        requests(k + 1, k + 3, a.G);
        stamp(4);
        stamp(1);
    };

    if (is_gxt) {
        // ---- gxt role
        const int u0 = wave * kStreamUnits;
        const int NU = a.NMT * a.KST;
        u32x4 wrh[kStreamUnits], wrl[kStreamUnits], wih[kStreamUnits], wil[kStreamUnits];
        {
            const u32x4* const ws = reinterpret_cast<const u32x4*>(a.wst) + (size_t)(f * a.G + wave) * (kStreamUnits * 4 * kWave) + lane;
#pragma unroll
            for (int ui = 0; ui < kStreamUnits; ++ui) {
                wrh[ui] = ws[(ui * 4 + 0) * kWave];
                wrl[ui] = ws[(ui * 4 + 1) * kWave];
                wih[ui] = ws[(ui * 4 + 2) * kWave];
                wil[ui] = ws[(ui * 4 + 3) * kWave];
            }
        }
        first_requests();
        // (first_requests waited for everything this wavefront had requested: the fragments are in.  Pinned here so that the compiler's own
        // wait for them stands in front of the loop, not at their first use inside it -- behind a record's freshly issued requests)
#pragma unroll
        for (int ui = 0; ui < kStreamUnits; ++ui) asm volatile("" : "+v"(wrh[ui]), "+v"(wrl[ui]), "+v"(wih[ui]), "+v"(wil[ui]));
        const int hbase = fr * KSI + 8 * fq;
        for (int k = 0; k < nrec; ++k) {
            head(k);
            // gxt of the record before leaves now -- vector, LDS and store work while the gW wavefronts of my SIMD are in their matrix
            // products; my own products follow (the roles take the matrix pipe in turns: -6.5 us of 108 on one box, four runs each)
            if (k > 0) reduce_gxt(k - 1);
            const lds_f16* const img = (const lds_f16*)(smem + (k & 1) * a.rec_bytes) + hbase;
            if (!(kDevSwitches && (a.dbg & 2))) {
                float* const pp = part0 + (k & 1) * part_floats + (4 * fq) * kStreamPartStride + 2 * fr;
                const float isv = tail0[(k & 3) * 64 + kTile + fr];      // 1 / s_v of my column of the partial tiles (a power of two, or 0)
                // re = (Wre Hre) - (Wim Him): the two sums in accumulators of their own, subtracted when the tile leaves (no sign flips of
                // the fragments)
                f32x4 are = {0.f, 0.f, 0.f, 0.f}, aim = are, arn = are;
                int mt_prev = u0 / a.KST;
                // (D layout of a partial tile: column = vertex = lane&15, row = 4*(lane>>4)+j; slot = wavefront + row tile)
                auto put = [&](const int mt_) {
                    float* q = pp + (wave + mt_) * kTile * kStreamPartStride;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float2*>(q + j * kStreamPartStride) = make_float2((are[j] - arn[j]) * isv, aim[j] * isv);
                };
                // H fragments of a unit: row fr of the image, k block kb, the four planes (re_hi, re_lo, im_hi, im_lo).  ONE set of registers:
                // the next unit's real planes are requested as soon as this unit's six instructions on them are issued, the imaginary
                // planes likewise (a second set does not fit beside the filter fragments)
                u32x4 h[4];
                auto hrow_of = [&](const int ui) { return img + (min(u0 + ui, NU - 1) % a.KST) * 32; };
                {
                    const lds_f16* hrow = hrow_of(0);
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl) h[pl] = *reinterpret_cast<lds_u32x4*>(hrow + pl * KP);
                }
#pragma unroll
                for (int ui = 0; ui < kStreamUnits; ++ui) {
                    const int u = u0 + ui;
                    const bool live = u < NU;
                    if (live) {
                        const int mt = u / a.KST;
                        if (mt != mt_prev) {
                            put(mt_prev);
                            are = f32x4{0.f, 0.f, 0.f, 0.f};
                            aim = are;
                            arn = are;
                            mt_prev = mt;
                        }
                        are = mfma32h(wrl[ui], h[0], are); aim = mfma32h(wil[ui], h[0], aim);
                        are = mfma32h(wrh[ui], h[1], are); aim = mfma32h(wih[ui], h[1], aim);
                        are = mfma32h(wrh[ui], h[0], are); aim = mfma32h(wih[ui], h[0], aim);
                    }
                    if (ui + 1 < kStreamUnits) {
                        const lds_f16* hn = hrow_of(ui + 1);
                        h[0] = *reinterpret_cast<lds_u32x4*>(hn);
                        h[1] = *reinterpret_cast<lds_u32x4*>(hn + KP);
                    }
                    if (live) {
                        aim = mfma32h(wrl[ui], h[2], aim); arn = mfma32h(wil[ui], h[2], arn);
                        aim = mfma32h(wrh[ui], h[3], aim); arn = mfma32h(wih[ui], h[3], arn);
                        aim = mfma32h(wrh[ui], h[2], aim); arn = mfma32h(wih[ui], h[2], arn);
                    }
                    if (ui + 1 < kStreamUnits) {
                        const lds_f16* hn = hrow_of(ui + 1);
                        h[2] = *reinterpret_cast<lds_u32x4*>(hn + 2 * KP);
                        h[3] = *reinterpret_cast<lds_u32x4*>(hn + 3 * KP);
                    }
                }
                put(mt_prev);
            }
            stamp(3);
        }
    } else {
        // ---- gW role: my tiles all lie in ONE column tile (i0), the row tiles of that column are dealt round-robin to the wavefronts that
        // share it (fc_backward_filter_half2_kernel's arithmetic)
        first_requests();
        const int jw = wave - a.G;
        const int my_ct = jw % a.NMT, my_idx = jw / a.NMT;
        const int ct_waves = (a.NW - my_ct + a.NMT - 1) / a.NMT;
        const int i0 = my_ct * 16;
        int gw_h[T];                // first k of my row tiles; -1: unused slot
#pragma unroll
        for (int n = 0; n < T; ++n) {
            const int rt = my_idx + n * ct_waves;
            gw_h[n] = (rt * 16 < KP) ? rt * 16 : -1;
        }
        const int a_lane = (8 * (fq & 1) + ((lane & 15) >> 2)) * KSI + 4 * (lane & 3) + (fq >= 2 ? KP : 0);
        const int a_hi_off = fq >= 2 ? KP : 0;               // (re_hi | im_hi) = the (re) fragment's address + this
        const int b_lane = i0 * kXbStride + fr * kXbStride + 8 * (fq & 1);    // B fragment: plane[i = i0 + fr][vertices 8*(g&1) .. +7]
        const bool upper = fq >= 2;
        const uint32_t lo_im_sign = upper ? 0u : 0x80008000u;
        f32x4 gre[T], gim[T];
#pragma unroll
        for (int n = 0; n < T; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }
        for (int k = 0; k < nrec; ++k) {
            head(k);
            const lds_f16* const img = (const lds_f16*)(smem + (k & 1) * a.rec_bytes) + a_lane;
            const lds_f16* const bp = xb0 + (k & 1) * 4 * xplane + b_lane;
            if (!(kDevSwitches && (a.dbg & 4))) {
                // second operand, once per record.  The MFMA's 32 k entries are two blocks of the 16 vertices; with H = a + ib and
                // X = c + id in halves, re = a c + b d takes THREE instructions: (a_hi | a_lo)[c_hi; c_hi] + (b_hi | b_lo)[d_hi; d_hi] +
                // (a_hi | b_hi)[c_lo; d_lo] -- the two hi*lo products share one -- and im = b c - a d likewise with [-d_lo; c_lo].
                const u32x4 c_hh = *reinterpret_cast<lds_u32x4*>(bp);
                const u32x4 d_hh = *reinterpret_cast<lds_u32x4*>(bp + 2 * xplane);
                const u32x4 lo_re = *reinterpret_cast<lds_u32x4*>(bp + (upper ? 3 : 1) * xplane);      // [c_lo; d_lo]
                u32x4 lo_im = *reinterpret_cast<lds_u32x4*>(bp + (upper ? 1 : 3) * xplane);            // [-d_lo; c_lo]
                const float it = tinv0[(k & 1) * 64 + i0 + fr];
                const u32x4 sign = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                const u32x4 nd_hh = d_hh ^ sign;
                lo_im ^= u32x4{lo_im_sign, lo_im_sign, lo_im_sign, lo_im_sign};
                // first operand: (hi | lo) of the real and of the imaginary part of H^T, and (re_hi | im_hi), by transposing reads; the
                // next tile's are requested before this tile's matrix instructions
                u32x4 af[2][3];
                auto load_a = [&](const int n, u32x4 (&dst)[3]) {
                    const lds_f16* ap = img + (gw_h[n] >= 0 ? gw_h[n] : 0);
                    const lds_f16* ah = ap + a_hi_off;
                    const u32x2 r0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap)));
                    const u32x2 r1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 4 * KSI)));
                    const u32x2 i0_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 2 * KP)));
                    const u32x2 i1_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 2 * KP + 4 * KSI)));
                    const u32x2 h0_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ah)));
                    const u32x2 h1_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ah + 4 * KSI)));
                    dst[0] = u32x4{r0.x, r0.y, r1.x, r1.y};
                    dst[1] = u32x4{i0_.x, i0_.y, i1_.x, i1_.y};
                    dst[2] = u32x4{h0_.x, h0_.y, h1_.x, h1_.y};
                };
                load_a(0, af[0]);
#pragma unroll
                for (int n = 0; n < T; ++n) {
                    if (n + 1 < T) load_a(n + 1, af[(n + 1) & 1]);
                    if (gw_h[n] >= 0) {
                        const u32x4 are = af[n & 1][0], aim = af[n & 1][1], ahi = af[n & 1][2];
                        // H conj(X), H = a + ib, X = c + id:  re = a c + b d,  im = b c - a d
                        f32x4 re = {0.f, 0.f, 0.f, 0.f}, im = re;
                        re = mfma32h(ahi, lo_re, re);  im = mfma32h(ahi, lo_im, im);
                        re = mfma32h(are, c_hh, re);   im = mfma32h(aim, c_hh, im);
                        re = mfma32h(aim, d_hh, re);   im = mfma32h(are, nd_hh, im);
                        // (packed: two fp32 products per instruction)
                        {
                            const f32x2 it2 = {it, it};
                            f32x2 g0 = {gre[n].x, gre[n].y}, g1 = {gre[n].z, gre[n].w}, g2 = {gim[n].x, gim[n].y}, g3 = {gim[n].z, gim[n].w};
                            g0 = __builtin_elementwise_fma(f32x2{re.x, re.y}, it2, g0);
                            g1 = __builtin_elementwise_fma(f32x2{re.z, re.w}, it2, g1);
                            g2 = __builtin_elementwise_fma(f32x2{im.x, im.y}, it2, g2);
                            g3 = __builtin_elementwise_fma(f32x2{im.z, im.w}, it2, g3);
                            gre[n] = f32x4{g0.x, g0.y, g1.x, g1.y};
                            gim[n] = f32x4{g2.x, g2.y, g3.x, g3.y};
                        }
                    }
                }
            }
            // the second operand of record k + 1 and the column magnitudes of record k + 2 BEHIND my matrix work, not in front of it: the gxt
            // wavefronts of my SIMD spend the head of the interval on the requests, and two roles that both start with vector work and
            // both end with matrix work leave each pipe idle half of the time (-4.6 us of 111 on one box, four runs each)
            pair_rows(k + 1);
            pair_max(k + 2);
            stamp(3);
        }
        // flush my gW partial
#pragma unroll
        for (int n = 0; n < T; ++n) {
            if (gw_h[n] >= 0) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int kk = gw_h[n] + 4 * fq + jj;
                    const int i = i0 + fr;
                    ggwp[(((size_t)blockIdx.x * F + f) * KP + kk) * IP + i] = make_float2(gre[n][jj], gim[n][jj]);
                }
            }
        }
    }
    __syncthreads();
    if (nrec > 0) reduce_gxt(nrec - 1);
    stamp(30);
    stamp.realtime(31);
}

// ------------------------------------------------------------------------------------------------ gx
// gx from the F gxt slices and x (fc_common.hpp: gx_from_slices) -- as a launch of its own behind the per-kernel entry points; behind
// fc_backward_all with module parameters the same arithmetic rides in the launch that finishes the pass (fc_pack.hip)
template <int B>
__global__ __launch_bounds__(256) void fc_backward_gx_kernel(const float2* __restrict__ gx_, const float2* __restrict__ ggxt,
                                                             float2* __restrict__ ggx, const size_t count /* N*I */, const int ks) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    ggx[idx] = gx_from_slices<B>(gx_[idx], ggxt, idx, count, ks);
}

}  // namespace fc
