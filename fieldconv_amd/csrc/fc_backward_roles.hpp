// Ring-major backward data kernel with WAVEFRONT ROLES (round 3; same workspace layout, packed image and filter-gradient
// kernel as fc_backward_ring.hpp).
//
// What bounds fc_backward_ring_data_kernel and the frequency-major kernel alike is the register file: sixteen wavefronts per
// CU leave 128 registers each, every wavefront carries ring sums AND filter fragments AND accumulators, so only two cotangent
// rows per wavefront are in flight and the gather waits for L2.  Here a workgroup has TWELVE wavefronts
// (168 registers each) and they split the work:
//
//   wavefronts 0..7    GATHER: four source vertices each (rows w + 8 j of the 32-row tile), their record streams walked in
//                      turns -- sixteen cotangent rows in flight per wavefront, requested and waited for with the kernel's own
//                      count (below) -- two rings of each response in registers, conversion into the LDS slab when a ring is
//                      final.  No filter fragments, no accumulators, no stores to HBM.
//   wavefronts 8..11   CONTRACT: the (input-channel tile, frequency) pairs m, m + 4, m + 8, m + 12 each, all R slabs of a tile
//                      accumulated in registers, three filter fragment sets in flight; they also copy the slab to the workspace
//                      for the filter-gradient kernel (after their MFMAs, so that the stores never sit in front of a fragment
//                      load they wait for) and write its scales.
//
// One slab buffer, two barriers per ring: "slab free" (the gatherers arrive after run q+1, the contractors after slab q) and
// "slab full".  While the contractors work on slab q the gatherers walk run q+1: the matrix pipe, the L2 filter stream and the
// slab's write stream run beside the gather instead of after it.
//
// MEASURED (config 2, isolated launches): 247 us against 225 us for fc_backward_ring_data_kernel and
// 171 us for the frequency-major kernel -- parity-green, opt-in (FC_BWD_RING with FC_BWD_ROLES=1, band limit <= 2).  The phases
// do overlap (the contraction, 11 k cycles per slab, and the slab copy run under the next run's gather), but the eight
// gathering wavefronts are bound by their own instruction issue: switching off the row requests, the record reads or the
// accumulating FMAs of the walk (FC_ROLES_EXP builds, tools/build_variants.sh) changes the kernel by 4, 13 and 29 us -- a
// wavefront issues one instruction every ~5 cycles whatever the instruction is, the walk has ~70 per slot of which 22 are
// vector arithmetic, and two gathering wavefronts per SIMD cannot hide each other the way four do.
#pragma once
#include "fc_backward_ring.hpp"

#ifndef FC_ROLES_EXP
#define FC_ROLES_EXP 0      // development builds only (results are wrong): 1 no row requests, 2 no record reads, 3 no accumulation, 4 no waits
#endif

namespace fc {

constexpr int kRoleWaves = 12;
constexpr int kRoleThreads = kRoleWaves * kWave;          // 768
constexpr int kBrGatherWaves = 8;
constexpr int kBrMmaWaves = kRoleWaves - kBrGatherWaves;  // 4
constexpr int kBrStreams = kBrRows / kBrGatherWaves;      // 4 source vertices per gathering wavefront
constexpr int kBrMaxPairs = 4;                            // (i tile, frequency) pairs per contracting wavefront
constexpr int kBrMaxJobs = 2 * kBrMaxPairs;               // filter blocks per contracting wavefront and slab
constexpr int kBrEntries = 2;                             // gx entries per thread: 32 I <= 2 * 768

struct BrRolesLds {
    char* slab;
    float* vinv;
    float* vsc;
    uint32_t* cmax;
    int* runs;          // [8 wavefronts][4 streams][2 tile parities][8]
    float* xmag;        // [32 rows][64]: |x[row][lane]| of the tile in hand
    float* ring;        // [8 wavefronts][4 streams][nr][256]
};
__host__ __device__ inline size_t br_roles_lds_bytes(const BrGeom& g, int nr) {
    return (size_t)br_region_bytes(g) + (2 * kBrRows + 2 * 64) * 4 + kBrRows * 16 * 4 + kBrRows * 64 * 4 + (size_t)kBrRows * nr * 1024;
}
__host__ __device__ inline bool br_roles_shape_ok(const BrGeom& g) {
    // (band limit <= 2: with 2B+1 = 7 the ring sums of four sources leave no room for sixteen rows in flight in 168 registers)
    return g.F <= 5 && g.NMT * g.F <= kBrMmaWaves * kBrMaxPairs && kBrRows * g.I <= kBrEntries * kRoleThreads;
}
__device__ __forceinline__ BrRolesLds br_roles_lds(char* smem, const BrArgs& a) {
    BrRolesLds l;
    l.slab = smem;
    l.vinv = reinterpret_cast<float*>(smem + a.region_bytes);
    l.vsc = l.vinv + kBrRows;
    l.cmax = reinterpret_cast<uint32_t*>(l.vsc + kBrRows);
    l.runs = reinterpret_cast<int*>(l.cmax + 2 * 64);
    l.xmag = reinterpret_cast<float*>(l.runs + kBrRows * 16);
    l.ring = l.xmag + kBrRows * 64;
    return l;
}

// Loads whose completion the KERNEL counts (the gathering wavefronts keep sixteen cotangent rows and up to eight record chunks in
// flight across branches and loop iterations; hipcc's own counting gives up at the first control-flow merge and drains
// everything, s_waitcnt vmcnt(0), in front of every use).  hipcc takes the destination for written when the request is made: no
// instruction it generates may read or move those registers before the kernel's own wait -- the build checks the ISA for that
// (tools/check_counted_loads.py).
__device__ __forceinline__ void counted_row_load(f32x2& dst, const float2* __restrict__ base, const uint32_t byte_off) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
// 1 KiB of records, 16 bytes per lane, into LDS at a wave-uniform address; the uniform part of the source in scalar registers
__device__ __forceinline__ void counted_chunk_dma(const float* __restrict__ src_uniform, const uint32_t lane_bytes, const void* lds_dst_uniform) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_uniform);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(lane_bytes), "s"(src_uniform) : "memory", "m0");
}
// at most N loads outstanding; `row` is the value about to be used
template <int N>
__device__ __forceinline__ void counted_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

template <int R, int B>
__global__ __launch_bounds__(kRoleThreads) void fc_backward_roles_data_kernel(
    const float2* __restrict__ gx_, const float2* __restrict__ ggy, const float* __restrict__ grec,
    const int32_t* __restrict__ growptr, const int32_t* __restrict__ gruns, const float* __restrict__ gwpk,
    float2* __restrict__ ggx, char* __restrict__ hdump, const BrArgs a) {
    constexpr int F = 2 * B + 1;
    constexpr int RECF = factored_record_floats(B);
    constexpr int LOG_CR = factored_log_chunk_records(B);
    constexpr int CR = 1 << LOG_CR;
    constexpr int NT = kRoleThreads;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BrGeom& g = a.g;
    const BrRolesLds l = br_roles_lds(smem, a);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_gather = wave < kBrGatherWaves;
    const int mw = wave - kBrGatherWaves;             // contracting wavefront 0..3
    const int nr = a.nr;
    const int I = a.I, O = a.O;
    const int KS = g.KS;

    for (int idx = tid; idx < (int)(a.region_bytes / 16); idx += NT)
        reinterpret_cast<f32x4*>(l.slab)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < kBrRows) { l.vinv[tid] = 0.f; l.vsc[tid] = 1.f; }
    if (tid < 2 * 64) l.cmax[tid] = 0u;
    __syncthreads();

    const int ol = lane < O ? lane : 0;
    Stamper stamp{(a.stamps && blockIdx.x == 0) ? a.stamps + wave * 256 : nullptr, 0};

    // gx epilogue (every thread): entries e = tid + 768 n of the tile's [32][I] block
    const int nent = kBrRows * I;
    int e_id[kBrEntries];
#pragma unroll
    for (int n = 0; n < kBrEntries; ++n) {
        const int e = tid + NT * n;
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

    // ------------------------------------------------------------------------------------------- gathering wavefronts
    auto row_of = [&](const int j) { return wave + kBrGatherWaves * j; };
    auto ring_of = [&](const int j) { return l.ring + (wave * kBrStreams + j) * nr * 256; };
    const uint32_t lane16 = lane * 16;
    int nissued = 0;                                  // loads requested by this (gathering) wavefront
    auto dma_chunk = [&](const int j, const int first, const int ch) {
        const float* src = grec + ((size_t)first + (size_t)ch * CR) * RECF;
        counted_chunk_dma(src, lane16, ring_of(j) + (ch & (nr - 1)) * 256);
        ++nissued;
    };
    auto row_load = [&](f32x2& dst, const int vertex) {
        uint32_t off;                               // vertex * row bytes + lane part, one full-rate instruction (N < 2^24: the plan checks)
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(vertex), "s"(8u * O), "v"(8u * ol));
        counted_row_load(dst, ggy, off);
        ++nissued;
    };
    auto slot_range = [&](const int vt, const int j, const int par, int& b, int& e) {
        b = 0;
        e = 0;
        int run[R];
#pragma unroll
        for (int q = 0; q < R; ++q) run[q] = 0;
        if (vt < a.nv_total) {
            const int t = br_vertex(vt, row_of(j), a.nv_full, a.N);
            if (t < a.N) {
                b = growptr[t];
                e = growptr[t + 1];
#pragma unroll
                for (int q = 0; q < R; ++q) run[q] = gruns[(size_t)t * kRunStride + q];
            }
        }
        if (lane == 0) {
            int* lro = l.runs + (wave * kBrStreams + j) * 16 + par * 8;
#pragma unroll
            for (int q = 0; q < R; ++q) lro[q] = run[q];
        }
    };
    auto rec_ptr = [&](const float* ring, const int s) {
        if constexpr (CR * RECF == 256) return ring + ((s * RECF) & (nr * 256 - 1));
        else return ring + ((s >> LOG_CR) & (nr - 1)) * 256 + (s & (CR - 1)) * RECF;
    };
    // the rows of a stream's first four slots (its first record chunk has landed)
    auto first_rows = [&](const int j, const int nslots, f32x2 (&rows)[4]) {
        const float* ring = ring_of(j);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (a stream without slots requests row 0: its registers are never used)
            const int n = nslots > 0 ? __float_as_int(ring[min(i, nslots - 1) * RECF + 3]) : 0;
            row_load(rows[i], n);
        }
    };
    // ring values c[f] of my stream-j source -> its row of the slab; xmag: |x[row][lane]| (column scales, see fc_backward_ring.hpp)
    auto flush_row = [&](const f32x2 (&c)[F], const int j, const float xmag, const int sp) {
        const int row_i = row_of(j);
        float mx = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) mx = fmaxf(mx, fmaxf(fabsf(c[f].x), fabsf(c[f].y)));
        mx = wave_max_nonneg(mx);
        float scale, inv;
        split_scale(mx, scale, inv);
        if (mx == 0.f) inv = 0.f;                        // an all-zero row drops out of the filter kernel's second operand
        if (lane == 0) { l.vinv[row_i] = inv; l.vsc[row_i] = scale; }
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

    // ------------------------------------------------------------------------------------------- contracting wavefronts
    // pair slot ps: pair id mw + 4 ps -> (i tile it, frequency f); job jb = 2 ps + s: block s of pair slot ps
    const int npairs = g.NMT * F;
    const rsrc_t wimg = make_rsrc(gwpk, a.wpk_bytes);
    const int planes0 = g.IP * 4;
    const int wplane = g.BT * g.IP * 64;
    const int wblk = g.IP * 64;
    auto pair_it = [&](const int ps) { return (mw + kBrMmaWaves * ps) % g.NMT; };
    auto pair_f = [&](const int ps) { return (mw + kBrMmaWaves * ps) / g.NMT; };
    auto pair_on = [&](const int ps) { return mw + kBrMmaWaves * ps < npairs; };
    auto job_on = [&](const int jb) { return jb < kBrMaxJobs && pair_on(jb >> 1) && (jb & 1) < g.nb[pair_f(jb >> 1)]; };
    auto load_job = [&](const int rq, const int jb, u32x4 (&dst)[4]) {
        const int ps = jb >> 1, s = jb & 1;
        const int wv = ((pair_it(ps) * 16 + (lane & 15)) * 32 + 8 * (lane >> 4)) * 2;
        const int base = planes0 + rq * (int)a.ring_bytes_w + wblk * (g.boff[pair_f(ps)] + s);
#pragma unroll
        for (int pnum = 0; pnum < 4; ++pnum) dst[pnum] = buffer_load16(wimg, wv, base + pnum * wplane);
    };
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
    f32x4 tot_re[kBrMaxPairs][2], tot_im[kBrMaxPairs][2];       // (live in the contracting wavefronts' loop only)
    u32x4 wf[3][4];                                              // three filter fragment sets: job jb lives in set jb % 3
    // my pairs' share of slab rq; the fragments of jobs 0..2 were requested before the slab barriers
    auto contract = [&](const int rq, const bool half) {
        if (a.dbg & 2) return;
        f32x4 tre[2], tim[2];
        static_for<0, kBrMaxJobs>([&](auto jc) {
            constexpr int JB = decltype(jc)::value;
            constexpr int PS = JB >> 1, S = JB & 1;
            if (job_on(JB)) {
                const int f = pair_f(PS);
                if (S == 0) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) { tre[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; tim[rt] = tre[rt]; }
                }
                mma_block(0, g.kb0[f] + S, wf[JB % 3], tre[0], tim[0]);
                if (!half) mma_block(1, g.kb0[f] + S, wf[JB % 3], tre[1], tim[1]);
                if (S == 1 || !job_on(JB + 1)) {         // the pair's last block: its product joins the running sum
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const float inv = l.vinv[16 * rt + (lane & 15)];
                        tot_re[PS][rt] += tre[rt] * inv;
                        tot_im[PS][rt] += tim[rt] * inv;
                    }
                }
            }
            if (job_on(JB + 3)) load_job(rq, JB + 3, wf[JB % 3]);
        });
    };
    // the slab (as it lies in LDS, without the row pads) and its scales -> workspace; rows mw, mw + 4, ... per contracting wavefront
    auto keep_slab = [&](const int vt, const int rq, const int sp) {
        char* const dst = hdump + ((size_t)vt * R + rq) * a.hs_bytes;
        if (!(a.dbg & 8)) {
            const int row_pieces = g.KP / 2;
            for (int row = mw; row < kBrRows; row += kBrMmaWaves) {
                const __attribute__((address_space(3))) char* src = (const __attribute__((address_space(3))) char*)l.slab + row * KS * 2;
                char* const drow = dst + (uint32_t)row * (uint32_t)(8 * g.KP);
                for (int pc = lane; pc < row_pieces; pc += kWave) {
                    const u32x4 v = *reinterpret_cast<lds_u32x4*>(src + pc * 16);
                    *reinterpret_cast<u32x4*>(drow + (uint32_t)pc * 16u) = v;
                }
            }
        }
        if (mw == kBrMmaWaves - 1) {
            float* const tail = reinterpret_cast<float*>(dst + (size_t)kBrRows * 8 * g.KP);
            if (lane < kBrRows) { tail[lane] = l.vsc[lane]; tail[kBrRows + lane] = l.vinv[lane]; }
            if (lane < g.IP) {
                const float cm = __uint_as_float(l.cmax[sp * 64 + lane]);
                float t, inv_t;
                split_scale(cm * 1.0000002f, t, inv_t);
                tail[2 * kBrRows + lane] = t;
                tail[2 * kBrRows + g.IP + lane] = inv_t;
                l.cmax[(sp ^ 1) * 64 + lane] = 0u;      // for the next slab (its updates come behind the next "slab free" barrier)
            }
        }
    };

    // the tile's tail, the same for both roles: gx epilogue from the exchange buffer, the region back to a slab
    auto tile_tail = [&](const int vt, const float2 (&exl)[kBrEntries]) {
        if (tid < 2 * 64) l.cmax[tid] = 0u;
#pragma unroll
        for (int n = 0; n < kBrEntries; ++n) {
            if (e_id[n] >= 0) {
                const int e_row = e_id[n] >> 8, e_i = e_id[n] & 255;
                const int vtx = br_vertex(vt, e_row, a.nv_full, a.N);
                const float2 xv = exl[n];
                const float2 u1 = unit_conj(xv);
                const float2 u2 = cmul(u1, u1);
                const float2 u3 = cmul(u2, u1);
                const float inv2 = is_origin(xv) ? 0.f : 1.f / (xv.x * xv.x + xv.y * xv.y);
                const float wk = gwpk[e_i];
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
        __syncthreads();                                 // the exchange buffer is consumed
        {
            const bool next_half = vt + (int)gridDim.x >= a.nv_full;
            const int kpad0 = (F * g.KI) >> 3, kpadn = (g.KP >> 3) - kpad0;
            for (int idx = tid; idx < kBrRows * kpadn * 4; idx += NT) {
                const int row = idx / (kpadn * 4), pc = idx - row * (kpadn * 4);
                *reinterpret_cast<f32x4*>(l.slab + (size_t)row * KS * 2 + kpad0 * 64 + pc * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (next_half) {
                const int row16 = KS * 2 / 16;
                for (int idx = tid; idx < 16 * row16; idx += NT)
                    *reinterpret_cast<f32x4*>(l.slab + (size_t)16 * KS * 2 + (size_t)idx * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                if (tid >= 16 && tid < 32) { l.vinv[tid] = 0.f; l.vsc[tid] = 1.f; }
            }
        }
        stamp(9);
        // (the zeroing is ordered before the next tile's first slab read by that tile's slab barriers)
    };

    stamp.realtime(29);
    stamp(28);
    // Two loops, one per role, with the same sequence of 2R + 3 barriers per tile: the roles' registers never live together
    if (is_gather) {
        int beg[kBrStreams], end[kBrStreams], par = 0;
        f32x2 gq[kBrStreams][4];                        // the four cotangent rows in flight per stream
        int last[kBrStreams];                           // nissued when the stream's last group ended
#pragma unroll
        for (int j = 0; j < kBrStreams; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[j][i] = f32x2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kBrStreams; ++j) {
            slot_range(first_tile_of_block(), j, 0, beg[j], end[j]);
            const int nch = (end[j] - beg[j] + CR - 1) >> LOG_CR;
            for (int ch = 0; ch < min(nch, nr); ++ch) dma_chunk(j, beg[j], ch);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < kBrStreams; ++j) first_rows(j, end[j] - beg[j], gq[j]);
#pragma unroll
        for (int j = 0; j < kBrStreams; ++j) last[j] = nissued;

        for (int vt = first_tile_of_block(); vt < a.nv_total; vt += gridDim.x) {
            const bool half = vt >= a.nv_full;
            const int nrows = half ? 16 : kBrRows;
            stamp(10);
            // (the rows requested at the end of the tile before have landed by now; nothing below may move a register with a
            //  request in flight, and hipcc does not know about them)
            counted_wait<0>();
            int nbeg[kBrStreams], nend[kBrStreams];
#pragma unroll
            for (int j = 0; j < kBrStreams; ++j) slot_range(vt + gridDim.x, j, par ^ 1, nbeg[j], nend[j]);
#pragma unroll
            for (int j = 0; j < kBrStreams; ++j) {      // |x| of my sources' rows for the column scales (read at every flush)
                const int vtx = br_vertex(vt, row_of(j), a.nv_full, a.N);
                float2 xv = make_float2(0.f, 0.f);
                if (vtx < a.N && lane < I) xv = gx_[(size_t)vtx * I + lane];
                l.xmag[row_of(j) * 64 + lane] = sqrtf(xv.x * xv.x + xv.y * xv.y);
            }
            f32x2 clo[kBrStreams][F], chi[kBrStreams][F];
#pragma unroll
            for (int j = 0; j < kBrStreams; ++j)
#pragma unroll
                for (int f = 0; f < F; ++f) { clo[j][f] = f32x2{0.f, 0.f}; chi[j][f] = clo[j][f]; }
            // The walk of one stream.  Row of slot s lives in gq[J][s & 3] and the rows of the four slots from the walk position
            // on are in flight: a slot consumes its row and requests the row of slot s + 4 into the same registers.  A GROUP is
            // the part of an aligned quadruple of slots that belongs to the current ring run: one LDS round trip for its
            // look-ahead vertex ids, then one per slot for its record (the other gathering wavefront of the SIMD fills the wait).
            struct Rec { f32x4 head; f32x2 ph[F]; };
            auto load_rec = [&](const float* ring, const int s_) {
                Rec r;
                const float* rp = rec_ptr(ring, s_);
                r.head = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
                for (int f = 0; f < F; ++f) r.ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                return r;
            };
            auto group = [&](auto jc, const int base, const int s0, const int e0) {
                constexpr int J = decltype(jc)::value;
                const float* const ring = ring_of(J);
                const int nslots = end[J] - beg[J];
                const int nch = (nslots + CR - 1) >> LOG_CR;
                // Loads return in order.  The look-ahead below enters the next record chunk, requested when the walk entered
                // this one, at least CR - 4 row requests ago: it has landed once no more than that many loads are outstanding
                if ((base & (CR - 1)) == CR - 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CR - 4) : "memory");
                int idn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) idn[i] = __float_as_int(rec_ptr(ring, min(base + 4 + i, nslots - 1))[3]);
                const int first = max(base, s0), stop = min(base + 4, e0);
                // The row of a slot has at least `young` requests behind it: the three rows after it of its own stream and
                // whatever the other streams requested since this stream's last group
                const int young = 3 + nissued - last[J];
                static_for<0, 4>([&](auto ic) {
                    constexpr int II = decltype(ic)::value;
                    const int s_ = base + II;
                    if (s_ >= first && s_ < stop) {
#if FC_ROLES_EXP == 2
                        Rec rec;
                        rec.head = f32x4{0.f, 0.5f, 0.25f, 0.f};
#pragma unroll
                        for (int f = 0; f < F; ++f) rec.ph[f] = f32x2{0.5f, 0.25f * f};
#else
                        const Rec rec = load_rec(ring, s_);
#endif
                        if ((s_ & (CR - 1)) == 0 && s_ > 0) {       // a chunk is entered: the one before it makes room
                            const int ch = s_ >> LOG_CR;
                            if (ch - 1 + nr < nch) dma_chunk(J, beg[J], ch - 1 + nr);
                        }
#if FC_ROLES_EXP != 4
                        if (young >= 15) counted_wait<15>();
                        else if (young >= 11) counted_wait<11>();
                        else if (young >= 7) counted_wait<7>();
                        else counted_wait<3>();
#endif
                        // (a real copy into other registers: the request below lands in the row's own registers)
                        f32x2 gv;
                        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(gv.x), "=&v"(gv.y) : "v"(gq[J][II].x), "v"(gq[J][II].y));
#if FC_ROLES_EXP != 1
                        row_load(gq[J][II], idn[II]);
#endif
                        const f32x2 w0v = f32x2{rec.head.y, rec.head.y}, w1v = f32x2{rec.head.z, rec.head.z};
                        f32x2 z[F];
#pragma unroll
                        for (int f = 0; f < F; ++f) z[f] = cmul_conj_pk_step1(gv, rec.ph[f]);
#pragma unroll
                        for (int f = 0; f < F; ++f) z[f] = cmul_conj_pk_step2(gv, rec.ph[f], z[f]);
#if FC_ROLES_EXP == 3
                        clo[J][0] += z[0] + z[F - 1] * w0v;
                        chi[J][0] += z[1] * w1v;
#else
#pragma unroll
                        for (int f = 0; f < F; ++f) clo[J][f] = __builtin_elementwise_fma(w0v, z[f], clo[J][f]);
#pragma unroll
                        for (int f = 0; f < F; ++f) chi[J][f] = __builtin_elementwise_fma(w1v, z[f], chi[J][f]);
#endif
                    }
                });
                last[J] = nissued;
            };
            for (int q = 0; q < R; ++q) {
                if (q < R - 1 && !(a.dbg & 1)) {
                    // ring run q of my streams, a group of each in turn: a stream's rows travel while the others' are used
                    int p[kBrStreams], e[kBrStreams];
#pragma unroll
                    for (int j = 0; j < kBrStreams; ++j) {
                        const int* lro = l.runs + (wave * kBrStreams + j) * 16 + par * 8;
                        p[j] = __builtin_amdgcn_readfirstlane(lro[q]);
                        e[j] = (q + 1 < R - 1) ? __builtin_amdgcn_readfirstlane(lro[q + 1]) : end[j] - beg[j];
                    }
                    bool more = true;
                    while (more) {
                        more = false;
                        static_for<0, kBrStreams>([&](auto jc) {
                            constexpr int J = decltype(jc)::value;
                            if (p[J] < e[J]) {
                                const int base = p[J] & ~3;
                                group(jc, base, p[J], e[J]);
                                p[J] = min(base + 4, e[J]);
                                more = true;
                            }
                            __builtin_amdgcn_sched_barrier(0);       // (keeps the streams' temporaries from piling up)
                        });
                    }
                }
                stamp(0);
                if (q == R - 2) {
                    // my sources are done: start streaming the first record chunks of my next tile's sources
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < kBrStreams; ++j) {
                        const int nnch = (nend[j] - nbeg[j] + CR - 1) >> LOG_CR;
                        for (int ch = 0; ch < min(nnch, nr); ++ch) dma_chunk(j, nbeg[j], ch);
                    }
                }
                __syncthreads();                         // slab free: the contracting wavefronts are done with slab q - 1
                stamp(4);
#pragma unroll
                for (int j = 0; j < kBrStreams; ++j)
                    if (row_of(j) < nrows) flush_row(clo[j], j, l.xmag[row_of(j) * 64 + lane], q & 1);
                if (q == R - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the next tile's first record chunks have landed)
#pragma unroll
                    for (int j = 0; j < kBrStreams; ++j) {
                        beg[j] = nbeg[j];
                        end[j] = nend[j];
                        first_rows(j, end[j] - beg[j], gq[j]);
                    }
#pragma unroll
                    for (int j = 0; j < kBrStreams; ++j) last[j] = nissued;
                }
                stamp(1);
                __syncthreads();                         // slab full
                stamp(2);
#pragma unroll
                for (int j = 0; j < kBrStreams; ++j)
#pragma unroll
                    for (int f = 0; f < F; ++f) { clo[j][f] = chi[j][f]; chi[j][f] = f32x2{0.f, 0.f}; }
            }
            __syncthreads();                             // every read of the slab is done: the region becomes the exchange buffer
            stamp(5);
            counted_wait<0>();
            float2 exl[kBrEntries];
#pragma unroll
            for (int n = 0; n < kBrEntries; ++n) exl[n] = load_entry(vt, n);
            __syncthreads();                             // gxt is in the exchange buffer
            tile_tail(vt, exl);
            par ^= 1;
        }
    } else {
#pragma unroll
        for (int ps = 0; ps < kBrMaxPairs; ++ps)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { tot_re[ps][rt] = f32x4{0.f, 0.f, 0.f, 0.f}; tot_im[ps][rt] = tot_re[ps][rt]; }
        {
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int st = 0; st < 3; ++st)
#pragma unroll
                for (int pnum = 0; pnum < 4; ++pnum) wf[st][pnum] = zero;
        }
        for (int vt = first_tile_of_block(); vt < a.nv_total; vt += gridDim.x) {
            const bool half = vt >= a.nv_full;
            stamp(10);
            for (int q = 0; q < R; ++q) {
                // (the first three blocks' fragments travel during the two barriers)
                if (job_on(0)) load_job(q, 0, wf[0]);
                if (job_on(1)) load_job(q, 1, wf[1]);
                if (job_on(2)) load_job(q, 2, wf[2]);
                __syncthreads();                         // slab free
                stamp(4);
                __syncthreads();                         // slab full
                stamp(2);
                contract(q, half);
                stamp(3);
                keep_slab(vt, q, q & 1);
                stamp(7);
            }
            __syncthreads();                             // every read of the slab is done
            stamp(5);
            float2 exl[kBrEntries];
#pragma unroll
            for (int n = 0; n < kBrEntries; ++n) exl[n] = load_entry(vt, n);
            {
                float2* const xb = reinterpret_cast<float2*>(l.slab);
                const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
                for (int ps = 0; ps < kBrMaxPairs; ++ps) {
                    if (pair_on(ps)) {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            float2* p = xb + (size_t)(16 * rt + fr) * a.xs + pair_f(ps) * g.IP + pair_it(ps) * 16 + 4 * fq;
                            *reinterpret_cast<f32x4*>(p) = f32x4{tot_re[ps][rt][0], tot_im[ps][rt][0], tot_re[ps][rt][1], tot_im[ps][rt][1]};
                            *reinterpret_cast<f32x4*>(p + 2) = f32x4{tot_re[ps][rt][2], tot_im[ps][rt][2], tot_re[ps][rt][3], tot_im[ps][rt][3]};
                            tot_re[ps][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
                            tot_im[ps][rt] = tot_re[ps][rt];
                        }
                    }
                }
            }
            __syncthreads();                             // gxt is in the exchange buffer
            tile_tail(vt, exl);
        }
    }
    stamp(30);
    stamp.realtime(31);
}

}  // namespace fc
