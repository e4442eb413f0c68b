// Microbenchmark: the per-edge walk of the FieldConv gathers in isolation (round 4 go/no-go for the record layout).
//
// One wavefront per vertex walks that vertex's ~30 records (wave-uniform: ring weights, phases, the neighbour), pulls the
// neighbour's feature row (8 B per lane) and accumulates z_f = row * conj(ph_f) into two rings of R x F accumulators --
// what fc_backward_data_kernel's gather does, without slabs, MFMA or barriers.  Variants switch single ingredients off to
// see what the walk is bound by; block sizes of 256 / 512 / 1024 threads give 1 / 2 / 4 wavefronts per SIMD.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o walk walk.hip && ./walk
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../fieldconv_amd/csrc/fc_common.hpp"

using namespace fc;

namespace fc {
// ---- scalar-memory record stream ----
// A walk's per-edge record is wave-uniform data.  Fetched with ONE s_load into SGPRs it costs the walk one instruction and
// the packed FMAs take it as their scalar operand -- against LDS-DMA + ring bookkeeping + several broadcast ds_reads +
// address moves per slot.  Scalar loads return out of order, so only lgkmcnt(0) is meaningful: a walk keeps ONE request in
// flight -- it touches the current record first (hipcc places the wait there), then requests the next one, the two pinned
// in that order with scheduling barriers.  Records are read as FLOAT vectors (hipcc 7.2 resolves every element of a uniform
// uint32 vector to element 0).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// a * b with b in an SGPR pair (a record field)
__device__ __forceinline__ f32x2 cmul_pk_sb(f32x2 a, f32x2 b_uniform) {
    const f32x2 t = f32x2{b_uniform.x, b_uniform.x} * a;
    f32x2 z;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(z) : "s"(b_uniform), "v"(a), "v"(t));
    return z;
}
// a * conj(b) with b in an SGPR pair (a record field): the same two instructions as cmul_conj_pk
__device__ __forceinline__ f32x2 cmul_conj_pk_sb(f32x2 a, f32x2 b_uniform) {
    const f32x2 t = f32x2{b_uniform.x, b_uniform.x} * a;
    f32x2 z;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(z) : "s"(b_uniform), "v"(a), "v"(t));
    return z;
}

// The same with the vertex in an SGPR (a field of a scalar-memory record) and the row size in a VGPR (one VALU instruction
// takes one scalar operand).
__device__ __forceinline__ float2 gather_row_uniform(const float2* __restrict__ base, int vertex_uniform, uint32_t row_bytes_vgpr, uint32_t lane_bytes) {
    uint32_t off;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "s"(vertex_uniform), "v"(row_bytes_vgpr), "v"(lane_bytes));
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + off);
}

}  // namespace fc

constexpr int R = 6, F = 5, RECF = 16;

struct Args {
    const float2* gy;
    const float* rec8b;      // [E][8]: c (re, im), g (re, im), w0, w1, own nbr, pad  -- the layout of the LDS-everything walk
    const float* rec8;       // [E][8]: nbr of slot+4, own nbr, w0, w1, c (re, im), g (re, im)  -- the layout the scalar-memory walk wants
    const float* rec;        // [E][16]: q, w0, w1, nbr, ph[5] (re, im), nbr of slot+2 (float 14), nbr of slot+4 (float 15)
    const int* rowptr;
    const int* runs;         // [N][8]
    float* out;              // [N][64]
    unsigned long long* clk; // [4]: s_memtime / s_memrealtime of block 0 wave 0 at start and end
    int N, C;
};

// V bits: 1 records through scalar memory (else through the LDS ring, as the shipped kernels do)   2 row loads   4 math
//         8 four rows in flight (look-ahead 4; scalar-memory records only)   16 records held in registers (no record fetch at all)
template <int V>
__global__ __launch_bounds__(1024) void walk(const float2* __restrict__ ggy, const float* __restrict__ grec, const float* __restrict__ grec8,
                                             const int* __restrict__ growptr, const int* __restrict__ gruns, const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int O = a.C;
    // lanes beyond the channel count: channel 0 (what the kernels do), or with V bit 512 the LAST channel (the 128-byte line the
    // neighbouring lanes touch anyway), or with bit 1024 consecutive addresses behind the row (a fourth line)
    const int ol = lane < O ? lane : ((V & 512) ? O - 1 : (V & 1024) ? lane : 0);
    uint32_t vrow = 8u * O;
    asm volatile("" : "+v"(vrow));
    float* const ring = reinterpret_cast<float*>(smem) + wave * 4 * 256;      // LDS record ring: 4 chunks of 16 records
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime();
        a.clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    for (int j = blockIdx.x * nw + wave; j < a.N; j += gridDim.x * nw) {
        const int beg = __builtin_amdgcn_readfirstlane(growptr[j]);
        const int nslots = __builtin_amdgcn_readfirstlane(growptr[j + 1]) - beg;
        int ro[R];
#pragma unroll
        for (int q = 0; q < R; ++q) ro[q] = __builtin_amdgcn_readfirstlane(gruns[j * 8 + q]);
        f32x2 h[R][F];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int f = 0; f < F; ++f) h[r][f] = f32x2{0.f, 0.f};

        if constexpr (V & 128) {
            // ---- what a tile-local row cache would give: the tile's distinct source rows staged ONCE in LDS (not timed here:
            // the cache holds whatever it holds), records carry an LDS-local row index, a slot reads its row with one ds_read_b64
            const int rowb = 8 * O;
            char* const recring = smem + wave * 2048;
            char* const rowcache = smem + 16 * 2048;                  // 128 rows shared by the workgroup
            {
                const int nch = (nslots + 31) >> 5;
                for (int ch = 0; ch < min(nch, 2); ++ch)
                    lds_dma16_untracked(a.rec8b + ((size_t)beg + ch * 32) * 8 + lane * 4, recring + ch * 1024);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const uint32_t lane_row = (uint32_t)(uintptr_t)rowcache + lane * 8;
            auto slot = [&](auto qc, const int s_) {
                constexpr int Q = decltype(qc)::value;
                const char* ra = recring + ((s_ * 32) & 2047);
                const f32x4 cg = *reinterpret_cast<const f32x4*>(ra);
                const f32x4 wn = *reinterpret_cast<const f32x4*>(ra + 16);          // w0, w1, row index, pad
                uint32_t rr;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rr) : "v"(__float_as_int(wn.z) & 127), "v"(vrow), "v"(lane_row));
                const f32x2 gv = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(rr);
                if (V & 4) {
                    const f32x2 c = {cg.x, cg.y}, gg = {cg.z, cg.w};
                    f32x2 z[F];
                    z[2] = cmul_conj_pk(gv, c);
                    z[3] = cmul_conj_pk(z[2], gg);
                    z[1] = cmul_pk(z[2], gg);
                    z[4] = cmul_conj_pk(z[3], gg);
                    z[0] = cmul_pk(z[1], gg);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q][f] = __builtin_elementwise_fma(f32x2{wn.x, wn.x}, z[f], h[Q][f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q + 1][f] = __builtin_elementwise_fma(f32x2{wn.y, wn.y}, z[f], h[Q + 1][f]);
                } else {
                    h[Q][0] += gv * f32x2{wn.x, wn.y};
                }
            };
            static_for<0, R - 1>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int s = ro[Q];
                const int stop = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                for (; s + 1 < stop; s += 2) {
                    slot(qc, s);
                    slot(qc, s + 1);
                }
                if (s < stop) {
                    slot(qc, s);
                    ++s;
                }
            });
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if constexpr (V & 64) {
            // ---- candidate: records AND rows through LDS.  Records: 32-byte geometric records, the vertex's chunks of 32 DMA'd
            // into a 64-record ring.  Rows: a ring of 8 rows per wavefront, filled by ONE global_load_lds_dwordx4 per PAIR of
            // slots (lanes [0, C/2) fetch the row of slot 2p, lanes [C/2, C) that of slot 2p+1, 16 bytes each), four pairs in
            // flight; a slot reads its row with one ds_read_b64.  No row registers, no register rotation, waits counted by hand.
            const int rowb = 8 * O;                                   // bytes per row
            char* const recring = smem + wave * 6144;
            char* const rowring = recring + 2048;
            const int hl = O / 2;                                     // lanes per row
            const uint32_t lane_rec = (uint32_t)(uintptr_t)recring + (lane >= hl ? 32u : 0u) + 24u;     // LDS address of "my" nbr field in pair 0
            const uint32_t lane_byte = (uint32_t)((lane >= hl ? lane - hl : lane) * 16);
            const unsigned long long pairmask = (2 * hl >= 64) ? ~0ull : ((1ull << (2 * hl)) - 1);
            const uint32_t lane_row = (uint32_t)(uintptr_t)rowring + lane * 8;
            const int P = (nslots + 1) >> 1;
            auto issue_pair = [&](const int p) {
                const uint32_t ra = lane_rec + (uint32_t)((p * 64) & 2047);
                const int nb = *reinterpret_cast<const __attribute__((address_space(3))) int*>(ra);
                uint32_t off;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(nb), "v"(vrow), "v"(lane_byte));
                const uint32_t dst = (uint32_t)(uintptr_t)rowring + (uint32_t)((p & 3) * 2 * rowb);
                asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                             : : "s"(dst), "v"(off), "s"(ggy), "s"(pairmask) : "memory", "m0");
            };
            {
                const int nch = (nslots + 31) >> 5;
                for (int ch = 0; ch < min(nch, 2); ++ch)
                    lds_dma16_untracked(a.rec8b + ((size_t)beg + ch * 32) * 8 + lane * 4, recring + ch * 1024);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int p = 0; p < min(P, 4); ++p) issue_pair(p);
            }
            auto slot = [&](auto qc, const int s_) {
                constexpr int Q = decltype(qc)::value;
                const char* ra = recring + ((s_ * 32) & 2047);
                const f32x4 cg = *reinterpret_cast<const f32x4*>(ra);
                const f32x2 w = *reinterpret_cast<const f32x2*>(ra + 16);
                const uint32_t rr = lane_row + (uint32_t)((s_ & 7) * rowb);
                const f32x2 gv = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(rr);
                if (V & 4) {
                    const f32x2 c = {cg.x, cg.y}, gg = {cg.z, cg.w};
                    f32x2 z[F];
                    z[2] = cmul_conj_pk(gv, c);
                    z[3] = cmul_conj_pk(z[2], gg);
                    z[1] = cmul_pk(z[2], gg);
                    z[4] = cmul_conj_pk(z[3], gg);
                    z[0] = cmul_pk(z[1], gg);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q][f] = __builtin_elementwise_fma(f32x2{w.x, w.x}, z[f], h[Q][f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q + 1][f] = __builtin_elementwise_fma(f32x2{w.y, w.y}, z[f], h[Q + 1][f]);
                } else {
                    h[Q][0] += gv * w;
                }
            };
            auto even_wait = [&](const int s_) {           // before an even slot: its pair's rows have landed
                if (P - 1 - (s_ >> 1) >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            };
            auto odd_done = [&](const int s_) {            // behind an odd slot: its pair's ring slot is refilled four pairs ahead
                const int pn = (s_ >> 1) + 4;
                if (pn < P) issue_pair(pn);
            };
            static_for<0, R - 1>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int s = ro[Q];
                const int stop = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                if ((s & 1) && s < stop) {
                    slot(qc, s);
                    odd_done(s);
                    ++s;
                }
                for (; s + 1 < stop; s += 2) {
                    even_wait(s);
                    slot(qc, s);
                    slot(qc, s + 1);
                    odd_done(s + 1);
                }
                if (s < stop) {
                    even_wait(s);
                    slot(qc, s);
                    ++s;
                }
            });
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if constexpr (V & 32) {
            // ---- candidate: 32-byte geometric records through scalar memory, one s_load_dwordx16 per PAIR of slots (PA: slots
            // 0,1 mod 4, PB: slots 2,3 mod 4; requested two slots before their first use), four row registers (slot s reads
            // r[s & 3] and requests the row of slot s + 4 into the same register when it is done with it)
            const float* const rp = grec8 + (size_t)beg * 8;
            uint32_t po = 0;                                        // float offset of the pair (slots 4m, 4m+1)
            f32x16 PA = {}, PB = {};
            float2 r0 = make_float2(1.f, 2.f), r1 = r0, r2 = r0, r3 = r0;
            if (nslots > 0) {
                PA = *reinterpret_cast<const f32x16*>(rp);
                if (V & 2) {
                    r0 = gather_row(ggy, __float_as_int(rp[1]), 8u * O, 8u * ol);
                    r1 = gather_row(ggy, __float_as_int(rp[min(1, nslots - 1) * 8 + 1]), 8u * O, 8u * ol);
                    r2 = gather_row(ggy, __float_as_int(rp[min(2, nslots - 1) * 8 + 1]), 8u * O, 8u * ol);
                    r3 = gather_row(ggy, __float_as_int(rp[min(3, nslots - 1) * 8 + 1]), 8u * O, 8u * ol);
                }
            }
            auto slot = [&](auto qc, auto kc, float2& row) {
                constexpr int Q = decltype(qc)::value;
                constexpr int K = decltype(kc)::value;
                constexpr int b = 8 * (K & 1);
                const f32x16& P = (K < 2) ? PA : PB;
                __builtin_amdgcn_sched_barrier(0);
                const f32x2 gv = f32x2{row.x, row.y};
                // t = gy conj(c): the first use of this pair's registers (the wait for them lands here)
                const f32x2 t = cmul_conj_pk_sb(gv, f32x2{P[b + 4], P[b + 5]});
                if constexpr (K == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    PB = *reinterpret_cast<const f32x16*>(rp + po + 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (K == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    po += 32;
                    PA = *reinterpret_cast<const f32x16*>(rp + po);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (V & 4) {
                    const f32x2 gg = f32x2{P[b + 6], P[b + 7]};
                    f32x2 z[F];
                    z[2] = t;
                    z[3] = cmul_conj_pk_sb(t, gg);
                    z[1] = cmul_pk_sb(t, gg);
                    z[4] = cmul_conj_pk_sb(z[3], gg);
                    z[0] = cmul_pk_sb(z[1], gg);
                    const float w0 = P[b + 2], w1 = P[b + 3];
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q][f] = __builtin_elementwise_fma(f32x2{w0, w0}, z[f], h[Q][f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q + 1][f] = __builtin_elementwise_fma(f32x2{w1, w1}, z[f], h[Q + 1][f]);
                } else {
                    h[Q][0] += t;
                }
                if (V & 2) row = gather_row_uniform(ggy, __float_as_int(P[b]), vrow, 8u * ol);      // the row of slot + 4
            };
            using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
            static_for<0, R - 1>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int s = ro[Q];
                const int stop = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                while (s < stop && (s & 3)) {
                    if ((s & 3) == 1) slot(qc, K1{}, r1);
                    else if ((s & 3) == 2) slot(qc, K2{}, r2);
                    else slot(qc, K3{}, r3);
                    ++s;
                }
                for (; s + 3 < stop; s += 4) {
                    slot(qc, K0{}, r0);
                    slot(qc, K1{}, r1);
                    slot(qc, K2{}, r2);
                    slot(qc, K3{}, r3);
                }
                if (s < stop) { slot(qc, K0{}, r0); ++s; }
                if (s < stop) { slot(qc, K1{}, r1); ++s; }
                if (s < stop) { slot(qc, K2{}, r2); ++s; }
            });
        } else if constexpr (V & 1) {
            const float* rp = grec + (size_t)beg * RECF;
            float2 ga = make_float2(1.f, 2.f), gb = ga, gc = ga, gd = ga;
            f32x16 recA = {}, recB = {};
            if (nslots > 0) {
                recA = *reinterpret_cast<const f32x16*>(rp);
                if (V & 2) {
                    ga = gather_row(ggy, __float_as_int(recA[3]), 8u * O, 8u * ol);
                    gb = gather_row(ggy, __float_as_int(rp[min(1, nslots - 1) * RECF + 3]), 8u * O, 8u * ol);
                    if (V & 8) {
                        gc = gather_row(ggy, __float_as_int(rp[min(2, nslots - 1) * RECF + 3]), 8u * O, 8u * ol);
                        gd = gather_row(ggy, __float_as_int(rp[min(3, nslots - 1) * RECF + 3]), 8u * O, 8u * ol);
                    }
                }
            }
            auto slot = [&](auto qc, const f32x16& cur, f32x16& nxt, float2& gcur) {
                constexpr int Q = decltype(qc)::value;
                __builtin_amdgcn_sched_barrier(0);
                const f32x2 gv = f32x2{gcur.x, gcur.y};
                if (V & 2) gcur = gather_row_uniform(ggy, __float_as_int(cur[(V & 8) ? 15 : 14]), vrow, 8u * ol);
                __builtin_amdgcn_sched_barrier(0);
                rp += RECF;
                nxt = *reinterpret_cast<const f32x16*>(rp);
                __builtin_amdgcn_sched_barrier(0);
                if (V & 4) {
                    const float w0 = cur[1], w1 = cur[2];
                    f32x2 z[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) z[f] = cmul_conj_pk_sb(gv, f32x2{cur[4 + 2 * f], cur[5 + 2 * f]});
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q][f] = __builtin_elementwise_fma(f32x2{w0, w0}, z[f], h[Q][f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q + 1][f] = __builtin_elementwise_fma(f32x2{w1, w1}, z[f], h[Q + 1][f]);
                } else {
                    h[Q][0] += gv * f32x2{cur[1], cur[2]};
                }
            };
            static_for<0, R - 1>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int s = ro[Q];
                const int stop = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                if constexpr (V & 8) {
                    // four rows in flight: slot s uses row register s & 3 and record register s & 1
                    while (s < stop && (s & 3)) {
                        if ((s & 3) == 1) slot(qc, recB, recA, gb);
                        else if ((s & 3) == 2) slot(qc, recA, recB, gc);
                        else slot(qc, recB, recA, gd);
                        ++s;
                    }
                    for (; s + 3 < stop; s += 4) {
                        slot(qc, recA, recB, ga);
                        slot(qc, recB, recA, gb);
                        slot(qc, recA, recB, gc);
                        slot(qc, recB, recA, gd);
                    }
                    if (s < stop) { slot(qc, recA, recB, ga); ++s; }
                    if (s < stop) { slot(qc, recB, recA, gb); ++s; }
                    if (s < stop) { slot(qc, recA, recB, gc); ++s; }
                } else {
                    if ((s & 1) && s < stop) {
                        slot(qc, recB, recA, gb);
                        ++s;
                    }
                    for (; s + 1 < stop; s += 2) {
                        slot(qc, recA, recB, ga);
                        slot(qc, recB, recA, gb);
                    }
                    if (s < stop) {
                        slot(qc, recA, recB, ga);
                        ++s;
                    }
                }
            });
        } else if constexpr (V & 16) {
            // no record fetch at all: the record is a set of registers, the neighbour a running index
            float2 ga = make_float2(1.f, 2.f), gb = ga;
            const float w0 = 0.25f + j * 1e-9f, w1 = 0.75f;
            int nb = j;
            auto slot = [&](auto qc, float2& gcur) {
                constexpr int Q = decltype(qc)::value;
                const f32x2 gv = f32x2{gcur.x, gcur.y};
                nb = nb + 7 < a.N ? nb + 7 : 0;
                if (V & 2) gcur = gather_row_uniform(ggy, nb, vrow, 8u * ol);
                if (V & 4) {
                    f32x2 z[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) z[f] = cmul_conj_pk(gv, f32x2{w0 + f, w1});
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q][f] = __builtin_elementwise_fma(f32x2{w0, w0}, z[f], h[Q][f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q + 1][f] = __builtin_elementwise_fma(f32x2{w1, w1}, z[f], h[Q + 1][f]);
                } else {
                    h[Q][0] += gv;
                }
            };
            static_for<0, R - 1>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int s = ro[Q];
                const int stop = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                if ((s & 1) && s < stop) { slot(qc, gb); ++s; }
                for (; s + 1 < stop; s += 2) { slot(qc, ga); slot(qc, gb); }
                if (s < stop) { slot(qc, ga); ++s; }
            });
        } else {
            // the shipped walk: records DMA'd into an LDS ring of 4 chunks x 16 records, read back with broadcast ds_reads
            constexpr int LOG_CR = 4, CR = 16, NR = 4;
            const int nch = (nslots + CR - 1) >> LOG_CR;
            auto dma_chunk = [&](const int ch) {
                const float* src = grec + ((size_t)beg + ((size_t)ch << LOG_CR)) * RECF + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (ch & (NR - 1)) * 256), 16, 0, 0);
            };
            for (int ch = 0; ch < min(nch, NR); ++ch) dma_chunk(ch);
            auto rec_ptr = [&](const int s) { return ring + ((s * RECF) & (NR * 256 - 1)); };
            float2 ga = make_float2(1.f, 2.f), gb = ga;
            if (nslots > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (V & 2) {
                    ga = gather_row(ggy, __float_as_int(rec_ptr(0)[3]), 8u * O, 8u * ol);
                    gb = gather_row(ggy, __float_as_int(rec_ptr(min(1, nslots - 1))[3]), 8u * O, 8u * ol);
                }
            }
            const unsigned long long cmask = O >= 64 ? ~0ull : ((1ull << O) - 1);
            auto slot = [&](auto qc, const int s, float2& gcur) {
                constexpr int Q = decltype(qc)::value;
                const float* rp = rec_ptr(s);
                const f32x4 head = *reinterpret_cast<const f32x4*>(rp);
                const int d2 = __float_as_int(rec_ptr(min(s + 2, nslots - 1))[3]);
                const f32x2 gv = f32x2{gcur.x, gcur.y};
                if (V & 2) gcur = gather_row(ggy, d2, 8u * O, 8u * ol);
                const f32x2 w0v = f32x2{head.y, head.y}, w1v = f32x2{head.z, head.z};
                if (V & 4) {
                    f32x2 ph[F], z[F];
                    if (V & 256) {
#pragma unroll
                        for (int f = 0; f < F; ++f) ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_mov_b64 exec, %0" : : "s"(cmask) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        ph[f] = *reinterpret_cast<const f32x2*>(rp + 4 + 2 * f);
                        z[f] = cmul_conj_pk_step1(gv, ph[f]);
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) z[f] = cmul_conj_pk_step2(gv, ph[f], z[f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q][f] = __builtin_elementwise_fma(w0v, z[f], h[Q][f]);
#pragma unroll
                    for (int f = 0; f < F; ++f) h[Q + 1][f] = __builtin_elementwise_fma(w1v, z[f], h[Q + 1][f]);
                    if (V & 256) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_mov_b64 exec, -1" : : : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    h[Q][0] += gv * w0v;
                }
            };
            static_for<0, R - 1>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int s = ro[Q];
                const int run_end = (Q + 1 < R - 1) ? ro[Q + 1] : nslots;
                while (s < run_end) {
                    const int m = s & (CR - 1);
                    if (m == 0 && s > 0) {
                        const int ch = s >> LOG_CR;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (ch - 1 + NR < nch) dma_chunk(ch - 1 + NR);
                    }
                    const int stop = min(run_end, s - m + CR);
                    if ((s & 1) && s < stop) { slot(qc, s, gb); ++s; }
                    for (; s + 1 < stop; s += 2) { slot(qc, s, ga); slot(qc, s + 1, gb); }
                    if (s < stop) { slot(qc, s, ga); ++s; }
                }
            });
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        // sink
        f32x2 t = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int f = 0; f < F; ++f) t += h[r][f];
        a.out[(size_t)j * 64 + lane] = t.x + t.y;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.clk[2] = __builtin_amdgcn_s_memtime();
        a.clk[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int V>
static void run(const char* name, const Args& a, long E, int threads, int lds_bytes, int grid = 256) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(walk<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(walk<V>, dim3(grid), dim3(threads), lds_bytes, 0, a.gy, a.rec, a.rec8, a.rowptr, a.runs, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(walk<V>, dim3(grid), dim3(threads), lds_bytes, 0, a.gy, a.rec, a.rec8, a.rowptr, a.runs, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[4];
    hipMemcpy(clk, a.clk, sizeof(clk), hipMemcpyDeviceToHost);
    const double ghz = (double)(clk[2] - clk[0]) / ((double)(clk[3] - clk[1]) * 10.0);      // shader cycles per ns (100 MHz reference)
    const double us = ms * 1e3 / reps;
    const double slots_per_simd = (double)E / 1024.0;
    float chk = 0.f;
    hipMemcpy(&chk, a.out + 64 * 100 + 5, sizeof(float), hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD=%d  %7.1f us  %.2f GHz  %6.1f cycles/slot/SIMD   (out %.4g)\n", name, threads / 256 * grid / 256, us, ghz,
           us * 1e3 * ghz / slots_per_simd, chk);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}

int main(int argc, char** argv) {
    const int N = 20000, C = argc > 2 ? atoi(argv[2]) : 48;
    // neighbour window: the neighbours of the 16 vertices of a tile are drawn from W rows around the tile (801: the rows of a tile
    // are all distinct, what consecutive indices of a Fibonacci-lattice sphere give; ~100: a spatially compact tile on a surface)
    const int W = argc > 1 ? atoi(argv[1]) : 801;
    srand(1);
    std::vector<int> rowptr(N + 1), runs((size_t)N * 8);
    std::vector<float> rec, rec8, rec8b;
    rowptr[0] = 0;
    std::vector<int> nbrs;
    for (int j = 0; j < N; ++j) {
        const int deg = 24 + rand() % 13;
        rowptr[j + 1] = rowptr[j] + deg;
        // ring of every slot, sorted
        std::vector<int> q(deg);
        for (int s = 0; s < deg; ++s) q[s] = rand() % (R - 1);
        std::sort(q.begin(), q.end());
        for (int r = 0; r < 8; ++r) {
            int first = deg;
            for (int s = deg - 1; s >= 0; --s) if (q[s] >= r) first = s;
            runs[(size_t)j * 8 + r] = first;
        }
        runs[(size_t)j * 8] = 0;
        std::vector<int> nb(deg);
        for (int s = 0; s < deg; ++s) nb[s] = (((j & ~15) + 8 + (rand() % W) - W / 2) % N + N) % N;
        for (int s = 0; s < deg; ++s) {
            float r16[16];
            *reinterpret_cast<int*>(&r16[0]) = q[s];
            r16[1] = 0.3f; r16[2] = 0.7f;
            *reinterpret_cast<int*>(&r16[3]) = nb[s];
            for (int f = 0; f < 5; ++f) { r16[4 + 2 * f] = 0.1f * (f + 1); r16[5 + 2 * f] = 0.05f * f; }
            *reinterpret_cast<int*>(&r16[14]) = nb[std::min(s + 2, deg - 1)];
            *reinterpret_cast<int*>(&r16[15]) = nb[std::min(s + 4, deg - 1)];
            rec.insert(rec.end(), r16, r16 + 16);
            float r8[8];
            *reinterpret_cast<int*>(&r8[0]) = nb[std::min(s + 4, deg - 1)];
            *reinterpret_cast<int*>(&r8[1]) = nb[s];
            r8[2] = 0.3f; r8[3] = 0.7f; r8[4] = 0.1f; r8[5] = 0.02f; r8[6] = 0.8f; r8[7] = 0.6f;
            rec8.insert(rec8.end(), r8, r8 + 8);
            float r8b[8] = {0.1f, 0.02f, 0.8f, 0.6f, 0.3f, 0.7f, 0.f, 0.f};
            *reinterpret_cast<int*>(&r8b[6]) = nb[s];
            rec8b.insert(rec8b.end(), r8b, r8b + 8);
        }
    }
    const long E = rowptr[N];
    rec.resize(rec.size() + 4096, 0.f);
    rec8.resize(rec8.size() + 4096, 0.f);
    rec8b.resize(rec8b.size() + 4096, 0.f);
    std::vector<float> gy((size_t)N * C * 2);
    for (auto& v : gy) v = (rand() % 2001 - 1000) * 1e-3f;
    Args a;
    float *dgy, *drec, *drec8, *drec8b, *dout;
    int *drow, *druns;
    unsigned long long* dclk;
    hipMalloc(&dgy, gy.size() * 4); hipMemcpy(dgy, gy.data(), gy.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&drec, rec.size() * 4); hipMemcpy(drec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&drec8, rec8.size() * 4); hipMemcpy(drec8, rec8.data(), rec8.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&drec8b, rec8b.size() * 4); hipMemcpy(drec8b, rec8b.data(), rec8b.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&drow, rowptr.size() * 4); hipMemcpy(drow, rowptr.data(), rowptr.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&druns, runs.size() * 4); hipMemcpy(druns, runs.data(), runs.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, (size_t)N * 64 * 4);
    hipMalloc(&dclk, 4 * 8);
    a.gy = reinterpret_cast<const float2*>(dgy); a.rec = drec; a.rec8 = drec8; a.rec8b = drec8b; a.rowptr = drow; a.runs = druns; a.out = dout; a.clk = dclk; a.N = N; a.C = C;
    printf("walk microbenchmark: N=%d, E=%ld (%.1f slots per vertex), C=%d, neighbour window %d rows per tile\n", N, E, (double)E / N, C, W);
    // the clock governor needs ~100 ms of load to settle: run the shipped walk for ~0.4 s first, and compare variants
    // by cycles (in-kernel clock) rather than by microseconds -- a variant's clock depends on what ran before it
    for (int i = 0; i < 6000; ++i) hipLaunchKernelGGL(walk<0 | 2 | 4>, dim3(256), dim3(1024), 100 * 1024, 0, a.gy, a.rec, a.rec8, a.rowptr, a.runs, a);
    hipDeviceSynchronize();
    for (int threads : {1024, 1024, 512, 256}) {
        const int lds = 100 * 1024;        // one workgroup per CU
        run<0 | 2 | 4>("LDS ring, rows, math (shipped walk)", a, E, threads, lds);
        run<0 | 2 | 4 | 256>("shipped walk, math under an exec mask of C lanes", a, E, threads, lds);
        run<128 | 2 | 4>("LDS geo records + rows from an LDS row cache, math", a, E, threads, lds);
        run<128 | 2>("LDS geo records + rows from an LDS row cache, no math", a, E, threads, lds);
        run<64 | 2 | 4>("LDS geo records + LDS row ring (DMA pairs), math", a, E, threads, lds);
        run<64 | 2>("LDS geo records + LDS row ring, no math", a, E, threads, lds);
        run<32 | 2 | 4>("pairs of scalar geo records, 4 rows, math", a, E, threads, lds);
        run<32 | 4>("pairs of scalar geo records, no rows, math", a, E, threads, lds);
        run<32 | 2>("pairs of scalar geo records, 4 rows, no math", a, E, threads, lds);
        run<1 | 2 | 4>("scalar records, rows, math", a, E, threads, lds);
        run<1 | 2 | 4 | 8>("scalar records, 4 rows in flight, math", a, E, threads, lds);
        run<1 | 4>("scalar records, no rows, math", a, E, threads, lds);
        run<1 | 2>("scalar records, rows, no math", a, E, threads, lds);
        run<16 | 2 | 4>("no records, rows, math", a, E, threads, lds);
        run<16 | 4>("no records, no rows, math only", a, E, threads, lds);
        run<16 | 2>("no records, rows only", a, E, threads, lds);
        run<16 | 2 | 512>("rows only, idle lanes read the last channel", a, E, threads, lds);
        run<16 | 2 | 1024>("rows only, idle lanes read on behind the row", a, E, threads, lds);
        run<0 | 2 | 4 | 512>("shipped walk, idle lanes read the last channel", a, E, threads, lds);
    }
    // two workgroups of 1024 per CU is impossible at 128 registers; 2 x 512 gives the same 4 waves/SIMD with independent workgroups
    run<1 | 2 | 4>("scalar records, rows, math, 2 WG x 512 / CU", a, E, 512, 60 * 1024, 512);
    run<0 | 2 | 4>("LDS ring, rows, math, 2 WG x 512 / CU", a, E, 512, 60 * 1024, 512);
    run<32 | 2 | 4>("pairs of scalar geo records, 4 rows, math, 2 WG x 512", a, E, 512, 60 * 1024, 512);
    return 0;
}
