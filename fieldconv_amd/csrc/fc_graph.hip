// Support-graph build for the kernels of this library: from the operator's inputs
//   supp_edges (E,2) int64 (col 0 = source, col 1 = target; reference nn/field_conv.py:104-121) and
//   supp_sten (E,R,F) complex64 (reference transforms/fc_precomp.py:95)
// to the edges grouped by target and by source (rowptr / nbr / ring-run offsets, include/fieldconv_hip.h: fc_csr), the
// factored per-edge records in slot order and the geometric-phase records of the forward pass.
// The torch version of this (fieldconv_amd/graph.py) is ~100 small launches and two host synchronisations, 1.5-2.5 ms
// per mesh -- as long as a whole training step of the segmentation network, and paid on every step of an epoch over
// different meshes.  Here: one analysis kernel per edge (factorisation + verification + keys + bucket counts), one
// kernel per vertex (ring-run offsets, degrees), two scans (rocPRIM through hipCUB: plumbing), a counting sort (below) and
// one placement kernel per side.  Slots are ordered by (vertex, lower ring q, original edge index).  That order used to
// come from two stable radix sorts of E (key, edge) pairs -- 20 merge passes each at E = 600 k, 240 us of the 510 us
// a config-2 mesh took -- although the bucket counts the analysis kernel takes anyway already say where every
// (vertex, ring) run starts: the atomic that counts an edge into its bucket returns its arrival index, which scatters
// the edge into its run (graph_scatter_kernel), and the runs -- a handful of edges each -- are then put into ascending
// edge order by rank (graph_order_kernel).  Integer atomics decide only the arrival order, which the ranking removes:
// the result is deterministic and identical to the sorted one.
#include <hipcub/hipcub.hpp>
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kGraphMaxR = 8;
constexpr int kGraphMaxF = 7;
constexpr float kGraphTol = 2e-6f;          // same acceptance threshold as graph.py: factor_stencil / geometric_phases

struct GraphArgs {
    int N, E, R, F, recf;
    int pad_rec, pad_geo;      // rows behind the E-th record that graph_place_kernel zero-fills (the kernels stream past the end)
};

__device__ __forceinline__ float cabs2(float2 z) { return z.x * z.x + z.y * z.y; }

// ---- per edge: factorisation of the stencil row, its verification, sort keys and bucket counts
// A workgroup takes kAnalyzeEdges consecutive edges; their stencil rows are one contiguous piece of memory, staged
// through LDS with coalesced 8-byte loads (a thread reading its own 240-byte row straight from memory runs at a
// seventh of the bandwidth), rows padded by one complex number against bank conflicts.
constexpr int kAnalyzeEdges = 128;
__global__ __launch_bounds__(kAnalyzeEdges) void graph_analyze_kernel(
    const int64_t* __restrict__ edges, const float2* __restrict__ sten, float* __restrict__ rec, float* __restrict__ geo,
    uint32_t* __restrict__ key_t, uint32_t* __restrict__ key_s, uint32_t* __restrict__ arr_t, uint32_t* __restrict__ arr_s,
    int32_t* __restrict__ cnt_t, int32_t* __restrict__ cnt_s, int32_t* __restrict__ flags, const GraphArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* const rows = reinterpret_cast<float2*>(smem);              // [kAnalyzeEdges][R*F + 1]
    const int R = a.R, F = a.F, B = (F - 1) / 2;
    const int e0 = blockIdx.x * kAnalyzeEdges;
    const int e = e0 + threadIdx.x;
    if (sten) {
        const int RF = R * F, stride = RF + 1;
        const int count = min(kAnalyzeEdges, a.E - e0) * RF;             // complex numbers of this workgroup's rows
        const float2* src = sten + (size_t)e0 * RF;
        for (int idx = threadIdx.x; idx < count; idx += kAnalyzeEdges) {
            const int r = idx / RF;
            rows[r * stride + (idx - r * RF)] = src[idx];
        }
        __syncthreads();
    }
    if (e >= a.E) return;
    int64_t src = edges[2 * (size_t)e], dst = edges[2 * (size_t)e + 1];
    if (src < 0 || src >= a.N || dst < 0 || dst >= a.N) {
        atomicOr(flags, 4);
        src = min(max(src, (int64_t)0), (int64_t)a.N - 1);
        dst = min(max(dst, (int64_t)0), (int64_t)a.N - 1);
    }
    int q = 0;
    if (sten) {
        const float2* row = rows + threadIdx.x * (R * F + 1);
        // ring magnitudes: first non-zero ring, nothing outside {q, q+1}
        float mag[kGraphMaxR];
        float scale = 0.f;
#pragma unroll
        for (int r = 0; r < kGraphMaxR; ++r) {
            mag[r] = 0.f;
            if (r < R) {
                float m2 = 0.f;
                for (int f = 0; f < F; ++f) m2 = fmaxf(m2, cabs2(row[r * F + f]));
                mag[r] = m2;
                scale = fmaxf(scale, m2);
            }
        }
        scale = sqrtf(scale);
        int first = 0;
        bool found = false;
#pragma unroll
        for (int r = 0; r < kGraphMaxR; ++r)
            if (r < R && !found && mag[r] > 0.f) { first = r; found = true; }
        q = min(first, R - 2);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < kGraphMaxR; ++r)
            if (r < R && mag[r] > 0.f && (r < q || r > q + 1)) bad = true;
        float2 s0[kGraphMaxF], s1[kGraphMaxF], ph[kGraphMaxF];
        float den = 0.f, n0 = 0.f, n1 = 0.f;
#pragma unroll
        for (int f = 0; f < kGraphMaxF; ++f)
            if (f < F) {
                s0[f] = row[q * F + f];
                s1[f] = row[(q + 1) * F + f];
                ph[f] = make_float2(s0[f].x + s1[f].x, s0[f].y + s1[f].y);       // = phase * (w_q + w_{q+1})
                den += cabs2(ph[f]);
                n0 += s0[f].x * ph[f].x + s0[f].y * ph[f].y;                      // Re(s0 conj(ph))
                n1 += s1[f].x * ph[f].x + s1[f].y * ph[f].y;
            }
        const float w0 = den > 0.f ? n0 / den : 0.f, w1 = den > 0.f ? n1 / den : 0.f;
        float err2 = 0.f;
#pragma unroll
        for (int f = 0; f < kGraphMaxF; ++f)
            if (f < F) {
                err2 = fmaxf(err2, cabs2(make_float2(s0[f].x - w0 * ph[f].x, s0[f].y - w0 * ph[f].y)));
                err2 = fmaxf(err2, cabs2(make_float2(s1[f].x - w1 * ph[f].x, s1[f].y - w1 * ph[f].y)));
            }
        if (bad || sqrtf(err2) > kGraphTol * scale) atomicOr(flags, 1);
        float* rp = rec + (size_t)e * a.recf;
        rp[0] = __int_as_float(q);
        rp[1] = w0;
        rp[2] = w1;
        rp[3] = 0.f;
#pragma unroll
        for (int f = 0; f < kGraphMaxF; ++f)
            if (f < F) { rp[4 + 2 * f] = ph[f].x; rp[5 + 2 * f] = ph[f].y; }
        for (int k = 4 + 2 * F; k < a.recf; ++k) rp[k] = 0.f;
        // geometric phases: ph[f] = c g^(f-B), |g| = 1
        if (geo) {
            bool badgeo = B < 1;
            float2 c = make_float2(0.f, 0.f), g = make_float2(1.f, 0.f);
            if (B >= 1) {
                c = ph[B];
                const float cm2 = cabs2(c);
                const bool live = cm2 > 0.f;
                if (live) {
                    const float2 n = ph[B + 1];
                    g = make_float2((n.x * c.x + n.y * c.y) / cm2, (n.y * c.x - n.x * c.y) / cm2);      // ph[B+1] / c
                }
                float pscale = 0.f;
#pragma unroll
                for (int f = 0; f < kGraphMaxF; ++f)
                    if (f < F) pscale = fmaxf(pscale, cabs2(ph[f]));
                pscale = sqrtf(pscale);
                float err = fabsf(sqrtf(cabs2(g)) - 1.f) * sqrtf(cm2);
                float2 p = c, pc = c;
#pragma unroll
                for (int m = 1; m <= (kGraphMaxF - 1) / 2; ++m)
                    if (m <= B) {
                        p = cmul(p, g);
                        pc = cmul_conj(pc, g);
                        const float2 up = ph[B + m], dn = ph[B - m];
                        err = fmaxf(err, sqrtf(cabs2(make_float2(up.x - p.x, up.y - p.y))));
                        err = fmaxf(err, sqrtf(cabs2(make_float2(dn.x - pc.x, dn.y - pc.y))));
                    }
                badgeo = err > kGraphTol * pscale || (!live && pscale > 0.f);
            }
            if (badgeo) atomicOr(flags, 2);
            float4* gp = reinterpret_cast<float4*>(geo + (size_t)e * 8);
            gp[0] = make_float4(__int_as_float(q), w0, w1, 0.f);
            gp[1] = make_float4(c.x, c.y, g.x, g.y);
        }
    }
    const uint32_t kt = (uint32_t)dst * 8u + (uint32_t)q, ks = (uint32_t)src * 8u + (uint32_t)q;
    key_t[e] = kt;
    key_s[e] = ks;
    arr_t[e] = (uint32_t)atomicAdd(cnt_t + kt, 1);          // arrival index inside the (vertex, ring) run
    arr_s[e] = (uint32_t)atomicAdd(cnt_s + ks, 1);
}

// ---- per input edge, straight from FCPrecomp's inputs (reference transforms/fc_precomp.py:53-97): what graph_analyze_kernel
// recovers from a materialised (E,R,F) stencil is known in closed form here -- lower ring q, the two interpolation weights,
// c = wxp, g = e^{i theta} -- so the dense stencil is never written or read.  Kept edges (r <= epsilon) keep their order:
// slot = pos[e].  Also emits what the module API returns per edge: supp_edges, ln, wxp, and the factor table
// [q bits, w_q, w_{q+1}, 0, Re c, Im c, Re g, Im g] in that order (FactoredStencil materialises from it on demand).
struct FactorArgs {
    const float* log_mag;
    const float* log_ang;
    const float2* xp;
    const float* w;
    const int64_t* edges_in;
    const int32_t* keep;
    const int32_t* pos;
    const float* total;
    float eps;
    int E_in;
    int64_t* edges_out;
    float2* ln;
    float2* wxp;
    float* factors;
};

__global__ void graph_factor_kernel(const FactorArgs p, float* __restrict__ rec, float* __restrict__ geo, uint32_t* __restrict__ key_t,
                                    uint32_t* __restrict__ key_s, uint32_t* __restrict__ arr_t, uint32_t* __restrict__ arr_s,
                                    int32_t* __restrict__ cnt_t, int32_t* __restrict__ cnt_s, int32_t* __restrict__ flags,
                                    const GraphArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E_in || !p.keep[e]) return;
    const int slot = p.pos[e];
    const int R = a.R, F = a.F, B = (F - 1) / 2;
    int64_t src = p.edges_in[2 * (size_t)e], dst = p.edges_in[2 * (size_t)e + 1];
    if (src < 0 || src >= a.N || dst < 0 || dst >= a.N) {
        atomicOr(flags, 4);
        src = min(max(src, (int64_t)0), (int64_t)a.N - 1);
        dst = min(max(dst, (int64_t)0), (int64_t)a.N - 1);
    }
    // every load first, then the two returning atomics, then the arithmetic, then the stores: memory operations return in
    // issue order, so a load behind the atomics would wait for them (88 us for this kernel at config 2), while the
    // arithmetic below does not
    const float r = p.log_mag[e] / p.eps, theta = p.log_ang[e];
    const float2 x = p.xp[e];
    const float wsrc = p.w[src], tot = p.total[dst];
    // upper knot: the first knot >= r, never knot 0 (csrc/fc_precomp.hip: precomp_stencil_kernel)
    int hi = R - 1;
    for (int k = R - 1; k >= 1; --k)
        if (sqrtf((float)k / (float)(R - 1)) >= r) hi = k;
    const int q = hi - 1;
    const uint32_t kt = (uint32_t)dst * 8u + (uint32_t)q, ks = (uint32_t)src * 8u + (uint32_t)q;
    const uint32_t at = (uint32_t)atomicAdd(cnt_t + kt, 1), as = (uint32_t)atomicAdd(cnt_s + ks, 1);
    float sn, cs;
    sincosf(theta, &sn, &cs);
    const float scale = wsrc / (1e-12f + tot);
    const float2 c = make_float2(scale * x.x, scale * x.y);
    const float k_lo = sqrtf((float)(hi - 1) / (float)(R - 1)), k_hi = sqrtf((float)hi / (float)(R - 1));
    const float w1 = (r - k_lo) / (k_hi - k_lo), w0 = 1.f - w1;
    const float4 head = make_float4(__int_as_float(q), w0, w1, 0.f);
    const float4 cg = make_float4(c.x, c.y, cs, sn);
    float row[4 + 2 * kGraphMaxF + 2];            // the record: head, F phases, zero padding up to recf (<= 20)
    row[0] = head.x; row[1] = w0; row[2] = w1; row[3] = 0.f;
#pragma unroll
    for (int f = 0; f < kGraphMaxF; ++f) {      // ph_f = e^{i (f-B) theta} * c, in the reference's order of operations
        float2 ph = make_float2(0.f, 0.f);
        if (f < F) {
            float s_, c_;
            sincosf((float)(f - B) * theta, &s_, &c_);
            ph = cmul(make_float2(c_, s_), c);
        }
        row[4 + 2 * f] = ph.x;
        row[5 + 2 * f] = ph.y;
    }
    row[4 + 2 * kGraphMaxF] = row[5 + 2 * kGraphMaxF] = 0.f;
    p.ln[slot] = make_float2(r * cs, r * sn);
    p.wxp[slot] = c;
    p.edges_out[2 * (size_t)slot] = src;
    p.edges_out[2 * (size_t)slot + 1] = dst;
    float4* fp = reinterpret_cast<float4*>(p.factors + (size_t)slot * 8);
    fp[0] = head;
    fp[1] = cg;
    if (geo) {
        float4* gp = reinterpret_cast<float4*>(geo + (size_t)slot * 8);
        gp[0] = head;
        gp[1] = cg;
    }
    float4* rp = reinterpret_cast<float4*>(rec + (size_t)slot * a.recf);
#pragma unroll
    for (int k = 0; k < (4 + 2 * kGraphMaxF + 2) / 4; ++k)
        if (4 * k < a.recf) rp[k] = make_float4(row[4 * k], row[4 * k + 1], row[4 * k + 2], row[4 * k + 3]);
    key_t[slot] = kt;
    key_s[slot] = ks;
    arr_t[slot] = at;          // arrival index inside the (vertex, ring) run
    arr_s[slot] = as;
}

// ---- per vertex: ring-run offsets (exclusive over q) and the degree, both sides
__global__ void graph_runs_kernel(const int32_t* __restrict__ cnt_t, const int32_t* __restrict__ cnt_s, int32_t* __restrict__ runs_t,
                                  int32_t* __restrict__ runs_s, int32_t* __restrict__ deg_t, int32_t* __restrict__ deg_s, int N) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > N) return;
    if (v == N) { deg_t[N] = 0; deg_s[N] = 0; return; }         // the scans run over N+1 entries: rowptr[N] = E
    int at = 0, as = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ct = cnt_t[(size_t)v * 8 + q], cs = cnt_s[(size_t)v * 8 + q];
        runs_t[(size_t)v * 8 + q] = at;
        runs_s[(size_t)v * 8 + q] = as;
        at += ct;
        as += cs;
    }
    deg_t[v] = at;
    deg_s[v] = as;
}

// ---- counting sort, step 1: every edge into its (vertex, ring) run, in arrival order
__global__ void graph_scatter_kernel(const uint32_t* __restrict__ key_t, const uint32_t* __restrict__ key_s,
                                     const uint32_t* __restrict__ arr_t, const uint32_t* __restrict__ arr_s,
                                     const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ rowptr_s,
                                     const int32_t* __restrict__ runs_t, const int32_t* __restrict__ runs_s,
                                     uint32_t* __restrict__ raw_t, uint32_t* __restrict__ raw_s, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint32_t kt = key_t[e], ks = key_s[e];
    raw_t[(uint32_t)rowptr_t[kt >> 3] + (uint32_t)runs_t[kt] + arr_t[e]] = (uint32_t)e;
    raw_s[(uint32_t)rowptr_s[ks >> 3] + (uint32_t)runs_s[ks] + arr_s[e]] = (uint32_t)e;
}

// ---- counting sort, step 2: every run into ascending edge order.  The edge ids of a run are distinct, so an id's place is
// the number of smaller ids in the run.  One thread per run of up to kShortRun edges (a k-NN support has ~k/R per run);
// longer runs (a vertex that collects thousands of edges in one ring) are listed and ranked by a whole workgroup each.
constexpr int kShortRun = 32;
__global__ void graph_order_kernel(const int32_t* __restrict__ cnt_t, const int32_t* __restrict__ cnt_s,
                                   const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ rowptr_s,
                                   const int32_t* __restrict__ runs_t, const int32_t* __restrict__ runs_s,
                                   const uint32_t* __restrict__ raw_t, const uint32_t* __restrict__ raw_s,
                                   uint32_t* __restrict__ ids_t, uint32_t* __restrict__ ids_s, int32_t* __restrict__ long_runs, int N) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // side * 8N + key
    if (idx >= (long)N * 16) return;
    const bool side_s = idx >= (long)N * 8;
    const uint32_t key = (uint32_t)(idx - (side_s ? (long)N * 8 : 0));
    const int len = (side_s ? cnt_s : cnt_t)[key];
    if (len == 0) return;
    const uint32_t beg = (uint32_t)(side_s ? rowptr_s : rowptr_t)[key >> 3] + (uint32_t)(side_s ? runs_s : runs_t)[key];
    const uint32_t* in = (side_s ? raw_s : raw_t) + beg;
    uint32_t* out = (side_s ? ids_s : ids_t) + beg;
    if (len > kShortRun) {
        const int at = atomicAdd(long_runs, 1);
        long_runs[1 + at] = (int32_t)idx;                               // at most 16 N entries; ranked by graph_order_long_kernel
        return;
    }
    for (int i = 0; i < len; ++i) {
        const uint32_t mine = in[i];
        int rank = 0;
        for (int j = 0; j < len; ++j) rank += in[j] < mine ? 1 : 0;
        out[rank] = mine;
    }
}

__global__ __launch_bounds__(256) void graph_order_long_kernel(
    const int32_t* __restrict__ cnt_t, const int32_t* __restrict__ cnt_s, const int32_t* __restrict__ rowptr_t,
    const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ runs_t, const int32_t* __restrict__ runs_s,
    const uint32_t* __restrict__ raw_t, const uint32_t* __restrict__ raw_s, uint32_t* __restrict__ ids_t, uint32_t* __restrict__ ids_s,
    const int32_t* __restrict__ long_runs, int N) {
    const int n_long = long_runs[0];
    for (int l = blockIdx.x; l < n_long; l += gridDim.x) {
        const long idx = long_runs[1 + l];
        const bool side_s = idx >= (long)N * 8;
        const uint32_t key = (uint32_t)(idx - (side_s ? (long)N * 8 : 0));
        const int len = (side_s ? cnt_s : cnt_t)[key];
        const uint32_t beg = (uint32_t)(side_s ? rowptr_s : rowptr_t)[key >> 3] + (uint32_t)(side_s ? runs_s : runs_t)[key];
        const uint32_t* in = (side_s ? raw_s : raw_t) + beg;
        uint32_t* out = (side_s ? ids_s : ids_t) + beg;
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            const uint32_t mine = in[i];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += in[j] < mine ? 1 : 0;
            out[rank] = mine;
        }
    }
}

// ---- per slot: the other endpoint, the slot -> edge permutation and the records in slot order
__global__ __launch_bounds__(256) void graph_place_kernel(
    const uint32_t* __restrict__ sorted_val, const int64_t* __restrict__ edges, int other_col, const float* __restrict__ rec,
    const float* __restrict__ geo, int32_t* __restrict__ nbr, int64_t* __restrict__ perm, float* __restrict__ rec_out,
    float* __restrict__ geo_out, const GraphArgs a) {
    const int pieces = a.recf / 4;                  // float4 pieces of a record
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t slot = idx / pieces;
    const int piece = (int)(idx - slot * pieces);
    if (slot >= (size_t)a.E) {                      // padding rows: zeros
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rec_out && slot < (size_t)a.E + a.pad_rec) *reinterpret_cast<float4*>(rec_out + slot * a.recf + 4 * piece) = z;
        if (geo_out && piece < 2 && slot < (size_t)a.E + a.pad_geo) *reinterpret_cast<float4*>(geo_out + slot * 8 + 4 * piece) = z;
        return;
    }
    const uint32_t e = sorted_val[slot];
    int64_t o = edges[2 * (size_t)e + other_col];
    o = min(max(o, (int64_t)0), (int64_t)a.N - 1);
    if (piece == 0) {
        nbr[slot] = (int32_t)o;
        perm[slot] = (int64_t)e;
    }
    if (rec_out) {
        float4 v = *reinterpret_cast<const float4*>(rec + (size_t)e * a.recf + 4 * piece);
        if (piece == 0) v.w = __int_as_float((int32_t)o);
        *reinterpret_cast<float4*>(rec_out + slot * a.recf + 4 * piece) = v;
    }
    if (geo_out && piece < 2) {
        float4 v = *reinterpret_cast<const float4*>(geo + (size_t)e * 8 + 4 * piece);
        if (piece == 0) v.w = __int_as_float((int32_t)o);
        *reinterpret_cast<float4*>(geo_out + slot * 8 + 4 * piece) = v;
    }
}

static size_t align256(size_t b) { return (b + 255) / 256 * 256; }

struct GraphPlan {
    size_t cnt, deg, keys, ids, longs, rec, geo, cub, total;
    size_t cub_bytes;
};

static GraphPlan plan_graph(int N, int E, int recf, bool with_sten) {
    GraphPlan p;
    size_t scan_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, N + 1);
    p.cub_bytes = scan_bytes;
    size_t off = 0;
    p.longs = off; off += align256(((size_t)N * 16 + 1) * 4);          // count + list of the runs longer than kShortRun
    p.cnt = off;   off += align256((size_t)N * 8 * 4 * 2);             // cnt_t | cnt_s
    p.deg = off;   off += align256((size_t)(N + 1) * 4 * 2);           // deg_t | deg_s
    p.keys = off;  off += align256((size_t)E * 4) * 2;                 // key_t | key_s
    p.ids = off;   off += align256((size_t)E * 4) * 4;                 // arrival index, later ordered ids: t | s; ids in arrival order: t | s
    p.rec = off;   off += with_sten ? align256((size_t)E * recf * 4) : 0;
    p.geo = off;   off += with_sten ? align256((size_t)E * 8 * 4) : 0;
    p.cub = off;   off += align256(p.cub_bytes);
    p.total = off + 256;
    return p;
}

}  // namespace fc

extern "C" {

size_t fc_graph_workspace_bytes(int32_t N, int32_t E, int32_t R, int32_t F, int32_t with_stencil) {
    if (N <= 0 || E < 0) return 0;
    const int recf = (4 + 2 * F + 3) / 4 * 4;
    return fc::plan_graph(N, E, recf, with_stencil != 0).total;
}

// pad_rec / pad_geo: rows behind the E-th of rec_t, rec_s / geo_t that are zero-filled
static int graph_build_impl(const int64_t* supp_edges, const float* supp_sten, const fc::FactorArgs* factors, int32_t N, int32_t E,
                            int32_t pad_rec, int32_t pad_geo, int32_t R, int32_t F, int32_t* rowptr_t, int32_t* nbr_t, int32_t* runs_t, int64_t* perm_t, int32_t* rowptr_s,
                            int32_t* nbr_s, int32_t* runs_s, int64_t* perm_s, float* rec_t, float* rec_s, float* geo_t, int32_t* flags,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (N <= 0 || E < 0 || !rowptr_t || !rowptr_s || !runs_t || !runs_s || !flags || !workspace) return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!supp_edges || !nbr_t || !nbr_s || !perm_t || !perm_s)) return FC_ERR_BAD_ARGUMENT;
    if ((uint64_t)N * 16 >= ((uint64_t)1 << 31)) return FC_ERR_UNSUPPORTED;
    const bool with_sten = supp_sten != nullptr || factors != nullptr;
    if (with_sten && (R < 2 || R > fc::kGraphMaxR || F < 1 || F > fc::kGraphMaxF || (F & 1) == 0 || !rec_t || !rec_s)) return FC_ERR_UNSUPPORTED;
    const int recf = (4 + 2 * F + 3) / 4 * 4;
    const fc::GraphPlan p = fc::plan_graph(N, E, recf, with_sten);
    if (workspace_bytes < p.total) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(workspace);
    int32_t* cnt_t = reinterpret_cast<int32_t*>(w + p.cnt);
    int32_t* cnt_s = cnt_t + (size_t)N * 8;
    int32_t* deg_t = reinterpret_cast<int32_t*>(w + p.deg);
    int32_t* deg_s = deg_t + (N + 1);
    const size_t ke = (((size_t)E * 4 + 255) / 256 * 256) / 4;
    uint32_t* key_t = reinterpret_cast<uint32_t*>(w + p.keys);
    uint32_t* key_s = key_t + ke;
    uint32_t* arr_t = reinterpret_cast<uint32_t*>(w + p.ids);         // arrival indices; the ordered ids take their place
    uint32_t* arr_s = arr_t + ke;
    uint32_t* raw_t = arr_s + ke;
    uint32_t* raw_s = raw_t + ke;
    int32_t* long_runs = reinterpret_cast<int32_t*>(w + p.longs);
    float* rec = with_sten ? reinterpret_cast<float*>(w + p.rec) : nullptr;
    float* geo = (with_sten && geo_t) ? reinterpret_cast<float*>(w + p.geo) : nullptr;
    void* cub = w + p.cub;
    size_t cub_bytes = p.cub_bytes;
    if (pad_rec < 0 || pad_geo < 0) return FC_ERR_BAD_ARGUMENT;
    const fc::GraphArgs a{N, E, R, F, recf, pad_rec, pad_geo};

    // one fill: the long-run list (its counter is what matters) and the bucket counts are adjacent
    if (hipMemsetAsync(w + p.longs, 0, (p.cnt - p.longs) + (size_t)N * 8 * 4 * 2, s) != hipSuccess) return FC_ERR_LAUNCH;
    if (hipMemsetAsync(flags, 0, 4, s) != hipSuccess) return FC_ERR_LAUNCH;
    if (factors) {
        if (factors->E_in > 0)
            hipLaunchKernelGGL(fc::graph_factor_kernel, dim3((factors->E_in + 255) / 256), dim3(256), 0, s, *factors, rec, geo, key_t, key_s,
                               arr_t, arr_s, cnt_t, cnt_s, flags, a);
    } else if (E > 0) {
        hipLaunchKernelGGL(fc::graph_analyze_kernel, dim3((E + fc::kAnalyzeEdges - 1) / fc::kAnalyzeEdges), dim3(fc::kAnalyzeEdges),
                           with_sten ? (size_t)fc::kAnalyzeEdges * (R * F + 1) * sizeof(float2) : 0, s, supp_edges,
                           reinterpret_cast<const float2*>(supp_sten), rec, geo, key_t, key_s, arr_t, arr_s, cnt_t, cnt_s, flags, a);
    }
    hipLaunchKernelGGL(fc::graph_runs_kernel, dim3((N + 1 + 255) / 256), dim3(256), 0, s, cnt_t, cnt_s, runs_t, runs_s, deg_t, deg_s, N);
    if (hipcub::DeviceScan::ExclusiveSum(cub, cub_bytes, deg_t, rowptr_t, N + 1, s) != hipSuccess) return FC_ERR_LAUNCH;
    cub_bytes = p.cub_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(cub, cub_bytes, deg_s, rowptr_s, N + 1, s) != hipSuccess) return FC_ERR_LAUNCH;
    if (E > 0) {
        hipLaunchKernelGGL(fc::graph_scatter_kernel, dim3((E + 255) / 256), dim3(256), 0, s, key_t, key_s, arr_t, arr_s, rowptr_t, rowptr_s,
                           runs_t, runs_s, raw_t, raw_s, E);
        uint32_t* ids_t = arr_t;
        uint32_t* ids_s = arr_s;
        hipLaunchKernelGGL(fc::graph_order_kernel, dim3((unsigned)(((size_t)N * 16 + 255) / 256)), dim3(256), 0, s, cnt_t, cnt_s, rowptr_t,
                           rowptr_s, runs_t, runs_s, raw_t, raw_s, ids_t, ids_s, long_runs, N);
        hipLaunchKernelGGL(fc::graph_order_long_kernel, dim3(256), dim3(256), 0, s, cnt_t, cnt_s, rowptr_t, rowptr_s, runs_t, runs_s, raw_t,
                           raw_s, ids_t, ids_s, long_runs, N);
        const size_t threads = ((size_t)E + (pad_rec > pad_geo ? pad_rec : pad_geo)) * (recf / 4);
        const dim3 grid((unsigned)((threads + 255) / 256));
        hipLaunchKernelGGL(fc::graph_place_kernel, grid, dim3(256), 0, s, ids_t, supp_edges, 0, rec, geo, nbr_t, perm_t,
                           with_sten ? rec_t : nullptr, geo ? geo_t : nullptr, a);
        hipLaunchKernelGGL(fc::graph_place_kernel, grid, dim3(256), 0, s, ids_s, supp_edges, 1, rec, nullptr, nbr_s, perm_s,
                           with_sten ? rec_s : nullptr, nullptr, a);
    }
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_graph_build(const int64_t* supp_edges, const float* supp_sten, int32_t N, int32_t E, int32_t R, int32_t F,
                   int32_t* rowptr_t, int32_t* nbr_t, int32_t* runs_t, int64_t* perm_t, int32_t* rowptr_s, int32_t* nbr_s,
                   int32_t* runs_s, int64_t* perm_s, float* rec_t, float* rec_s, float* geo_t, int32_t* flags, void* workspace,
                   size_t workspace_bytes, void* stream) {
    return graph_build_impl(supp_edges, supp_sten, nullptr, N, E, 0, 0, R, F, rowptr_t, nbr_t, runs_t, perm_t, rowptr_s, nbr_s, runs_s, perm_s,
                            rec_t, rec_s, geo_t, flags, workspace, workspace_bytes, stream);
}

int fc_precomp_graph(const float* log_mag, const float* log_ang, const float* xp, const float* w, const int64_t* supp_edges,
                     float epsilon, int32_t N, int32_t E, int32_t E_kept, int32_t R, int32_t F, int32_t rec_pad_rows,
                     int32_t geo_pad_rows, int64_t* supp_edges_out, float* ln,
                     float* wxp, float* factors, int32_t* rowptr_t, int32_t* nbr_t, int32_t* runs_t, int64_t* perm_t, int32_t* rowptr_s,
                     int32_t* nbr_s, int32_t* runs_s, int64_t* perm_s, float* rec_t, float* rec_s, float* geo_t, int32_t* flags,
                     void* precomp_workspace, size_t precomp_workspace_bytes, void* graph_workspace, size_t graph_workspace_bytes,
                     void* stream) {
    if (!log_mag || !log_ang || !xp || !w || !supp_edges || !precomp_workspace || N <= 0 || E <= 0 || E_kept <= 0 || E_kept > E)
        return FC_ERR_BAD_ARGUMENT;
    if (!supp_edges_out || !ln || !wxp || !factors || !(epsilon > 0.f)) return FC_ERR_BAD_ARGUMENT;
    if (precomp_workspace_bytes < fc_precomp_workspace_bytes(N, E)) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // layout of the selection left by fc_precomp_mark (csrc/fc_precomp.hip): keep | pos | total (N)
    const size_t seg = ((size_t)(E + 2) * 4 + 255) / 256 * 256;
    char* wsp = static_cast<char*>(precomp_workspace);
    fc::FactorArgs fa;
    fa.log_mag = log_mag; fa.log_ang = log_ang; fa.xp = reinterpret_cast<const float2*>(xp); fa.w = w; fa.edges_in = supp_edges;
    fa.keep = reinterpret_cast<const int32_t*>(wsp);
    fa.pos = reinterpret_cast<const int32_t*>(wsp + seg);
    float* total = reinterpret_cast<float*>(wsp + 2 * seg);
    fa.total = total;
    fa.eps = epsilon; fa.E_in = E;
    fa.edges_out = supp_edges_out; fa.ln = reinterpret_cast<float2*>(ln); fa.wxp = reinterpret_cast<float2*>(wxp); fa.factors = factors;
    const int rc = fc::precomp_area_sums(supp_edges, fa.keep, w, total, N, E, s);
    if (rc != FC_OK) return rc;
    return graph_build_impl(supp_edges_out, nullptr, &fa, N, E_kept, rec_pad_rows, geo_pad_rows, R, F, rowptr_t, nbr_t, runs_t, perm_t, rowptr_s, nbr_s, runs_s, perm_s,
                            rec_t, rec_s, geo_t, flags, graph_workspace, graph_workspace_bytes, stream);
}

}  // extern "C"
