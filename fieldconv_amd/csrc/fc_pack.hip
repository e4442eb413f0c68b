// Filter packing for the MFMA contractions, and the parameter-gradient chain.
//   W_eff (O,I,R,F) complex64 is what reference nn/field_conv.py:10-33 (weightContrib*) assembles from
//   (zonal, spherical, phase); the 1/(2B+1) of :14,:25,:33 is folded into the packed values.
//   fc_pack_filter          packs a given W_eff
//   fc_pack_filter_params   assembles W_eff from the parameters on the fly (one kernel instead of the
//                           cat / flip / conj / polar / mul chain in torch) and packs it
//   fc_filter_param_grads   pulls dL/dW_eff back to (zonal, spherical, phase) -- the autograd twin
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"
#include "fc_forward_ring.hpp"

namespace fc {

// W_eff[o,i,r,f] from the reference parameter tensors (layouts: nn/field_conv.py:73-93).
//   ftype 0: [conj(sph) reversed | zonal | sph]                           (:12)
//   ftype 1: the same times exp(i*phase[o,i,|f-B|])                        (:18,:23,:25)
//   ftype 2: [sph[..:B] | zonal_c | sph[..B:]]  (complex zonal, 2B spherical) (:31)
__device__ __forceinline__ float2 filter_entry(const float* __restrict__ zonal, const float* __restrict__ sph,
                                               const float* __restrict__ phase, int ftype, int B, int R, int I, int o, int i,
                                               int r, int f) {
    const size_t oir = ((size_t)o * I + i) * R + r;
    if (ftype == 2) {
        if (f == B) return make_float2(zonal[oir * 2], zonal[oir * 2 + 1]);
        const int b = f < B ? f : f - 1;
        const float* p = sph + (oir * (2 * B) + b) * 2;
        return make_float2(p[0], p[1]);
    }
    // every request unconditional and from a valid address (the coefficient of |f - B| = 0 is the zonal one: the spherical slot read for it
    // is entry 0 and dropped by a select): behind data-dependent branches the loads wait for one another -- the packing and finishing
    // launches are chains of such round trips, not bandwidth
    const int q = f >= B ? f - B : B - f;
    const float z = zonal[oir];
    float2 sp = make_float2(0.f, 0.f);
    if (B > 0) sp = *reinterpret_cast<const float2*>(sph + (oir * B + (q > 0 ? q - 1 : 0)) * 2);
    float ph = 0.f;
    if (ftype == 1) ph = phase[((size_t)o * I + i) * (B + 1) + q];
    float2 c = q == 0 ? make_float2(z, 0.f) : make_float2(sp.x, f > B ? sp.y : -sp.y);
    if (ftype == 1) {
        float s, co;
        sincosf(ph, &s, &co);
        c = cmul(c, make_float2(co, s));
    }
    return c;
}

// Packed images (layouts: fc_tile.hpp packed_image_floats):
//   fwd image: rows m = o, k = r*I + i, value W/F          bwd image: rows m = i, k = r*O + o, value conj(W)/F
// An fp32 image is written one element per thread; a split image one row per workgroup (the row's power-of-two
// scale needs the row maximum first).
struct PackArgs {
    int O, I, R, F, B, ftype;
    MmaGeom gf, gb;          // forward / backward contraction geometry (with their modes)
    unsigned blocks_f;       // workgroups that write the forward image; the rest write the backward image
    int ring_f;              // forward image in ring-major layout (fc_forward_ring.hpp): one slab of planes per RING,
                             // k = f*KI + i inside it, instead of one per frequency with k = r*KI + i
    int o0, i0, Ifull;       // the (O, I) filter is the block [o0, o0 + O) x [i0, i0 + I) of parameter tensors with Ifull input
                             // channels (wide layers run as channel blocks, fc_wide.hip); a whole layer: 0, 0, I
};

constexpr int kPackThreads = 256;

// WITH_PHASE = false: the same entry without the ftype-1 phase factor (same modulus, no sincos)
template <bool FROM_PARAMS, bool WITH_PHASE>
__device__ __forceinline__ float2 packed_value(const float2* __restrict__ w, const float* __restrict__ zonal,
                                               const float* __restrict__ sph, const float* __restrict__ phase,
                                               const PackArgs& a, bool is_bwd, int m, int k, int slab) {
    const int KI = is_bwd ? a.gb.KI : a.gf.KI;          // k = r*KI + c, channels c >= their count are padding
    const bool ring = !is_bwd && a.ring_f;              // ring-major: slab = r, k = f*KI + c
    const int kk = k / KI, c = k - kk * KI;
    const int r = ring ? slab : kk, f = ring ? kk : slab;
    if (f >= a.F) return make_float2(0.f, 0.f);
    const int o = is_bwd ? c : m, i = is_bwd ? m : c;
    if (o >= a.O || i >= a.I || r >= a.R) return make_float2(0.f, 0.f);
    const float2 v = FROM_PARAMS ? filter_entry(zonal, sph, phase, (!WITH_PHASE && a.ftype == 1) ? 0 : a.ftype, a.B, a.R, a.Ifull, a.o0 + o,
                                                a.i0 + i, r, f)
                                 : w[(((size_t)(a.o0 + o) * a.Ifull + a.i0 + i) * a.R + r) * a.F + f];
    const float sc = 1.f / (float)a.F;
    return make_float2(v.x * sc, (is_bwd ? -v.y : v.y) * sc);
}

// max over the row of |W_eff|^2 / F^2: forward image row m = o (over i, r, f), backward image row m = i (over o, r, f).
// From the parameters the scan runs over the UNIQUE coefficients -- |coeff[o,i,r,+-b]| = |spherical[o,i,r,b]|, the phase
// factor has modulus 1 -- i.e. C*R entries of 1 + B (ftype 2: 1 + 2B) numbers instead of C*R*F assembled filter entries.
template <bool FROM_PARAMS>
__device__ __forceinline__ float row_max2(const float2* __restrict__ w, const float* __restrict__ zonal, const float* __restrict__ sph,
                                          const PackArgs& a, bool is_bwd, int m) {
    const int M = is_bwd ? a.I : a.O, C = is_bwd ? a.O : a.I;
    float mx = 0.f;
    if (m >= M) return mx;
    if (FROM_PARAMS) {
        for (int idx = threadIdx.x; idx < C * a.R; idx += kPackThreads) {
            const int c = idx / a.R, r = idx - c * a.R;
            const int o = is_bwd ? c : m, i = is_bwd ? m : c;
            const size_t oir = ((size_t)(a.o0 + o) * a.Ifull + a.i0 + i) * a.R + r;
            if (a.ftype == 2) {
                mx = fmaxf(mx, zonal[oir * 2] * zonal[oir * 2] + zonal[oir * 2 + 1] * zonal[oir * 2 + 1]);
                for (int b = 0; b < 2 * a.B; ++b) {
                    const float* p = sph + (oir * (2 * a.B) + b) * 2;
                    mx = fmaxf(mx, p[0] * p[0] + p[1] * p[1]);
                }
            } else {
                mx = fmaxf(mx, zonal[oir] * zonal[oir]);
                for (int b = 0; b < a.B; ++b) {
                    const float* p = sph + (oir * a.B + b) * 2;
                    mx = fmaxf(mx, p[0] * p[0] + p[1] * p[1]);
                }
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < C * a.R * a.F; idx += kPackThreads) {
            const int c = idx / (a.R * a.F), rf = idx - c * (a.R * a.F);
            const int o = is_bwd ? c : m, i = is_bwd ? m : c;
            const float2 v = w[((size_t)(a.o0 + o) * a.Ifull + a.i0 + i) * a.R * a.F + rf];
            mx = fmaxf(mx, v.x * v.x + v.y * v.y);
        }
    }
    const float sc = 1.f / (float)a.F;
    return mx * sc * sc;
}

template <bool FROM_PARAMS>
__device__ __forceinline__ void pack_filter_body(const float2* __restrict__ w, const float* __restrict__ zonal, const float* __restrict__ sph,
                                                 const float* __restrict__ phase, float* __restrict__ fwd, float* __restrict__ bwd,
                                                 const PackArgs& a, const unsigned block) {
    const bool is_bwd = block >= a.blocks_f;
    const unsigned blk = is_bwd ? block - a.blocks_f : block;
    const MmaGeom& g = is_bwd ? a.gb : a.gf;
    float* const img = is_bwd ? bwd : fwd;
    if (!g.split) {
        const size_t j = (size_t)blk * kPackThreads + threadIdx.x;
        if (j >= (size_t)a.F * 2 * g.MP * g.KP) return;
        const int k = j % g.KP;
        const int m = (j / g.KP) % g.MP;
        const int pl = (j / ((size_t)g.KP * g.MP)) % 2;
        const int f = j / ((size_t)g.KP * g.MP * 2);
        const float2 v = packed_value<FROM_PARAMS, true>(w, zonal, sph, phase, a, is_bwd, m, k, f);
        img[j] = pl == 0 ? v.x : v.y;
        return;
    }
    // split image: workgroup = (row m, frequency f).  The row scale comes from the largest MODULUS of the row
    // (all frequencies): it bounds both components, needs no phase factor (|c e^{i phi}| = |c|), and every
    // workgroup of the row recomputes it cheaply instead of synchronising.
    __shared__ float red[kPackThreads / kWave];
    const int nslab = (!is_bwd && a.ring_f) ? a.R : a.F;     // slabs of planes in the image
    const int m = blk / nslab, f = blk - m * nslab;
    float mx = row_max2<FROM_PARAMS>(w, zonal, sph, a, is_bwd, m);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = sqrtf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))) * 1.0000002f;     // never below the true modulus
    float scale, inv;
    split_scale(mx, scale, inv);
    if (threadIdx.x == 0 && f == 0) img[m] = inv;
    _Float16* const planes = reinterpret_cast<_Float16*>(img + g.MP);
    const size_t plane = (size_t)g.MP * g.KP;
    for (int k = threadIdx.x; k < g.KP; k += kPackThreads) {
        const float2 v = packed_value<FROM_PARAMS, true>(w, zonal, sph, phase, a, is_bwd, m, k, f);
        _Float16 rh, rl, ih, il;
        split_halves(v.x * scale, rh, rl);
        split_halves(v.y * scale, ih, il);
        _Float16* p = planes + (size_t)f * 2 * g.split * plane + ((size_t)(k >> 5) * g.MP + m) * 32 + (k & 31);       // k-block major
        if (g.split == 2) {
            p[0] = rh;
            p[plane] = rl;
            p[2 * plane] = ih;
            p[3 * plane] = il;
        } else {
            p[0] = rh;
            p[plane] = ih;
        }
    }
}

template <bool FROM_PARAMS>
__global__ __launch_bounds__(kPackThreads) void fc_pack_filter_kernel(const float2* __restrict__ w, const float* __restrict__ zonal,
                                                                      const float* __restrict__ sph,
                                                                      const float* __restrict__ phase, float* __restrict__ fwd,
                                                                      float* __restrict__ bwd, const PackArgs a) {
    pack_filter_body<FROM_PARAMS>(w, zonal, sph, phase, fwd, bwd, a, blockIdx.x);
}

// The filters of TWO layers from one launch (the two convolutions of an FCResNetBlock, csrc/fc_blocks.hip): the first `split` workgroups
// pack layer 0, the rest layer 1 -- the same per-workgroup work as two launches of the kernel above, one launch fewer per block and pass.
struct PackPtrs { const float* zonal; const float* sph; const float* phase; float* fwd; float* bwd; };
__global__ __launch_bounds__(kPackThreads) void fc_pack_filter_pair_kernel(const PackPtrs p0, const PackArgs a0, const PackPtrs p1,
                                                                           const PackArgs a1, const unsigned split) {
    if (blockIdx.x < split) pack_filter_body<true>(nullptr, p0.zonal, p0.sph, p0.phase, p0.fwd, p0.bwd, a0, blockIdx.x);
    else pack_filter_body<true>(nullptr, p1.zonal, p1.sph, p1.phase, p1.fwd, p1.bwd, a1, blockIdx.x - split);
}

static unsigned pack_blocks(const MmaGeom& g, int F) {      // F: slabs of planes in the image
    return g.split ? (unsigned)(g.MP * F) : (unsigned)(((size_t)F * 2 * g.MP * g.KP + kPackThreads - 1) / kPackThreads);
}

static bool ring_forward_image(const fc_dims* d, int records) { return (records & 1) && forward_ring_fits(d); }

size_t packed_filter_floats_fwd(const fc_dims* d, int records) {
    if (ring_forward_image(d, records)) return packed_ring_image_floats(d->O, 2 * d->B + 1, d->I, d->R, halves_of(d));
    return packed_image_floats(d->O, d->R, d->I, 2 * d->B + 1, halves_of(d));
}
size_t packed_filter_floats_bwd(const fc_dims* d, int records) {
    (void)records;
    return packed_image_floats(d->I, d->R, d->O, 2 * d->B + 1, halves_of(d));
}

// -> workgroups of the launch (0: nothing to write)
static unsigned pack_args(PackArgs& a, int ftype, bool want_fwd, bool want_bwd, const fc_dims* d, int records, int o0, int i0, int Ifull) {
    a.O = d->O; a.I = d->I; a.R = d->R; a.B = d->B; a.F = 2 * d->B + 1; a.ftype = ftype;
    a.o0 = o0; a.i0 = i0; a.Ifull = Ifull > 0 ? Ifull : d->I;
    a.ring_f = ring_forward_image(d, records & 1) ? 1 : 0;
    a.gf = a.ring_f ? ring_geom(d->O, a.F, d->I, halves_of(d)) : make_mma_geom(d->O, d->R, d->I, halves_of(d));
    a.gb = make_mma_geom(d->I, d->R, d->O, halves_of(d));
    a.blocks_f = want_fwd ? pack_blocks(a.gf, a.ring_f ? a.R : a.F) : 0u;                // no forward image wanted: backward image only
    const unsigned blocks_b = pack_blocks(a.gb, a.F);
    return a.blocks_f + (want_bwd ? blocks_b : 0u);                                       // no backward image wanted: forward image only
}

int pack_filter_params_pair_impl(const fc_filter_params& f0, float* fwd0, float* bwd0, const fc_dims* d0, const fc_filter_params& f1,
                                 float* fwd1, float* bwd1, const fc_dims* d1, int records, hipStream_t stream) {
    PackArgs a0, a1;
    const unsigned n0 = pack_args(a0, f0.ftype, fwd0 != nullptr, bwd0 != nullptr, d0, records, 0, 0, 0);
    const unsigned n1 = pack_args(a1, f1.ftype, fwd1 != nullptr, bwd1 != nullptr, d1, records, 0, 0, 0);
    if (n0 == 0 || n1 == 0) return FC_ERR_BAD_ARGUMENT;
    const PackPtrs p0{f0.zonal, f0.spherical, f0.phase, fwd0, bwd0}, p1{f1.zonal, f1.spherical, f1.phase, fwd1, bwd1};
    hipLaunchKernelGGL(fc_pack_filter_pair_kernel, dim3(n0 + n1), dim3(kPackThreads), 0, stream, p0, a0, p1, a1, n0);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <bool FROM_PARAMS>
static int launch_pack(const float* w_eff, const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd,
                       float* wpk_bwd, const fc_dims* d, int records, hipStream_t stream, int o0 = 0, int i0 = 0, int Ifull = 0) {
    PackArgs a;
    const unsigned blocks = pack_args(a, ftype, wpk_fwd != nullptr, wpk_bwd != nullptr, d, records, o0, i0, Ifull);
    if (blocks == 0) return FC_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(fc_pack_filter_kernel<FROM_PARAMS>, dim3(blocks), dim3(kPackThreads), 0, stream,
                       reinterpret_cast<const float2*>(w_eff), zonal, sph, phase, wpk_fwd, wpk_bwd, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int pack_filter_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, int records, hipStream_t stream) {
    return launch_pack<false>(w_eff, nullptr, nullptr, nullptr, 0, wpk_fwd, wpk_bwd, d, records, stream);
}

int pack_filter_params_impl(const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd,
                            float* wpk_bwd, const fc_dims* d, int records, hipStream_t stream) {
    return launch_pack<true>(nullptr, zonal, sph, phase, ftype, wpk_fwd, wpk_bwd, d, records, stream);
}

int pack_filter_params_block_impl(const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd, float* wpk_bwd,
                                  const fc_dims* d, int records, int o0, int i0, int Ifull, hipStream_t stream) {
    return launch_pack<true>(nullptr, zonal, sph, phase, ftype, wpk_fwd, wpk_bwd, d, records, stream, o0, i0, Ifull);
}

int pack_filter_block_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, int records, int o0, int i0, int Ifull,
                           hipStream_t stream) {
    return launch_pack<false>(w_eff, nullptr, nullptr, nullptr, 0, wpk_fwd, wpk_bwd, d, records, stream, o0, i0, Ifull);
}


// One thread per (o, i, r): reads gW_eff[o,i,r,:] (F complex) and writes the parameter gradients of
// its ring; the phase gradient sums over the rings of (o,i) through LDS in a fixed order.
// (torch convention for a real loss: g = dL/dRe + i dL/dIm; for a real parameter p of a complex
// w(p): g_p = Re(conj(dw/dp) g_w)).
constexpr int kGradPairs = 32;      // (o,i) pairs per workgroup; blockDim = kGradPairs * R
constexpr int kMaxB = 3;

// Parameter gradients of ring r of the pair (o, i) from g = gW_eff[o,i,r,:] (F complex numbers, global or LDS); the phase
// gradient's per-ring pieces go to gph_row[0..B] (summed over the rings by the caller).
__device__ __forceinline__ void param_grads_entry(const float2* g, const float* __restrict__ zonal, const float* __restrict__ sph,
                                                  const float* __restrict__ phase, int ftype, float* __restrict__ g_zonal,
                                                  float* __restrict__ g_sph, float* gph_row, int I, int R, int B, int o, int i, int r) {
    const size_t oi = (size_t)o * I + i;
    const size_t oir = oi * R + r;
    if (ftype == 2) {
        g_zonal[oir * 2] = g[B].x;
        g_zonal[oir * 2 + 1] = g[B].y;
        for (int b = 0; b < 2 * B; ++b) {
            const float2 v = g[b < B ? b : b + 1];
            g_sph[(oir * (2 * B) + b) * 2] = v.x;
            g_sph[(oir * (2 * B) + b) * 2 + 1] = v.y;
        }
        return;
    }
    // coefficient gradient g_coeff = gW conj(P); phase gradient pieces Im(conj(P) gP), gP = gW conj(coeff)
    for (int q = 0; q <= B; ++q) {
        float s = 0.f, co = 1.f;
        if (ftype == 1) sincosf(phase[oi * (B + 1) + q], &s, &co);
        const float2 P = make_float2(co, s);
        const float2 gp = g[B + q];
        const float2 cp = filter_entry(zonal, sph, phase, 0, B, R, I, o, i, r, B + q);     // without the phase
        float2 gP = cmul_conj(gp, cp);
        const float2 gcp = cmul_conj(gp, P);
        float acc = P.x * gP.y - P.y * gP.x;
        if (q == 0) {
            g_zonal[oir] = gcp.x;
        } else {
            const float2 gm = g[B - q];
            const float2 cm = filter_entry(zonal, sph, phase, 0, B, R, I, o, i, r, B - q);
            gP = cmul_conj(gm, cm);
            const float2 gcm = cmul_conj(gm, P);
            acc += P.x * gP.y - P.y * gP.x;
            // f = B+q reads sph[q-1] directly, f = B-q its conjugate
            g_sph[(oir * B + (q - 1)) * 2] = gcp.x + gcm.x;
            g_sph[(oir * B + (q - 1)) * 2 + 1] = gcp.y - gcm.y;
        }
        gph_row[q] = acc;
    }
}

__global__ void fc_filter_param_grads_kernel(const float2* __restrict__ gw, const float* __restrict__ zonal,
                                             const float* __restrict__ sph, const float* __restrict__ phase, int ftype,
                                             float* __restrict__ g_zonal, float* __restrict__ g_sph,
                                             float* __restrict__ g_phase, int O, int I, int R, int B) {
    __shared__ float gph[kGradPairs * 8 * (kMaxB + 1)];      // [pair][r][q]
    const int pair = threadIdx.x / R, r = threadIdx.x - pair * R;
    const int oi = blockIdx.x * kGradPairs + pair;
    const bool valid = oi < O * I;
    const int F = 2 * B + 1;
    const int o = oi / I, i = oi - o * I;
    if (valid)
        param_grads_entry(gw + ((size_t)oi * R + r) * F, zonal, sph, phase, ftype, g_zonal, g_sph, gph + (pair * 8 + r) * (kMaxB + 1), I, R, B,
                          o, i, r);
    __syncthreads();
    if (valid && ftype == 1 && r == 0) {
        for (int q = 0; q <= B; ++q) {
            float acc = 0.f;
            for (int rr = 0; rr < R; ++rr) acc += gph[(pair * 8 + rr) * (kMaxB + 1) + q];
            g_phase[(size_t)oi * (B + 1) + q] = acc;
        }
    }
}

// The fixed-order sum of the filter-gradient kernels' per-workgroup partials AND the parameter-gradient chain in one launch
// (fewer and fatter small kernels).  Workgroup = (output channel o, 16 input channels): every thread sums
// the P partials of its (i, r, f) entries -- partial (p, r, f, o, i) lies at gwp[p*sp + r*sr + f*sf + o*so + i], or with the
// rings in pairs (RpStrides::pairs) -- writes gW_eff (when asked for) and leaves it in LDS, from where the
// workgroup's (pair, ring) threads pull it back to (zonal, spherical, phase) exactly as fc_filter_param_grads_kernel does.
constexpr int kRpPairs = 16;          // (8 / 4 input channels per workgroup: 16 / 26 us instead of 10 at config 2 -- the partials' 128-byte segments shrink)
constexpr int kRpThreads = 1024;
constexpr int kRpGroups = 4;           // an entry's P partials are summed in four consecutive groups, by four threads
constexpr int kRpMaxPer = 16;          // partials per group held in flight
struct RpStrides { size_t sp, sr, sf, so; int pairs; };       // pairs: k = dump_k(r, o) (fc_kernels.hpp)

__global__ __launch_bounds__(kRpThreads) void fc_reduce_param_grads_kernel(
    const float2* __restrict__ gwp, const RpStrides st, const int P, float2* __restrict__ gw_out, const float* __restrict__ zonal,
    const float* __restrict__ sph, const float* __restrict__ phase, const int ftype, float* __restrict__ g_zonal,
    float* __restrict__ g_sph, float* __restrict__ g_phase, const int O, const int I, const int R, const int B, const int po0,
    const int pi0, const int Ifull, const float* __restrict__ bias_partials, const int bias_nparts, float* __restrict__ g_bias,
    const int bias_block0, const f32x4* __restrict__ gx_parts, f32x4* __restrict__ gx, const size_t gx_count4, const size_t gx_stride4,
    const int gx_nparts, const size_t gx_tail_floats, const int gx_block0, const float2* __restrict__ gx_x, const int dbg) {
    // (po0, pi0, Ifull: the filter is the block [po0, po0 + O) x [pi0, pi0 + I) of parameter tensors with Ifull input channels)
    __shared__ float2 part[kRpGroups][kRpPairs * 8 * 7];     // [group][(r*F + f)*16 + pair]
    __shared__ float2 gws[kRpPairs * 8 * 7];                 // [pair][r][f]
    __shared__ float gph[kRpPairs * 8 * (kMaxB + 1)];        // [pair][r][q]
    if ((int)blockIdx.x >= gx_block0 && gx_nparts < 0) {
        // second rider, H-streaming arrangement (fc_backward_stream.hpp): gx from the F gxt slices the streaming kernel left and x --
        // fc_backward_gx_kernel's arithmetic; gx_parts: the slices, gx_count4: complex numbers per slice, -gx_nparts: the band limit
        const size_t idx = (size_t)((int)blockIdx.x - gx_block0) * kRpThreads + threadIdx.x;
        if (idx >= gx_count4) return;
        const float2* gxt = reinterpret_cast<const float2*>(gx_parts);
        float2* out = reinterpret_cast<float2*>(gx);
        const float2 xv = gx_x[idx];
        // (-gx_nparts = band limit + 4 * (slices per frequency - 1))
        const int bl = (-gx_nparts) & 3, ks = 1 + ((-gx_nparts) >> 2);
        out[idx] = bl == 1 ? gx_from_slices<1>(xv, gxt, idx, gx_count4, ks)
                 : bl == 2 ? gx_from_slices<2>(xv, gxt, idx, gx_count4, ks) : gx_from_slices<3>(xv, gxt, idx, gx_count4, ks);
        return;
    }
    if ((int)blockIdx.x >= gx_block0) {
        // second rider: gx = the sum of the backward data kernel's partial arrays in part order (fc_sum_parts_kernel's arithmetic)
        const size_t idx = (size_t)((int)blockIdx.x - gx_block0) * kRpThreads + threadIdx.x;
        if (idx >= gx_count4) return;
        f32x4 s = gx_parts[idx];
        for (int p = 1; p < gx_nparts; ++p) s += gx_parts[(size_t)p * gx_stride4 + idx];
        if (idx + 1 < gx_count4 || gx_tail_floats == 0) gx[idx] = s;
        else {
            float* dd = reinterpret_cast<float*>(gx + idx);
            for (size_t k = 0; k < gx_tail_floats; ++k) dd[k] = s[k];
        }
        return;
    }
    if ((int)blockIdx.x >= bias_block0) {
        // the rider (fc_filter_params::bias_partials): one wavefront per channel -- lanes take every 64th partial in order, then a fixed
        // butterfly -- exactly tangent_nonlin_gb_reduce_kernel's arithmetic
        const int lane = threadIdx.x & 63;
        const int c = ((int)blockIdx.x - bias_block0) * (kRpThreads / 64) + (threadIdx.x >> 6);
        if (c >= O) return;
        float s = 0.f;
        for (int p = lane; p < bias_nparts; p += 64) s += bias_partials[(size_t)p * O + c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) g_bias[c] = s;
        return;
    }
    if (dbg & 8) return;
    const int F = 2 * B + 1;
    const int nit = (I + kRpPairs - 1) / kRpPairs;
    const int o = blockIdx.x / nit, i0 = (blockIdx.x - o * nit) * kRpPairs;
    const int nent = kRpPairs * R * F;
    const int per = (P + kRpGroups - 1) / kRpGroups;         // partials per group
    // every (entry, group) pair sums its partials in order, all loads in flight at once (the sum of 51 partials is latency:
    // two rounds of loads here instead of seven)
    for (int idx = threadIdx.x; idx < nent * kRpGroups; idx += kRpThreads) {
        const int grp = idx / nent, e = idx - grp * nent;
        const int pi = e % kRpPairs, rf = e / kRpPairs;
        const int f = rf % F, r = rf / F;
        const int i = i0 + pi;
        float2 s = make_float2(0.f, 0.f);
        if (i < I && !(dbg & 1)) {
            const size_t ro = st.pairs ? (size_t)dump_k(r, o, R, O, true) * st.so : (size_t)r * st.sr + (size_t)o * st.so;
            const float2* src = gwp + ro + (size_t)f * st.sf + i;
            const int pend = min((grp + 1) * per, P);
            for (int p0 = grp * per; p0 < pend; p0 += kRpMaxPer) {
                float2 v[kRpMaxPer];
#pragma unroll
                for (int u = 0; u < kRpMaxPer; ++u) v[u] = (p0 + u < pend) ? src[(size_t)(p0 + u) * st.sp] : make_float2(0.f, 0.f);
#pragma unroll
                for (int u = 0; u < kRpMaxPer; ++u) { s.x += v[u].x; s.y += v[u].y; }
            }
        }
        part[grp][e] = s;
    }
    __syncthreads();
    if (dbg & 4) { if (threadIdx.x == 0 && g_zonal) g_zonal[blockIdx.x] = part[0][0].x; return; }
    const float sc = 1.f / (float)F;
    for (int e = threadIdx.x; e < nent; e += kRpThreads) {
        const int pi = e % kRpPairs, rf = e / kRpPairs;
        const int f = rf % F, r = rf / F;
        float2 s = part[0][e];
#pragma unroll
        for (int grp = 1; grp < kRpGroups; ++grp) { s.x += part[grp][e].x; s.y += part[grp][e].y; }      // fixed order
        s.x *= sc;
        s.y *= sc;
        gws[(pi * 8 + r) * 7 + f] = s;
    }
    __syncthreads();
    if (gw_out) {
        // gW_eff[o][i0 .. i0+15][r][f] is ONE contiguous block of the output: written in its own order from LDS (entry by entry from the
        // loop above the 8-byte stores lie R*F*8 bytes apart -- 11 of this launch's 16.7 us at 64 channels, band limit 3)
        const int RF = R * F;
        const int npi = min(kRpPairs, I - i0);
        float2* const dst = gw_out + ((size_t)o * I + i0) * RF;
        for (int j = threadIdx.x; j < npi * RF; j += kRpThreads) {
            const int pi = j / RF, rem = j - pi * RF;
            const int r = rem / F, f = rem - r * F;
            dst[j] = gws[(pi * 8 + r) * 7 + f];
        }
    }
    const int pair = threadIdx.x / R, r = threadIdx.x - pair * R;
    const bool valid = pair < kRpPairs && i0 + pair < I;
    if (valid && !(dbg & 2))
        param_grads_entry(gws + (pair * 8 + r) * 7, zonal, sph, phase, ftype, g_zonal, g_sph, gph + (pair * 8 + r) * (kMaxB + 1), Ifull, R, B,
                          po0 + o, pi0 + i0 + pair, r);
    __syncthreads();
    if (valid && ftype == 1 && r == 0) {
        for (int q = 0; q <= B; ++q) {
            float acc = 0.f;
            for (int rr = 0; rr < R; ++rr) acc += gph[(pair * 8 + rr) * (kMaxB + 1) + q];
            g_phase[((size_t)(po0 + o) * Ifull + pi0 + i0 + pair) * (B + 1) + q] = acc;
        }
    }
}

// gwp: the partials, P of them, entry (p, r, f, o, i) at gwp[p*sp + r*sr + f*sf + o*so + i]
int reduce_param_grads_impl(const float* gwp, size_t sp, size_t sr, size_t sf, size_t so, bool ring_pairs, int P, float* gw_eff,
                            const float* zonal, const float* sph, const float* phase, int ftype, float* g_zonal, float* g_sph,
                            float* g_phase, const fc_dims* d, hipStream_t stream, int o0, int i0, int Ifull, const float* bias_partials,
                            int bias_nparts, float* g_bias, const float* gx_parts, float* gx, size_t gx_count, size_t gx_stride, int gx_nparts,
                            const float* gx_x) {
    if (d->R > 8 || d->B > kMaxB || kRpPairs * d->R > kRpThreads) return FC_ERR_UNSUPPORTED;
    const int nit = (d->I + kRpPairs - 1) / kRpPairs;
    const RpStrides st{sp, sr, sf, so, ring_pairs ? 1 : 0};
    const int main_blocks = d->O * nit;
    static const int dbg = [] { const char* e = dev_env("FC_DEBUG_RP"); return e ? atoi(e) : 0; }();      // development: 1 no partial loads, 2 no parameter chain
    const int bias_blocks = (bias_partials && bias_nparts > 0 && g_bias) ? (d->O + kRpThreads / 64 - 1) / (kRpThreads / 64) : 0;
    const bool sum_gx = gx_parts && gx && gx_nparts > 1;
    const bool from_slices = gx_parts && gx && gx_x && gx_nparts < 0;      // (gx_nparts = -band limit: gx from the gxt slices, one complex number per thread)
    const size_t gx_floats = gx_count * 2, gx_count4 = from_slices ? gx_count : sum_gx ? (gx_floats + 3) / 4 : 0;
    const int gx_blocks = (int)((gx_count4 + kRpThreads - 1) / kRpThreads);
    hipLaunchKernelGGL(fc_reduce_param_grads_kernel, dim3(main_blocks + bias_blocks + gx_blocks), dim3(kRpThreads), 0, stream,
                       reinterpret_cast<const float2*>(gwp), st, P, reinterpret_cast<float2*>(gw_eff), zonal, sph, phase, ftype, g_zonal, g_sph,
                       g_phase, d->O, d->I, d->R, d->B, o0, i0, Ifull > 0 ? Ifull : d->I, bias_partials, bias_nparts, g_bias, main_blocks,
                       reinterpret_cast<const f32x4*>(gx_parts), reinterpret_cast<f32x4*>(gx), gx_count4, gx_stride / 2, gx_nparts,
                       gx_floats % 4, main_blocks + bias_blocks, reinterpret_cast<const float2*>(gx_x), dbg);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int filter_param_grads_impl(const float* gw_eff, const float* zonal, const float* sph, const float* phase, int ftype,
                            float* g_zonal, float* g_sph, float* g_phase, const fc_dims* d, hipStream_t stream) {
    if (d->R > 8 || d->B > kMaxB) return FC_ERR_UNSUPPORTED;
    const int total = d->O * d->I;
    hipLaunchKernelGGL(fc_filter_param_grads_kernel, dim3((total + kGradPairs - 1) / kGradPairs), dim3(kGradPairs * d->R), 0, stream,
                       reinterpret_cast<const float2*>(gw_eff), zonal, sph, phase, ftype, g_zonal, g_sph, g_phase, d->O, d->I,
                       d->R, d->B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc
