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
    float2 c;
    if (f == B) c = make_float2(zonal[oir], 0.f);
    else if (f > B) { const float* p = sph + (oir * B + (f - B - 1)) * 2; c = make_float2(p[0], p[1]); }
    else { const float* p = sph + (oir * B + (B - 1 - f)) * 2; c = make_float2(p[0], -p[1]); }
    if (ftype == 1) {
        const int q = f >= B ? f - B : B - f;
        float s, co;
        sincosf(phase[((size_t)o * I + i) * (B + 1) + q], &s, &co);
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
};

constexpr int kPackThreads = 256;
constexpr int kPackRowValues = 16;       // (f, k) entries per thread of a split row: F * KP <= 16 * 256

template <bool FROM_PARAMS>
__device__ __forceinline__ float2 packed_value(const float2* __restrict__ w, const float* __restrict__ zonal,
                                               const float* __restrict__ sph, const float* __restrict__ phase,
                                               const PackArgs& a, bool is_bwd, int m, int k, int f) {
    const int inner = is_bwd ? a.O : a.I;
    const int r = k / inner, c = k - r * inner;
    const int o = is_bwd ? c : m, i = is_bwd ? m : c;
    if (o >= a.O || i >= a.I || r >= a.R) return make_float2(0.f, 0.f);
    const float2 v = FROM_PARAMS ? filter_entry(zonal, sph, phase, a.ftype, a.B, a.R, a.I, o, i, r, f)
                                 : w[(((size_t)o * a.I + i) * a.R + r) * a.F + f];
    const float sc = 1.f / (float)a.F;
    return make_float2(v.x * sc, (is_bwd ? -v.y : v.y) * sc);
}

template <bool FROM_PARAMS>
__global__ __launch_bounds__(kPackThreads) void fc_pack_filter_kernel(const float2* __restrict__ w, const float* __restrict__ zonal,
                                                                      const float* __restrict__ sph,
                                                                      const float* __restrict__ phase, float* __restrict__ fwd,
                                                                      float* __restrict__ bwd, const PackArgs a) {
    const bool is_bwd = blockIdx.x >= a.blocks_f;
    const unsigned blk = is_bwd ? blockIdx.x - a.blocks_f : blockIdx.x;
    const MmaGeom& g = is_bwd ? a.gb : a.gf;
    float* const img = is_bwd ? bwd : fwd;
    if (!g.split) {
        const size_t j = (size_t)blk * kPackThreads + threadIdx.x;
        if (j >= (size_t)a.F * 2 * g.MP * g.KP) return;
        const int k = j % g.KP;
        const int m = (j / g.KP) % g.MP;
        const int pl = (j / ((size_t)g.KP * g.MP)) % 2;
        const int f = j / ((size_t)g.KP * g.MP * 2);
        const float2 v = packed_value<FROM_PARAMS>(w, zonal, sph, phase, a, is_bwd, m, k, f);
        img[j] = pl == 0 ? v.x : v.y;
        return;
    }
    // split image, row m = blk
    __shared__ float red[kPackThreads / kWave];
    const int m = blk;
    const int total = a.F * g.KP;
    float2 vals[kPackRowValues];
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < kPackRowValues; ++u) {
        const int idx = u * kPackThreads + threadIdx.x;
        vals[u] = make_float2(0.f, 0.f);
        if (idx < total) {
            const int f = idx / g.KP, k = idx - f * g.KP;
            vals[u] = packed_value<FROM_PARAMS>(w, zonal, sph, phase, a, is_bwd, m, k, f);
        }
        mx = fmaxf(mx, fmaxf(fabsf(vals[u].x), fabsf(vals[u].y)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float scale, inv;
    split_scale(mx, scale, inv);
    if (threadIdx.x == 0) img[m] = inv;
    _Float16* const planes = reinterpret_cast<_Float16*>(img + g.MP);
    const size_t plane = (size_t)g.MP * g.KP;
#pragma unroll
    for (int u = 0; u < kPackRowValues; ++u) {
        const int idx = u * kPackThreads + threadIdx.x;
        if (idx < total) {
            const int f = idx / g.KP, k = idx - f * g.KP;
            _Float16 rh, rl, ih, il;
            split_halves(vals[u].x * scale, rh, rl);
            split_halves(vals[u].y * scale, ih, il);
            _Float16* p = planes + (size_t)f * 4 * plane + (size_t)m * g.KP + k;
            p[0] = rh;
            p[plane] = rl;
            p[2 * plane] = ih;
            p[3 * plane] = il;
        }
    }
}

static unsigned pack_blocks(const MmaGeom& g, int F) {
    return g.split ? (unsigned)g.MP : (unsigned)(((size_t)F * 2 * g.MP * g.KP + kPackThreads - 1) / kPackThreads);
}

template <bool FROM_PARAMS>
static int launch_pack(const float* w_eff, const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd,
                       float* wpk_bwd, const fc_dims* d, hipStream_t stream) {
    PackArgs a;
    a.O = d->O; a.I = d->I; a.R = d->R; a.B = d->B; a.F = 2 * d->B + 1; a.ftype = ftype;
    a.gf = make_mma_geom(d->O, d->R * d->I, split_mode());
    a.gb = make_mma_geom(d->I, d->R * d->O, split_mode());
    if ((a.gf.split && a.F * a.gf.KP > kPackRowValues * kPackThreads) || (a.gb.split && a.F * a.gb.KP > kPackRowValues * kPackThreads))
        return FC_ERR_UNSUPPORTED;
    a.blocks_f = pack_blocks(a.gf, a.F);
    const unsigned blocks = a.blocks_f + pack_blocks(a.gb, a.F);
    hipLaunchKernelGGL(fc_pack_filter_kernel<FROM_PARAMS>, dim3(blocks), dim3(kPackThreads), 0, stream,
                       reinterpret_cast<const float2*>(w_eff), zonal, sph, phase, wpk_fwd, wpk_bwd, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int pack_filter_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, hipStream_t stream) {
    return launch_pack<false>(w_eff, nullptr, nullptr, nullptr, 0, wpk_fwd, wpk_bwd, d, stream);
}

int pack_filter_params_impl(const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd,
                            float* wpk_bwd, const fc_dims* d, hipStream_t stream) {
    return launch_pack<true>(nullptr, zonal, sph, phase, ftype, wpk_fwd, wpk_bwd, d, stream);
}

bool split_mode() {
    static const bool mode = [] {
        const char* e = getenv("FC_MFMA");
        return !(e && e[0] == 'f' && e[1] == '3' && e[2] == '2');
    }();
    return mode;
}

// One thread per (o,i): reads gW_eff[o,i,:,:] (R*F complex) and writes the parameter gradients
// (torch convention for a real loss: g = dL/dRe + i dL/dIm; for a real parameter p of a complex
// w(p): g_p = Re(conj(dw/dp) g_w)).
__global__ void fc_filter_param_grads_kernel(const float2* __restrict__ gw, const float* __restrict__ zonal,
                                             const float* __restrict__ sph, const float* __restrict__ phase, int ftype,
                                             float* __restrict__ g_zonal, float* __restrict__ g_sph,
                                             float* __restrict__ g_phase, int O, int I, int R, int B) {
    const int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= O * I) return;
    const int F = 2 * B + 1;
    const int o = oi / I, i = oi - o * I;
    if (ftype == 2) {
        for (int r = 0; r < R; ++r) {
            const size_t oir = (size_t)oi * R + r;
            const float2* g = gw + oir * F;
            g_zonal[oir * 2] = g[B].x;
            g_zonal[oir * 2 + 1] = g[B].y;
            for (int b = 0; b < 2 * B; ++b) {
                const float2 v = g[b < B ? b : b + 1];
                g_sph[(oir * (2 * B) + b) * 2] = v.x;
                g_sph[(oir * (2 * B) + b) * 2 + 1] = v.y;
            }
        }
        return;
    }
    // per-|m| phase gradient accumulators (ftype 1): g_phase[q] = sum_{f: |f-B| = q} Re(conj(i P_f) gP_f),
    // gP_f = sum_r gW[r,f] conj(coeff[r,f])
    for (int q = 0; q <= B; ++q) {
        float acc = 0.f;
        float s = 0.f, co = 1.f;
        if (ftype == 1) sincosf(phase[(size_t)oi * (B + 1) + q], &s, &co);
        const float2 P = make_float2(co, s);
        for (int sign = 0; sign < (q == 0 ? 1 : 2); ++sign) {
            const int f = sign == 0 ? B + q : B - q;
            float2 gP = make_float2(0.f, 0.f);
            for (int r = 0; r < R; ++r) {
                const size_t oir = (size_t)oi * R + r;
                const float2 g = gw[oir * F + f];
                const float2 coeff = filter_entry(zonal, sph, phase, 0, B, R, I, o, i, r, f);   // without the phase
                const float2 t = cmul_conj(g, coeff);
                gP.x += t.x;
                gP.y += t.y;
                // coefficient gradient g_coeff = gW conj(P)
                const float2 gc = cmul_conj(g, P);
                if (q == 0) g_zonal[oir] = gc.x;
                else if (sign == 0) {                         // f = B + q: sph[b = q-1] direct
                    g_sph[(oir * B + (q - 1)) * 2] = gc.x;
                    g_sph[(oir * B + (q - 1)) * 2 + 1] = gc.y;
                } else {                                      // f = B - q: conj(sph[b = q-1]) -> add conj(g_coeff)
                    g_sph[(oir * B + (q - 1)) * 2] += gc.x;
                    g_sph[(oir * B + (q - 1)) * 2 + 1] -= gc.y;
                }
            }
            // d/dphi exp(i phi) = i P ; Re(conj(i P) gP) = Re((-i conj(P)) gP) = Im(conj(P) gP)... expanded:
            acc += (P.x * gP.y - P.y * gP.x);
        }
        if (ftype == 1) g_phase[(size_t)oi * (B + 1) + q] = acc;
    }
}

int filter_param_grads_impl(const float* gw_eff, const float* zonal, const float* sph, const float* phase, int ftype,
                            float* g_zonal, float* g_sph, float* g_phase, const fc_dims* d, hipStream_t stream) {
    const int total = d->O * d->I;
    hipLaunchKernelGGL(fc_filter_param_grads_kernel, dim3((total + 127) / 128), dim3(128), 0, stream,
                       reinterpret_cast<const float2*>(gw_eff), zonal, sph, phase, ftype, g_zonal, g_sph, g_phase, d->O, d->I,
                       d->R, d->B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc
