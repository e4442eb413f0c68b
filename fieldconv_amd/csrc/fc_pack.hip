// Filter packing for the MFMA contractions, and the parameter-gradient chain.
//   W_eff (O,I,R,F) complex64 is what reference nn/field_conv.py:10-33 (weightContrib*) assembles from
//   (zonal, spherical, phase); the 1/(2B+1) of :14,:25,:33 is folded into the packed values.
//   fc_pack_filter          packs a given W_eff
//   fc_pack_filter_params   assembles W_eff from the parameters on the fly (one kernel instead of the
//                           cat / flip / conj / polar / mul chain in torch) and packs it
//   fc_filter_param_grads   pulls dL/dW_eff back to (zonal, spherical, phase) -- the autograd twin
#include "fc_common.hpp"
#include "fc_kernels.hpp"

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

// fwd image: [F][2][OP][KPf], k = r*I + i, value W/F.   bwd image: [F][2][IP][KPb], k = r*O + o, conj(W)/F.
template <bool FROM_PARAMS>
__global__ void fc_pack_filter_kernel(const float2* __restrict__ w, const float* __restrict__ zonal,
                                      const float* __restrict__ sph, const float* __restrict__ phase, int ftype, int B,
                                      float* __restrict__ fwd, float* __restrict__ bwd, int O, int I, int R, int F, int OP,
                                      int KPf, int IP, int KPb) {
    const size_t nf = (size_t)F * 2 * OP * KPf;
    const size_t nb = (size_t)F * 2 * IP * KPb;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float sc = 1.f / (float)F;
    int o, i, r, f, pl;
    bool valid, is_bwd;
    size_t j;
    if (idx < nf) {
        is_bwd = false;
        j = idx;
        const int k = j % KPf;
        o = (j / KPf) % OP;
        pl = (j / ((size_t)KPf * OP)) % 2;
        f = j / ((size_t)KPf * OP * 2);
        r = k / I;
        i = k - r * I;
        valid = o < O && k < R * I;
    } else if (idx < nf + nb) {
        is_bwd = true;
        j = idx - nf;
        const int k = j % KPb;
        i = (j / KPb) % IP;
        pl = (j / ((size_t)KPb * IP)) % 2;
        f = j / ((size_t)KPb * IP * 2);
        r = k / O;
        o = k - r * O;
        valid = i < I && k < R * O;
    } else {
        return;
    }
    float v = 0.f;
    if (valid) {
        const float2 c = FROM_PARAMS ? filter_entry(zonal, sph, phase, ftype, B, R, I, o, i, r, f)
                                     : w[(((size_t)o * I + i) * R + r) * F + f];
        v = (pl == 0 ? c.x : (is_bwd ? -c.y : c.y)) * sc;
    }
    (is_bwd ? bwd : fwd)[j] = v;
}

static void pack_geometry(const fc_dims* d, int& F, int& OP, int& KPf, int& IP, int& KPb, size_t& total) {
    F = 2 * d->B + 1;
    OP = round_up(d->O, 16);
    KPf = round_up(d->R * d->I, 16);
    IP = round_up(d->I, 16);
    KPb = round_up(d->R * d->O, 16);
    total = (size_t)F * 2 * OP * KPf + (size_t)F * 2 * IP * KPb;
}

int pack_filter_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, hipStream_t stream) {
    int F, OP, KPf, IP, KPb;
    size_t total;
    pack_geometry(d, F, OP, KPf, IP, KPb, total);
    hipLaunchKernelGGL(fc_pack_filter_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(w_eff), nullptr, nullptr, nullptr, 0, d->B, wpk_fwd, wpk_bwd, d->O,
                       d->I, d->R, F, OP, KPf, IP, KPb);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int pack_filter_params_impl(const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd,
                            float* wpk_bwd, const fc_dims* d, hipStream_t stream) {
    int F, OP, KPf, IP, KPb;
    size_t total;
    pack_geometry(d, F, OP, KPf, IP, KPb, total);
    hipLaunchKernelGGL(fc_pack_filter_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, nullptr,
                       zonal, sph, phase, ftype, d->B, wpk_fwd, wpk_bwd, d->O, d->I, d->R, F, OP, KPf, IP, KPb);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
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
