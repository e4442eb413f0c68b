// Filter packing: W_eff (O,I,R,F) complex64 -> the two MFMA operand images (see fieldconv_hip.h).
// W_eff itself is what reference nn/field_conv.py:10-33 assembles; the 1/(2B+1) of those lines is
// folded into the packed values.
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

// fwd image: [F][2][OP][KPf], k = r*I + i, value W/F.   bwd image: [F][2][IP][KPb], k = r*O + o, conj(W)/F.
__global__ void fc_pack_filter_kernel(const float2* __restrict__ w, float* __restrict__ fwd, float* __restrict__ bwd,
                                      int O, int I, int R, int F, int OP, int KPf, int IP, int KPb) {
    const size_t nf = (size_t)F * 2 * OP * KPf;
    const size_t nb = (size_t)F * 2 * IP * KPb;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float sc = 1.f / (float)F;
    if (idx < nf) {
        const int k = idx % KPf;
        const int o = (idx / KPf) % OP;
        const int pl = (idx / ((size_t)KPf * OP)) % 2;
        const int f = idx / ((size_t)KPf * OP * 2);
        float v = 0.f;
        if (o < O && k < R * I) {
            const int r = k / I, i = k - r * I;
            const float2 c = w[(((size_t)o * I + i) * R + r) * F + f];
            v = (pl == 0 ? c.x : c.y) * sc;
        }
        fwd[idx] = v;
    } else if (idx < nf + nb) {
        const size_t j = idx - nf;
        const int k = j % KPb;
        const int i = (j / KPb) % IP;
        const int pl = (j / ((size_t)KPb * IP)) % 2;
        const int f = j / ((size_t)KPb * IP * 2);
        float v = 0.f;
        if (i < I && k < R * O) {
            const int r = k / O, o = k - r * O;
            const float2 c = w[(((size_t)o * I + i) * R + r) * F + f];
            v = (pl == 0 ? c.x : -c.y) * sc;
        }
        bwd[j] = v;
    }
}

int pack_filter_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, hipStream_t stream) {
    const int F = 2 * d->B + 1;
    const int OP = round_up(d->O, 16), KPf = round_up(d->R * d->I, 16);
    const int IP = round_up(d->I, 16), KPb = round_up(d->R * d->O, 16);
    const size_t total = (size_t)F * 2 * OP * KPf + (size_t)F * 2 * IP * KPb;
    hipLaunchKernelGGL(fc_pack_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(w_eff), wpk_fwd, wpk_bwd, d->O, d->I, d->R, F, OP, KPf, IP, KPb);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc
