// FieldConv forward, fp32-MFMA instantiation and the mode dispatch (kernels: fc_forward_kernels.hpp).
#include "fc_forward_kernels.hpp"

namespace fc {

template int forward_impl_mode<false>(const float*, const float*, const fc_csr*, const float*, float*, const fc_dims*, int,
                                      void*, size_t, const fc_epilogue*, hipStream_t);
extern template int forward_impl_mode<true>(const float*, const float*, const fc_csr*, const float*, float*, const fc_dims*,
                                            int, void*, size_t, const fc_epilogue*, hipStream_t);

int forward_impl(const float* x, const float* sten, const fc_csr* g, const float* wpk, float* y, const fc_dims* d,
                 int kind, void* ws, size_t ws_bytes, const fc_epilogue* epi, hipStream_t stream) {
    if (kind != 0 && forward_ring_fits(d)) return forward_ring_impl(x, sten, g, wpk, y, d, kind, ws, ws_bytes, epi, stream);
    return halves_of(d) ? forward_impl_mode<true>(x, sten, g, wpk, y, d, kind, ws, ws_bytes, epi, stream)
                        : forward_impl_mode<false>(x, sten, g, wpk, y, d, kind, ws, ws_bytes, epi, stream);
}

size_t forward_workspace_bytes(const fc_dims* d, int kind) { return forward_workspace_bytes_impl(d, kind); }

bool forward_fits(const fc_dims* d) {
    const MmaGeom g = make_mma_geom(d->O, d->R, d->I, halves_of(d));
    return forward_lds_floats(g, 1) * sizeof(float) + (size_t)kWaves * kRingChunks * 1024 <= kMaxLds;
}

}  // namespace fc
