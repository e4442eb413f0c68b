// FieldConv forward, fp32-MFMA instantiation and the mode dispatch (kernels: fc_forward_kernels.hpp).
#include "fc_forward_kernels.hpp"

namespace fc {

template int forward_impl_mode<false>(const float*, const float*, const fc_csr*, const float*, float*, const fc_dims*, int,
                                      hipStream_t);
extern template int forward_impl_mode<true>(const float*, const float*, const fc_csr*, const float*, float*, const fc_dims*,
                                            int, hipStream_t);

int forward_impl(const float* x, const float* sten, const fc_csr* g, const float* wpk, float* y, const fc_dims* d,
                 int kind, hipStream_t stream) {
    return split_mode() ? forward_impl_mode<true>(x, sten, g, wpk, y, d, kind, stream)
                        : forward_impl_mode<false>(x, sten, g, wpk, y, d, kind, stream);
}

}  // namespace fc
