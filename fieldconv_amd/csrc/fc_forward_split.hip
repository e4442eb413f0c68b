// FieldConv forward, split-half MFMA instantiation (kernels: fc_forward_kernels.hpp).
#include "fc_forward_kernels.hpp"

namespace fc {

template int forward_impl_mode<true>(const float*, const float*, const fc_csr*, const float*, float*, const fc_dims*, int,
                                     void*, size_t, const fc_epilogue*, hipStream_t);

}  // namespace fc
