// FieldConv backward data kernel, split-half MFMA instantiation (kernel: fc_backward_kernels.hpp).
#include "fc_backward_kernels.hpp"

namespace fc {

template int backward_data_impl_mode<true>(const float*, const float*, const float*, const fc_csr*, const float*, float*, void*,
                                           size_t, const fc_dims*, bool, hipStream_t, bool);

}  // namespace fc
