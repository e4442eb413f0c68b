// Internal declarations shared by the .hip translation units of libfieldconv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fieldconv_hip.h"

// (n_rings, band_limit) pairs with compiled kernels: every n_rings in 2..8 for band limits 1..3 (the
// reference's notebooks use (6,1), (6,2), (6,3)).  Anything else returns FC_ERR_UNSUPPORTED
// (fc_supported() == 0).
#define FC_FOR_EACH_SHAPE(X) \
    X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(7, 1) X(8, 1) \
    X(2, 2) X(3, 2) X(4, 2) X(5, 2) X(6, 2) X(7, 2) X(8, 2) \
    X(2, 3) X(3, 3) X(4, 3) X(5, 3) X(6, 3) X(7, 3) X(8, 3)

namespace fc {

constexpr size_t kMaxLds = 160 * 1024;
constexpr int kDefaultCUs = 256;     // MI355X: 8 XCDs x 32 CUs (used when no device can be queried, e.g. size queries on a CPU-only box)
constexpr int kMaxDevices = 16;
// Compute units of the current device (hipDeviceAttributeMultiprocessorCount; a partitioned or partial-CU device reports
// fewer than 256): sizes the persistent grids, the edge split and the ring-major threshold.  Queried once per device.
inline int num_cus() {
    static int cached[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return kDefaultCUs;
    if (dev < kMaxDevices && cached[dev] > 0) return cached[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = kDefaultCUs;
    if (dev < kMaxDevices) cached[dev] = n;
    return n;
}
constexpr int kMaxChannels = 64;     // one channel per lane in the gather phases

// kind: 0 dense stencil rows, 1 factored records, 2 geometric-phase records
// ws: optional scratch of forward_workspace_bytes(d, kind) bytes that lets several workgroups share a tile on small meshes
int forward_impl(const float* x, const float* sten, const fc_csr* g, const float* wpk, float* y,
                 const fc_dims* d, int kind, void* ws, size_t ws_bytes, const fc_epilogue* epi, hipStream_t stream);
size_t forward_workspace_bytes(const fc_dims* d, int kind);
size_t backward_workspace_bytes(const fc_dims* d);
// (defer_gx_sum: when tiles are shared -- edge parts, frequency groups -- the partial gx arrays stay in the workspace and the launch that
//  finishes the pass adds them: backward_finish_params_impl's gx_deferred)
int backward_data_impl(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk, float* gx,
                       void* ws, size_t ws_bytes, const fc_dims* d, bool factored, hipStream_t stream, bool defer_gx_sum = false);
// (factored: the launch is record-driven -- with plan_stream() the H-streaming arrangement has done the filter kernel's work inside
//  backward_data_impl, and the partials lie k = o*R + r)
int backward_filter_impl(const float* x, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream, bool factored = false);
bool backward_streams(const fc_dims* d, bool factored);
int backward_stream_impl(const float* x, const float* gy, const float* rec, const fc_csr* g, const float* wpk, float* gx, void* ws,
                         size_t ws_bytes, const fc_dims* d, hipStream_t stream, int stages);
int backward_finish_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream, bool factored = false);
// records != 0: images for the record-driven entry points (fc_forward_factored / _geometric, fc_backward_fused)
int pack_filter_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, int records, hipStream_t stream);
int pack_filter_params_impl(const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd,
                            float* wpk_bwd, const fc_dims* d, int records, hipStream_t stream);
// the filter images of TWO layers from one launch (an FCResNetBlock's two convolutions)
int pack_filter_params_pair_impl(const fc_filter_params& f0, float* fwd0, float* bwd0, const fc_dims* d0, const fc_filter_params& f1,
                                 float* fwd1, float* bwd1, const fc_dims* d1, int records, hipStream_t stream);
size_t packed_filter_floats_fwd(const fc_dims* d, int records);
size_t packed_filter_floats_bwd(const fc_dims* d, int records);

// ring-major record kernels (fc_forward_ring.hpp)
bool ring_enabled(const fc_dims* d);
bool forward_ring_fits(const fc_dims* d);
size_t packed_ring_image_floats(int M, int F, int channels, int R, int halves);
int forward_ring_impl(const float* x, const float* rec, const fc_csr* g, const float* wpk, float* y, const fc_dims* d, int kind,
                      void* ws, size_t ws_bytes, const fc_epilogue* epi, hipStream_t stream);
int filter_param_grads_impl(const float* gw_eff, const float* zonal, const float* sph, const float* phase, int ftype,
                            float* g_zonal, float* g_sph, float* g_phase, const fc_dims* d, hipStream_t stream);

// reduction of the filter-gradient partials fused with the parameter-gradient chain (fc_pack.hip); *_finish_params_impl
// pick the partial layout of their kernel family
// (o0, i0, Ifull: the filter is the block [o0, o0 + O) x [i0, i0 + I) of parameter tensors with Ifull input channels; whole
// layer: 0, 0, 0)
// (ring_pairs: the partials' k index is dump_k(r, o) * so instead of r * sr + o * so)
// (bias_partials / bias_nparts / g_bias: fc_filter_params' rider -- extra workgroups of the same launch sum the modReLU's bias-gradient
//  partials, bit-identically to tangent_nonlin_gb_reduce_kernel; gx_parts ...: a second rider, the deferred sum of the backward data
//  kernel's partial gx arrays -- gx_count complex numbers, gx_stride apart, fc_sum_parts_kernel's arithmetic)
int reduce_param_grads_impl(const float* gwp, size_t sp, size_t sr, size_t sf, size_t so, bool ring_pairs, int P, float* gw_eff,
                            const float* zonal, const float* sph, const float* phase, int ftype, float* g_zonal, float* g_sph,
                            float* g_phase, const fc_dims* d, hipStream_t stream, int o0 = 0, int i0 = 0, int Ifull = 0,
                            const float* bias_partials = nullptr, int bias_nparts = 0, float* g_bias = nullptr,
                            const float* gx_parts = nullptr, float* gx = nullptr, size_t gx_count = 0, size_t gx_stride = 0, int gx_nparts = 0,
                            const float* gx_x = nullptr);
// the rider as a launch of its own (fc_pointwise.hip): g_bias[c] = fixed-order sum over p of partials[p][c]
int bias_partials_reduce_impl(const float* partials, int nparts, int C, float* g_bias, hipStream_t stream);

// k index of entry (ring r, channel o) in a row of the kept H slabs, hence of the filter-gradient partials.  The split modes keep the
// rings in PAIRS -- (2p, o) and (2p + 1, o) adjacent -- so that the data kernel stores two entries with one instruction; the last ring
// of an odd count, and every ring in fp32 mode, lie ring-major.
__host__ __device__ inline int dump_k(int r, int o, int R, int O, bool pairs) {
    return (pairs && (r | 1) < R) ? (r >> 1) * 2 * O + 2 * o + (r & 1) : r * O + o;
}
int backward_finish_params_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, const fc_filter_params* fp, hipStream_t stream,
                                int o0 = 0, int i0 = 0, int Ifull = 0,
                                float* gx_deferred = nullptr, bool factored = false, const float* x_for_gx = nullptr);
int pack_filter_block_impl(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* d, int records, int o0, int i0, int Ifull,
                           hipStream_t stream);
int pack_filter_params_block_impl(const float* zonal, const float* sph, const float* phase, int ftype, float* wpk_fwd, float* wpk_bwd,
                                  const fc_dims* d, int records, int o0, int i0, int Ifull, hipStream_t stream);

bool shape_compiled(int R, int B);
// the record-driven kernels form 32-bit row offsets with a 24-bit multiply: N < 2^24 and N * channels * 8 < 4 GiB
bool rows_fit_32bit(const fc_dims* d);
// one-line descriptions of the kernels a launch with these dims selects (fc_describe_kernels)
void describe_forward(const fc_dims* d, int kind, char* buf, size_t n);
void describe_backward(const fc_dims* d, int records, char* buf, size_t n);

// TangentLin's backward pass (fc_pointwise.hip) with an optional addend of the input gradient (may be gx itself): the block-level
// entry points (fc_blocks.hip) add the residual branch's input gradient to the convolution's inside this launch
int tangent_lin_backward_impl(const float* x, const float* gy, const float* re_w, const float* im_w, float* gx, const float* gx_addend,
                              float* g_re, float* g_im, void* workspace, size_t workspace_bytes, int N, int I, int O, hipStream_t s);

// FCPrecomp's area sums total[dst] += w[src] over the kept edges (csrc/fc_precomp.hip), shared with the fused build
int precomp_area_sums(const int64_t* edges, const int32_t* keep, const float* w, float* total, int N, int E, hipStream_t s);

// development: device buffer that receives in-kernel time stamps (fc_debug_stamp_buffer), or nullptr
unsigned long long* debug_stamp_buffer();

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised -- once per
// function and device, not per launch (the call costs a few microseconds of host time).  `done`: a static flag array of
// the calling launcher (one per kernel instantiation).
inline bool allow_full_lds(const void* fn, size_t lds_bytes, bool (&done)[kMaxDevices]) {
    if (lds_bytes <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 0 && dev < kMaxDevices && done[dev]) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds) != hipSuccess) return false;
    if (dev >= 0 && dev < kMaxDevices) done[dev] = true;
    return true;
}
// LDS budget of the factored kernels in the current MFMA mode (the dense kernels need less)
bool forward_fits(const fc_dims* d);
bool backward_fits(const fc_dims* d);

}  // namespace fc
