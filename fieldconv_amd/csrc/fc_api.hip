// extern "C" surface of libfieldconv_hip.so; see include/fieldconv_hip.h for the contract.
#include <stdio.h>
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

bool shape_compiled(int R, int B) {
#define FC_CASE(RR, BB) if (R == RR && B == BB) return true;
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    return false;
}

static bool dims_valid(const fc_dims* d) {
    return d && d->N > 0 && d->E >= 0 && d->I > 0 && d->O > 0 && d->R > 0 && d->B >= 0 && d->mode >= FC_MFMA_SPLIT_F16 && d->mode <= FC_MFMA_F16;
}

static bool dims_supported(const fc_dims* d) {
    if (!dims_valid(d)) return false;
    if (!shape_compiled(d->R, d->B)) return false;
    if (d->I > kMaxChannels || d->O > kMaxChannels) return false;
    // slab, partial sums and record ring must fit the CU's LDS (e.g. 8 rings x 57..64 channels do not in split mode:
    // the callers run such layers in narrower channel blocks, which the operator's linearity allows)
    return forward_fits(d) && backward_fits(d);
}

// The factored kernels address feature rows with 32-bit byte offsets formed by a 24-bit multiply: N * C * 8 must stay
// below 4 GiB (8 million vertices at 64 channels) and N below 2^24.
bool rows_fit_32bit(const fc_dims* d) {
    const uint64_t c = (uint64_t)(d->I > d->O ? d->I : d->O);
    return (uint64_t)d->N * c * 8 < (1ull << 32) && d->N < (1 << 24);
}

static unsigned long long* g_stamps = nullptr;
unsigned long long* debug_stamp_buffer() { return g_stamps; }

}  // namespace fc

extern "C" {

void fc_debug_stamp_buffer(void* device_buffer) { fc::g_stamps = static_cast<unsigned long long*>(device_buffer); }

int fc_abi_version(void) { return 11; }

int fc_dev_switches(void) { return fc::kDevSwitches ? 1 : 0; }

const char* fc_status_string(int s) {
    switch (s) {
        case FC_OK: return "ok";
        case FC_ERR_BAD_ARGUMENT: return "bad argument (null pointer or inconsistent dims)";
        case FC_ERR_UNSUPPORTED: return "unsupported (n_rings, band_limit, channels) for the compiled kernels";
        case FC_ERR_LAUNCH: return "HIP kernel launch failed";
        case FC_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}

int fc_supported(const fc_dims* dims) { return fc::dims_supported(dims) ? 1 : 0; }

int fc_describe_kernels(const fc_dims* dims, int32_t kind, char* buffer, size_t buffer_bytes) {
    if (!buffer || buffer_bytes < 2 || kind < 0 || kind > 2 || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims)) return FC_ERR_UNSUPPORTED;
    char fwd[256], bwd[768];
    fc::describe_forward(dims, kind, fwd, sizeof(fwd));
    fc::describe_backward(dims, kind != 0, bwd, sizeof(bwd));
    snprintf(buffer, buffer_bytes, "%s; %s", fwd, bwd);
    return FC_OK;
}

size_t fc_packed_filter_floats_fwd(const fc_dims* d, int32_t records) {
    if (!fc::dims_valid(d)) return 0;
    return fc::packed_filter_floats_fwd(d, records);
}

size_t fc_packed_filter_floats_bwd(const fc_dims* d, int32_t records) {
    if (!fc::dims_valid(d)) return 0;
    return fc::packed_filter_floats_bwd(d, records);
}

int fc_pack_filter(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* dims, int32_t records, void* stream) {
    if (!w_eff || (!wpk_fwd && !wpk_bwd) || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    return fc::pack_filter_impl(w_eff, wpk_fwd, wpk_bwd, dims, records, static_cast<hipStream_t>(stream));
}

int fc_pack_filter_params(const float* zonal, const float* spherical, const float* phase, int32_t ftype, float* wpk_fwd,
                          float* wpk_bwd, const fc_dims* dims, int32_t records, void* stream) {
    if (!zonal || !spherical || (!wpk_fwd && !wpk_bwd) || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (ftype < 0 || ftype > 2 || (ftype == 1 && !phase)) return FC_ERR_BAD_ARGUMENT;
    return fc::pack_filter_params_impl(zonal, spherical, phase, ftype, wpk_fwd, wpk_bwd, dims, records,
                                       static_cast<hipStream_t>(stream));
}

int fc_filter_param_grads(const float* gw_eff, const float* zonal, const float* spherical, const float* phase, int32_t ftype,
                          float* g_zonal, float* g_spherical, float* g_phase, const fc_dims* dims, void* stream) {
    if (!gw_eff || !zonal || !spherical || !g_zonal || !g_spherical || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (ftype < 0 || ftype > 2 || (ftype == 1 && (!phase || !g_phase))) return FC_ERR_BAD_ARGUMENT;
    return fc::filter_param_grads_impl(gw_eff, zonal, spherical, phase, ftype, g_zonal, g_spherical, g_phase, dims,
                                       static_cast<hipStream_t>(stream));
}

static bool epilogue_valid(const fc_epilogue* e) { return !e || !e->modrelu_bias || e->activated; }

int fc_forward(const float* x, const float* sten, const fc_csr* by_target, const float* wpk_fwd, float* y,
               const fc_dims* dims, const fc_epilogue* epilogue, void* stream) {
    if (!epilogue_valid(epilogue)) return FC_ERR_BAD_ARGUMENT;
    if (!x || !y || !wpk_fwd || !by_target || !by_target->rowptr || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (dims->E > 0 && (!sten || !by_target->nbr)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims)) return FC_ERR_UNSUPPORTED;
    return fc::forward_impl(x, sten, by_target, wpk_fwd, y, dims, 0, nullptr, 0, epilogue, static_cast<hipStream_t>(stream));
}

int fc_factored_record_floats(int32_t band_limit) { return band_limit >= 0 ? fc::factored_record_floats(band_limit) : 0; }

size_t fc_forward_workspace_bytes(const fc_dims* dims) {
    if (!fc::dims_valid(dims) || !fc::dims_supported(dims)) return 0;
    return fc::forward_workspace_bytes(dims, 1);
}

int fc_forward_factored(const float* x, const float* rec_t, const fc_csr* by_target, const float* wpk_fwd, float* y,
                        void* workspace, size_t workspace_bytes, const fc_dims* dims, const fc_epilogue* epilogue, void* stream) {
    if (!epilogue_valid(epilogue)) return FC_ERR_BAD_ARGUMENT;
    if (!x || !y || !wpk_fwd || !by_target || !by_target->rowptr || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (dims->E > 0 && (!rec_t || !by_target->runs)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims) || dims->R > 8 || !fc::rows_fit_32bit(dims)) return FC_ERR_UNSUPPORTED;
    return fc::forward_impl(x, rec_t, by_target, wpk_fwd, y, dims, 1, workspace, workspace_bytes, epilogue, static_cast<hipStream_t>(stream));
}

int fc_geometric_record_floats(void) { return fc::kGeoRecordFloats; }

int fc_forward_geometric(const float* x, const float* geo_t, const fc_csr* by_target, const float* wpk_fwd, float* y,
                         void* workspace, size_t workspace_bytes, const fc_dims* dims, const fc_epilogue* epilogue, void* stream) {
    if (!epilogue_valid(epilogue)) return FC_ERR_BAD_ARGUMENT;
    if (!x || !y || !wpk_fwd || !by_target || !by_target->rowptr || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (dims->E > 0 && (!geo_t || !by_target->runs)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims) || dims->R > 8 || !fc::rows_fit_32bit(dims)) return FC_ERR_UNSUPPORTED;
    return fc::forward_impl(x, geo_t, by_target, wpk_fwd, y, dims, 2, workspace, workspace_bytes, epilogue, static_cast<hipStream_t>(stream));
}

int32_t fc_records_flags(const fc_dims* dims, int32_t record_driven) {
    if (!record_driven) return 0;
    (void)dims;
    return 1;
}

size_t fc_backward_workspace_bytes(const fc_dims* dims, int32_t records) {
    if (!fc::dims_supported(dims)) return 0;
    (void)records;
    return fc::backward_workspace_bytes(dims);
}

static int check_bwd(const float* x, const float* gy, const float* sten, const fc_csr* by_source, const float* wpk_bwd,
                     float* gx, const fc_dims* dims) {
    if (!x || !gy || !gx || !wpk_bwd || !by_source || !by_source->rowptr || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (dims->E > 0 && !sten) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims)) return FC_ERR_UNSUPPORTED;
    return FC_OK;
}

int fc_backward_data(const float* x, const float* gy, const float* sten_s, const fc_csr* by_source, const float* wpk_bwd,
                     float* gx, void* workspace, size_t workspace_bytes, const fc_dims* dims, void* stream) {
    const int rc = check_bwd(x, gy, sten_s, by_source, wpk_bwd, gx, dims);
    if (rc != FC_OK) return rc;
    if (dims->E > 0 && !by_source->nbr) return FC_ERR_BAD_ARGUMENT;
    return fc::backward_data_impl(x, gy, sten_s, by_source, wpk_bwd, gx, workspace, workspace_bytes, dims, false,
                                  static_cast<hipStream_t>(stream));
}

int fc_backward_data_factored(const float* x, const float* gy, const float* rec_s, const fc_csr* by_source,
                              const float* wpk_bwd, float* gx, void* workspace, size_t workspace_bytes,
                              const fc_dims* dims, int32_t records, void* stream) {
    const int rc = check_bwd(x, gy, rec_s, by_source, wpk_bwd, gx, dims);
    if (rc != FC_OK) return rc;
    if (dims->E > 0 && !by_source->runs) return FC_ERR_BAD_ARGUMENT;
    if (dims->R > 8 || !fc::rows_fit_32bit(dims)) return FC_ERR_UNSUPPORTED;
    (void)records;
    return fc::backward_data_impl(x, gy, rec_s, by_source, wpk_bwd, gx, workspace, workspace_bytes, dims, true,
                                  static_cast<hipStream_t>(stream));
}

int fc_backward_filter(const float* x, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records, void* stream) {
    if (!x || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims)) return FC_ERR_UNSUPPORTED;
    return fc::backward_filter_impl(x, workspace, workspace_bytes, dims, static_cast<hipStream_t>(stream), records != 0);
}

int32_t fc_backward_streams(const fc_dims* dims, int32_t records) {
    return (fc::dims_supported(dims) && dims->R <= 8 && fc::rows_fit_32bit(dims) && fc::backward_streams(dims, records != 0)) ? 1 : 0;
}

int fc_backward_gather(const float* gy, const float* rec_s, const fc_csr* by_source, const float* wpk_bwd, void* workspace,
                       size_t workspace_bytes, const fc_dims* dims, void* stream) {
    if (!gy || !by_source || !by_source->rowptr || !wpk_bwd || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (dims->E > 0 && (!rec_s || !by_source->runs)) return FC_ERR_BAD_ARGUMENT;
    if (!fc_backward_streams(dims, 1)) return FC_ERR_UNSUPPORTED;
    return fc::backward_stream_impl(nullptr, gy, rec_s, by_source, wpk_bwd, nullptr, workspace, workspace_bytes, dims,
                                    static_cast<hipStream_t>(stream), 1);
}

int fc_backward_stream(const float* x, const float* wpk_bwd, float* gx, void* workspace, size_t workspace_bytes, const fc_dims* dims,
                       void* stream) {
    if (!x || !wpk_bwd || !gx || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (!fc_backward_streams(dims, 1)) return FC_ERR_UNSUPPORTED;
    return fc::backward_stream_impl(x, nullptr, nullptr, nullptr, wpk_bwd, gx, workspace, workspace_bytes, dims,
                                    static_cast<hipStream_t>(stream), 2);
}

int fc_backward_finish(float* gw_eff, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records, void* stream) {
    if (!gw_eff || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims)) return FC_ERR_UNSUPPORTED;
    return fc::backward_finish_impl(gw_eff, workspace, workspace_bytes, dims, static_cast<hipStream_t>(stream), records != 0);
}

static int check_finish_params(const fc_filter_params* params, const fc_dims* dims) {
    if (!params || !fc::dims_valid(dims)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::dims_supported(dims)) return FC_ERR_UNSUPPORTED;
    if (params->ftype < 0 || params->ftype > 2 || !params->zonal || !params->spherical || !params->g_zonal || !params->g_spherical ||
        (params->ftype == 1 && (!params->phase || !params->g_phase)))
        return FC_ERR_BAD_ARGUMENT;
    if (params->bias_partials && (params->bias_nparts <= 0 || !params->g_bias)) return FC_ERR_BAD_ARGUMENT;
    return FC_OK;
}

int fc_backward_finish_params(float* gw_eff, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records,
                              const fc_filter_params* params, void* stream) {
    {
        const int rc = check_finish_params(params, dims);
        if (rc != FC_OK) return rc;
    }
    // the partials' fixed-order sum and the parameter-gradient chain in ONE launch (FC_SPLIT_FINISH=1: the two kernels)
    static const bool split_finish = [] { const char* e = fc::dev_env("FC_SPLIT_FINISH"); return e && atoi(e) != 0; }();
    if (split_finish) {
        if (!gw_eff) return FC_ERR_BAD_ARGUMENT;        // (the two-kernel development variant passes gW_eff from one to the other)
        int rc = fc_backward_finish(gw_eff, workspace, workspace_bytes, dims, records, stream);
        if (rc != FC_OK) return rc;
        rc = fc_filter_param_grads(gw_eff, params->zonal, params->spherical, params->phase, params->ftype, params->g_zonal,
                                   params->g_spherical, params->g_phase, dims, stream);
        if (rc != FC_OK || !params->bias_partials) return rc;
        return fc::bias_partials_reduce_impl(params->bias_partials, params->bias_nparts, dims->O, params->g_bias, static_cast<hipStream_t>(stream));
    }
    return fc::backward_finish_params_impl(gw_eff, workspace, workspace_bytes, dims, params, static_cast<hipStream_t>(stream), 0, 0, 0, nullptr,
                                           records != 0);
}

int fc_backward_all(const float* x, const float* gy, const float* sten_or_rec_s, const fc_csr* by_source, int32_t records,
                    const float* wpk_bwd, float* gx, float* gw_eff, const fc_filter_params* params, void* workspace,
                    size_t workspace_bytes, const fc_dims* dims, void* stream) {
    if (!gw_eff && !params) return FC_ERR_BAD_ARGUMENT;         // with parameters gW_eff is optional: nullptr = not wanted
    // With parameters the pass ends in ONE launch (fc_backward_finish_params' kernel): when tiles are shared, the sum of the data kernel's
    // partial gx arrays rides there too instead of a launch of its own between the kernels (nobody reads gx before this call returns)
    static const bool split_finish = [] { const char* e = fc::dev_env("FC_SPLIT_FINISH"); return e && atoi(e) != 0; }();
    const bool defer = params != nullptr && !split_finish;
    int rc = check_bwd(x, gy, sten_or_rec_s, by_source, wpk_bwd, gx, dims);
    if (rc != FC_OK) return rc;
    if (params) {           // before anything is enqueued: a bad params struct must not leave gx half-finished in the workspace
        rc = check_finish_params(params, dims);
        if (rc != FC_OK) return rc;
    }
    if (records) {
        if (dims->E > 0 && !by_source->runs) return FC_ERR_BAD_ARGUMENT;
        if (dims->R > 8 || !fc::rows_fit_32bit(dims)) return FC_ERR_UNSUPPORTED;
    } else if (dims->E > 0 && !by_source->nbr) return FC_ERR_BAD_ARGUMENT;
    rc = fc::backward_data_impl(x, gy, sten_or_rec_s, by_source, wpk_bwd, gx, workspace, workspace_bytes, dims, records != 0,
                                static_cast<hipStream_t>(stream), defer);
    if (rc != FC_OK) return rc;
    rc = fc_backward_filter(x, workspace, workspace_bytes, dims, records, stream);
    if (rc != FC_OK) return rc;
    if (!params) return fc_backward_finish(gw_eff, workspace, workspace_bytes, dims, records, stream);
    if (!defer) return fc_backward_finish_params(gw_eff, workspace, workspace_bytes, dims, records, params, stream);
    return fc::backward_finish_params_impl(gw_eff, workspace, workspace_bytes, dims, params, static_cast<hipStream_t>(stream), 0, 0, 0, gx,
                                           records != 0, x);
}

int fc_forward_params(const float* x, const float* sten_or_records, const fc_csr* by_target, int32_t kind,
                      const fc_filter_params* params, float* wpk_fwd, float* wpk_bwd, float* y, void* workspace,
                      size_t workspace_bytes, const fc_dims* dims, int32_t records, const fc_epilogue* epilogue, void* stream) {
    if (!params || kind < 0 || kind > 2 || ((kind != 0) != ((records & 1) != 0))) return FC_ERR_BAD_ARGUMENT;
    const int rc = fc_pack_filter_params(params->zonal, params->spherical, params->phase, params->ftype, wpk_fwd, wpk_bwd, dims,
                                         records, stream);
    if (rc != FC_OK) return rc;
    if (kind == 2) return fc_forward_geometric(x, sten_or_records, by_target, wpk_fwd, y, workspace, workspace_bytes, dims, epilogue, stream);
    if (kind == 1) return fc_forward_factored(x, sten_or_records, by_target, wpk_fwd, y, workspace, workspace_bytes, dims, epilogue, stream);
    return fc_forward(x, sten_or_records, by_target, wpk_fwd, y, dims, epilogue, stream);
}

}  // extern "C"
