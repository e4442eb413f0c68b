// FieldConv layers wider than the kernels' channel block (64 channels: one channel per lane in the gather phases), run
// NATIVELY as channel blocks (reference nn/field_conv.py:62 takes any width).  The operator is linear in the input channels and
// independent across output channels:
//
//     y[:, ob] = sum_ib conv(x[:, ib]; W[ob, ib]),
//
// so a wide layer is nob x nib launches of the ordinary kernels on contiguous channel blocks.  Everything a block launch needs
// is produced here, on the caller's stream, from ONE foreign call per pass: strided 2-D copies cut x / gy into blocks, the
// filter images are packed straight from the (o0, i0) block of the full parameter tensors, the sum over input blocks rides in
// the convolution's residual epilogue (addend = the output block itself), the sum over output blocks of the input gradient is
// the fixed-order sum of per-block partials, and the parameter gradients of a block land in their block of the full gradient
// tensors.  The gather is repeated per output block (the fused kernels contract at most 64 output channels per gather).
#include "fc_common.hpp"
#include "fc_kernels.hpp"
#include "fc_tile.hpp"

namespace fc {

struct WidePlan {
    int blk, nib, nob;
    fc_dims db;                    // dims of a full block (I = O = blk)
    size_t x_blocks, y_block, gy_blocks, gx_parts, gx_block, wpk, gw_block, conv_ws, total_fwd, total_bwd;
};

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

static bool wide_plan(const fc_dims* d, int blk, int records, WidePlan& p) {
    if (blk <= 0 || blk > kMaxChannels) return false;
    p.blk = blk;
    p.nib = (d->I + blk - 1) / blk;
    p.nob = (d->O + blk - 1) / blk;
    p.db = *d;
    p.db.I = blk;
    p.db.O = blk;
    const size_t rows = (size_t)d->N;
    p.x_blocks = align256(rows * p.nib * blk * 8);          // x cut into contiguous (N, blk) blocks (the last one narrower)
    p.y_block = align256(rows * blk * 8);
    p.gy_blocks = align256(rows * p.nob * blk * 8);
    p.gx_parts = align256(rows * blk * 8) * p.nob;         // one partial input-gradient block per output block
    p.gx_block = align256(rows * blk * 8);
    const size_t wf = packed_filter_floats_fwd(&p.db, records), wb = packed_filter_floats_bwd(&p.db, records);
    p.wpk = align256((wf > wb ? wf : wb) * sizeof(float));
    p.gw_block = align256((size_t)blk * blk * d->R * (2 * d->B + 1) * 8);
    const size_t fws = records ? forward_workspace_bytes(&p.db, 1) : 0, bws = backward_workspace_bytes(&p.db);
    p.conv_ws = align256(fws > bws ? fws : bws);
    p.total_fwd = p.x_blocks + p.y_block + p.wpk + align256(fws);
    p.total_bwd = p.x_blocks + p.gy_blocks + p.gx_parts + p.gx_block + p.wpk + p.gw_block + align256(bws);
    return true;
}

// (rows, cols) complex numbers from a matrix with `ld_src` columns per row to one with `ld_dst`
static int copy_block(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, hipStream_t stream) {
    if (rows == 0 || cols == 0) return FC_OK;
    return hipMemcpy2DAsync(dst, (size_t)ld_dst * 8, src, (size_t)ld_src * 8, (size_t)cols * 8, rows, hipMemcpyDeviceToDevice, stream) ==
                   hipSuccess
               ? FC_OK
               : FC_ERR_LAUNCH;
}

}  // namespace fc

extern "C" {

size_t fc_wide_workspace_bytes(const fc_dims* dims, int32_t block, int32_t records, int32_t backward) {
    fc::WidePlan p;
    if (!dims || dims->N <= 0 || dims->I <= 0 || dims->O <= 0 || !fc::wide_plan(dims, block, records, p)) return 0;
    return backward ? p.total_bwd : p.total_fwd;
}

int fc_forward_wide(const float* x, const float* sten_or_records, const fc_csr* by_target, int32_t kind, const fc_filter_params* params,
                    const float* w_eff, float* y, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records, int32_t block,
                    void* stream) {
    if (!x || !y || (!params == !w_eff) || !by_target || !dims || kind < 0 || kind > 2 || ((kind != 0) != (records != 0)))
        return FC_ERR_BAD_ARGUMENT;
    fc::WidePlan p;
    if (!fc::wide_plan(dims, block, records, p)) return FC_ERR_BAD_ARGUMENT;
    if (!fc_supported(&p.db)) return FC_ERR_UNSUPPORTED;
    if (!by_target->rowptr || (dims->E > 0 && (!sten_or_records || (records ? !by_target->runs : !by_target->nbr)))) return FC_ERR_BAD_ARGUMENT;
    if (records && (dims->R > 8 || !fc::rows_fit_32bit(&p.db))) return FC_ERR_UNSUPPORTED;          // the guards of the per-block entry points
    if (!workspace || workspace_bytes < p.total_fwd) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(workspace);
    float* xb = reinterpret_cast<float*>(w);
    float* yb = reinterpret_cast<float*>(w + p.x_blocks);
    float* wpk = reinterpret_cast<float*>(w + p.x_blocks + p.y_block);
    void* cws = w + p.x_blocks + p.y_block + p.wpk;
    const size_t cws_bytes = p.total_fwd - (p.x_blocks + p.y_block + p.wpk);
    const int N = dims->N, I = dims->I, O = dims->O, blk = p.blk;
    const size_t xstride = (size_t)N * blk * 2;              // floats between the x blocks
    for (int ib = 0; ib < p.nib; ++ib) {
        const int bi = (I - ib * blk) < blk ? (I - ib * blk) : blk;
        const int rc = fc::copy_block(xb + ib * xstride, bi, x + 2 * (size_t)ib * blk, I, N, bi, st);
        if (rc != FC_OK) return rc;
    }
    for (int ob = 0; ob < p.nob; ++ob) {
        const int bo = (O - ob * blk) < blk ? (O - ob * blk) : blk;
        for (int ib = 0; ib < p.nib; ++ib) {
            const int bi = (I - ib * blk) < blk ? (I - ib * blk) : blk;
            fc_dims d = *dims;
            d.I = bi;
            d.O = bo;
            int rc = params ? fc::pack_filter_params_block_impl(params->zonal, params->spherical, params->phase, params->ftype, wpk, nullptr, &d,
                                                                records, ob * blk, ib * blk, I, st)
                            : fc::pack_filter_block_impl(w_eff, wpk, nullptr, &d, records, ob * blk, ib * blk, I, st);
            if (rc != FC_OK) return rc;
            fc_epilogue epi = {ib > 0 ? yb : nullptr, nullptr, nullptr};      // the sum over the input blocks: y_block += this block's output
            const size_t fws = records ? fc::forward_workspace_bytes(&d, kind) : 0;
            rc = fc::forward_impl(xb + ib * xstride, sten_or_records, by_target, wpk, yb, &d, kind, fws && fws <= cws_bytes ? cws : nullptr,
                                  fws && fws <= cws_bytes ? fws : 0, ib > 0 ? &epi : nullptr, st);
            if (rc != FC_OK) return rc;
        }
        const int rc = fc::copy_block(y + 2 * (size_t)ob * blk, O, yb, bo, N, bo, st);
        if (rc != FC_OK) return rc;
    }
    return FC_OK;
}

int fc_backward_wide(const float* x, const float* gy, const float* sten_or_rec_s, const fc_csr* by_source, int32_t records,
                     const fc_filter_params* params, const float* w_eff, float* gw_eff, float* gx, void* workspace, size_t workspace_bytes,
                     const fc_dims* dims, int32_t block, void* stream) {
    if (!x || !gy || !gx || (!params == !w_eff) || !by_source || !dims) return FC_ERR_BAD_ARGUMENT;
    if (params ? (!params->g_zonal || !params->g_spherical) : !gw_eff) return FC_ERR_BAD_ARGUMENT;
    fc::WidePlan p;
    if (!fc::wide_plan(dims, block, records, p)) return FC_ERR_BAD_ARGUMENT;
    if (!fc_supported(&p.db)) return FC_ERR_UNSUPPORTED;
    if (!by_source->rowptr || (dims->E > 0 && (!sten_or_rec_s || (records ? !by_source->runs : !by_source->nbr)))) return FC_ERR_BAD_ARGUMENT;
    if (records && (dims->R > 8 || !fc::rows_fit_32bit(&p.db))) return FC_ERR_UNSUPPORTED;          // the guards of the per-block entry points
    if (!workspace || workspace_bytes < p.total_bwd) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(workspace);
    float* xb = reinterpret_cast<float*>(w);
    float* gyb = reinterpret_cast<float*>(w + p.x_blocks);
    float* parts = reinterpret_cast<float*>(w + p.x_blocks + p.gy_blocks);
    float* gxb = reinterpret_cast<float*>(w + p.x_blocks + p.gy_blocks + p.gx_parts);
    float* wpk = reinterpret_cast<float*>(w + p.x_blocks + p.gy_blocks + p.gx_parts + p.gx_block);
    float* gwb = reinterpret_cast<float*>(w + p.x_blocks + p.gy_blocks + p.gx_parts + p.gx_block + p.wpk);
    void* cws = w + p.x_blocks + p.gy_blocks + p.gx_parts + p.gx_block + p.wpk + p.gw_block;
    const size_t cws_bytes = p.total_bwd - (p.x_blocks + p.gy_blocks + p.gx_parts + p.gx_block + p.wpk + p.gw_block);
    const int N = dims->N, I = dims->I, O = dims->O, blk = p.blk;
    const size_t bstride = (size_t)N * blk * 2;              // floats between the x / gy blocks
    const size_t pstride = p.gx_parts / p.nob / sizeof(float);
    for (int ib = 0; ib < p.nib; ++ib) {
        const int bi = (I - ib * blk) < blk ? (I - ib * blk) : blk;
        const int rc = fc::copy_block(xb + ib * bstride, bi, x + 2 * (size_t)ib * blk, I, N, bi, st);
        if (rc != FC_OK) return rc;
    }
    for (int ob = 0; ob < p.nob; ++ob) {
        const int bo = (O - ob * blk) < blk ? (O - ob * blk) : blk;
        const int rc = fc::copy_block(gyb + ob * bstride, bo, gy + 2 * (size_t)ob * blk, O, N, bo, st);
        if (rc != FC_OK) return rc;
    }
    for (int ib = 0; ib < p.nib; ++ib) {
        const int bi = (I - ib * blk) < blk ? (I - ib * blk) : blk;
        for (int ob = 0; ob < p.nob; ++ob) {
            const int bo = (O - ob * blk) < blk ? (O - ob * blk) : blk;
            fc_dims d = *dims;
            d.I = bi;
            d.O = bo;
            int rc = params ? fc::pack_filter_params_block_impl(params->zonal, params->spherical, params->phase, params->ftype, nullptr, wpk, &d,
                                                                records, ob * blk, ib * blk, I, st)
                            : fc::pack_filter_block_impl(w_eff, nullptr, wpk, &d, records, ob * blk, ib * blk, I, st);
            if (rc != FC_OK) return rc;
            float* part = p.nob == 1 ? gxb : parts + ob * pstride;
            rc = fc::backward_data_impl(xb + ib * bstride, gyb + ob * bstride, sten_or_rec_s, by_source, wpk, part, cws, cws_bytes, &d, records != 0,
                                        st);
            if (rc != FC_OK) return rc;
            rc = fc::backward_filter_impl(xb + ib * bstride, cws, cws_bytes, &d, st, records != 0);
            if (rc != FC_OK) return rc;
            if (params) {
                fc_filter_params block_params = *params;             // (the bias rider belongs to the whole layer: after the blocks)
                block_params.bias_partials = nullptr;
                block_params.bias_nparts = 0;
                block_params.g_bias = nullptr;
                rc = fc::backward_finish_params_impl(gwb, cws, cws_bytes, &d, &block_params, st, ob * blk, ib * blk, I, nullptr, records != 0);
            } else {            // explicit filter: the block's gradient goes into its block of gw_eff (rows o, bi*R*F numbers each)
                rc = fc::backward_finish_impl(gwb, cws, cws_bytes, &d, st, records != 0);
                const int rf = dims->R * (2 * dims->B + 1);
                if (rc == FC_OK)
                    rc = fc::copy_block(gw_eff + 2 * (((size_t)ob * blk * I + (size_t)ib * blk) * rf), I * rf, gwb, bi * rf, bo, bi * rf, st);
            }
            if (rc != FC_OK) return rc;
        }
        if (p.nob > 1) {            // the sum over the output blocks, in block order
            const int rc = fc::sum_parts(parts, gxb, (size_t)N * bi, pstride / 2, p.nob, st);
            if (rc != FC_OK) return rc;
        }
        const int rc = fc::copy_block(gx + 2 * (size_t)ib * blk, I, gxb, bi, N, bi, st);
        if (rc != FC_OK) return rc;
    }
    if (params && params->bias_partials) {
        if (params->bias_nparts <= 0 || !params->g_bias) return FC_ERR_BAD_ARGUMENT;
        return fc::bias_partials_reduce_impl(params->bias_partials, params->bias_nparts, O, params->g_bias, st);
    }
    return FC_OK;
}

}  // extern "C"
