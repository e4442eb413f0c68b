// Whole network blocks from ONE foreign call per pass (include/fieldconv_hip.h, "whole blocks").
//
//   FCResNetBlock  reference nn/fc_resnet_block.py:65-88   out = modReLU_2(res(x) + conv2(modReLU_1(conv1(x))))
//   ECHOBlock      reference nn/echo_block.py:93-94        desc = ECHO(modReLU(conv(x)))   (the MLP behind it is the reference's nn.Linear)
//   LiftBlock      reference nn/lift_block.py:53-55        out = modReLU(TransField(x))
//
// Nothing here computes: every function lays the caller's `saved` / `workspace` buffers out and enqueues the kernels of the
// per-operator entry points in the order the reference applies the operators.  What it removes is the HOST side of a block --
// sixteen foreign calls, a dozen tensor allocations and five autograd nodes per FCResNetBlock pass pair -- which on the
// reference's ~1k-vertex meshes (a new one every step, so a captured HIP graph does not apply) cost more than the GPU work.
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

static size_t blk_align(size_t v) { return (v + 255) / 256 * 256; }

struct Carver {             // hands out 256-byte aligned pieces of one caller-owned buffer
    char* base;
    size_t used = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    float* take(size_t bytes) {
        float* r = reinterpret_cast<float*>(base + used);
        used += blk_align(bytes);
        return r;
    }
};

static bool mesh_valid(const fc_mesh* m, bool need_records) {
    if (!m || m->N <= 0 || m->E < 0 || m->R <= 0 || m->B < 0 || m->kind < 0 || m->kind > 2) return false;
    if (!m->by_target || !m->by_source || !m->by_target->rowptr || !m->by_source->rowptr) return false;
    if (need_records && m->E > 0 && (!m->fwd || !m->bwd)) return false;
    return true;
}

static fc_dims conv_dims(const fc_mesh* m, int I, int O) { return fc_dims{m->N, m->E, I, O, m->R, m->B, m->mode}; }

// ---- FCResNetBlock ----------------------------------------------------------------------------------------------------------
struct ResnetPlan {
    fc_dims d1, d2;
    int records;
    size_t pre1, act1, pre2, wpk_b1, wpk_b2, saved;                    // bytes (aligned) of the pieces of `saved`
    size_t wpk_f1, wpk_f2, res_out, fwd_ws, ws_fwd;                    // forward workspace
    size_t g_pre2, g_h, g_pre1, part1, part2, gw, lin_ws, conv_ws, ws_bwd;      // backward workspace
};

static bool resnet_plan(const fc_mesh* m, const fc_resnet_block_params* p, ResnetPlan& pl) {
    if (!mesh_valid(m, true) || !p || p->C_in <= 0 || p->C_mid <= 0 || p->C_out <= 0) return false;
    pl.d1 = conv_dims(m, p->C_in, p->C_mid);
    pl.d2 = conv_dims(m, p->C_mid, p->C_out);
    if (!fc_supported(&pl.d1) || !fc_supported(&pl.d2)) return false;
    pl.records = m->kind != 0 ? 1 : 0;
    const size_t N = (size_t)m->N;
    pl.pre1 = blk_align(N * p->C_mid * 8);
    pl.act1 = pl.pre1;
    pl.pre2 = blk_align(N * p->C_out * 8);
    pl.wpk_b1 = blk_align(packed_filter_floats_bwd(&pl.d1, pl.records) * sizeof(float));
    pl.wpk_b2 = blk_align(packed_filter_floats_bwd(&pl.d2, pl.records) * sizeof(float));
    pl.saved = pl.pre1 + pl.act1 + pl.pre2 + pl.wpk_b1 + pl.wpk_b2;
    pl.wpk_f1 = blk_align(packed_filter_floats_fwd(&pl.d1, pl.records) * sizeof(float));
    pl.wpk_f2 = blk_align(packed_filter_floats_fwd(&pl.d2, pl.records) * sizeof(float));
    pl.res_out = blk_align(N * p->C_out * 8);
    const size_t f1 = pl.records ? forward_workspace_bytes(&pl.d1, m->kind) : 0, f2 = pl.records ? forward_workspace_bytes(&pl.d2, m->kind) : 0;
    pl.fwd_ws = blk_align(f1 > f2 ? f1 : f2);
    pl.ws_fwd = pl.wpk_f1 + pl.wpk_f2 + pl.res_out + pl.fwd_ws;
    pl.g_pre2 = blk_align(N * p->C_out * 8);
    pl.g_h = blk_align(N * p->C_mid * 8);
    pl.g_pre1 = pl.g_h;
    pl.part1 = blk_align(fc_tangent_nonlin_backward_workspace_bytes(m->N, p->C_mid));
    pl.part2 = blk_align(fc_tangent_nonlin_backward_workspace_bytes(m->N, p->C_out));
    pl.gw = 0;           // (the convolutions' gW_eff tensors are not wanted: fc_backward_all pulls the partial sums back to the parameters)
    pl.lin_ws = blk_align(fc_tangent_lin_backward_workspace_bytes(m->N, p->C_in, p->C_out));
    const size_t b1 = backward_workspace_bytes(&pl.d1), b2 = backward_workspace_bytes(&pl.d2);
    pl.conv_ws = blk_align(b1 > b2 ? b1 : b2);
    pl.ws_bwd = pl.g_pre2 + pl.g_h + pl.g_pre1 + pl.part1 + pl.part2 + pl.gw + pl.lin_ws + pl.conv_ws;
    return true;
}

static bool filter_params_ok(const fc_filter_params& f, bool backward) {
    if (!f.zonal || !f.spherical || f.ftype < 0 || f.ftype > 2 || (f.ftype == 1 && !f.phase)) return false;
    if (backward && (!f.g_zonal || !f.g_spherical || (f.ftype == 1 && !f.g_phase))) return false;
    return true;
}

// packed = true: the images were written by an earlier launch of this pass (pack_filter_params_pair_impl)
static int conv_forward(const float* x, const fc_mesh* m, const fc_dims* d, const fc_filter_params& f, float* wpk_f, float* wpk_b, float* y,
                        const fc_epilogue* epi, void* ws, size_t ws_bytes, int records, hipStream_t st, bool packed = false) {
    if (!packed) {
        const int rc = pack_filter_params_impl(f.zonal, f.spherical, f.phase, f.ftype, wpk_f, wpk_b, d, records, st);
        if (rc != FC_OK) return rc;
    }
    const size_t need = records ? forward_workspace_bytes(d, m->kind) : 0;
    const bool give = need != 0 && need <= ws_bytes;
    return forward_impl(x, m->fwd, m->by_target, wpk_f, y, d, m->kind, give ? ws : nullptr, give ? need : 0, epi, st);
}

}  // namespace fc

extern "C" {

size_t fc_resnet_block_saved_bytes(const fc_mesh* mesh, const fc_resnet_block_params* p) {
    fc::ResnetPlan pl;
    return fc::resnet_plan(mesh, p, pl) ? pl.saved : 0;
}

size_t fc_resnet_block_workspace_bytes(const fc_mesh* mesh, const fc_resnet_block_params* p, int32_t backward) {
    fc::ResnetPlan pl;
    if (!fc::resnet_plan(mesh, p, pl)) return 0;
    return backward ? pl.ws_bwd : pl.ws_fwd;
}

int fc_resnet_block_forward(const float* x, const fc_mesh* mesh, const fc_resnet_block_params* p, float* out, void* saved,
                            size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !fc::mesh_valid(mesh, true) || !p) return FC_ERR_BAD_ARGUMENT;
    if (!fc::filter_params_ok(p->conv1, false) || !fc::filter_params_ok(p->conv2, false) || !p->bias1 || !p->bias2 || !p->res_re || !p->res_im)
        return FC_ERR_BAD_ARGUMENT;
    fc::ResnetPlan pl;
    if (!fc::resnet_plan(mesh, p, pl)) return FC_ERR_UNSUPPORTED;
    if (!saved || saved_bytes < pl.saved || !workspace || workspace_bytes < pl.ws_fwd) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    fc::Carver sv(saved), ws(workspace);
    float* pre1 = sv.take(pl.pre1);
    float* act1 = sv.take(pl.act1);
    float* pre2 = sv.take(pl.pre2);
    float* wpk_b1 = sv.take(pl.wpk_b1);
    float* wpk_b2 = sv.take(pl.wpk_b2);
    float* wpk_f1 = ws.take(pl.wpk_f1);
    float* wpk_f2 = ws.take(pl.wpk_f2);
    float* res_out = ws.take(pl.res_out);
    void* fws = ws.take(pl.fwd_ws);
    // both convolutions' filter images (forward and backward) from one launch: the parameters are all known now
    int rc = fc::pack_filter_params_pair_impl(p->conv1, wpk_f1, wpk_b1, &pl.d1, p->conv2, wpk_f2, wpk_b2, &pl.d2, pl.records, st);
    if (rc != FC_OK) return rc;
    // h = modReLU_1(conv1(x)): the modReLU in the convolution's epilogue (pre1 = the pre-activation its VJP needs)
    fc_epilogue e1 = {nullptr, p->bias1, act1};
    rc = fc::conv_forward(x, mesh, &pl.d1, p->conv1, wpk_f1, wpk_b1, pre1, &e1, fws, pl.fwd_ws, pl.records, st, true);
    if (rc != FC_OK) return rc;
    // res(x)
    rc = fc_tangent_lin_forward(x, p->res_re, p->res_im, res_out, mesh->N, p->C_in, p->C_out, stream);
    if (rc != FC_OK) return rc;
    // out = modReLU_2(res(x) + conv2(h)): residual add and modReLU in conv2's epilogue
    fc_epilogue e2 = {res_out, p->bias2, out};
    return fc::conv_forward(act1, mesh, &pl.d2, p->conv2, wpk_f2, wpk_b2, pre2, &e2, fws, pl.fwd_ws, pl.records, st, true);
}

int fc_resnet_block_backward(const float* x, const float* g_out, const fc_mesh* mesh, const fc_resnet_block_params* p, const void* saved,
                             size_t saved_bytes, float* gx, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !g_out || !gx || !fc::mesh_valid(mesh, true) || !p) return FC_ERR_BAD_ARGUMENT;
    if (!fc::filter_params_ok(p->conv1, true) || !fc::filter_params_ok(p->conv2, true) || !p->bias1 || !p->bias2 || !p->res_re || !p->res_im ||
        !p->g_bias1 || !p->g_bias2 || !p->g_res_re || !p->g_res_im)
        return FC_ERR_BAD_ARGUMENT;
    fc::ResnetPlan pl;
    if (!fc::resnet_plan(mesh, p, pl)) return FC_ERR_UNSUPPORTED;
    if (!saved || saved_bytes < pl.saved || !workspace || workspace_bytes < pl.ws_bwd) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    fc::Carver sv(const_cast<void*>(saved)), ws(workspace);
    const float* pre1 = sv.take(pl.pre1);
    const float* act1 = sv.take(pl.act1);
    const float* pre2 = sv.take(pl.pre2);
    const float* wpk_b1 = sv.take(pl.wpk_b1);
    const float* wpk_b2 = sv.take(pl.wpk_b2);
    float* g_pre2 = ws.take(pl.g_pre2);
    float* g_h = ws.take(pl.g_h);
    float* g_pre1 = ws.take(pl.g_pre1);
    float* part1 = ws.take(pl.part1);
    float* part2 = ws.take(pl.part2);
    void* lin_ws = ws.take(pl.lin_ws);
    void* cws = ws.take(pl.conv_ws);
    const int N = mesh->N;
    // modReLU_2: g_pre2 and the bias-gradient partials, which conv2's finishing launch sums (fc_filter_params' rider)
    int rc = fc_tangent_nonlin_backward_partial(pre2, p->bias2, g_out, g_pre2, part2, pl.part2, N, p->C_out, stream);
    if (rc != FC_OK) return rc;
    fc_filter_params f2 = p->conv2;
    f2.bias_partials = part2;
    f2.bias_nparts = fc_tangent_nonlin_backward_groups(N);
    f2.g_bias = p->g_bias2;
    rc = fc_backward_all(act1, g_pre2, mesh->bwd, mesh->by_source, pl.records, wpk_b2, g_h, nullptr, &f2, cws, pl.conv_ws, &pl.d2, stream);
    if (rc != FC_OK) return rc;
    // modReLU_1 and conv1: gx = conv1's input gradient ...
    rc = fc_tangent_nonlin_backward_partial(pre1, p->bias1, g_h, g_pre1, part1, pl.part1, N, p->C_mid, stream);
    if (rc != FC_OK) return rc;
    fc_filter_params f1 = p->conv1;
    f1.bias_partials = part1;
    f1.bias_nparts = fc_tangent_nonlin_backward_groups(N);
    f1.g_bias = p->g_bias1;
    rc = fc_backward_all(x, g_pre1, mesh->bwd, mesh->by_source, pl.records, wpk_b1, gx, nullptr, &f1, cws, pl.conv_ws, &pl.d1, stream);
    if (rc != FC_OK) return rc;
    // ... plus the residual branch's (res sees the same cotangent as conv2: g_pre2), added by the TangentLin kernel itself
    return fc::tangent_lin_backward_impl(x, g_pre2, p->res_re, p->res_im, gx, gx, p->g_res_re, p->g_res_im, lin_ws, pl.lin_ws, N, p->C_in,
                                         p->C_out, st);
}

}  // extern "C"

// ---- ECHOBlock (tangent-feature half) ---------------------------------------------------------------------------------------
namespace fc {

struct EchoPlan {
    fc_dims d;
    int records, dS;
    size_t pre, act, hist, wpk_b, saved;
    size_t wpk_f, fwd_ws, ws_fwd;
    size_t g_act, g_pre, part, gw, gh, conv_ws, ws_bwd;
};

static bool echo_plan(const fc_mesh* m, const fc_echo_block_params* p, EchoPlan& pl) {
    if (!mesh_valid(m, true) || !p || p->C_in <= 0 || p->n_des <= 0) return false;
    pl.dS = fc_echo_hist_dim(p->n_bins);
    if (pl.dS == 0) return false;
    pl.d = conv_dims(m, p->C_in, p->n_des);
    if (!fc_supported(&pl.d)) return false;
    pl.records = m->kind != 0 ? 1 : 0;
    const size_t N = (size_t)m->N;
    pl.pre = blk_align(N * p->n_des * 8);
    pl.act = pl.pre;
    pl.hist = blk_align(N * p->n_des * pl.dS * 8);
    pl.wpk_b = blk_align(packed_filter_floats_bwd(&pl.d, pl.records) * sizeof(float));
    pl.saved = pl.pre + pl.act + pl.hist + pl.wpk_b;
    pl.wpk_f = blk_align(packed_filter_floats_fwd(&pl.d, pl.records) * sizeof(float));
    pl.fwd_ws = blk_align(pl.records ? forward_workspace_bytes(&pl.d, m->kind) : 0);
    pl.ws_fwd = pl.wpk_f + pl.fwd_ws;
    pl.g_act = pl.pre;
    pl.g_pre = pl.pre;
    pl.part = blk_align(fc_tangent_nonlin_backward_workspace_bytes(m->N, p->n_des));
    pl.gw = 0;
    pl.gh = pl.hist;
    pl.conv_ws = blk_align(backward_workspace_bytes(&pl.d));
    pl.ws_bwd = pl.g_act + pl.g_pre + pl.part + pl.gw + pl.gh + pl.conv_ws;
    return true;
}

}  // namespace fc

extern "C" {

size_t fc_echo_block_saved_bytes(const fc_mesh* mesh, const fc_echo_block_params* p) {
    fc::EchoPlan pl;
    return fc::echo_plan(mesh, p, pl) ? pl.saved : 0;
}

size_t fc_echo_block_workspace_bytes(const fc_mesh* mesh, const fc_echo_block_params* p, int32_t backward) {
    fc::EchoPlan pl;
    if (!fc::echo_plan(mesh, p, pl)) return 0;
    return backward ? pl.ws_bwd : pl.ws_fwd;
}

int fc_echo_block_forward(const float* x, const fc_mesh* mesh, const float* ln_t, const float* wxp_t, const fc_echo_block_params* p,
                          float* desc, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !desc || !fc::mesh_valid(mesh, true) || !p || !fc::filter_params_ok(p->conv, false) || !p->bias) return FC_ERR_BAD_ARGUMENT;
    if (mesh->E > 0 && (!ln_t || !wxp_t || !mesh->by_target->nbr)) return FC_ERR_BAD_ARGUMENT;
    fc::EchoPlan pl;
    if (!fc::echo_plan(mesh, p, pl)) return FC_ERR_UNSUPPORTED;
    if (!saved || saved_bytes < pl.saved || !workspace || workspace_bytes < pl.ws_fwd) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    fc::Carver sv(saved), ws(workspace);
    float* pre = sv.take(pl.pre);
    float* act = sv.take(pl.act);
    float* hist = sv.take(pl.hist);
    float* wpk_b = sv.take(pl.wpk_b);
    float* wpk_f = ws.take(pl.wpk_f);
    void* fws = ws.take(pl.fwd_ws);
    fc_epilogue e = {nullptr, p->bias, act};
    int rc = fc::conv_forward(x, mesh, &pl.d, p->conv, wpk_f, wpk_b, pre, &e, fws, pl.fwd_ws, pl.records, st);
    if (rc != FC_OK) return rc;
    return fc_echo_forward(act, ln_t, wxp_t, mesh->by_target, hist, desc, mesh->N, mesh->E, p->n_des, p->n_bins, stream);
}

int fc_echo_block_backward(const float* x, const float* g_desc, const fc_mesh* mesh, const float* ln_s, const float* wxp_s,
                           const fc_echo_block_params* p, const void* saved, size_t saved_bytes, float* gx, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!x || !g_desc || !gx || !fc::mesh_valid(mesh, true) || !p || !fc::filter_params_ok(p->conv, true) || !p->bias || !p->g_bias)
        return FC_ERR_BAD_ARGUMENT;
    if (mesh->E > 0 && (!ln_s || !wxp_s || !mesh->by_source->nbr)) return FC_ERR_BAD_ARGUMENT;
    fc::EchoPlan pl;
    if (!fc::echo_plan(mesh, p, pl)) return FC_ERR_UNSUPPORTED;
    if (!saved || saved_bytes < pl.saved || !workspace || workspace_bytes < pl.ws_bwd) return FC_ERR_WORKSPACE;
    fc::Carver sv(const_cast<void*>(saved)), ws(workspace);
    const float* pre = sv.take(pl.pre);
    const float* act = sv.take(pl.act);
    const float* hist = sv.take(pl.hist);
    const float* wpk_b = sv.take(pl.wpk_b);
    float* g_act = ws.take(pl.g_act);
    float* g_pre = ws.take(pl.g_pre);
    float* part = ws.take(pl.part);
    float* gh = ws.take(pl.gh);
    void* cws = ws.take(pl.conv_ws);
    const int N = mesh->N;
    int rc = fc_echo_backward(act, ln_s, wxp_s, mesh->by_source, hist, g_desc, g_act, gh, N, mesh->E, p->n_des, p->n_bins, stream);
    if (rc != FC_OK) return rc;
    rc = fc_tangent_nonlin_backward_partial(pre, p->bias, g_act, g_pre, part, pl.part, N, p->n_des, stream);
    if (rc != FC_OK) return rc;
    fc_filter_params f = p->conv;
    f.bias_partials = part;
    f.bias_nparts = fc_tangent_nonlin_backward_groups(N);
    f.g_bias = p->g_bias;
    return fc_backward_all(x, g_pre, mesh->bwd, mesh->by_source, pl.records, wpk_b, gx, nullptr, &f, cws, pl.conv_ws, &pl.d, stream);
}

}  // extern "C"

// ---- LiftBlock --------------------------------------------------------------------------------------------------------------
namespace fc {

struct LiftPlan {
    size_t pre, ang, mag, s1sum, saved;
    size_t g_pre, nl_ws, tf_ws, ws_bwd;
};

static bool lift_plan(const fc_mesh* m, const fc_lift_block_params* p, LiftPlan& pl) {
    if (!m || m->N <= 0 || m->E < 0 || m->R <= 0 || !m->by_target || !m->by_source || !p || p->C_in <= 0 || p->C_out <= 0) return false;
    const size_t N = (size_t)m->N;
    pl.pre = blk_align(N * p->C_out * 8);
    pl.ang = blk_align(N * p->C_in * m->R * 8);
    pl.mag = blk_align(N * p->C_in * m->R * 4);
    pl.s1sum = blk_align(N * m->R * 8);
    pl.saved = pl.pre + pl.ang + pl.mag + pl.s1sum;
    pl.g_pre = pl.pre;
    pl.nl_ws = blk_align(fc_tangent_nonlin_backward_workspace_bytes(m->N, p->C_out));
    pl.tf_ws = blk_align(fc_trans_field_backward_workspace_bytes(m->N, p->C_in, p->C_out, m->R));
    pl.ws_bwd = pl.g_pre + pl.nl_ws + pl.tf_ws;
    return true;
}

}  // namespace fc

extern "C" {

size_t fc_lift_block_saved_bytes(const fc_mesh* mesh, const fc_lift_block_params* p) {
    fc::LiftPlan pl;
    return fc::lift_plan(mesh, p, pl) ? pl.saved : 0;
}

size_t fc_lift_block_workspace_bytes(const fc_mesh* mesh, const fc_lift_block_params* p, int32_t backward) {
    fc::LiftPlan pl;
    if (!fc::lift_plan(mesh, p, pl)) return 0;
    return backward ? pl.ws_bwd : 0;
}

int fc_lift_block_forward(const float* x, const float* lift_sten, int32_t sten_stride, const fc_mesh* mesh, const int64_t* slot_to_edge_t,
                          const fc_lift_block_params* p, float* out, void* saved, size_t saved_bytes, void* stream) {
    fc::LiftPlan pl;
    if (!x || !out || !fc::lift_plan(mesh, p, pl) || !p->zonal_ang || !p->zonal_mag || !p->phase || !p->bias) return FC_ERR_BAD_ARGUMENT;
    if (!saved || saved_bytes < pl.saved) return FC_ERR_WORKSPACE;
    fc::Carver sv(saved);
    float* pre = sv.take(pl.pre);
    float* ang = sv.take(pl.ang);
    float* mag = sv.take(pl.mag);
    float* s1sum = sv.take(pl.s1sum);
    int rc = fc_trans_field_forward(x, lift_sten, mesh->by_target, slot_to_edge_t, p->zonal_ang, p->zonal_mag, p->phase, pre, ang, mag, s1sum,
                                    mesh->N, mesh->E, p->C_in, p->C_out, mesh->R, sten_stride, stream);
    if (rc != FC_OK) return rc;
    return fc_tangent_nonlin_forward(pre, p->bias, out, mesh->N, p->C_out, stream);
}

int fc_lift_block_backward(const float* g_out, const float* lift_sten, int32_t sten_stride, const fc_mesh* mesh, const int64_t* slot_to_edge_s,
                           const fc_lift_block_params* p, const void* saved, size_t saved_bytes, float* gx, void* workspace,
                           size_t workspace_bytes, void* stream) {
    fc::LiftPlan pl;
    if (!g_out || !gx || !fc::lift_plan(mesh, p, pl) || !p->zonal_ang || !p->zonal_mag || !p->phase || !p->bias || !p->g_zonal_ang ||
        !p->g_zonal_mag || !p->g_bias || (p->ftype != 0 && !p->g_phase))
        return FC_ERR_BAD_ARGUMENT;
    if (!saved || saved_bytes < pl.saved || !workspace || workspace_bytes < pl.ws_bwd) return FC_ERR_WORKSPACE;
    fc::Carver sv(const_cast<void*>(saved)), ws(workspace);
    const float* pre = sv.take(pl.pre);
    const float* ang = sv.take(pl.ang);
    const float* mag = sv.take(pl.mag);
    const float* s1sum = sv.take(pl.s1sum);
    float* g_pre = ws.take(pl.g_pre);
    void* nl_ws = ws.take(pl.nl_ws);
    void* tf_ws = ws.take(pl.tf_ws);
    int rc = fc_tangent_nonlin_backward(pre, p->bias, g_out, g_pre, p->g_bias, nl_ws, pl.nl_ws, mesh->N, p->C_out, stream);
    if (rc != FC_OK) return rc;
    return fc_trans_field_backward(lift_sten, mesh->by_source, slot_to_edge_s, p->zonal_ang, p->zonal_mag, p->phase, ang, mag, s1sum, g_pre, gx,
                                   p->g_zonal_ang, p->g_zonal_mag, p->ftype != 0 ? p->g_phase : nullptr, tf_ws, pl.tf_ws, mesh->N, mesh->E,
                                   p->C_in, p->C_out, mesh->R, sten_stride, p->ftype, stream);
}

}  // extern "C"
