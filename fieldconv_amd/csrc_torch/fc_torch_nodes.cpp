// Autograd nodes for the block-level entry points of libfieldconv_hip.so, in C++ (a torch extension: host-side plumbing only).
//
// On the reference's ~1k-vertex meshes (segmentation.ipynb:120,137: batch size 1, a different mesh every step) the GPU needs ~1.4 ms for a
// forward + backward of the segmentation network and a Python `torch.autograd.Function` costs the host ~10 us per differentiable tensor
// argument per node (62 parameter tensors in that network) on top of the interpreter's own time: with the nodes of fieldconv_amd/blocks.py
// the step is host-bound on all but the fastest hosts.  The same nodes here: argument wrapping, saved tensors, gradient buffers and
// the ONE call into the C ABI per block and pass happen without the interpreter, and the backward pass runs on the autograd engine's
// device thread without the GIL.  No arithmetic lives in this file; the library is reached through function pointers handed over by
// the Python binding (fieldconv_amd/_lib.py), so the nodes use exactly the library instance the package loaded.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "../../include/fieldconv_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Api {
    decltype(&fc_resnet_block_forward) resnet_fwd = nullptr;
    decltype(&fc_resnet_block_backward) resnet_bwd = nullptr;
    decltype(&fc_echo_block_forward) echo_fwd = nullptr;
    decltype(&fc_echo_block_backward) echo_bwd = nullptr;
    decltype(&fc_lift_block_forward) lift_fwd = nullptr;
    decltype(&fc_lift_block_backward) lift_bwd = nullptr;
    decltype(&fc_lift_block_workspace_bytes) lift_ws = nullptr;
    decltype(&fc_lift_block_saved_bytes) lift_saved = nullptr;
    decltype(&fc_soft_abs_forward) soft_abs_fwd = nullptr;
    decltype(&fc_soft_abs_backward) soft_abs_bwd = nullptr;
    decltype(&fc_echo_head_forward) head_fwd = nullptr;
    decltype(&fc_echo_head_backward) head_bwd = nullptr;
    decltype(&fc_echo_head_forward_workspace_bytes) head_fwd_ws = nullptr;
    decltype(&fc_echo_head_backward_workspace_bytes) head_bwd_ws = nullptr;
    decltype(&fc_status_string) status_string = nullptr;
} api;

template <typename F>
void take(F& slot, const std::map<std::string, int64_t>& addrs, const char* name) {
    auto it = addrs.find(name);
    TORCH_CHECK(it != addrs.end() && it->second != 0, "fc_torch_nodes.bind: no address for ", name);
    slot = reinterpret_cast<F>(static_cast<intptr_t>(it->second));
}

void bind(const std::map<std::string, int64_t>& addrs) {
    take(api.resnet_fwd, addrs, "fc_resnet_block_forward");
    take(api.resnet_bwd, addrs, "fc_resnet_block_backward");
    take(api.echo_fwd, addrs, "fc_echo_block_forward");
    take(api.echo_bwd, addrs, "fc_echo_block_backward");
    take(api.lift_fwd, addrs, "fc_lift_block_forward");
    take(api.lift_bwd, addrs, "fc_lift_block_backward");
    take(api.lift_ws, addrs, "fc_lift_block_workspace_bytes");
    take(api.lift_saved, addrs, "fc_lift_block_saved_bytes");
    take(api.soft_abs_fwd, addrs, "fc_soft_abs_forward");
    take(api.soft_abs_bwd, addrs, "fc_soft_abs_backward");
    take(api.head_fwd, addrs, "fc_echo_head_forward");
    take(api.head_bwd, addrs, "fc_echo_head_backward");
    take(api.head_fwd_ws, addrs, "fc_echo_head_forward_workspace_bytes");
    take(api.head_bwd_ws, addrs, "fc_echo_head_backward_workspace_bytes");
    take(api.status_string, addrs, "fc_status_string");
}

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed: ", api.status_string ? api.status_string(rc) : "?", " (status ", rc, ")");
}

void* stream_of(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

const float* fp(const at::Tensor& t) { return t.defined() && t.numel() ? static_cast<const float*>(t.data_ptr()) : nullptr; }
float* fpm(const at::Tensor& t) { return t.defined() && t.numel() ? static_cast<float*>(t.data_ptr()) : nullptr; }

// One mesh's support graph as the block-level calls take it: the grouping arrays and records (kept alive here), sizes, record kind.
// Built once per (graph, band limit) by the Python side and handed to every node of that mesh.
struct GraphRef {
    std::vector<at::Tensor> t;          // rowptr_t, nbr_t, runs_t, rowptr_s, nbr_s, runs_s, fwd records / rows, bwd records / rows
    int64_t N, E, R, B, kind;          // kind: record kind | fc_mfma_mode << 8 (the arithmetic mode of the mesh's convolutions)
    GraphRef(std::vector<at::Tensor> tensors, int64_t n, int64_t e, int64_t r, int64_t b, int64_t k)
        : t(std::move(tensors)), N(n), E(e), R(r), B(b), kind(k) {
        TORCH_CHECK(t.size() == 8, "GraphRef takes 8 tensors");
    }
    void fill(fc_mesh& m, fc_csr& ct, fc_csr& cs) const {
        auto ip = [](const at::Tensor& x) { return x.defined() && x.numel() ? static_cast<const int32_t*>(x.data_ptr()) : nullptr; };
        ct = fc_csr{ip(t[0]), ip(t[1]), ip(t[2])};
        cs = fc_csr{ip(t[3]), ip(t[4]), ip(t[5])};
        m = fc_mesh{(int32_t)N, (int32_t)E, (int32_t)R, (int32_t)B, (int32_t)(kind & 255), &ct, &cs, fp(t[6]), fp(t[7]), (int32_t)(kind >> 8)};
    }
};

// the mesh of a node travels from forward to backward in the context's saved_data (the tensors keep the arrays alive)
void save_graph(AutogradContext* ctx, const GraphRef& g) {
    ctx->saved_data["graph_tensors"] = g.t;
    ctx->saved_data["graph_ints"] = std::vector<int64_t>{g.N, g.E, g.R, g.B, g.kind};
}
GraphRef load_graph(AutogradContext* ctx) {
    const auto ints = ctx->saved_data["graph_ints"].toIntVector();
    return GraphRef(ctx->saved_data["graph_tensors"].toTensorVector(), ints[0], ints[1], ints[2], ints[3], ints[4]);
}

// views of one flat float buffer, one per parameter-shaped gradient (undefined entries take no room); pieces 16-byte aligned
struct GradViews {
    at::Tensor flat;
    std::vector<at::Tensor> v;
    GradViews(const std::vector<const at::Tensor*>& like, const at::Tensor& ref, bool zero = false) {
        int64_t total = 0;
        for (auto* p : like)
            if (p) total += (p->numel() + 3) / 4 * 4;
        const auto opts = ref.options().dtype(at::kFloat);
        flat = zero ? at::zeros({total}, opts) : at::empty({total}, opts);
        int64_t off = 0;
        for (auto* p : like) {
            if (!p) {
                v.emplace_back();
                continue;
            }
            v.push_back(flat.narrow(0, off, p->numel()).view(p->sizes()));
            off += (p->numel() + 3) / 4 * 4;
        }
    }
};

fc_filter_params filter_params(const at::Tensor& z, const at::Tensor& s, const at::Tensor& p, int64_t ftype, const at::Tensor* gz = nullptr,
                               const at::Tensor* gs = nullptr, const at::Tensor* gp = nullptr) {
    fc_filter_params f{};
    f.zonal = fp(z);
    f.spherical = fp(s);
    f.phase = fp(p);
    f.ftype = (int32_t)ftype;
    f.g_zonal = gz ? fpm(*gz) : nullptr;
    f.g_spherical = gs ? fpm(*gs) : nullptr;
    f.g_phase = (gp && gp->defined()) ? fpm(*gp) : nullptr;
    return f;
}

at::Tensor bytes(int64_t n, const at::Tensor& ref) { return at::empty({n > 0 ? n : 1}, ref.options().dtype(at::kByte)); }

// ---------------------------------------------------------------------------------------------------------------- FCResNetBlock
struct ResnetBlockFn : public torch::autograd::Function<ResnetBlockFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x_, const at::Tensor& z1_, const at::Tensor& s1_, const at::Tensor& p1_,
                              const at::Tensor& b1_, const at::Tensor& z2_, const at::Tensor& s2_, const at::Tensor& p2_, const at::Tensor& b2_,
                              const at::Tensor& re_, const at::Tensor& im_, const std::shared_ptr<GraphRef>& g, int64_t ftype,
                              int64_t saved_bytes, int64_t ws_fwd, int64_t ws_bwd) {
        const at::Tensor x = x_.contiguous(), z1 = z1_.contiguous(), s1 = s1_.contiguous(), p1 = p1_.contiguous(), b1 = b1_.contiguous(),
                         z2 = z2_.contiguous(), s2 = s2_.contiguous(), p2 = p2_.contiguous(), b2 = b2_.contiguous(), re = re_.contiguous(),
                         im = im_.contiguous();
        c10::DeviceGuard guard(x.device());
        fc_mesh m;
        fc_csr ct, cs;
        g->fill(m, ct, cs);
        fc_resnet_block_params bp{};
        bp.C_mid = (int32_t)z1.size(0);
        bp.C_in = (int32_t)z1.size(1);
        bp.C_out = (int32_t)z2.size(0);
        bp.conv1 = filter_params(z1, s1, p1, ftype);
        bp.conv2 = filter_params(z2, s2, p2, ftype);
        bp.bias1 = fp(b1);
        bp.bias2 = fp(b2);
        bp.res_re = fp(re);
        bp.res_im = fp(im);
        at::Tensor out = at::empty({g->N, z2.size(0)}, x.options());
        at::Tensor saved = bytes(saved_bytes, x), ws = bytes(ws_fwd, x);
        check(api.resnet_fwd(fp(x), &m, &bp, fpm(out), saved.data_ptr(), (size_t)saved_bytes, ws.data_ptr(), (size_t)ws_fwd, stream_of(x)),
              "fc_resnet_block_forward");
        ctx->save_for_backward({x, saved, z1, s1, p1, b1, z2, s2, p2, b2, re, im});
        ctx->saved_data["ftype"] = ftype;
        ctx->saved_data["saved_bytes"] = saved_bytes;
        ctx->saved_data["ws_bwd"] = ws_bwd;
        save_graph(ctx, *g);
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &x = sv[0], &saved = sv[1], &z1 = sv[2], &s1 = sv[3], &p1 = sv[4], &b1 = sv[5], &z2 = sv[6], &s2 = sv[7], &p2 = sv[8],
                         &b2 = sv[9], &re = sv[10], &im = sv[11];
        const int64_t ftype = ctx->saved_data["ftype"].toInt(), saved_bytes = ctx->saved_data["saved_bytes"].toInt(),
                      ws_bwd = ctx->saved_data["ws_bwd"].toInt();
        const GraphRef gr = load_graph(ctx);
        const GraphRef* g = &gr;
        const at::Tensor g_out = grads[0].contiguous();
        c10::DeviceGuard guard(x.device());
        fc_mesh m;
        fc_csr ct, cs;
        g->fill(m, ct, cs);
        const bool ph = ftype == 1;
        GradViews gv({&z1, &s1, ph ? &p1 : nullptr, &b1, &z2, &s2, ph ? &p2 : nullptr, &b2, &re, &im}, x);
        at::Tensor gx = at::empty_like(x), ws = bytes(ws_bwd, x);
        fc_resnet_block_params bp{};
        bp.C_mid = (int32_t)z1.size(0);
        bp.C_in = (int32_t)z1.size(1);
        bp.C_out = (int32_t)z2.size(0);
        bp.conv1 = filter_params(z1, s1, p1, ftype, &gv.v[0], &gv.v[1], &gv.v[2]);
        bp.conv2 = filter_params(z2, s2, p2, ftype, &gv.v[4], &gv.v[5], &gv.v[6]);
        bp.bias1 = fp(b1);
        bp.bias2 = fp(b2);
        bp.res_re = fp(re);
        bp.res_im = fp(im);
        bp.g_bias1 = fpm(gv.v[3]);
        bp.g_bias2 = fpm(gv.v[7]);
        bp.g_res_re = fpm(gv.v[8]);
        bp.g_res_im = fpm(gv.v[9]);
        check(api.resnet_bwd(fp(x), fp(g_out), &m, &bp, saved.data_ptr(), (size_t)saved_bytes, fpm(gx), ws.data_ptr(), (size_t)ws_bwd,
                             stream_of(x)),
              "fc_resnet_block_backward");
        return {gx, gv.v[0], gv.v[1], gv.v[2], gv.v[3], gv.v[4], gv.v[5], gv.v[6], gv.v[7], gv.v[8], gv.v[9], at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor()};
    }

};

// -------------------------------------------------------------------------------------------------- ECHOBlock, tangent-feature half
// desc = ECHO(modReLU(conv(x))), reference nn/echo_block.py:93-94 (fc_echo_block_forward / _backward)
struct EchoBlockFn : public torch::autograd::Function<EchoBlockFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x_, const at::Tensor& z_, const at::Tensor& s_, const at::Tensor& p_,
                              const at::Tensor& bias_, const std::shared_ptr<GraphRef>& g, const std::vector<at::Tensor>& slots, int64_t ftype,
                              int64_t n_des, int64_t n_bins, int64_t dS, int64_t saved_bytes, int64_t ws_fwd, int64_t ws_bwd) {
        const at::Tensor x = x_.contiguous(), z = z_.contiguous(), s = s_.contiguous(), p = p_.contiguous(), bias = bias_.contiguous();
        TORCH_CHECK(slots.size() == 4, "EchoBlockFn takes ln_t, wxp_t, ln_s, wxp_s");
        c10::DeviceGuard guard(x.device());
        fc_mesh m;
        fc_csr ct, cs;
        g->fill(m, ct, cs);
        fc_echo_block_params bp{};
        bp.C_in = (int32_t)z.size(1);
        bp.n_des = (int32_t)n_des;
        bp.n_bins = (int32_t)n_bins;
        bp.conv = filter_params(z, s, p, ftype);
        bp.bias = fp(bias);
        at::Tensor desc = at::empty({g->N, n_des, dS}, x.options().dtype(at::kFloat));
        at::Tensor saved = bytes(saved_bytes, x), ws = bytes(ws_fwd, x);
        check(api.echo_fwd(fp(x), &m, fp(slots[0]), fp(slots[1]), &bp, fpm(desc), saved.data_ptr(), (size_t)saved_bytes, ws.data_ptr(),
                           (size_t)ws_fwd, stream_of(x)),
              "fc_echo_block_forward");
        ctx->save_for_backward({x, saved, z, s, p, bias, slots[2], slots[3]});
        ctx->saved_data["ints"] = std::vector<int64_t>{ftype, n_des, n_bins, saved_bytes, ws_bwd};
        save_graph(ctx, *g);
        return desc;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &x = sv[0], &saved = sv[1], &z = sv[2], &s = sv[3], &p = sv[4], &bias = sv[5], &ln_s = sv[6], &wxp_s = sv[7];
        const auto ints = ctx->saved_data["ints"].toIntVector();
        const int64_t ftype = ints[0], n_des = ints[1], n_bins = ints[2], saved_bytes = ints[3], ws_bwd = ints[4];
        const GraphRef g = load_graph(ctx);
        const at::Tensor g_desc = grads[0].contiguous();
        c10::DeviceGuard guard(x.device());
        fc_mesh m;
        fc_csr ct, cs;
        g.fill(m, ct, cs);
        // the module's bias has in_channels entries of which the first n_des act (reference nn/echo_block.py:57,93): the rest get zero
        GradViews gv({&z, &s, ftype == 1 ? &p : nullptr, &bias}, x, bias.numel() > n_des);
        at::Tensor gx = at::empty_like(x), ws = bytes(ws_bwd, x);
        fc_echo_block_params bp{};
        bp.C_in = (int32_t)z.size(1);
        bp.n_des = (int32_t)n_des;
        bp.n_bins = (int32_t)n_bins;
        bp.conv = filter_params(z, s, p, ftype, &gv.v[0], &gv.v[1], &gv.v[2]);
        bp.bias = fp(bias);
        bp.g_bias = fpm(gv.v[3]);
        check(api.echo_bwd(fp(x), fp(g_desc), &m, fp(ln_s), fp(wxp_s), &bp, saved.data_ptr(), (size_t)saved_bytes, fpm(gx), ws.data_ptr(),
                           (size_t)ws_bwd, stream_of(x)),
              "fc_echo_block_backward");
        return {gx, gv.v[0], gv.v[1], gv.v[2], gv.v[3], at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// ---------------------------------------------------------------------------------------------------------------------- LiftBlock
// out = modReLU(TransField(x)), reference nn/lift_block.py:53-55 (fc_lift_block_forward / _backward).  csr: rowptr_t, nbr_t, perm_t,
// rowptr_s, nbr_s, perm_s of the edge grouping; sten: the (E,R,2) columns (a strided view is read in place) or, with stride 0, FCPrecomp's
// (E,8) factor table.
struct LiftBlockFn : public torch::autograd::Function<LiftBlockFn> {
    static void mesh_of(const std::vector<at::Tensor>& csr, int64_t N, int64_t E, int64_t R, fc_mesh& m, fc_csr& ct, fc_csr& cs) {
        auto ip = [](const at::Tensor& x) { return x.defined() && x.numel() ? static_cast<const int32_t*>(x.data_ptr()) : nullptr; };
        ct = fc_csr{ip(csr[0]), ip(csr[1]), nullptr};
        cs = fc_csr{ip(csr[3]), ip(csr[4]), nullptr};
        m = fc_mesh{(int32_t)N, (int32_t)E, (int32_t)R, 0, 0, &ct, &cs, nullptr, nullptr, 0};
    }
    static const int64_t* lp(const at::Tensor& t) { return t.defined() && t.numel() ? static_cast<const int64_t*>(t.data_ptr()) : nullptr; }

    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x_, const at::Tensor& sten, const at::Tensor& za_, const at::Tensor& zm_,
                              const at::Tensor& ph_, const at::Tensor& bias_, const std::vector<at::Tensor>& csr, int64_t stride, int64_t ftype,
                              int64_t E) {
        const at::Tensor x = x_.contiguous(), za = za_.contiguous(), zm = zm_.contiguous(), ph = ph_.contiguous(), bias = bias_.contiguous();
        TORCH_CHECK(csr.size() == 6, "LiftBlockFn takes rowptr_t, nbr_t, perm_t, rowptr_s, nbr_s, perm_s");
        const int64_t N = x.size(0), C_in = x.size(1), C_out = za.size(0), R = za.size(2);
        c10::DeviceGuard guard(x.device());
        fc_mesh m;
        fc_csr ct, cs;
        mesh_of(csr, N, E, R, m, ct, cs);
        fc_lift_block_params bp{};
        bp.C_in = (int32_t)C_in;
        bp.C_out = (int32_t)C_out;
        bp.ftype = (int32_t)ftype;
        bp.zonal_ang = fp(za);
        bp.zonal_mag = fp(zm);
        bp.phase = fp(ph);
        bp.bias = fp(bias);
        const int64_t saved_bytes = (int64_t)api.lift_saved(&m, &bp);
        at::Tensor out = at::empty({N, C_out}, x.options().dtype(at::kComplexFloat));
        at::Tensor saved = bytes(saved_bytes, x);
        check(api.lift_fwd(fp(x), fp(sten), (int32_t)stride, &m, lp(csr[2]), &bp, fpm(out), saved.data_ptr(), (size_t)saved_bytes,
                           stream_of(x)),
              "fc_lift_block_forward");
        ctx->save_for_backward({sten, saved, za, zm, ph, bias, csr[3], csr[4], csr[5], csr[0], csr[1]});
        ctx->saved_data["ints"] = std::vector<int64_t>{N, C_in, stride, ftype, E, saved_bytes};
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &sten = sv[0], &saved = sv[1], &za = sv[2], &zm = sv[3], &ph = sv[4], &bias = sv[5];
        const auto ints = ctx->saved_data["ints"].toIntVector();
        const int64_t N = ints[0], C_in = ints[1], stride = ints[2], ftype = ints[3], E = ints[4], saved_bytes = ints[5];
        const at::Tensor g_out = grads[0].contiguous();
        c10::DeviceGuard guard(g_out.device());
        fc_mesh m;
        fc_csr ct, cs;
        mesh_of({sv[9], sv[10], at::Tensor(), sv[6], sv[7], sv[8]}, N, E, za.size(2), m, ct, cs);
        GradViews gv({&za, &zm, ftype != 0 ? &ph : nullptr, &bias}, g_out);
        at::Tensor gx = at::empty({N, C_in}, g_out.options().dtype(at::kFloat));
        fc_lift_block_params bp{};
        bp.C_in = (int32_t)C_in;
        bp.C_out = (int32_t)za.size(0);
        bp.ftype = (int32_t)ftype;
        bp.zonal_ang = fp(za);
        bp.zonal_mag = fp(zm);
        bp.phase = fp(ph);
        bp.bias = fp(bias);
        bp.g_zonal_ang = fpm(gv.v[0]);
        bp.g_zonal_mag = fpm(gv.v[1]);
        bp.g_phase = gv.v[2].defined() ? fpm(gv.v[2]) : nullptr;
        bp.g_bias = fpm(gv.v[3]);
        const int64_t ws_bytes = (int64_t)api.lift_ws(&m, &bp, 1);
        at::Tensor ws = bytes(ws_bytes, g_out);
        check(api.lift_bwd(fp(g_out), fp(sten), (int32_t)stride, &m, lp(sv[8]), &bp, saved.data_ptr(), (size_t)saved_bytes, fpm(gx),
                           ws.data_ptr(), (size_t)ws_bytes, stream_of(g_out)),
              "fc_lift_block_backward");
        return {gx, at::Tensor(), gv.v[0], gv.v[1], gv.v[2], gv.v[3], at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// ------------------------------------------------------------------------------------------------------------- ECHOBlock, dense tail
// lin3(relu(lin2(relu(lin1(d))))) + res(softAbs(x)), reference nn/echo_block.py:95-103: the reference's four dense layers (ATen GEMMs, i.e.
// hipBLASLt) and softAbs (fc_soft_abs_*) as ONE node with the backward pass written out
struct EchoTailFn : public torch::autograd::Function<EchoTailFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& d, const at::Tensor& x_, const at::Tensor& w1, const at::Tensor& b1,
                              const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3, const at::Tensor& wr,
                              const at::Tensor& br) {
        const at::Tensor x = x_.contiguous();
        c10::DeviceGuard guard(x.device());
        at::Tensor a = at::empty(x.sizes(), x.options().dtype(at::kFloat));
        check(api.soft_abs_fwd(fp(x), fpm(a), (size_t)x.numel(), stream_of(x)), "fc_soft_abs_forward");
        at::Tensor h1 = at::addmm(b1, d, w1.t()).relu_();
        at::Tensor h2 = at::addmm(b2, h1, w2.t()).relu_();
        at::Tensor y = at::addmm(b3, h2, w3.t()).add_(at::addmm(br, a, wr.t()));         // lin3(h2) + res(a), in the reference's order
        ctx->save_for_backward({d, x, a, h1, h2, w1, w2, w3, wr});
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &d = sv[0], &x = sv[1], &a = sv[2], &h1 = sv[3], &h2 = sv[4], &w1 = sv[5], &w2 = sv[6], &w3 = sv[7], &wr = sv[8];
        const at::Tensor g = grads[0].contiguous();
        c10::DeviceGuard guard(x.device());
        at::Tensor gb = g.sum(0);                            // lin3.bias and res.bias see the same cotangent
        at::Tensor g_w3 = g.t().mm(h2);
        at::Tensor g_h2 = g.mm(w3).mul_(h2 > 0);
        at::Tensor g_b2 = g_h2.sum(0);
        at::Tensor g_w2 = g_h2.t().mm(h1);
        at::Tensor g_h1 = g_h2.mm(w2).mul_(h1 > 0);
        at::Tensor g_b1 = g_h1.sum(0);
        at::Tensor g_w1 = g_h1.t().mm(d);
        at::Tensor g_d = g_h1.mm(w1);
        at::Tensor g_wr = g.t().mm(a);
        at::Tensor g_a = g.mm(wr);
        at::Tensor gx = at::empty_like(x);
        check(api.soft_abs_bwd(fp(x), fp(g_a), fpm(gx), (size_t)x.numel(), stream_of(x)), "fc_soft_abs_backward");
        return {g_d, gx, g_w1, g_b1, g_w2, g_b2, g_w3, gb, g_wr, gb.clone()};
    }
};

// The same tail through the library's own kernels (fc_echo_head_forward / fc_echo_head_backward, csrc/fc_head.hip): three launches per pass
struct EchoHeadFn : public torch::autograd::Function<EchoHeadFn> {
    static fc_echo_head_params params(const at::Tensor& d, const at::Tensor& x, const at::Tensor& w1, const at::Tensor& w2, const at::Tensor& w3,
                                      const at::Tensor& wr) {
        fc_echo_head_params p{};
        p.D = (int32_t)d.size(1);
        p.H1 = (int32_t)w1.size(0);
        p.H2 = (int32_t)w2.size(0);
        p.C_in = (int32_t)x.size(1);
        p.C_out = (int32_t)w3.size(0);
        p.w1 = fp(w1); p.w2 = fp(w2); p.w3 = fp(w3); p.wr = fp(wr);
        return p;
    }
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& d_, const at::Tensor& x_, const at::Tensor& w1, const at::Tensor& b1,
                              const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3, const at::Tensor& wr,
                              const at::Tensor& br) {
        const at::Tensor d = d_.contiguous(), x = x_.contiguous();
        c10::DeviceGuard guard(x.device());
        fc_echo_head_params p = params(d, x, w1, w2, w3, wr);
        p.b1 = fp(b1); p.b2 = fp(b2); p.b3 = fp(b3); p.br = fp(br);
        const int64_t N = d.size(0);
        const int64_t nws = (int64_t)api.head_fwd_ws((int32_t)N, &p);
        const auto opts = d.options().dtype(at::kFloat);
        at::Tensor buf = at::empty({N * (p.H1 + p.H2) + (nws + 3) / 4}, opts);
        at::Tensor h1 = buf.narrow(0, 0, N * p.H1).view({N, p.H1}), h2 = buf.narrow(0, N * p.H1, N * p.H2).view({N, p.H2});
        at::Tensor y = at::empty({N, p.C_out}, opts);                 // (its own storage: the caller may write into it)
        check(api.head_fwd(fp(d), fp(x), &p, fpm(h1), fpm(h2), fpm(y), nws ? fpm(buf) + N * (p.H1 + p.H2) : nullptr, (size_t)nws, (int32_t)N,
                           stream_of(x)), "fc_echo_head_forward");
        ctx->save_for_backward({d, x, h1, h2, w1, w2, w3, wr});
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &d = sv[0], &x = sv[1], &h1 = sv[2], &h2 = sv[3], &w1 = sv[4], &w2 = sv[5], &w3 = sv[6], &wr = sv[7];
        const at::Tensor g = grads[0].contiguous();
        c10::DeviceGuard guard(x.device());
        fc_echo_head_params p = params(d, x, w1, w2, w3, wr);
        const int64_t N = d.size(0);
        const auto opts = d.options().dtype(at::kFloat);
        // the parameter gradients in a buffer of their own (they may live on as .grad); the flowing gradients and scratch in another
        const int64_t psz[8] = {(int64_t)p.H1 * p.D, p.H1, (int64_t)p.H2 * p.H1, p.H2, (int64_t)p.C_out * p.H2, p.C_out, (int64_t)p.C_out * p.C_in,
                                p.C_out};
        int64_t poff[9];
        poff[0] = 0;
        for (int i = 0; i < 8; ++i) poff[i + 1] = poff[i] + (psz[i] + 3) / 4 * 4;
        at::Tensor pbuf = at::empty({poff[8]}, opts);
        auto pg = [&](int i) { return pbuf.narrow(0, poff[i], psz[i]); };
        const int64_t sizes[3] = {N * p.D, 2 * N * p.C_in, N * p.H1};           // g_d, gx, g_h1 (scratch)
        int64_t off[4];
        off[0] = 0;
        for (int i = 0; i < 3; ++i) off[i + 1] = off[i] + (sizes[i] + 3) / 4 * 4;
        const int64_t nws = (int64_t)api.head_bwd_ws((int32_t)N, &p);
        at::Tensor buf = at::empty({off[3] + (nws + 3) / 4}, opts);
        auto piece = [&](int i) { return buf.narrow(0, off[i], sizes[i]); };
        p.g_w1 = fpm(pg(0)); p.g_b1 = fpm(pg(1)); p.g_w2 = fpm(pg(2)); p.g_b2 = fpm(pg(3));
        p.g_w3 = fpm(pg(4)); p.g_b3 = fpm(pg(5)); p.g_wr = fpm(pg(6)); p.g_br = fpm(pg(7));
        check(api.head_bwd(fp(d), fp(x), fp(h1), fp(h2), fp(g), &p, fpm(piece(0)), fpm(piece(1)), fpm(piece(2)), fpm(buf) + off[3], (size_t)nws,
                           (int32_t)N, stream_of(x)), "fc_echo_head_backward");
        return {piece(0).view({N, p.D}), at::view_as_complex(piece(1).view({N, p.C_in, 2})), pg(0).view({p.H1, p.D}), pg(1),
                pg(2).view({p.H2, p.H1}), pg(3), pg(4).view({p.C_out, p.H2}), pg(5), pg(6).view({p.C_out, p.C_in}), pg(7)};
    }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ autograd nodes for libfieldconv_hip.so's block-level entry points";
    m.def("bind", &bind);
    pybind11::class_<GraphRef, std::shared_ptr<GraphRef>>(m, "GraphRef")
        .def(pybind11::init<std::vector<at::Tensor>, int64_t, int64_t, int64_t, int64_t, int64_t>());
    m.def("resnet_block", [](const at::Tensor& x, const at::Tensor& z1, const at::Tensor& s1, const at::Tensor& p1, const at::Tensor& b1,
                             const at::Tensor& z2, const at::Tensor& s2, const at::Tensor& p2, const at::Tensor& b2, const at::Tensor& re,
                             const at::Tensor& im, const std::shared_ptr<GraphRef>& g, int64_t ftype, int64_t saved_bytes, int64_t ws_fwd,
                             int64_t ws_bwd) {
        return ResnetBlockFn::apply(x, z1, s1, p1, b1, z2, s2, p2, b2, re, im, g, ftype, saved_bytes, ws_fwd, ws_bwd);
    });
    m.def("echo_block", [](const at::Tensor& x, const at::Tensor& z, const at::Tensor& s, const at::Tensor& p, const at::Tensor& bias,
                           const std::shared_ptr<GraphRef>& g, const std::vector<at::Tensor>& slots, int64_t ftype, int64_t n_des, int64_t n_bins,
                           int64_t dS, int64_t saved_bytes, int64_t ws_fwd, int64_t ws_bwd) {
        return EchoBlockFn::apply(x, z, s, p, bias, g, slots, ftype, n_des, n_bins, dS, saved_bytes, ws_fwd, ws_bwd);
    });
    m.def("lift_block", [](const at::Tensor& x, const at::Tensor& sten, const at::Tensor& za, const at::Tensor& zm, const at::Tensor& ph,
                           const at::Tensor& bias, const std::vector<at::Tensor>& csr, int64_t stride, int64_t ftype, int64_t E) {
        return LiftBlockFn::apply(x, sten, za, zm, ph, bias, csr, stride, ftype, E);
    });
    m.def("echo_tail", [](const at::Tensor& d, const at::Tensor& x, const at::Tensor& w1, const at::Tensor& b1, const at::Tensor& w2,
                          const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3, const at::Tensor& wr, const at::Tensor& br) {
        return EchoTailFn::apply(d, x, w1, b1, w2, b2, w3, b3, wr, br);
    });
    m.def("echo_head", [](const at::Tensor& d, const at::Tensor& x, const at::Tensor& w1, const at::Tensor& b1, const at::Tensor& w2,
                          const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3, const at::Tensor& wr, const at::Tensor& br) {
        return EchoHeadFn::apply(d, x, w1, b1, w2, b2, w3, b3, wr, br);
    });
}
