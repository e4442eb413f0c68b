// FieldConv forward on per-edge records, ring-major kernels (fc_forward_ring.hpp): instantiation and launch.
#include <stdio.h>
#include "fc_forward_ring.hpp"
#include "fc_forward_kernels.hpp"

namespace fc {

// The record-driven forward pass runs the ring-major kernels (two 8-wavefront workgroups per CU) in the default
// two-halves mode; FC_RING=0 keeps the frequency-major ones (one 16-wavefront workgroup per CU; also what FC_MFMA=f32 / f16
// run).  Read once per process.  Config 2 on MI355X: 138 us against 150 us.
bool ring_enabled(const fc_dims* d) {
    static const bool on = [] { const char* e = dev_env("FC_RING"); return !(e && atoi(e) == 0); }();
    return on && halves_of(d) == 2;
}

// Meshes of up to 256 tiles (4096 vertices) are one round of the frequency-major kernels (one 16-vertex tile per CU, with
// the edge split for the smallest ones), which they keep: 41 against 46 us at C = 48, k = 32.  From 257 tiles on the
// frequency-major grid needs a second round while the ring-major one (two workgroups per CU) still takes one: 59 against
// 66 us between 4097 and 8192 vertices, 91 against 94 up to 12288 (tools/ring_threshold.py).  FC_RING=2 forces the
// ring-major kernels for any size (tests).
bool forward_ring_fits(const fc_dims* d) {
    static const bool force = [] { const char* e = dev_env("FC_RING"); return e && atoi(e) == 2; }();
    if (!ring_enabled(d) || !plan_ring(d->O, 2 * d->B + 1, d->I, halves_of(d)).ok) return false;
    return force || (d->N + kTile - 1) / kTile > num_cus();
}

// Which kernel a forward launch with these dims takes (fc_describe_kernels); kind as in forward_impl.
void describe_forward(const fc_dims* d, int kind, char* buf, size_t n) {
    const char* mode = halves_of(d) == 2 ? "split-f16" : halves_of(d) == 1 ? "f16" : "f32";
    const char* rec = kind == 2 ? "geometric records" : kind == 1 ? "factored records" : "dense rows";
    if (kind != 0 && forward_ring_fits(d)) {
        const RingPlan p = plan_ring(d->O, 2 * d->B + 1, d->I, halves_of(d));
        snprintf(buf, n, "fc_forward_ring_kernel<%s,%s> (ring-major, 2 workgroups x 8 wavefronts per CU, %zu B LDS, %d %s record chunks%s)",
                 rec, mode, p.lds, p.nr, p.logh ? "half-size" : "1 KiB", p.alias ? ", partials aliased onto the slab" : "");
    } else {
        snprintf(buf, n, "%s<%s,%s> (frequency-major, 16 wavefronts per CU, parts=%d)", kind ? "fc_forward_factored_kernel" : "fc_forward_kernel",
                 rec, mode, 1 << forward_parts_log2(d, kind));
    }
}

size_t packed_ring_image_floats(int M, int F, int channels, int R, int halves) {
    const MmaGeom g = ring_geom(M, F, channels, halves);
    return (size_t)g.MP + (size_t)R * halves * g.MP * g.KP;
}

template <int R, int B, bool GEO, int LOGH>
static int launch_forward_ring(const float2* x, const float* rec, const fc_csr* g, const float* wpk, float2* y, const RingArgs& a,
                               size_t lds, int grid, hipStream_t stream) {
    auto kern = fc_forward_ring_kernel<R, B, GEO, LOGH>;
    static bool lds_ok[kMaxDevices] = {};
    if (!allow_full_lds(reinterpret_cast<const void*>(kern), lds, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kDuoThreads), lds, stream, x, rec, g->rowptr, g->runs, wpk, y, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int forward_ring_impl(const float* x, const float* rec, const fc_csr* g, const float* wpk, float* y, const fc_dims* d, int kind,
                      void* ws, size_t ws_bytes, const fc_epilogue* epi, hipStream_t stream) {
    const int F = 2 * d->B + 1;
    const RingPlan p = plan_ring(d->O, F, d->I, halves_of(d));
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    RingArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.g = p.g;
    a.ntiles = (d->N + kTile - 1) / kTile;
    a.parts_log2 = (ws && ws_bytes >= forward_workspace_bytes_impl(d, kind)) ? forward_parts_log2(d, kind) : 0;
    a.part_stride = (uint32_t)forward_part_stride(d);
    a.epi = make_epi(epi);
    a.nr = p.nr;
    a.alias_part = p.alias;
    a.wpk_bytes = (uint32_t)(packed_ring_image_floats(d->O, F, d->I, d->R, p.g.split) * sizeof(float));
    a.slab_bytes_w = (uint32_t)(2 * p.g.split * p.g.MP * p.g.KP * 2);
    static const int dbg = [] { const char* e = dev_env("FC_DEBUG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    a.stamps = debug_stamp_buffer();
    const int nvt = a.ntiles << a.parts_log2;
    int grid = nvt < 2 * num_cus() ? nvt : 2 * num_cus();            // persistent: two workgroups per CU
    // the last, partly filled round: as half tiles when those still fit one round
    const int rem = nvt % grid;
    a.nv_full = nvt;
    a.nv_total = nvt;
    a.spread_cus = 0;
    static const bool halves = !(dev_env("FC_RING_HALVES") && atoi(dev_env("FC_RING_HALVES")) == 0);
    if (halves && a.parts_log2 == 0 && rem > 0 && 2 * rem <= grid) {
        a.nv_full = nvt - rem;
        a.nv_total = a.nv_full + 2 * rem;
    }
    // Between one and two workgroups per CU (a FAUST-sized mesh: 313 tiles on 256 CUs) whole tiles would leave some CUs with two of
    // them and the rest with one: as many half tiles as fill every slot instead -- W whole + 2 (nvt - W) half tiles = 2 cus items, one per
    // workgroup, a CU holding at most a whole and a half tile (RingArgs::spread_cus)
    const int cus = num_cus();
    if (halves && a.parts_log2 == 0 && nvt > cus && nvt < 2 * cus) {
        a.nv_full = 2 * nvt - 2 * cus;
        a.nv_total = 2 * cus;
        a.spread_cus = cus;
        grid = 2 * cus;
    }
    int rc = FC_ERR_UNSUPPORTED;
    const float2* x2 = reinterpret_cast<const float2*>(x);
    float2* y2 = reinterpret_cast<float2*>(a.parts_log2 ? static_cast<float*>(ws) : y);
#define FC_CASE(RR, BB)                                                                                            \
    if (d->R == RR && d->B == BB) {                                                                                \
        if constexpr (BB >= 2) {                                                                                   \
            if (p.logh)                                                                                            \
                rc = kind == 2 ? launch_forward_ring<RR, BB, true, 1>(x2, rec, g, wpk, y2, a, p.lds, grid, stream)  \
                               : launch_forward_ring<RR, BB, false, 1>(x2, rec, g, wpk, y2, a, p.lds, grid, stream); \
        }                                                                                                          \
        if (!p.logh)                                                                                               \
            rc = kind == 2 ? launch_forward_ring<RR, BB, true, 0>(x2, rec, g, wpk, y2, a, p.lds, grid, stream)      \
                           : launch_forward_ring<RR, BB, false, 0>(x2, rec, g, wpk, y2, a, p.lds, grid, stream);    \
    }
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    if (rc != FC_OK || a.parts_log2 == 0) return rc;
    return sum_parts_epilogue(static_cast<const float*>(ws), y, (size_t)d->N * d->O, a.part_stride, 1 << a.parts_log2, d->O, a.epi, stream);
}

}  // namespace fc
