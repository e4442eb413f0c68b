// FieldConv backward, ring-major path (fc_backward_ring.hpp): the filter-gradient kernel on the kept slabs, the reduction
// of its partials, plans and launchers.
#include "fc_backward_roles.hpp"

namespace fc {

// ---------------------------------------------------------------------------------- filter gradient
// gW[o,i,q,f] = 1/F sum_j H[j,(f,o),q] conj(x~_f[j,i]) for ONE ring q per workgroup (blockIdx.y): a persistent workgroup
// walks the work items, streams the kept slab of (item, q) -- 32 rows of halves exactly as the data kernel's matrix-pipe
// operand lay in LDS -- back with LDS-DMA into one of two images, and accumulates in MFMA registers.  The matrix
// instruction's k dimension is the 32 vertices of the tile:
//   first operand : H^T, rows (f, o): k-major fragments straight out of the image with ds_read_b64_tr_b16 (no conversion);
//   second operand: x~_f[j,i] / s_j (the slab rows carry the scale s_j, which cannot be factored out of a sum over j, so it
//                   moves to the other factor) times a power-of-two scale t_i per column and slab (computed by the data
//                   kernel, kept behind the slab), split into halves IN REGISTERS: wavefront w owns the (i tile,
//                   frequency) pair w, its lanes hold 8 vertices of one column each.
// A wavefront keeps gW of its pair for all OT row tiles (16 output channels each); the products of a slab start from zero
// and are added to the fp32 running sums with the factor 1/t_i.
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline int br_image_row_bytes(const BrGeom& g) { return 8 * g.KP + 32; }      // + 32: transposed reads of 8 rows hit distinct banks

// x~_m = x u^m, u = exp(-i angle(x)) = conj(x) / |x| (1 inside the origin box), for a WAVE-UNIFORM m: closed forms instead of
// unit vector + powers -- x u = |x|, x u^2 = conj(x), x u^3 = conj(x)^2 / |x|, x conj(u) = x^2 / |x|, x conj(u)^2 = x^3 / |x|^2,
// x conj(u)^3 = x^4 / |x|^3.
__device__ __forceinline__ float2 rotated_feature(const float2 x, const int m) {
    if (m == 0 || is_origin(x)) return x;
    const float n2 = x.x * x.x + x.y * x.y;
    if (m == 1) return make_float2(sqrtf(n2), 0.f);
    if (m == 2) return make_float2(x.x, -x.y);
    const float rn = __frsqrt_rn(n2);
    if (m == 3) { const float2 c = make_float2(x.x, -x.y); const float2 c2 = cmul(c, c); return make_float2(c2.x * rn, c2.y * rn); }
    const float2 x2 = cmul(x, x);
    if (m == -1) return make_float2(x2.x * rn, x2.y * rn);
    const float2 x3 = cmul(x2, x);
    if (m == -2) { const float r2 = rn * rn; return make_float2(x3.x * r2, x3.y * r2); }
    const float2 x4 = cmul(x2, x2);
    const float r3 = rn * rn * rn;
    return make_float2(x4.x * r3, x4.y * r3);
}

template <int OT>
__global__ __launch_bounds__(kThreads) void fc_backward_ring_filter_kernel(
    const float2* __restrict__ gx_, const char* __restrict__ hdump, float2* __restrict__ ggwp /* [P][R][F][OT*16][IP] */,
    const BrArgs a, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BrGeom& g = a.g;
    const int F = g.F, R = g.R, I = a.I;
    const int RS = br_image_row_bytes(g);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.y;
    const int fr = lane & 15, fq = lane >> 4;

    const bool active = wave < g.NMT * F;
    const int it = active ? wave % g.NMT : 0;
    const int mf = active ? wave / g.NMT : 0;
    const int m = mf - B;
    const int i_col = it * 16 + fr;                    // my column of x

    // rows are DMA'd piecewise (1 KiB per wavefront instruction, never across a row: the LDS rows carry a pad)
    const int row_bytes = 8 * g.KP;
    const int npr = (row_bytes + 1023) / 1024;
    const int npieces = kBrRows * npr;
    auto dma_slab = [&](const int vt, const int buf) {
        const char* src = hdump + ((size_t)vt * R + q) * a.hs_bytes;
        char* img = smem + (size_t)buf * kBrRows * RS;
        for (int p = wave; p < npieces; p += kWaves) {
            const int row = p / npr, part = p - row * npr;
            const int off = part * 1024 + lane * 16;
            if (off < row_bytes) lds_dma16_untracked(src + (size_t)row * row_bytes + off, img + row * RS + part * 1024);
        }
    };

    f32x4 gre[OT], gim[OT];
#pragma unroll
    for (int n = 0; n < OT; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }

    // one slab ahead in registers: the 8 entries x[j][i_col], j = 8 fq .. + 7, of my column and the scales behind the slab
    float2 xn[8];
    f32x4 sinv[2];
    float pt = 1.f, pit = 1.f;
    auto prefetch = [&](const int vt) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int vtx = br_vertex(vt, 8 * fq + jj, a.nv_full, a.N);
            xn[jj] = (vtx < a.N && i_col < I) ? gx_[(size_t)vtx * I + i_col] : make_float2(0.f, 0.f);
        }
        const float* tail = reinterpret_cast<const float*>(hdump + ((size_t)vt * R + q) * a.hs_bytes + (size_t)kBrRows * row_bytes);
        sinv[0] = *reinterpret_cast<const f32x4*>(tail + kBrRows + 8 * fq);
        sinv[1] = *reinterpret_cast<const f32x4*>(tail + kBrRows + 8 * fq + 4);
        pt = tail[2 * kBrRows + i_col];
        pit = tail[2 * kBrRows + g.IP + i_col];
    };

    // A fragment addressing (ds_read_b64_tr_b16): lane 4 r + p of a 16-lane group addresses row (vb + r), entries k0 + 4 p .. + 3
    // of a plane, and receives entry k0 + lane % 16 of rows vb .. vb + 3; rows vb = 8 fq and 8 fq + 4 make the lane's 8 vertices.
    // In the image a row holds [k / 8][plane][k % 8] halves.
    const int a_lane = (8 * fq + (fr >> 2)) * RS + (((4 * (fr & 3)) >> 3) * 32 + ((4 * (fr & 3)) & 7)) * 2;      // bytes

    if ((int)blockIdx.x < a.nv_total) {
        dma_slab(blockIdx.x, 0);
        if (active) prefetch(blockIdx.x);
    }
    int buf = 0;
    for (int vt = blockIdx.x; vt < a.nv_total; vt += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // my pieces of this slab (and my prefetched registers) have landed
        __syncthreads();                                               // everyone's have; the previous slab's reads are done
        const int vn = vt + gridDim.x;
        if (vn < a.nv_total) dma_slab(vn, buf ^ 1);                    // the next slab streams in under this one's arithmetic
        // ---- second operand of my pair in registers: c + i d = x~ / s_j * t_i, halves
        u32x4 c_hi, c_lo, d_hi, d_lo;
        const float it_scale = pit;
        if (active) {
            f16x2 h[8], lo8[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float2 xv = xn[jj];
                const float2 xt = rotated_feature(xv, m);
                const float sc = sinv[jj >> 2][jj & 3] * pt;
                split_halves2(f32x2{xt.x, xt.y}, sc, h[jj], lo8[jj]);
            }
            auto pack2 = [](_Float16 x0, _Float16 x1) {
                const f16x2 v = {x0, x1};
                return __builtin_bit_cast(uint32_t, v);
            };
            c_hi = u32x4{pack2(h[0].x, h[1].x), pack2(h[2].x, h[3].x), pack2(h[4].x, h[5].x), pack2(h[6].x, h[7].x)};
            d_hi = u32x4{pack2(h[0].y, h[1].y), pack2(h[2].y, h[3].y), pack2(h[4].y, h[5].y), pack2(h[6].y, h[7].y)};
            c_lo = u32x4{pack2(lo8[0].x, lo8[1].x), pack2(lo8[2].x, lo8[3].x), pack2(lo8[4].x, lo8[5].x), pack2(lo8[6].x, lo8[7].x)};
            d_lo = u32x4{pack2(lo8[0].y, lo8[1].y), pack2(lo8[2].y, lo8[3].y), pack2(lo8[4].y, lo8[5].y), pack2(lo8[6].y, lo8[7].y)};
        } else {
            c_hi = u32x4{0u, 0u, 0u, 0u}; c_lo = c_hi; d_hi = c_hi; d_lo = c_hi;
        }
        if (vn < a.nv_total && active) prefetch(vn);
        if (active && !(a.dbg & 4)) {
            const u32x4 sign = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
            const u32x4 nd_hi = d_hi ^ sign, nd_lo = d_lo ^ sign;
            const char* img = smem + (size_t)buf * kBrRows * RS;
#pragma unroll
            for (int n = 0; n < OT; ++n) {
                const int k0 = mf * g.KI + 16 * n;                      // first (f, o) row of this tile inside the slab's k
                const __attribute__((address_space(3))) char* ap =
                    (const __attribute__((address_space(3))) char*)img + a_lane + (k0 >> 3) * 64;
                auto tr = [&](const int plane, const int rows4) {
                    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + plane * 16 + rows4 * 4 * RS)));
                };
                const u32x2 r0 = tr(0, 0), r1 = tr(0, 1), r2 = tr(1, 0), r3 = tr(1, 1);
                const u32x2 i0 = tr(2, 0), i1 = tr(2, 1), i2 = tr(3, 0), i3 = tr(3, 1);
                const u32x4 a_hi = {r0.x, r0.y, r1.x, r1.y}, a_lo = {r2.x, r2.y, r3.x, r3.y};
                const u32x4 b_hi = {i0.x, i0.y, i1.x, i1.y}, b_lo = {i2.x, i2.y, i3.x, i3.y};
                // H conj(X), H = a + ib, X = c + id:  re = a c + b d,  im = b c - a d; each product lo*hi + hi*lo + hi*hi
                f32x4 re = {0.f, 0.f, 0.f, 0.f}, im = re;
                re = mfma32h(a_lo, c_hi, re); im = mfma32h(b_lo, c_hi, im);
                re = mfma32h(a_hi, c_lo, re); im = mfma32h(b_hi, c_lo, im);
                re = mfma32h(a_hi, c_hi, re); im = mfma32h(b_hi, c_hi, im);
                re = mfma32h(b_lo, d_hi, re); im = mfma32h(a_lo, nd_hi, im);
                re = mfma32h(b_hi, d_lo, re); im = mfma32h(a_hi, nd_lo, im);
                re = mfma32h(b_hi, d_hi, re); im = mfma32h(a_hi, nd_hi, im);
                gre[n] += re * it_scale;
                gim[n] += im * it_scale;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // my partial of gW[:, :, q, mf]: rows (o = 16 n + 4 fq + jj), column i_col
    if (active) {
#pragma unroll
        for (int n = 0; n < OT; ++n)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int o = 16 * n + 4 * fq + jj;
                ggwp[((((size_t)blockIdx.x * R + q) * F + mf) * (OT * 16) + o) * g.IP + i_col] = make_float2(gre[n][jj], gim[n][jj]);
            }
    }
}

// gw_eff[o][i][r][f] = 1/F sum_p gwp[p][r][f][o][i]
__global__ void fc_reduce_gw_ring_kernel(const float2* __restrict__ gwp, float2* __restrict__ gw, int P, int F, int R, int O, int I,
                                         int OP, int IP) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over (r, f, o, i), i fastest
    const int total = R * F * O * I;
    if (idx >= total) return;
    const int i = idx % I;
    const int o = (idx / I) % O;
    const int f = (idx / (I * O)) % F;
    const int r = idx / (I * O * F);
    float2 s = make_float2(0.f, 0.f);
    const size_t stride = (size_t)R * F * OP * IP;
    const float2* src = gwp + (((size_t)r * F + f) * OP + o) * IP + i;
    int p = 0;
    for (; p + 8 <= P; p += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; }
    }
    for (; p < P; ++p) {
        const float2 v = src[(size_t)p * stride];
        s.x += v.x;
        s.y += v.y;
    }
    const float sc = 1.f / (float)F;
    gw[(((size_t)o * I + i) * R + r) * F + f] = make_float2(s.x * sc, s.y * sc);
}

// ---------------------------------------------------------------------------------- plan
struct BrPlan {
    BrGeom g;
    BrItems items;
    int nr, P;
    size_t lds_data, lds_filter, hdump_bytes, gwp_bytes;
    bool ok;
    bool roles;         // the data kernel with gathering and contracting wavefronts (fc_backward_roles.hpp)
};

static BrPlan plan_br(const fc_dims* d) {
    BrPlan p;
    const int F = 2 * d->B + 1;
    p.g = br_geom(d->I, d->O, d->R, F);
    p.ok = false;
    p.roles = false;
    p.nr = 0;
    p.lds_data = p.lds_filter = p.hdump_bytes = p.gwp_bytes = 0;
    p.P = 1;
    p.items = br_items(d->N, num_cus());
    if (F > kBrMaxF || d->R > 8 || d->I > 255 || p.g.NMT * F > kWaves || p.g.OT > 4) return p;
    for (int f = 0; f < F; ++f)
        if (p.g.nb[f] > 2) return p;            // (two filter fragment sets per wavefront)
    for (int nr = 4; nr >= 2; nr >>= 1) {
        const size_t lds = br_lds_bytes(p.g, nr);
        if (lds <= kMaxLds) {
            p.nr = nr;
            p.lds_data = lds;
            break;
        }
    }
    static const int roles_mode = [] { const char* e = getenv("FC_BWD_ROLES"); return e ? atoi(e) : 0; }();
    if (roles_mode && br_roles_shape_ok(p.g) && d->N < (1 << 24)) {
        for (int nr = 4; nr >= 2; nr >>= 1) {
            const size_t lds = br_roles_lds_bytes(p.g, nr);
            if (lds <= kMaxLds) {
                p.roles = true;
                p.nr = nr;
                p.lds_data = lds;
                break;
            }
        }
    }
    p.lds_filter = (size_t)2 * kBrRows * br_image_row_bytes(p.g);
    if (!p.nr || p.lds_filter > kMaxLds) return p;
    int P = num_cus() / d->R;
    if (P < 1) P = 1;
    if (P > p.items.nv_total) P = p.items.nv_total;
    p.P = P;
    p.hdump_bytes = (size_t)p.items.nv_total * d->R * br_slab_bytes(p.g) + 256;
    p.gwp_bytes = (size_t)P * d->R * F * (p.g.OT * 16) * p.g.IP * sizeof(float2);
    p.ok = true;
    return p;
}

// The ring-major backward path: two-halves mode, record-driven graphs, shapes whose (i tile, frequency) pairs fit the
// sixteen wavefronts and whose 32-row slab fits the CU's LDS.  OPT-IN (FC_BWD_RING=1: meshes of more than one round of
// 16-vertex tiles; FC_BWD_RING=2: any mesh size, used by the tests): parity-green on the whole suite, but at config 2 it
// measures 225 + 98 us against 171 + 72 us for the frequency-major pair (with 128 registers per
// wavefront only two cotangent rows per wavefront are in flight and the gather waits for L2; one 66 KB slab in flight per
// CU leaves the filter kernel waiting for HBM latency), so the default stays the frequency-major kernels.
bool backward_ring_fits(const fc_dims* d) {
    static const int mode = [] { const char* e = getenv("FC_BWD_RING"); return e ? atoi(e) : 0; }();
    if (mode == 0 || split_mode() != 2) return false;
    if (!plan_br(d).ok) return false;
    return mode == 2 || (d->N + kTile - 1) / kTile > num_cus();
}

size_t packed_bwd_ring_image_floats(const fc_dims* d) { return br_image_floats(br_geom(d->I, d->O, d->R, 2 * d->B + 1)); }

size_t backward_ring_workspace_bytes(const fc_dims* d) {
    const BrPlan p = plan_br(d);
    return p.hdump_bytes + p.gwp_bytes + 256;
}

static BrArgs make_br_args(const fc_dims* d, const BrPlan& p) {
    BrArgs a;
    a.N = d->N; a.I = d->I; a.O = d->O;
    a.g = p.g;
    a.nv_full = p.items.nv_full;
    a.nv_total = p.items.nv_total;
    a.wpk_bytes = (uint32_t)(br_image_floats(p.g) * sizeof(float));
    a.ring_bytes_w = (uint32_t)(4 * p.g.BT * p.g.IP * 64);
    a.hs_bytes = (uint32_t)br_slab_bytes(p.g);
    a.nr = p.nr;
    a.xs = p.g.F * p.g.IP + kBrXPad;
    a.region_bytes = br_region_bytes(p.g);
    static const int dbg = [] { const char* e = getenv("FC_DEBUG_BWD"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    a.stamps = debug_stamp_buffer();
    return a;
}

template <int R, int B>
static int launch_br_data(const float2* x, const float2* gy, const float* rec, const fc_csr* g, const float* wpk, float2* gx,
                          char* hdump, const BrArgs& a, const BrPlan& p, hipStream_t stream) {
    if (p.roles) {
        auto kern = fc_backward_roles_data_kernel<R, B>;
        static bool lds_ok[kMaxDevices] = {};
        if (!allow_full_lds(reinterpret_cast<const void*>(kern), p.lds_data, lds_ok)) return FC_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(p.items.grid), dim3(kRoleThreads), p.lds_data, stream, x, gy, rec, g->rowptr, g->runs, wpk, gx, hdump, a);
        return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
    }
    auto kern = fc_backward_ring_data_kernel<R, B>;
    static bool lds_ok[kMaxDevices] = {};
    if (!allow_full_lds(reinterpret_cast<const void*>(kern), p.lds_data, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.items.grid), dim3(kThreads), p.lds_data, stream, x, gy, rec, g->rowptr, g->runs, wpk, gx, hdump, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_ring_data_impl(const float* x, const float* gy, const float* rec, const fc_csr* g, const float* wpk, float* gx,
                            void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BrPlan p = plan_br(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const BrArgs a = make_br_args(d, p);
    int rc = FC_ERR_UNSUPPORTED;
#define FC_CASE(RR, BB)                                                                                                 \
    if (d->R == RR && d->B == BB)                                                                                       \
        rc = launch_br_data<RR, BB>(reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(gy), rec, g, wpk, \
                                    reinterpret_cast<float2*>(gx), static_cast<char*>(ws), a, p, stream);
    FC_FOR_EACH_SHAPE(FC_CASE)
#undef FC_CASE
    return rc;
}

template <int OT>
static int launch_br_filter(const float2* x, const char* hdump, float2* gwp, const BrArgs& a, const BrPlan& p, const fc_dims* d,
                            hipStream_t stream) {
    auto kern = fc_backward_ring_filter_kernel<OT>;
    static bool lds_ok[kMaxDevices] = {};
    if (!allow_full_lds(reinterpret_cast<const void*>(kern), p.lds_filter, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.P, d->R), dim3(kThreads), p.lds_filter, stream, x, hdump, gwp, a, d->B);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_ring_filter_impl(const float* x, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BrPlan p = plan_br(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const BrArgs a = make_br_args(d, p);
    const char* hdump = static_cast<const char*>(ws);
    float2* gwp = reinterpret_cast<float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const float2* x2 = reinterpret_cast<const float2*>(x);
    switch (p.g.OT) {
        case 1: return launch_br_filter<1>(x2, hdump, gwp, a, p, d, stream);
        case 2: return launch_br_filter<2>(x2, hdump, gwp, a, p, d, stream);
        case 3: return launch_br_filter<3>(x2, hdump, gwp, a, p, d, stream);
        case 4: return launch_br_filter<4>(x2, hdump, gwp, a, p, d, stream);
    }
    return FC_ERR_UNSUPPORTED;
}

int backward_ring_finish_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream) {
    const BrPlan p = plan_br(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float2* gwp = reinterpret_cast<const float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const int F = 2 * d->B + 1;
    const int total = d->R * F * d->O * d->I;
    hipLaunchKernelGGL(fc_reduce_gw_ring_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gwp, reinterpret_cast<float2*>(gw_eff),
                       p.P, F, d->R, d->O, d->I, p.g.OT * 16, p.g.IP);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_ring_finish_params_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, const fc_filter_params* fp,
                                     hipStream_t stream) {
    const BrPlan p = plan_br(d);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float* gwp = reinterpret_cast<const float*>(static_cast<char*>(ws) + p.hdump_bytes);
    const int F = 2 * d->B + 1;
    const size_t OP = (size_t)p.g.OT * 16, IP = (size_t)p.g.IP;
    // partial (p, r, f, o, i) at (((p*R + r)*F + f)*OP + o)*IP + i
    return reduce_param_grads_impl(gwp, (size_t)d->R * F * OP * IP, (size_t)F * OP * IP, OP * IP, IP, p.P, gw_eff, fp->zonal, fp->spherical,
                                   fp->phase, fp->ftype, fp->g_zonal, fp->g_spherical, fp->g_phase, d, stream);
}

}  // namespace fc
