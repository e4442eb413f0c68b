// TransField, the learned 'gradient' that lifts scalar features to tangent-vector features (reference
// nn/trans_field.py:78-113 with weightContribReal / weightContribOffset :9-24):
//   ang[n,i,r] = - sum_{e: dst=n} (x[src_e,i] - x[n,i]) s1[e,r]          s1 = lift_sten[e,r,1]
//   mag[n,i,r] =   sum_{e: dst=n}  x[src_e,i] |s0[e,r]|                  s0 = lift_sten[e,r,0]
//   y[n,o]     =   sum_i |sum_r mag zonalMag[o,i,r]| * exp(i (angle(sum_r ang zonalAng[o,i,r]) + phase[o,i]))
// The reference does this with two scatter_adds over E*C_in*R elements, two einsums and polar / angle chains under
// autograd.  Here: one target-centric kernel (wavefront per vertex, lanes = (i,r) pairs for the aggregation, then
// lanes = o for the combination) and a backward of three kernels: per-vertex adjoint with per-wavefront parameter
// partials, a fixed-order reduction of those, and a source-centric gather for the input gradient.  No atomics.
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kTfWaves = 4;          // vertices per workgroup
constexpr int kTfMaxIn = 4;          // scalar input channels (the networks lift 3)
constexpr int kTfMaxR = 8;
constexpr int kTfGridWaves = 1024;   // wavefronts of the backward kernel = parameter-gradient partials

__device__ __forceinline__ float soft_abs(float2 z) { return is_origin(z) ? 0.f : sqrtf(z.x * z.x + z.y * z.y); }

struct TfArgs { int N, E, Cin, O, R, ftype, sten_stride; };

// lift_sten rows are addressed through the slot -> edge permutation: no permuted copy of the stencil
// (row stride in complex elements: 2 for a packed (E,R,2) array, 2B+1 when the caller passes the m = 0 column of the full stencil;
//  0: `lsten` is FCPrecomp's (E,8) FACTOR TABLE [q bits, w_q, w_{q+1}, 0, Re c, Im c, cos theta, sin theta] -- the stencil row is
//  w_r c e^{i m theta} (fc_precomp_graph), so the two columns are s0 = w_r c and s1 = w_r c g and no (E,R,2) array is ever built)
__device__ __forceinline__ void tf_load_sten(const float2* __restrict__ lsten, int edge, int R, int r, int stride, float2& s0, float2& s1) {
    if (stride == 0) {
        const float4* f = reinterpret_cast<const float4*>(lsten) + 2 * (size_t)edge;
        const float4 a = f[0], b = f[1];
        const int q = __float_as_int(a.x);
        const float w = r == q ? a.y : (r == q + 1 ? a.z : 0.f);
        const float2 c = make_float2(b.x, b.y);
        const float2 cg = cmul(c, make_float2(b.z, b.w));
        s0 = make_float2(w * c.x, w * c.y);
        s1 = make_float2(w * cg.x, w * cg.y);
        return;
    }
    const float2* p = lsten + ((size_t)edge * R + r) * stride;
    s0 = p[0];
    s1 = p[1];
}

template <bool FACTORS>
__device__ __forceinline__ void tf_load_sten_t(const float2* __restrict__ lsten, int edge, int R, int r, int stride, float2& s0, float2& s1) {
    tf_load_sten(lsten, edge, R, r, FACTORS ? 0 : (stride ? stride : 1), s0, s1);
}

// The per-edge chain of the two gathers (slot -> edge and neighbour -> stencil row and neighbour's value) is latency, not data: kTfAhead
// edges go through each stage together -- straight-line code, no branch per edge: an edge past the end is the last edge again and its
// contribution is dropped by a select -- so a group costs two round trips instead of two per edge.  Sums in edge order as before.
constexpr int kTfAhead = 8;

// the filters of the combination stages, staged once per workgroup: [o][i][r] as in memory
__device__ __forceinline__ void tf_stage_filters(float* s_zA, float* s_zM, const float* __restrict__ zA, const float* __restrict__ zM, int count) {
    for (int idx = threadIdx.x; idx < count; idx += kTfWaves * kWave) {
        s_zA[idx] = zA[idx];
        s_zM[idx] = zM[idx];
    }
}

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(kTfWaves * kWave) void trans_field_forward_kernel(
    const float* __restrict__ x, const float2* __restrict__ lsten, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ nbr, const int64_t* __restrict__ perm, const float* __restrict__ zA, const float* __restrict__ zM,
    const float* __restrict__ phase, float2* __restrict__ y, float2* __restrict__ ang_out, float* __restrict__ mag_out,
    float2* __restrict__ s1sum_out, const TfArgs a, const int wpv) {
    __shared__ float2 s_ang[kTfWaves][kTfMaxIn * kTfMaxR];
    __shared__ float s_mag[kTfWaves][kTfMaxIn * kTfMaxR];
    __shared__ float2 s_s1[kTfWaves][kTfMaxIn * kTfMaxR];
    __shared__ float2 s_cs[kWave * kTfMaxIn];              // (cos, sin) of phase[o][i]
    __shared__ float s_zA[kWave * kTfMaxIn * kTfMaxR], s_zM[kWave * kTfMaxIn * kTfMaxR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Cin = a.Cin, R = a.R, O = a.O, IR = Cin * R;
    for (int idx = threadIdx.x; idx < O * Cin; idx += kTfWaves * kWave) {
        float s, c;
        sincosf(phase[idx], &s, &c);
        s_cs[idx] = make_float2(c, s);
    }
    tf_stage_filters(s_zA, s_zM, zA, zM, O * IR);
    __syncthreads();
    // `wpv` (1, 2 or 4) wavefronts share a vertex, each with every wpv-th in-edge (meshes with few vertices and wide
    // supports, see fc_echo.hip); their sums are added in wavefront order
    const int n = (blockIdx.x * kTfWaves + wave) / wpv;
    const int sub = wave % wpv;
    const bool active = n < a.N;
    // ---- aggregation: lane = (i, r)
    const int li = lane < IR ? lane / R : 0, lr = lane < IR ? lane - (lane / R) * R : 0;
    float2 ang = make_float2(0.f, 0.f), s1sum = ang;
    float mag = 0.f;
    const float xd = active ? x[(size_t)n * Cin + li] : 0.f;
    const int beg = active ? rowptr[n] : 0, end = active ? rowptr[n + 1] : 0;
    auto gather = [&](auto factors) {
        constexpr bool FACTORS = decltype(factors)::value;
        for (int e0 = beg + sub; e0 < end; e0 += kTfAhead * wpv) {
            int src[kTfAhead], edge[kTfAhead];
#pragma unroll
            for (int u = 0; u < kTfAhead; ++u) {
                const int e = min(e0 + u * wpv, end - 1);
                src[u] = nbr[e];
                edge[u] = (int)perm[e];
            }
            float2 s0[kTfAhead], s1[kTfAhead];
            float xs[kTfAhead];
#pragma unroll
            for (int u = 0; u < kTfAhead; ++u) {
                tf_load_sten_t<FACTORS>(lsten, edge[u], R, lr, a.sten_stride, s0[u], s1[u]);
                xs[u] = x[(size_t)src[u] * Cin + li];
            }
#pragma unroll
            for (int u = 0; u < kTfAhead; ++u) {
                // (the updated sums are formed exactly as in a per-edge loop -- same fused multiply-adds -- and dropped by a select past the end)
                const bool on = e0 + u * wpv < end;
                const float d = xs[u] - xd;
                float2 na = ang, ns = s1sum;
                float nm = mag;
                na.x += d * s1[u].x;
                na.y += d * s1[u].y;
                nm += xs[u] * soft_abs(s0[u]);
                ns.x += s1[u].x;
                ns.y += s1[u].y;
                ang = on ? na : ang;
                mag = on ? nm : mag;
                s1sum = on ? ns : s1sum;
            }
        }
    };
    if (a.sten_stride == 0) gather(std::true_type{});
    else gather(std::false_type{});
    ang = make_float2(-ang.x, -ang.y);
    if (wpv > 1) {
        if (lane < IR) { s_ang[wave][lane] = ang; s_mag[wave][lane] = mag; s_s1[wave][lane] = s1sum; }
        __syncthreads();
        if (sub != 0 || !active) return;
        if (lane < IR)
            for (int s = 1; s < wpv; ++s) {
                const float2 pa = s_ang[wave + s][lane], ps = s_s1[wave + s][lane];
                ang.x += pa.x; ang.y += pa.y;
                mag += s_mag[wave + s][lane];
                s1sum.x += ps.x; s1sum.y += ps.y;
            }
    } else if (!active) {
        return;
    }
    if (lane < IR) {
        s_ang[wave][lane] = ang;
        s_mag[wave][lane] = mag;
        ang_out[(size_t)n * IR + lane] = ang;
        mag_out[(size_t)n * IR + lane] = mag;
        if (li == 0) s1sum_out[(size_t)n * R + lr] = s1sum;
    }
    // (same wavefront wrote and reads s_ang / s_mag: LDS operations of a wavefront complete in order)
    // ---- combination: lane = o
    if (lane < O) {
        float2 out = make_float2(0.f, 0.f);
        for (int i = 0; i < Cin; ++i) {
            float2 A = make_float2(0.f, 0.f);
            float M = 0.f;
            const float* wa = s_zA + (lane * Cin + i) * R;
            const float* wm = s_zM + (lane * Cin + i) * R;
            for (int r = 0; r < R; ++r) {
                const float2 sa = s_ang[wave][i * R + r];
                A.x += sa.x * wa[r];
                A.y += sa.y * wa[r];
                M += s_mag[wave][i * R + r] * wm[r];
            }
            // exp(i softAngle(A)) = A / |A|, or 1 inside the origin box (reference utils/field.py:40-48)
            const float2 u = is_origin(A) ? make_float2(1.f, 0.f) : make_float2(A.x * __frsqrt_rn(A.x * A.x + A.y * A.y), A.y * __frsqrt_rn(A.x * A.x + A.y * A.y));
            const float2 E = cmul(u, s_cs[lane * Cin + i]);
            const float rho = fabsf(M);
            out.x += rho * E.x;
            out.y += rho * E.y;
        }
        y[(size_t)n * O + lane] = out;
    }
}

// ------------------------------------------------------------------------------------------ backward, per vertex
// Persistent wavefronts; lane = o keeps its rows of the parameter gradients in registers and writes one partial at the end.
__global__ __launch_bounds__(kTfWaves * kWave) void trans_field_backward_vertex_kernel(
    const float2* __restrict__ ang_in, const float* __restrict__ mag_in, const float2* __restrict__ s1sum,
    const float* __restrict__ zA, const float* __restrict__ zM, const float* __restrict__ phase, const float2* __restrict__ gy,
    float2* __restrict__ g_ang, float* __restrict__ g_mag, float* __restrict__ gx_dst, float* __restrict__ partial,
    const TfArgs a, const int total_waves) {
    __shared__ float2 s_ang[kTfWaves][kTfMaxIn * kTfMaxR];
    __shared__ float s_mag[kTfWaves][kTfMaxIn * kTfMaxR];
    __shared__ float2 s_gA[kTfWaves][kWave * kTfMaxIn];
    __shared__ float s_gM[kTfWaves][kWave * kTfMaxIn];
    __shared__ float s_t[kTfWaves][kTfMaxIn * kTfMaxR];
    __shared__ float2 s_cs[kWave * kTfMaxIn];
    __shared__ float s_zA[kWave * kTfMaxIn * kTfMaxR], s_zM[kWave * kTfMaxIn * kTfMaxR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Cin = a.Cin, R = a.R, O = a.O, IR = Cin * R;
    for (int idx = threadIdx.x; idx < O * Cin; idx += kTfWaves * kWave) {
        float s, c;
        sincosf(phase[idx], &s, &c);
        s_cs[idx] = make_float2(c, s);
    }
    tf_stage_filters(s_zA, s_zM, zA, zM, O * IR);
    __syncthreads();
    const int gw = blockIdx.x * kTfWaves + wave;
    const int li = lane < IR ? lane / R : 0, lr = lane < IR ? lane - (lane / R) * R : 0;
    float gzA[kTfMaxIn][kTfMaxR], gzM[kTfMaxIn][kTfMaxR], gph[kTfMaxIn];
#pragma unroll
    for (int i = 0; i < kTfMaxIn; ++i) {
        gph[i] = 0.f;
#pragma unroll
        for (int r = 0; r < kTfMaxR; ++r) { gzA[i][r] = 0.f; gzM[i][r] = 0.f; }
    }
    for (int n = gw; n < a.N; n += total_waves) {
        if (lane < IR) {
            s_ang[wave][lane] = ang_in[(size_t)n * IR + lane];
            s_mag[wave][lane] = mag_in[(size_t)n * IR + lane];
        }
        const float2 g = lane < O ? gy[(size_t)n * O + lane] : make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kTfMaxIn; ++i) {
            if (i < Cin) {
                float2 A = make_float2(0.f, 0.f);
                float M = 0.f;
                const int oc = lane < O ? lane : 0;
                const float* wa = s_zA + (oc * Cin + i) * R;
                const float* wm = s_zM + (oc * Cin + i) * R;
#pragma unroll
                for (int r = 0; r < kTfMaxR; ++r)
                    if (r < R) {
                        const float2 sa = s_ang[wave][i * R + r];
                        A.x += sa.x * wa[r];
                        A.y += sa.y * wa[r];
                        M += s_mag[wave][i * R + r] * wm[r];
                    }
                const bool orgA = is_origin(A);
                const float n2 = A.x * A.x + A.y * A.y;
                const float inv = orgA ? 0.f : __frsqrt_rn(n2);
                const float2 u = orgA ? make_float2(1.f, 0.f) : make_float2(A.x * inv, A.y * inv);
                const float2 E = cmul(u, s_cs[oc * Cin + i]);
                const float rho = fabsf(M);
                const float g_rho = g.x * E.x + g.y * E.y;                      // Re(conj(g) E)
                const float g_phi = rho * (g.y * E.x - g.x * E.y);              // Re(conj(g) i rho E)
                const float gM = M < 0.f ? -g_rho : g_rho;                      // softAbsolute (reference utils/field.py:18-26)
                // angle(A): g_A = g_phi * i A / |A|^2 ; none inside the origin box
                const float sA = orgA ? 0.f : g_phi * inv * inv;
                const float2 gA = make_float2(-A.y * sA, A.x * sA);
                if (lane < O) {
                    gph[i] += g_phi;
#pragma unroll
                    for (int r = 0; r < kTfMaxR; ++r)
                        if (r < R) {
                            const float2 sa = s_ang[wave][i * R + r];
                            gzA[i][r] += gA.x * sa.x + gA.y * sa.y;                 // zonalAng is real: Re(conj(gA) ang)
                            gzM[i][r] += gM * s_mag[wave][i * R + r];
                        }
                }
                s_gA[wave][lane * Cin + i] = lane < O ? gA : make_float2(0.f, 0.f);
                s_gM[wave][lane * Cin + i] = lane < O ? gM : 0.f;
            }
        }
        // ---- adjoint of the two contractions over o: lane = (i, r)
        float2 ga = make_float2(0.f, 0.f);
        float gm = 0.f;
        if (lane < IR) {
            for (int o = 0; o < O; ++o) {
                const float wa = s_zA[(o * Cin + li) * R + lr], wm = s_zM[(o * Cin + li) * R + lr];
                const float2 t = s_gA[wave][o * Cin + li];
                ga.x += t.x * wa;
                ga.y += t.y * wa;
                gm += s_gM[wave][o * Cin + li] * wm;
            }
            g_ang[(size_t)n * IR + lane] = ga;
            g_mag[(size_t)n * IR + lane] = gm;
            // ang depends on x[n,i] through + x[n,i] * sum_e s1[e,r]
            const float2 ss = s1sum[(size_t)n * R + lr];
            s_t[wave][lane] = ga.x * ss.x + ga.y * ss.y;
        }
        if (lane < Cin) {
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc += s_t[wave][lane * R + r];
            gx_dst[(size_t)n * Cin + lane] = acc;
        }
    }
    // partial[wave][o][Cin][2R+1]
    if (lane < O) {
        float* p = partial + ((size_t)gw * O + lane) * Cin * (2 * R + 1);
#pragma unroll
        for (int i = 0; i < kTfMaxIn; ++i)
            if (i < Cin) {
#pragma unroll
                for (int r = 0; r < kTfMaxR; ++r)
                    if (r < R) {
                        p[i * (2 * R + 1) + r] = gzA[i][r];
                        p[i * (2 * R + 1) + R + r] = gzM[i][r];
                    }
                p[i * (2 * R + 1) + 2 * R] = gph[i];
            }
    }
}

// fixed-order sum of the per-wavefront partials: 64 outputs (o, i, slot) per workgroup, 16 strided part-sums each, then a
// fixed-order sum of the 16
constexpr int kTfReduceWays = 16;
__global__ __launch_bounds__(kTfReduceWays * kWave) void trans_field_reduce_kernel(
    const float* __restrict__ partial, float* __restrict__ g_zA, float* __restrict__ g_zM, float* __restrict__ g_phase, int nparts,
    int O, int Cin, int R, int ftype) {
    __shared__ float s_part[kTfReduceWays][kWave];
    const int lane = threadIdx.x & 63, way = threadIdx.x >> 6;
    const int idx = blockIdx.x * kWave + lane;
    const int per = 2 * R + 1, total = O * Cin * per;
    float s = 0.f;
    if (idx < total)
        for (int p = way; p < nparts; p += kTfReduceWays) s += partial[(size_t)p * total + idx];
    s_part[way][lane] = s;
    __syncthreads();
    if (way != 0 || idx >= total) return;
    s = 0.f;
#pragma unroll
    for (int q = 0; q < kTfReduceWays; ++q) s += s_part[q][lane];
    const int oi = idx / per, slot = idx - oi * per;
    if (slot < R) g_zA[(size_t)oi * R + slot] = s;
    else if (slot < 2 * R) g_zM[(size_t)oi * R + slot - R] = s;
    else if (ftype != 0) g_phase[oi] = s;
}

// ------------------------------------------------------------------------------------------ backward, input gradient
// gx[j,i] = gx_dst[j,i] + sum_{e: src=j} sum_r ( -Re(conj(g_ang[dst_e,i,r]) s1[e,r]) + g_mag[dst_e,i,r] |s0[e,r]| )
__global__ __launch_bounds__(kTfWaves * kWave) void trans_field_backward_input_kernel(
    const float2* __restrict__ lsten, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr,
    const int64_t* __restrict__ perm, const float2* __restrict__ g_ang, const float* __restrict__ g_mag,
    const float* __restrict__ gx_dst, float* __restrict__ gx, const TfArgs a, const int wpv) {
    __shared__ float s_t[kTfWaves][kTfMaxIn * kTfMaxR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Cin = a.Cin, R = a.R, IR = Cin * R;
    const int j = (blockIdx.x * kTfWaves + wave) / wpv;
    const int sub = wave % wpv;
    const bool active = j < a.N;
    const int lr = lane < IR ? lane - (lane / R) * R : 0;
    const int lc = lane < IR ? lane : 0;
    float acc = 0.f;
    const int beg = active ? rowptr[j] : 0, end = active ? rowptr[j + 1] : 0;
    auto gather = [&](auto factors) {
        constexpr bool FACTORS = decltype(factors)::value;
        for (int e0 = beg + sub; e0 < end; e0 += kTfAhead * wpv) {
            int dst[kTfAhead], edge[kTfAhead];
#pragma unroll
            for (int u = 0; u < kTfAhead; ++u) {
                const int e = min(e0 + u * wpv, end - 1);
                dst[u] = nbr[e];
                edge[u] = (int)perm[e];
            }
            float2 s0[kTfAhead], s1[kTfAhead], ga[kTfAhead];
            float gm[kTfAhead];
#pragma unroll
            for (int u = 0; u < kTfAhead; ++u) {
                tf_load_sten_t<FACTORS>(lsten, edge[u], R, lr, a.sten_stride, s0[u], s1[u]);
                ga[u] = g_ang[(size_t)dst[u] * IR + lc];
                gm[u] = g_mag[(size_t)dst[u] * IR + lc];
            }
#pragma unroll
            for (int u = 0; u < kTfAhead; ++u) {
                float nacc = acc;
                nacc += gm[u] * soft_abs(s0[u]) - (ga[u].x * s1[u].x + ga[u].y * s1[u].y);       // (as in a per-edge loop; dropped past the end)
                acc = (e0 + u * wpv < end) ? nacc : acc;
            }
        }
    };
    if (a.sten_stride == 0) gather(std::true_type{});
    else gather(std::false_type{});
    if (lane < IR) s_t[wave][lane] = acc;
    if (wpv > 1) __syncthreads();
    if (sub != 0 || !active) return;
    if (lane < Cin) {
        float s = gx_dst[(size_t)j * Cin + lane];
        for (int w = 0; w < wpv; ++w)
            for (int r = 0; r < R; ++r) s += s_t[wave + w][lane * R + r];
        gx[(size_t)j * Cin + lane] = s;
    }
}

static int tf_waves_per_vertex(int N, int E) {
    const long deg = N > 0 ? (long)E / N : 0;
    if (deg >= 64 && N < 65536) return 4;
    if (deg >= 32 && N < 131072) return 2;
    return 1;
}

static bool tf_supported(int Cin, int O, int R) { return Cin >= 1 && Cin <= kTfMaxIn && R >= 1 && R <= kTfMaxR && O >= 1 && O <= kWave; }

}  // namespace fc

extern "C" {

int fc_trans_field_forward(const float* x, const float* lift_sten, const fc_csr* by_target, const int64_t* slot_to_edge,
                           const float* zonal_ang, const float* zonal_mag, const float* phase, float* y, float* ang, float* mag,
                           float* s1sum, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride, void* stream) {
    if (!x || !by_target || !by_target->rowptr || !zonal_ang || !zonal_mag || !phase || !y || !ang || !mag || !s1sum || N <= 0 || E < 0)
        return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!lift_sten || !by_target->nbr || !slot_to_edge)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::tf_supported(Cin, O, R)) return FC_ERR_UNSUPPORTED;
    if (sten_stride < 2 && sten_stride != 0) return FC_ERR_BAD_ARGUMENT;
    const fc::TfArgs a{N, E, Cin, O, R, 1, sten_stride};
    const int wpv = fc::tf_waves_per_vertex(N, E), per_wg = fc::kTfWaves / wpv;
    hipLaunchKernelGGL(fc::trans_field_forward_kernel, dim3((N + per_wg - 1) / per_wg), dim3(fc::kTfWaves * fc::kWave), 0,
                       static_cast<hipStream_t>(stream), x, reinterpret_cast<const float2*>(lift_sten), by_target->rowptr, by_target->nbr,
                       slot_to_edge, zonal_ang, zonal_mag, phase, reinterpret_cast<float2*>(y), reinterpret_cast<float2*>(ang), mag,
                       reinterpret_cast<float2*>(s1sum), a, wpv);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

size_t fc_trans_field_backward_workspace_bytes(int32_t N, int32_t Cin, int32_t O, int32_t R) {
    if (N <= 0 || !fc::tf_supported(Cin, O, R)) return 0;
    const size_t IR = (size_t)Cin * R;
    // g_ang (N,IR) c64 | g_mag (N,IR) f32 | gx_dst (N,Cin) f32 | partials [waves][O][Cin][2R+1]
    return (size_t)N * IR * 8 + (size_t)N * IR * 4 + (size_t)N * Cin * 4 + (size_t)fc::kTfGridWaves * O * Cin * (2 * R + 1) * 4 + 256;
}

int fc_trans_field_backward(const float* lift_sten, const fc_csr* by_source, const int64_t* slot_to_edge_s, const float* zonal_ang,
                            const float* zonal_mag, const float* phase, const float* ang, const float* mag, const float* s1sum,
                            const float* gy, float* gx, float* g_zonal_ang, float* g_zonal_mag, float* g_phase, void* workspace,
                            size_t workspace_bytes, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride,
                            int32_t ftype, void* stream) {
    if (!by_source || !by_source->rowptr || !zonal_ang || !zonal_mag || !phase || !ang || !mag || !s1sum || !gy || !gx || !g_zonal_ang ||
        !g_zonal_mag || N <= 0 || E < 0)
        return FC_ERR_BAD_ARGUMENT;
    if (ftype != 0 && !g_phase) return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!lift_sten || !by_source->nbr || !slot_to_edge_s)) return FC_ERR_BAD_ARGUMENT;
    if (!fc::tf_supported(Cin, O, R)) return FC_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < fc_trans_field_backward_workspace_bytes(N, Cin, O, R)) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t IR = (size_t)Cin * R;
    char* w = static_cast<char*>(workspace);
    float2* g_ang = reinterpret_cast<float2*>(w);
    float* g_mag = reinterpret_cast<float*>(w + (size_t)N * IR * 8);
    float* gx_dst = g_mag + (size_t)N * IR;
    float* partial = gx_dst + (size_t)N * Cin;
    if (sten_stride < 2 && sten_stride != 0) return FC_ERR_BAD_ARGUMENT;
    const fc::TfArgs a{N, E, Cin, O, R, ftype, sten_stride};
    // persistent wavefronts = parameter-gradient partials: a quarter of the vertices on small meshes
    int waves = (N < 4 * fc::kTfGridWaves) ? ((N + 3) / 4 + fc::kTfWaves - 1) / fc::kTfWaves * fc::kTfWaves : fc::kTfGridWaves;
    hipLaunchKernelGGL(fc::trans_field_backward_vertex_kernel, dim3(waves / fc::kTfWaves), dim3(fc::kTfWaves * fc::kWave), 0, s,
                       reinterpret_cast<const float2*>(ang), mag, reinterpret_cast<const float2*>(s1sum), zonal_ang, zonal_mag, phase,
                       reinterpret_cast<const float2*>(gy), g_ang, g_mag, gx_dst, partial, a, waves);
    const int total = O * Cin * (2 * R + 1);
    hipLaunchKernelGGL(fc::trans_field_reduce_kernel, dim3((total + fc::kWave - 1) / fc::kWave), dim3(fc::kTfReduceWays * fc::kWave), 0, s, partial, g_zonal_ang, g_zonal_mag,
                       g_phase, waves, O, Cin, R, ftype);
    const int wpv = fc::tf_waves_per_vertex(N, E), per_wg = fc::kTfWaves / wpv;
    hipLaunchKernelGGL(fc::trans_field_backward_input_kernel, dim3((N + per_wg - 1) / per_wg), dim3(fc::kTfWaves * fc::kWave), 0,
                       s, reinterpret_cast<const float2*>(lift_sten), by_source->rowptr, by_source->nbr, slot_to_edge_s, g_ang, g_mag,
                       gx_dst, gx, a, wpv);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
