// TransField and ECHO descriptors with run-time loops, in the tensors' own precision (float / double): the counterpart of
// fc_generic.hip for the two blocks around the convolutions.
//
// The specialised kernels (fc_trans_field.hip, fc_echo.hip) map (channel, ring) pairs / channels to the lanes of a wavefront and keep
// their tables in LDS: at most 4 scalar inputs, 64 output channels, 8 rings, 8 raster bins, float32.  The reference's modules take
// any sizes (nn/trans_field.py:78-113, nn/echo.py:94-148), and its TransField / LiftBlock run under .double() (its ECHO and FCPrecomp
// do not: both raise a dtype error in their index_put).  Everything outside the specialised kernels' range runs here: one thread per
// output entry, plain loops over edges / channels / rings, no LDS tables, no atomics, fixed summation order.  A correctness path.
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

template <typename T> struct Cx { T x, y; };
template <typename T> __device__ __forceinline__ Cx<T> cx(T a, T b) { Cx<T> r; r.x = a; r.y = b; return r; }
template <typename T> __device__ __forceinline__ Cx<T> mul(Cx<T> a, Cx<T> b) { return cx<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T> __device__ __forceinline__ Cx<T> mul_conj(Cx<T> a, Cx<T> b) { return cx<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a conj(b)
template <typename T> __device__ __forceinline__ bool origin(Cx<T> z) {     // reference utils/field.py:10-16, EPS = 1e-7 whatever the dtype
    const T e = (T)1e-7;
    return z.x < e && z.x > -e && z.y < e && z.y > -e;
}
template <typename T> __device__ __forceinline__ T gsqrt(T v);
template <> __device__ __forceinline__ float gsqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double gsqrt<double>(double v) { return sqrt(v); }
template <typename T> __device__ __forceinline__ T soft_abs_g(Cx<T> z) { return origin(z) ? (T)0 : gsqrt<T>(z.x * z.x + z.y * z.y); }
template <typename T> __device__ __forceinline__ void gsincos(T a, T& s, T& c);
template <> __device__ __forceinline__ void gsincos<float>(float a, float& s, float& c) { sincosf(a, &s, &c); }
template <> __device__ __forceinline__ void gsincos<double>(double a, double& s, double& c) { sincos(a, &s, &c); }

// ================================================================================================ TransField
struct TfgArgs { int N, E, Cin, O, R, ftype, stride; };

template <typename T>
__device__ __forceinline__ void tfg_sten(const Cx<T>* __restrict__ lsten, long edge, int R, int r, int stride, Cx<T>& s0, Cx<T>& s1) {
    const Cx<T>* p = lsten + (edge * R + r) * stride;
    s0 = p[0];
    s1 = p[1];
}

// ang[n,i,r] = - sum_e (x[src,i] - x[n,i]) s1[e,r];  mag[n,i,r] = sum_e x[src,i] |s0[e,r]|;  s1sum[n,r] = sum_e s1[e,r]
template <typename T>
__global__ void tfg_aggregate_kernel(const T* __restrict__ x, const Cx<T>* __restrict__ lsten, const int32_t* __restrict__ rowptr,
                                     const int32_t* __restrict__ nbr, const int64_t* __restrict__ perm, Cx<T>* __restrict__ ang,
                                     T* __restrict__ mag, Cx<T>* __restrict__ s1sum, const TfgArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int IR = a.Cin * a.R;
    if (idx >= (long)a.N * IR) return;
    const int n = (int)(idx / IR), ir = (int)(idx - (long)n * IR), i = ir / a.R, r = ir - i * a.R;
    const T xd = x[(long)n * a.Cin + i];
    Cx<T> acc = cx<T>(0, 0), ssum = cx<T>(0, 0);
    T m = 0;
    for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) {
        Cx<T> s0, s1;
        tfg_sten(lsten, (long)perm[e], a.R, r, a.stride, s0, s1);
        const T xs = x[(long)nbr[e] * a.Cin + i];
        const T d = xs - xd;
        acc.x += d * s1.x;
        acc.y += d * s1.y;
        m += xs * soft_abs_g(s0);
        ssum.x += s1.x;
        ssum.y += s1.y;
    }
    ang[idx] = cx<T>(-acc.x, -acc.y);
    mag[idx] = m;
    if (i == 0) s1sum[(long)n * a.R + r] = ssum;
}

// per (n, o, i): A = sum_r ang zA, M = sum_r mag zM, u = A/|A| (1 inside the origin box), E = u e^{i phase}
template <typename T>
__device__ __forceinline__ void tfg_terms(const Cx<T>* __restrict__ ang, const T* __restrict__ mag, const T* __restrict__ zA,
                                          const T* __restrict__ zM, const T* __restrict__ phase, long n, int o, int i, const TfgArgs& a,
                                          Cx<T>& A, T& M, Cx<T>& u, Cx<T>& E, bool& live, T& inv_abs) {
    A = cx<T>(0, 0);
    M = 0;
    const T* wa = zA + ((long)o * a.Cin + i) * a.R;
    const T* wm = zM + ((long)o * a.Cin + i) * a.R;
    const Cx<T>* pa = ang + (n * a.Cin + i) * a.R;
    const T* pm = mag + (n * a.Cin + i) * a.R;
    for (int r = 0; r < a.R; ++r) {
        A.x += pa[r].x * wa[r];
        A.y += pa[r].y * wa[r];
        M += pm[r] * wm[r];
    }
    live = !origin(A);
    inv_abs = live ? (T)1 / gsqrt<T>(A.x * A.x + A.y * A.y) : (T)0;
    u = live ? cx<T>(A.x * inv_abs, A.y * inv_abs) : cx<T>(1, 0);
    T s, c;
    gsincos<T>(phase[(long)o * a.Cin + i], s, c);
    E = mul(u, cx<T>(c, s));
}

// y[n,o] = sum_i |M| E      (reference nn/trans_field.py:9-24, weightContribReal / weightContribOffset)
template <typename T>
__global__ void tfg_combine_kernel(const Cx<T>* __restrict__ ang, const T* __restrict__ mag, const T* __restrict__ zA,
                                   const T* __restrict__ zM, const T* __restrict__ phase, Cx<T>* __restrict__ y, const TfgArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.N * a.O) return;
    const long n = idx / a.O;
    const int o = (int)(idx - n * a.O);
    Cx<T> out = cx<T>(0, 0);
    for (int i = 0; i < a.Cin; ++i) {
        Cx<T> A, u, E;
        T M, inv;
        bool live;
        tfg_terms(ang, mag, zA, zM, phase, n, o, i, a, A, M, u, E, live, inv);
        const T rho = M < 0 ? -M : M;
        out.x += rho * E.x;
        out.y += rho * E.y;
    }
    y[idx] = out;
}

// adjoint of the combination per (n, o, i): GA = dL/dA (complex), GM = dL/dM, GP = dL/dphase contribution
//   y = rho E, rho = |M|, E = u c:  g_rho = Re(conj(gy) E), GM = g_rho sign(M) (softAbsolute: -1 below zero, +1 otherwise);
//   GP = rho Re(conj(gy) i E) = -rho Im(conj(gy) E);  g_u = rho gy conj(c);  u = A/|A|: GA = i u Im(conj(u) g_u) / |A| (0 inside the box)
template <typename T>
__global__ void tfg_adjoint_kernel(const Cx<T>* __restrict__ ang, const T* __restrict__ mag, const T* __restrict__ zA,
                                   const T* __restrict__ zM, const T* __restrict__ phase, const Cx<T>* __restrict__ gy,
                                   Cx<T>* __restrict__ GA, T* __restrict__ GM, T* __restrict__ GP, const TfgArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.N * a.O * a.Cin) return;
    const long no = idx / a.Cin;
    const int i = (int)(idx - no * a.Cin);
    const long n = no / a.O;
    const int o = (int)(no - n * a.O);
    Cx<T> A, u, E;
    T M, inv;
    bool live;
    tfg_terms(ang, mag, zA, zM, phase, n, o, i, a, A, M, u, E, live, inv);
    const Cx<T> g = gy[no];
    const Cx<T> ge = mul_conj(E, g);                  // E conj(gy): Re = Re(conj(gy) E), Im = Im(conj(gy) E)
    const T rho = M < 0 ? -M : M;
    GM[idx] = ge.x * (M < 0 ? (T)-1 : (T)1);
    GP[idx] = -rho * ge.y;
    Cx<T> ga = cx<T>(0, 0);
    if (live) {
        T s, c;
        gsincos<T>(phase[(long)o * a.Cin + i], s, c);
        const Cx<T> gu = mul_conj(g, cx<T>(c, s));    // gy conj(c), times rho below
        const T t = rho * (u.x * gu.y - u.y * gu.x) * inv;       // rho Im(conj(u) g_u) / |A|
        ga = cx<T>(-u.y * t, u.x * t);                // i u t
    }
    GA[idx] = ga;
}

// parameter gradients: a workgroup per (o, i) sums over the vertices in a fixed order (threads take every 256th vertex, then a
// tree over the threads): g_zA[o,i,r] = sum_n Re(conj(GA) ang[n,i,r]), g_zM[o,i,r] = sum_n GM mag[n,i,r], g_phase[o,i] = sum_n GP
template <typename T>
__global__ __launch_bounds__(256) void tfg_param_grads_kernel(const Cx<T>* __restrict__ ang, const T* __restrict__ mag,
                                                               const Cx<T>* __restrict__ GA, const T* __restrict__ GM,
                                                               const T* __restrict__ GP, T* __restrict__ g_zA, T* __restrict__ g_zM,
                                                               T* __restrict__ g_phase, const TfgArgs a) {
    __shared__ T sh[256];
    const int o = blockIdx.x / a.Cin, i = blockIdx.x - o * a.Cin;
    for (int slot = 0; slot < 2 * a.R + 1; ++slot) {
        T s = 0;
        for (long n = threadIdx.x; n < a.N; n += 256) {
            const long noi = (n * a.O + o) * a.Cin + i;
            if (slot < a.R) {
                const Cx<T> g = GA[noi], v = ang[(n * a.Cin + i) * a.R + slot];
                s += g.x * v.x + g.y * v.y;
            } else if (slot < 2 * a.R) {
                s += GM[noi] * mag[(n * a.Cin + i) * a.R + slot - a.R];
            } else {
                s += GP[noi];
            }
        }
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int d = 128; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const long oi = (long)o * a.Cin + i;
            if (slot < a.R) g_zA[oi * a.R + slot] = sh[0];
            else if (slot < 2 * a.R) g_zM[oi * a.R + slot - a.R] = sh[0];
            else if (a.ftype != 0) g_phase[oi] = sh[0];
        }
        __syncthreads();
    }
}

// g_ang[n,i,r] = sum_o GA zA[o,i,r];  g_mag[n,i,r] = sum_o GM zM[o,i,r]
template <typename T>
__global__ void tfg_vertex_grads_kernel(const Cx<T>* __restrict__ GA, const T* __restrict__ GM, const T* __restrict__ zA,
                                        const T* __restrict__ zM, Cx<T>* __restrict__ g_ang, T* __restrict__ g_mag, const TfgArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int IR = a.Cin * a.R;
    if (idx >= (long)a.N * IR) return;
    const long n = idx / IR;
    const int ir = (int)(idx - n * IR), i = ir / a.R, r = ir - i * a.R;
    Cx<T> ga = cx<T>(0, 0);
    T gm = 0;
    for (int o = 0; o < a.O; ++o) {
        const long noi = (n * a.O + o) * a.Cin + i;
        const T wa = zA[((long)o * a.Cin + i) * a.R + r], wm = zM[((long)o * a.Cin + i) * a.R + r];
        ga.x += GA[noi].x * wa;
        ga.y += GA[noi].y * wa;
        gm += GM[noi] * wm;
    }
    g_ang[idx] = ga;
    g_mag[idx] = gm;
}

// gx[j,i] = sum_r Re(conj(g_ang[j,i,r]) s1sum[j,r]) + sum_{e: src=j} sum_r ( g_mag[dst,i,r] |s0[e,r]| - Re(conj(g_ang[dst,i,r]) s1[e,r]) )
template <typename T>
__global__ void tfg_input_grad_kernel(const Cx<T>* __restrict__ lsten, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr,
                                      const int64_t* __restrict__ perm, const Cx<T>* __restrict__ g_ang, const T* __restrict__ g_mag,
                                      const Cx<T>* __restrict__ s1sum, T* __restrict__ gx, const TfgArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.N * a.Cin) return;
    const long j = idx / a.Cin;
    const int i = (int)(idx - j * a.Cin);
    T s = 0;
    for (int r = 0; r < a.R; ++r) {
        const Cx<T> g = g_ang[(j * a.Cin + i) * a.R + r], v = s1sum[j * a.R + r];
        s += g.x * v.x + g.y * v.y;
    }
    for (int e = rowptr[j]; e < rowptr[j + 1]; ++e) {
        const long dst = nbr[e];
        for (int r = 0; r < a.R; ++r) {
            Cx<T> s0, s1;
            tfg_sten(lsten, (long)perm[e], a.R, r, a.stride, s0, s1);
            const Cx<T> g = g_ang[(dst * a.Cin + i) * a.R + r];
            s += g_mag[(dst * a.Cin + i) * a.R + r] * soft_abs_g(s0) - (g.x * s1.x + g.y * s1.y);
        }
    }
    gx[idx] = s;
}

static size_t g_align(size_t v) { return (v + 255) / 256 * 256; }

template <typename T>
static int tfg_forward(const void* x, const void* lsten, const fc_csr* t, const int64_t* perm, const void* zA, const void* zM, const void* phase,
                       void* y, void* ang, void* mag, void* s1sum, const TfgArgs& a, hipStream_t st) {
    const long nir = (long)a.N * a.Cin * a.R, no = (long)a.N * a.O;
    hipLaunchKernelGGL(tfg_aggregate_kernel<T>, dim3((unsigned)((nir + 255) / 256)), dim3(256), 0, st, static_cast<const T*>(x),
                       static_cast<const Cx<T>*>(lsten), t->rowptr, t->nbr, perm, static_cast<Cx<T>*>(ang), static_cast<T*>(mag),
                       static_cast<Cx<T>*>(s1sum), a);
    hipLaunchKernelGGL(tfg_combine_kernel<T>, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, static_cast<const Cx<T>*>(ang),
                       static_cast<const T*>(mag), static_cast<const T*>(zA), static_cast<const T*>(zM), static_cast<const T*>(phase),
                       static_cast<Cx<T>*>(y), a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <typename T>
static int tfg_backward(const void* lsten, const fc_csr* s, const int64_t* perm, const void* zA, const void* zM, const void* phase,
                        const void* ang, const void* mag, const void* s1sum, const void* gy, void* gx, void* g_zA, void* g_zM, void* g_phase,
                        void* ws, const TfgArgs& a, hipStream_t st) {
    const long noi = (long)a.N * a.O * a.Cin, nir = (long)a.N * a.Cin * a.R, ni = (long)a.N * a.Cin;
    char* w = static_cast<char*>(ws);
    Cx<T>* GA = reinterpret_cast<Cx<T>*>(w);
    w += g_align(noi * 2 * sizeof(T));
    T* GM = reinterpret_cast<T*>(w);
    w += g_align(noi * sizeof(T));
    T* GP = reinterpret_cast<T*>(w);
    w += g_align(noi * sizeof(T));
    Cx<T>* g_ang = reinterpret_cast<Cx<T>*>(w);
    w += g_align(nir * 2 * sizeof(T));
    T* g_mag = reinterpret_cast<T*>(w);
    hipLaunchKernelGGL(tfg_adjoint_kernel<T>, dim3((unsigned)((noi + 255) / 256)), dim3(256), 0, st, static_cast<const Cx<T>*>(ang),
                       static_cast<const T*>(mag), static_cast<const T*>(zA), static_cast<const T*>(zM), static_cast<const T*>(phase),
                       static_cast<const Cx<T>*>(gy), GA, GM, GP, a);
    hipLaunchKernelGGL(tfg_param_grads_kernel<T>, dim3(a.O * a.Cin), dim3(256), 0, st, static_cast<const Cx<T>*>(ang),
                       static_cast<const T*>(mag), GA, GM, GP, static_cast<T*>(g_zA), static_cast<T*>(g_zM), static_cast<T*>(g_phase), a);
    hipLaunchKernelGGL(tfg_vertex_grads_kernel<T>, dim3((unsigned)((nir + 255) / 256)), dim3(256), 0, st, GA, GM, static_cast<const T*>(zA),
                       static_cast<const T*>(zM), g_ang, g_mag, a);
    hipLaunchKernelGGL(tfg_input_grad_kernel<T>, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, st, static_cast<const Cx<T>*>(lsten),
                       s->rowptr, s->nbr, perm, g_ang, g_mag, static_cast<const Cx<T>*>(s1sum), static_cast<T*>(gx), a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

// ================================================================================================ ECHO descriptors
// cell -> bin of the (2n+1)^2 raster (cells outside the disk of radius n + 1/4 alias bin 0, reference nn/echo.py:11-27); one thread
__global__ void echog_dmap_kernel(int* __restrict__ dmap, int n) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int w = 2 * n + 1;
    int d = 0;
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) {
            const long r2 = 16L * ((long)(i - n) * (i - n) + (long)(j - n) * (j - n)), lim = (4L * n + 1) * (4L * n + 1);
            dmap[i * w + j] = r2 <= lim ? d : 0;
            d += r2 <= lim ? 1 : 0;
        }
}
static int echog_hist_dim(int n) {
    int d = 0;
    for (long i = -n; i <= n; ++i)
        for (long j = -n; j <= n; ++j) d += (16 * (i * i + j * j) <= (4L * n + 1) * (4L * n + 1)) ? 1 : 0;
    return d;
}

template <typename T> __device__ __forceinline__ T gceil(T v);
template <> __device__ __forceinline__ float gceil<float>(float v) { return ceilf(v); }
template <> __device__ __forceinline__ double gceil<double>(double v) { return ceil(v); }
template <typename T> __device__ __forceinline__ T gfloor(T v);
template <> __device__ __forceinline__ float gfloor<float>(float v) { return floorf(v); }
template <> __device__ __forceinline__ double gfloor<double>(double v) { return floor(v); }

// bilinear vote of the point p scaled to raster units (reference nn/echo.py:30-61): weights, cells, weight derivatives
template <typename T>
struct VoteG {
    T w[4], dq0[4], dq1[4];
    int cell[4];
};
template <typename T>
__device__ __forceinline__ VoteG<T> echog_rasterize(Cx<T> p, int n) {
    const T nf = (T)n;
    const T q0 = p.x * nf, q1 = p.y * nf;
    auto clampv = [&](T v) { return v < -nf ? -nf : (v > nf ? nf : v); };
    const T c0 = clampv(gceil<T>(q0)), c1 = clampv(gceil<T>(q1)), f0 = clampv(gfloor<T>(q0)), f1 = clampv(gfloor<T>(q1));
    const T up0 = c0 - q0, up1 = c1 - q1, dn0 = q0 - f0, dn1 = q1 - f1;
    const int w = 2 * n + 1;
    const int ic0 = (int)c0 + n, ic1 = (int)c1 + n, if0 = (int)f0 + n, if1 = (int)f1 + n;
    VoteG<T> v;
    v.w[0] = up0 * up1; v.cell[0] = w * if0 + if1; v.dq0[0] = -up1; v.dq1[0] = -up0;
    v.w[1] = dn0 * dn1; v.cell[1] = w * ic0 + ic1; v.dq0[1] = dn1;  v.dq1[1] = dn0;
    v.w[2] = dn0 * up1; v.cell[2] = w * ic0 + if1; v.dq0[2] = up1;  v.dq1[2] = -dn0;
    v.w[3] = up0 * dn1; v.cell[3] = w * if0 + ic1; v.dq0[3] = -dn1; v.dq1[3] = up0;
    return v;
}
template <typename T>
__device__ __forceinline__ Cx<T> frame_of(Cx<T> x, bool& live) {          // exp(-i softAngle(x)): conj(x)/|x|, 1 inside the origin box
    live = !origin(x);
    if (!live) return cx<T>(1, 0);
    const T inv = (T)1 / gsqrt<T>(x.x * x.x + x.y * x.y);
    return cx<T>(x.x * inv, -x.y * inv);
}

// one thread per (vertex, channel): its histogram row lives in `hist` itself (nobody else touches it)
template <typename T>
__global__ void echog_forward_kernel(const Cx<T>* __restrict__ x, const Cx<T>* __restrict__ ln_t, const Cx<T>* __restrict__ wxp_t,
                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr, const int* __restrict__ dmap,
                                     Cx<T>* __restrict__ hist, T* __restrict__ desc, int N, int C, int n, int dS) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * C) return;
    const long v = idx / C;
    const int c = (int)(idx - v * C);
    Cx<T>* row = hist + idx * dS;
    for (int b = 0; b < dS; ++b) row[b] = cx<T>(0, 0);
    for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        const Cx<T> xv = x[(long)nbr[e] * C + c];
        bool live;
        const Cx<T> fr = frame_of(xv, live);
        if (!live) continue;                                    // zero features do not vote (reference nn/echo.py:107-113)
        const VoteG<T> vt = echog_rasterize(mul(ln_t[e], fr), n);
        const Cx<T> xw = mul(xv, wxp_t[e]);
        for (int k = 0; k < 4; ++k) {
            Cx<T>* h = row + dmap[vt.cell[k]];
            h->x += xw.x * vt.w[k];
            h->y += xw.y * vt.w[k];
        }
    }
    for (int b = 0; b < dS; ++b) desc[idx * dS + b] = soft_abs_g(row[b]);
}

template <typename T>
__global__ void echog_hist_grad_kernel(const Cx<T>* __restrict__ hist, const T* __restrict__ g_desc, Cx<T>* __restrict__ gh, long count) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const Cx<T> h = hist[idx];
    Cx<T> out = cx<T>(0, 0);
    if (!origin(h)) {
        const T s = g_desc[idx] / gsqrt<T>(h.x * h.x + h.y * h.y);
        out = cx<T>(h.x * s, h.y * s);
    }
    gh[idx] = out;
}

// one thread per (source vertex, channel): the same two gradient paths as echo_backward_kernel (fc_echo.hip)
template <typename T>
__global__ void echog_backward_kernel(const Cx<T>* __restrict__ x, const Cx<T>* __restrict__ ln_s, const Cx<T>* __restrict__ wxp_s,
                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr, const int* __restrict__ dmap,
                                      const Cx<T>* __restrict__ gh_all, Cx<T>* __restrict__ gx, int N, int C, int n, int dS) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * C) return;
    const long j = idx / C;
    const int c = (int)(idx - j * C);
    const Cx<T> xv = x[idx];
    bool live;
    const Cx<T> fr = frame_of(xv, live);
    Cx<T> out = cx<T>(0, 0);
    if (live) {
        Cx<T> gval = cx<T>(0, 0), gframe = cx<T>(0, 0);
        for (int e = rowptr[j]; e < rowptr[j + 1]; ++e) {
            const Cx<T> le = ln_s[e], we = wxp_s[e];
            const VoteG<T> vt = echog_rasterize(mul(le, fr), n);
            const Cx<T> xw = mul(xv, we);
            const Cx<T>* ghrow = gh_all + ((long)nbr[e] * C + c) * dS;
            Cx<T> acc = cx<T>(0, 0);
            T gq0 = 0, gq1 = 0;
            for (int k = 0; k < 4; ++k) {
                const Cx<T> gh = ghrow[dmap[vt.cell[k]]];
                acc.x += vt.w[k] * gh.x;
                acc.y += vt.w[k] * gh.y;
                const T t = gh.x * xw.x + gh.y * xw.y;
                gq0 += t * vt.dq0[k];
                gq1 += t * vt.dq1[k];
            }
            const Cx<T> gv = mul_conj(acc, we);
            gval.x += gv.x;
            gval.y += gv.y;
            const Cx<T> gf = mul_conj(cx<T>((T)n * gq0, (T)n * gq1), le);
            gframe.x += gf.x;
            gframe.y += gf.y;
        }
        const T gth = gframe.x * fr.y - gframe.y * fr.x;
        const T inv2 = (T)1 / (xv.x * xv.x + xv.y * xv.y);
        out = cx<T>(gval.x - xv.y * gth * inv2, gval.y + xv.x * gth * inv2);
    }
    gx[idx] = out;
}

}  // namespace fc

extern "C" {

int fc_trans_field_forward_generic(const void* x, const void* lift_sten, const fc_csr* by_target, const int64_t* slot_to_edge,
                                   const void* zonal_ang, const void* zonal_mag, const void* phase, void* y, void* ang, void* mag,
                                   void* s1sum, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride, int32_t dtype,
                                   void* stream) {
    if (!x || !by_target || !by_target->rowptr || !zonal_ang || !zonal_mag || !phase || !y || !ang || !mag || !s1sum || N <= 0 || E < 0 ||
        Cin <= 0 || O <= 0 || R <= 0 || sten_stride < 2 || (dtype != FC_F32 && dtype != FC_F64))
        return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!lift_sten || !by_target->nbr || !slot_to_edge)) return FC_ERR_BAD_ARGUMENT;
    const fc::TfgArgs a{N, E, Cin, O, R, 1, sten_stride};
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == FC_F64 ? fc::tfg_forward<double>(x, lift_sten, by_target, slot_to_edge, zonal_ang, zonal_mag, phase, y, ang, mag, s1sum, a, st)
                           : fc::tfg_forward<float>(x, lift_sten, by_target, slot_to_edge, zonal_ang, zonal_mag, phase, y, ang, mag, s1sum, a, st);
}

size_t fc_trans_field_backward_generic_workspace_bytes(int32_t N, int32_t Cin, int32_t O, int32_t R, int32_t dtype) {
    if (N <= 0 || Cin <= 0 || O <= 0 || R <= 0) return 0;
    const size_t sz = dtype == FC_F64 ? 8 : 4;
    const size_t noi = (size_t)N * O * Cin, nir = (size_t)N * Cin * R;
    return fc::g_align(noi * 2 * sz) + 2 * fc::g_align(noi * sz) + fc::g_align(nir * 2 * sz) + fc::g_align(nir * sz);
}

int fc_trans_field_backward_generic(const void* lift_sten, const fc_csr* by_source, const int64_t* slot_to_edge_s, const void* zonal_ang,
                                    const void* zonal_mag, const void* phase, const void* ang, const void* mag, const void* s1sum,
                                    const void* gy, void* gx, void* g_zonal_ang, void* g_zonal_mag, void* g_phase, void* workspace,
                                    size_t workspace_bytes, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride,
                                    int32_t ftype, int32_t dtype, void* stream) {
    if (!by_source || !by_source->rowptr || !zonal_ang || !zonal_mag || !phase || !ang || !mag || !s1sum || !gy || !gx || !g_zonal_ang ||
        !g_zonal_mag || N <= 0 || E < 0 || Cin <= 0 || O <= 0 || R <= 0 || sten_stride < 2 || (dtype != FC_F32 && dtype != FC_F64))
        return FC_ERR_BAD_ARGUMENT;
    if (ftype != 0 && !g_phase) return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!lift_sten || !by_source->nbr || !slot_to_edge_s)) return FC_ERR_BAD_ARGUMENT;
    if (!workspace || workspace_bytes < fc_trans_field_backward_generic_workspace_bytes(N, Cin, O, R, dtype)) return FC_ERR_WORKSPACE;
    const fc::TfgArgs a{N, E, Cin, O, R, ftype, sten_stride};
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == FC_F64 ? fc::tfg_backward<double>(lift_sten, by_source, slot_to_edge_s, zonal_ang, zonal_mag, phase, ang, mag, s1sum, gy, gx,
                                                      g_zonal_ang, g_zonal_mag, g_phase, workspace, a, st)
                           : fc::tfg_backward<float>(lift_sten, by_source, slot_to_edge_s, zonal_ang, zonal_mag, phase, ang, mag, s1sum, gy, gx,
                                                     g_zonal_ang, g_zonal_mag, g_phase, workspace, a, st);
}

int fc_echo_hist_dim_generic(int32_t n_bins) { return n_bins >= 1 && n_bins <= 1024 ? fc::echog_hist_dim(n_bins) : 0; }

size_t fc_echo_generic_workspace_bytes(int32_t n_bins) {
    return n_bins >= 1 && n_bins <= 1024 ? fc::g_align((size_t)(2 * n_bins + 1) * (2 * n_bins + 1) * sizeof(int)) : 0;
}

int fc_echo_forward_generic(const void* x, const void* ln_t, const void* wxp_t, const fc_csr* by_target, void* hist, void* desc,
                            void* workspace, size_t workspace_bytes, int32_t N, int32_t E, int32_t C, int32_t n_bins, int32_t dtype,
                            void* stream) {
    if (!x || !by_target || !by_target->rowptr || !hist || !desc || N <= 0 || E < 0 || C <= 0 || (dtype != FC_F32 && dtype != FC_F64))
        return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!ln_t || !wxp_t || !by_target->nbr)) return FC_ERR_BAD_ARGUMENT;
    const size_t need = fc_echo_generic_workspace_bytes(n_bins);
    if (need == 0) return FC_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < need) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* dmap = static_cast<int*>(workspace);
    const int dS = fc::echog_hist_dim(n_bins);
    const long nc = (long)N * C;
    hipLaunchKernelGGL(fc::echog_dmap_kernel, dim3(1), dim3(64), 0, st, dmap, n_bins);
    if (dtype == FC_F64)
        hipLaunchKernelGGL(fc::echog_forward_kernel<double>, dim3((unsigned)((nc + 127) / 128)), dim3(128), 0, st, static_cast<const fc::Cx<double>*>(x),
                           static_cast<const fc::Cx<double>*>(ln_t), static_cast<const fc::Cx<double>*>(wxp_t), by_target->rowptr, by_target->nbr,
                           dmap, static_cast<fc::Cx<double>*>(hist), static_cast<double*>(desc), N, C, n_bins, dS);
    else
        hipLaunchKernelGGL(fc::echog_forward_kernel<float>, dim3((unsigned)((nc + 127) / 128)), dim3(128), 0, st, static_cast<const fc::Cx<float>*>(x),
                           static_cast<const fc::Cx<float>*>(ln_t), static_cast<const fc::Cx<float>*>(wxp_t), by_target->rowptr, by_target->nbr,
                           dmap, static_cast<fc::Cx<float>*>(hist), static_cast<float*>(desc), N, C, n_bins, dS);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int fc_echo_backward_generic(const void* x, const void* ln_s, const void* wxp_s, const fc_csr* by_source, const void* hist,
                             const void* g_desc, void* gx, void* hist_grad_workspace, void* workspace, size_t workspace_bytes, int32_t N,
                             int32_t E, int32_t C, int32_t n_bins, int32_t dtype, void* stream) {
    if (!x || !by_source || !by_source->rowptr || !hist || !g_desc || !gx || !hist_grad_workspace || N <= 0 || E < 0 || C <= 0 ||
        (dtype != FC_F32 && dtype != FC_F64))
        return FC_ERR_BAD_ARGUMENT;
    if (E > 0 && (!ln_s || !wxp_s || !by_source->nbr)) return FC_ERR_BAD_ARGUMENT;
    const size_t need = fc_echo_generic_workspace_bytes(n_bins);
    if (need == 0) return FC_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < need) return FC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* dmap = static_cast<int*>(workspace);
    const int dS = fc::echog_hist_dim(n_bins);
    const long nc = (long)N * C, count = nc * dS;
    hipLaunchKernelGGL(fc::echog_dmap_kernel, dim3(1), dim3(64), 0, st, dmap, n_bins);
    if (dtype == FC_F64) {
        hipLaunchKernelGGL(fc::echog_hist_grad_kernel<double>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                           static_cast<const fc::Cx<double>*>(hist), static_cast<const double*>(g_desc), static_cast<fc::Cx<double>*>(hist_grad_workspace), count);
        hipLaunchKernelGGL(fc::echog_backward_kernel<double>, dim3((unsigned)((nc + 127) / 128)), dim3(128), 0, st, static_cast<const fc::Cx<double>*>(x),
                           static_cast<const fc::Cx<double>*>(ln_s), static_cast<const fc::Cx<double>*>(wxp_s), by_source->rowptr, by_source->nbr, dmap,
                           static_cast<const fc::Cx<double>*>(hist_grad_workspace), static_cast<fc::Cx<double>*>(gx), N, C, n_bins, dS);
    } else {
        hipLaunchKernelGGL(fc::echog_hist_grad_kernel<float>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                           static_cast<const fc::Cx<float>*>(hist), static_cast<const float*>(g_desc), static_cast<fc::Cx<float>*>(hist_grad_workspace), count);
        hipLaunchKernelGGL(fc::echog_backward_kernel<float>, dim3((unsigned)((nc + 127) / 128)), dim3(128), 0, st, static_cast<const fc::Cx<float>*>(x),
                           static_cast<const fc::Cx<float>*>(ln_s), static_cast<const fc::Cx<float>*>(wxp_s), by_source->rowptr, by_source->nbr, dmap,
                           static_cast<const fc::Cx<float>*>(hist_grad_workspace), static_cast<fc::Cx<float>*>(gx), N, C, n_bins, dS);
    }
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
