// FCPrecomp on the device in three launches (reference transforms/fc_precomp.py:53-97): from the
// per-edge log-map (logMag, logAng), the transport xp, the vertex areas w and the support radius epsilon to the
// stencil the convolutions take:
//   keep edges with r = logMag / epsilon <= 1 (order preserved),
//   ln[e] = r e^{i theta},   wxp[e] = w[src] / (1e-12 + sum_{e': dst' = dst} w[src']) * xp[e],
//   supp_sten[e,q,m] = ring[e,q] * e^{i m theta} * wxp[e],  m = -B..B, with ring = linear interpolation weights on the
//   equal-area knots sqrt(q / (R-1)): two adjacent non-zeros (reference :10-27).
// The reference (and the torch version in fieldconv_amd/transforms/fc_precomp.py, which CPU tensors take and the tests
// compare this against) does it in ~40 elementwise / scatter launches.  The area sums use float atomics, as torch's
// index_add does on a GPU.
#include <hipcub/hipcub.hpp>
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

constexpr int kPrecompEdges = 128;          // edges per workgroup of the stencil kernel
constexpr int kPrecompMaxRF = 156;        // n_rings * (2 band_limit + 1) of the literal stencil builder: its rows live in LDS

// keep[e] = 1 for the edges inside the support radius; *bad_index is set when a kept edge refers to a vertex outside
// [0, N) (the reference would raise an IndexError; the kernels that follow clamp such ids)
__global__ void precomp_keep_kernel(const float* __restrict__ log_mag, const int64_t* __restrict__ edges, float eps,
                                    int32_t* __restrict__ keep, int32_t* __restrict__ bad_index, int N, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > E) return;
    const bool k = e < E && log_mag[e] / eps <= 1.0f;
    keep[e] = k ? 1 : 0;          // entry E: the scan's total lands behind it
    if (k && edges) {
        const int64_t src = edges[2 * (size_t)e], dst = edges[2 * (size_t)e + 1];
        if (src < 0 || src >= N || dst < 0 || dst >= N) atomicOr(bad_index, 1);
    }
}

// (an edge that refers to a vertex outside [0, N) is clamped here and reported by the kernels that follow: the reference
// would raise an IndexError)
__global__ void precomp_area_kernel(const int64_t* __restrict__ edges, const int32_t* __restrict__ keep, const float* __restrict__ w,
                                    float* __restrict__ total, int N, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !keep[e]) return;
    const int64_t src = min(max(edges[2 * (size_t)e], (int64_t)0), (int64_t)N - 1);
    const int64_t dst = min(max(edges[2 * (size_t)e + 1], (int64_t)0), (int64_t)N - 1);
    atomicAdd(total + dst, w[src]);
}

// thread per input edge; the kept edges of a workgroup occupy consecutive output slots, so their stencil rows are one
// contiguous piece of the output: built in LDS, written with coalesced stores
__global__ __launch_bounds__(kPrecompEdges) void precomp_stencil_kernel(
    const float* __restrict__ log_mag, const float* __restrict__ log_ang, const float2* __restrict__ xp, const float* __restrict__ w,
    const int64_t* __restrict__ edges, const int32_t* __restrict__ keep, const int32_t* __restrict__ pos, const float* __restrict__ total,
    float eps, int64_t* __restrict__ edges_out, float2* __restrict__ sten, float2* __restrict__ ln, float2* __restrict__ wxp_out, int N,
    int E, int R, int F) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* const rows = reinterpret_cast<float2*>(smem);              // [kPrecompEdges][R*F]
    const int RF = R * F, B = (F - 1) / 2;
    const int e0 = blockIdx.x * kPrecompEdges;
    const int e = e0 + threadIdx.x;
    const int first = pos[e0];
    const int last = pos[min(e0 + kPrecompEdges, E)];                   // slots [first, last) belong to this workgroup
    if (e < E && keep[e]) {
        const int slot = pos[e];
        int64_t src = edges[2 * (size_t)e], dst = edges[2 * (size_t)e + 1];
        src = min(max(src, (int64_t)0), (int64_t)N - 1);          // out-of-range ids were reported by fc_precomp_mark
        dst = min(max(dst, (int64_t)0), (int64_t)N - 1);
        const float r = log_mag[e] / eps, theta = log_ang[e];
        float sn, cs;
        sincosf(theta, &sn, &cs);
        ln[slot] = make_float2(r * cs, r * sn);
        const float scale = w[src] / (1e-12f + total[dst]);
        const float2 x = xp[e];
        const float2 wx = make_float2(scale * x.x, scale * x.y);
        wxp_out[slot] = wx;
        edges_out[2 * (size_t)slot] = src;
        edges_out[2 * (size_t)slot + 1] = dst;
        // upper knot: the first knot >= r, never knot 0
        int hi = R - 1;
        for (int q = R - 1; q >= 1; --q)
            if (sqrtf((float)q / (float)(R - 1)) >= r) hi = q;
        const float k_lo = sqrtf((float)(hi - 1) / (float)(R - 1)), k_hi = sqrtf((float)hi / (float)(R - 1));
        const float w_hi = (r - k_lo) / (k_hi - k_lo), w_lo = 1.f - w_hi;
        float2* row = rows + (slot - first) * RF;
        for (int q = 0; q < R; ++q) {
            const float rw = q == hi ? w_hi : (q == hi - 1 ? w_lo : 0.f);
            for (int f = 0; f < F; ++f) {
                float s, c;
                sincosf((float)(f - B) * theta, &s, &c);
                // (ring * freq) * wxp, in the reference's order of operations
                const float2 rf = make_float2(rw * c, rw * s);
                row[q * F + f] = cmul(rf, wx);
            }
        }
    }
    __syncthreads();
    const int count = (last - first) * RF;
    float2* out = sten + (size_t)first * RF;
    for (int idx = threadIdx.x; idx < count; idx += kPrecompEdges) out[idx] = rows[idx];
}

int precomp_area_sums(const int64_t* edges, const int32_t* keep, const float* w, float* total, int N, int E, hipStream_t s) {
    if (hipMemsetAsync(total, 0, (size_t)N * 4, s) != hipSuccess) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(precomp_area_kernel, dim3((E + 255) / 256), dim3(256), 0, s, edges, keep, w, total, N, E);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // namespace fc

// keep | pos, each (E+2) int32 rounded up to 256 bytes: pos[E] = number of kept edges, pos[E+1] = index-range flag
static size_t precomp_seg(int32_t E) { return ((size_t)(E + 2) * 4 + 255) / 256 * 256; }

extern "C" {

size_t fc_precomp_workspace_bytes(int32_t N, int32_t E) {
    if (N <= 0 || E < 0) return 0;
    size_t scan = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (const int32_t*)nullptr, (int32_t*)nullptr, E + 1);
    // keep | pos | total (N) | scan scratch
    return precomp_seg(E) * 2 + ((size_t)N * 4 + 255) / 256 * 256 + scan + 256;
}

// Step 1: marks the edges inside the support radius and scans them; afterwards the two int32 at
// fc_precomp_kept_count_ptr(workspace, E) hold the number of kept edges E' and a flag that is non-zero when a kept edge
// refers to a vertex outside [0, N) (the caller's one synchronisation reads both).  supp_edges may be NULL (no check).
int fc_precomp_mark(const float* log_mag, const int64_t* supp_edges, float epsilon, int32_t N, int32_t E, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (!log_mag || !workspace || N <= 0 || E < 0 || !(epsilon > 0.f)) return FC_ERR_BAD_ARGUMENT;
    if (workspace_bytes < fc_precomp_workspace_bytes(N, E)) return FC_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t seg = precomp_seg(E);
    char* w = static_cast<char*>(workspace);
    int32_t* keep = reinterpret_cast<int32_t*>(w);
    int32_t* pos = reinterpret_cast<int32_t*>(w + seg);
    void* scan = w + 2 * seg + ((size_t)N * 4 + 255) / 256 * 256;
    size_t scan_bytes = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep, pos, E + 1);
    if (hipMemsetAsync(pos + E + 1, 0, 4, s) != hipSuccess) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(fc::precomp_keep_kernel, dim3((E + 1 + 255) / 256), dim3(256), 0, s, log_mag, supp_edges, epsilon, keep,
                       pos + E + 1, N, E);
    if (hipcub::DeviceScan::ExclusiveSum(scan, scan_bytes, keep, pos, E + 1, s) != hipSuccess) return FC_ERR_LAUNCH;
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

const int32_t* fc_precomp_kept_count_ptr(const void* workspace, int32_t E) {
    return reinterpret_cast<const int32_t*>(static_cast<const char*>(workspace) + precomp_seg(E)) + E;
}

// Step 2: outputs sized for the E' kept edges: supp_edges_out (E',2) int64, supp_sten (E',R,F) c64, ln (E') c64, wxp (E') c64.
int fc_precomp_build(const float* log_mag, const float* log_ang, const float* xp, const float* w, const int64_t* supp_edges,
                     float epsilon, int32_t N, int32_t E, int32_t R, int32_t F, int64_t* supp_edges_out, float* supp_sten, float* ln,
                     float* wxp, void* workspace, size_t workspace_bytes, void* stream) {
    if (!log_mag || !log_ang || !xp || !w || !supp_edges || !workspace || N <= 0 || E < 0) return FC_ERR_BAD_ARGUMENT;
    if (R < 2 || F < 1 || (F & 1) == 0 || R * F > fc::kPrecompMaxRF) return FC_ERR_UNSUPPORTED;
    if (workspace_bytes < fc_precomp_workspace_bytes(N, E)) return FC_ERR_WORKSPACE;
    if (E == 0) return FC_OK;
    if (!supp_edges_out || !supp_sten || !ln || !wxp) return FC_ERR_BAD_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t seg = precomp_seg(E);
    char* wsp = static_cast<char*>(workspace);
    const int32_t* keep = reinterpret_cast<const int32_t*>(wsp);
    const int32_t* pos = reinterpret_cast<const int32_t*>(wsp + seg);
    float* total = reinterpret_cast<float*>(wsp + 2 * seg);
    if (fc::precomp_area_sums(supp_edges, keep, w, total, N, E, s) != FC_OK) return FC_ERR_LAUNCH;
    static bool lds_ok[fc::kMaxDevices] = {};
    if (!fc::allow_full_lds(reinterpret_cast<const void*>(fc::precomp_stencil_kernel), (size_t)fc::kPrecompEdges * R * F * sizeof(float2), lds_ok))
        return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(fc::precomp_stencil_kernel, dim3((E + fc::kPrecompEdges - 1) / fc::kPrecompEdges), dim3(fc::kPrecompEdges),
                       (size_t)fc::kPrecompEdges * R * F * sizeof(float2), s, log_mag, log_ang, reinterpret_cast<const float2*>(xp), w,
                       supp_edges, keep, pos, total, epsilon, supp_edges_out, reinterpret_cast<float2*>(supp_sten),
                       reinterpret_cast<float2*>(ln), reinterpret_cast<float2*>(wxp), N, E, R, F);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
