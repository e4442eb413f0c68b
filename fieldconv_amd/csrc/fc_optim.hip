// Fused Adam over one flat parameter buffer: the segmentation network has ~60 parameter tensors with
// ~1 M entries in total; torch's multi-tensor Adam is a dozen launches of list kernels (0.5 ms per step at config 3, a
// quarter of the whole forward + backward), this is two: a one-thread tick of the device-side step counter (so that the
// step is capturable in a HIP graph: nothing step-dependent comes from the host) and one elementwise update with the
// arithmetic of torch.optim.Adam (L2 weight decay, no amsgrad):
//   g' = g + wd p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

__global__ void adam_tick_kernel(float* __restrict__ step) { step[0] += 1.f; }

__global__ __launch_bounds__(256) void adam_update_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                          float4* __restrict__ v, const float* __restrict__ step, size_t n4, float lr,
                                                          float b1, float b2, float eps, float wd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const float t = step[0];
    const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    float4 pp = p[idx], mm = m[idx], vv = v[idx];
    const float4 gg = g[idx];
    float* pa = reinterpret_cast<float*>(&pp);
    float* ma = reinterpret_cast<float*>(&mm);
    float* va = reinterpret_cast<float*>(&vv);
    const float* ga = reinterpret_cast<const float*>(&gg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gk = ga[k] + wd * pa[k];
        ma[k] = b1 * ma[k] + (1.f - b1) * gk;
        va[k] = b2 * va[k] + (1.f - b2) * gk * gk;
        pa[k] -= step_size * (ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps));
    }
    p[idx] = pp;
    m[idx] = mm;
    v[idx] = vv;
}

}  // namespace fc

extern "C" {

int fc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step, size_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step || (n & 3)) return FC_ERR_BAD_ARGUMENT;
    if (n == 0) return FC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(fc::adam_tick_kernel, dim3(1), dim3(1), 0, s, step);
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(fc::adam_update_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<float4*>(params),
                       reinterpret_cast<const float4*>(grads), reinterpret_cast<float4*>(exp_avg), reinterpret_cast<float4*>(exp_avg_sq),
                       step, n4, lr, beta1, beta2, eps, weight_decay);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

}  // extern "C"
