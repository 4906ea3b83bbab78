// Fused AdamW + cosine schedule for the PSF-network fit (reference: deeplens/psfnet.py:85-108: AdamW(lr) +
// CosineAnnealingLR(T_max = iters, eta_min = 0), stepped once per iteration).
//
// torch's capturable AdamW needs ~60 launches per step for the 22 parameter tensors of the MLP (46 of them one-element
// divisions for lr / bias-correction), 0.3 ms of a 0.65 ms training step; autocast adds 47 cast kernels.  Here the
// parameters live in ONE flat fp32 buffer (the nn.Module's tensors are views of it) and one launch updates them all:
//     t      = *step (completed steps, device counter -> graph-capturable, no host-side scalar)
//     lr     = 0.5 lr0 (1 + cos(pi min(t, T) / T))                      (CosineAnnealingLR closed form, float64)
//     p     *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//     p     -= lr / (1 - b1^(t+1)) * m / (sqrt(v) / sqrt(1 - b2^(t+1)) + eps)            (torch.optim.AdamW, fp32)
// and, optionally, refreshes a bf16 copy of the parameters (the bf16 leg trains on it directly: no per-step casts).
// Gradients are fp32 or bf16 (the gradients of the bf16 copy).  A second one-thread launch increments the counter.
#include <cmath>
#include <cstdint>
#include "common.h"

namespace aadff {

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {        // round to nearest even; NaN stays NaN
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool GRAD_BF16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ p16, long n,
                                                    const int* __restrict__ step, float lr0, int T, float b1, float b2,
                                                    float eps, float wd) {
    const int t = *step;
    const double frac = (double)(t < T ? t : T) / (double)T;
    const float lr = (float)(0.5 * (double)lr0 * (1.0 + cos(M_PI * frac)));
    const double s1 = (double)(t + 1);
    const float bc1 = (float)(1.0 - pow((double)b1, s1));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, s1));
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = GRAD_BF16 ? bf16_to_f32(static_cast<const uint16_t*>(g)[i]) : static_cast<const float*>(g)[i];
        float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);                 // lerp, as torch
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        pi -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p16) p16[i] = f32_to_bf16(pi);
    }
}

__global__ void bump_step_kernel(int* step) {
    if (threadIdx.x == 0) *step += 1;
}

}  // namespace aadff

using namespace aadff;

extern "C" int aadff_adamw_step(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                                void* param_bf16_or_null, long n, int* step_dev, float lr0, int t_max, float beta1,
                                float beta2, float eps, float weight_decay, aadff_stream_t stream) {
    AADFF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_dev, "adamw_step: NULL pointer");
    AADFF_CHECK_ARG(n > 0 && t_max > 0, "adamw_step: bad sizes n=%ld T=%d", n, t_max);
    const int blocks = (int)std::min<long>((n + 255) / 256, 4096);
    if (grad_is_bf16)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                           static_cast<uint16_t*>(param_bf16_or_null), n, step_dev, lr0, t_max, beta1, beta2, eps, weight_decay);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                           static_cast<uint16_t*>(param_bf16_or_null), n, step_dev, lr0, t_max, beta1, beta2, eps, weight_decay);
    AADFF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev);
    AADFF_CHECK_LAUNCH();
    return 0;
}
