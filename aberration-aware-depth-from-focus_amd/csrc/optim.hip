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
// Gradients are fp32 or bf16 (the gradients of the bf16 copy).  A one-thread launch in front computes the step's scalars
// (float64 pow / cos once) into a 4-float scratch and increments the counter.
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

__global__ void adamw_prep_kernel(AdamwSchedule s) {
    if (threadIdx.x == 0) adamw_prepare(s);
}

template <bool GRAD_BF16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ p16, long n,
                                                    const float* __restrict__ scal, float b1, float b2, float eps) {
    const float step_size = scal[0], bc2s = scal[1], decay = scal[2];
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

// dz = dy * (y > 0) and db = column sums of dz in one pass (the ReLU backward and the bias gradient of a hidden layer:
// two torch kernels per layer).
template <typename T>
__device__ __forceinline__ float ld(const T* p, long i);
template <> __device__ __forceinline__ float ld<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t* p, long i) { return bf16_to_f32(p[i]); }
__device__ __forceinline__ void st(float* p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void st(uint16_t* p, long i, float v) { p[i] = f32_to_bf16(v); }

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dz,
                                                            T* __restrict__ db, int M, int N) {
    // Block = 4 columns x 64 row lanes: the whole problem is 128 x 256 elements, so what counts is latency, not
    // coalescing — every thread has at most M/64 independent loads and the grid has N/4 blocks (with 64 columns per block
    // and 32 rows per thread the kernel took 18 us, longer than the two torch kernels it replaces).
    __shared__ float part[64][4];
    const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    float acc = 0.f;
    if (c < N)
        for (int r = rl; r < M; r += 64) {
            const long i = (long)r * N + c;
            const float g = ld(y, i) > 0.f ? ld(dy, i) : 0.f;
            st(dz, i, g);
            acc += g;
        }
    part[rl][cl] = acc;
    __syncthreads();
    if (threadIdx.x < 4 && blockIdx.x * 4 + threadIdx.x < N) {
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += part[r][threadIdx.x];
        st(db, blockIdx.x * 4 + threadIdx.x, s);
    }
}

// Output head of the PSF network and its loss gradient in one launch (deeplens/psfnet_arch.py:41-47 + nn.MSELoss,
// deeplens/psfnet.py:94-106): per row  s = sigmoid(z), pred = s / max(sum |s|, 1e-12)  (F.normalize p=1),
// L = mean((pred - target)^2) over all B x N elements, and dz = dL/dz:
//   g = 2 (pred - t) / (B N);  dL/ds = (g - sum_j g_j pred_j) / S;  dz = dL/ds * s (1 - s).
// One wave per row (N <= 128: two columns per lane).  `s` is rounded to T like torch's sigmoid output.
template <typename T>
__global__ __launch_bounds__(64) void head_kernel(const T* __restrict__ z, const float* __restrict__ target, float* __restrict__ pred,
                                                  T* __restrict__ dz, int B, int N) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float s[2], t[2];
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        s[k] = 0.f; t[k] = 0.f;
        if (c < N) {
            const float v = 1.f / (1.f + __expf(-ld(z, (long)r * N + c)));
            s[k] = sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(v)) : v;
            t[k] = target[(long)r * N + c];
            S += s[k];
        }
    }
    S = fmaxf(wave_sum(S), 1e-12f);
    const float inv = 1.f / S, scale = 2.f / ((float)B * (float)N);
    float g[2], p[2], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        p[k] = s[k] * inv;
        g[k] = scale * (p[k] - t[k]);
        dot += g[k] * p[k];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        if (c < N) {
            pred[(long)r * N + c] = p[k];
            st(dz, (long)r * N + c, (g[k] - dot) * inv * s[k] * (1.f - s[k]));
        }
    }
}

}  // namespace aadff

using namespace aadff;

extern "C" int aadff_adamw_step(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                                void* param_bf16_or_null, long n, int* step_dev, float* scratch4, float lr0, int t_max, float beta1,
                                float beta2, float eps, float weight_decay, aadff_stream_t stream) {
    AADFF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_dev && scratch4, "adamw_step: NULL pointer");
    AADFF_CHECK_ARG(n > 0 && t_max > 0, "adamw_step: bad sizes n=%ld T=%d", n, t_max);
    hipLaunchKernelGGL(adamw_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, AdamwSchedule{step_dev, scratch4, lr0, t_max, beta1, beta2, weight_decay});
    AADFF_CHECK_LAUNCH();
    const int blocks = (int)std::min<long>((n + 255) / 256, 4096);
    if (grad_is_bf16)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                           static_cast<uint16_t*>(param_bf16_or_null), n, scratch4, beta1, beta2, eps);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                           static_cast<uint16_t*>(param_bf16_or_null), n, scratch4, beta1, beta2, eps);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_relu_bwd_bias(const void* dy, const void* y, void* dz, void* db, int M, int N, int is_bf16, aadff_stream_t stream) {
    AADFF_CHECK_ARG(dy && y && dz && db && M > 0 && N > 0, "relu_bwd_bias: bad arguments");
    const dim3 g((N + 3) / 4);
    if (is_bf16)
        hipLaunchKernelGGL(relu_bwd_bias_kernel<uint16_t>, g, dim3(256), 0, (hipStream_t)stream, static_cast<const uint16_t*>(dy),
                           static_cast<const uint16_t*>(y), static_cast<uint16_t*>(dz), static_cast<uint16_t*>(db), M, N);
    else
        hipLaunchKernelGGL(relu_bwd_bias_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(dy),
                           static_cast<const float*>(y), static_cast<float*>(dz), static_cast<float*>(db), M, N);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_psfnet_head_loss_grad(const void* z, const float* target, float* pred, void* dz, int B, int N, int is_bf16,
                                           aadff_stream_t stream) {
    AADFF_CHECK_ARG(z && target && pred && dz && B > 0 && N > 0 && N <= 128, "psfnet_head_loss_grad: bad arguments (N <= 128)");
    if (is_bf16)
        hipLaunchKernelGGL(head_kernel<uint16_t>, dim3(B), dim3(64), 0, (hipStream_t)stream, static_cast<const uint16_t*>(z), target, pred,
                           static_cast<uint16_t*>(dz), B, N);
    else
        hipLaunchKernelGGL(head_kernel<float>, dim3(B), dim3(64), 0, (hipStream_t)stream, static_cast<const float*>(z), target, pred,
                           static_cast<float*>(dz), B, N);
    AADFF_CHECK_LAUNCH();
    return 0;
}
