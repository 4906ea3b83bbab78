// Shared helpers for the aadff HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "aadff.h"

namespace aadff {

void set_error(const char* fmt, ...);

#define AADFF_CHECK_ARG(cond, ...)                    \
    do {                                              \
        if (!(cond)) {                                \
            ::aadff::set_error(__VA_ARGS__);          \
            return AADFF_EINVAL;                      \
        }                                             \
    } while (0)

#define AADFF_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::aadff::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                \
        }                                                                                  \
    } while (0)

#define AADFF_CHECK_LAUNCH() AADFF_CHECK_HIP(hipGetLastError())

constexpr int kWave = 64;   // gfx950 wavefront

// armed by aadff_time_next_launch, consumed (and cleared) by the next slice-batched convolution or PSF-grid launch of the thread
extern thread_local hipEvent_t g_time_start, g_time_stop;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Read of a small PER-LAUNCH parameter (a block the host re-uploads to the SAME device address before every launch: sensor planes,
// object points, count rows, job lists, lens states) as an agent-scope atomic load, i.e. from the coherence point instead of through the
// scalar cache.  Conservative, not a measured necessity: it was introduced while hunting round 6's run-to-run differences of strict /
// edge slices beside a busy second stream, whose cause turned out to be an instruction form (tools/check_isa.py); the A/B build with
// ordinary loads below is just as clean (tools/concurrency_probe.py: 0 of 17 700 stacks either way) and just as fast.
// fresh_uniform(): the same for a wave-uniform address, result in a scalar register.
template <typename T>
#ifdef AADFF_PLAIN_PARAM_LOADS                       // A/B build for tools/concurrency_probe.py: ordinary loads
__device__ __forceinline__ T fresh(const T* p) { return *p; }
#else
__device__ __forceinline__ T fresh(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__device__ __forceinline__ int fresh_uniform(const int* p) { return __builtin_amdgcn_readfirstlane(fresh(p)); }
__device__ __forceinline__ unsigned fresh_uniform(const unsigned* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)fresh(p)); }
__device__ __forceinline__ float fresh_uniform(const float* p) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fresh(p))));
}

// AdamW step scalars under the cosine schedule (torch.optim.AdamW + CosineAnnealingLR(T_max, eta_min 0) of
// deeplens/psfnet.py:85-90), computed ONCE per step by one thread (float64 pow / cos), then the step counter advances:
// scal = {lr / bias_correction1, sqrt(bias_correction2), 1 - lr * weight_decay, lr}.
struct AdamwSchedule { int* step; float* scal; float lr0; int T; float b1, b2, wd; };

__device__ inline void adamw_prepare(const AdamwSchedule& s) {
    const int t = *s.step;
    const double frac = (double)(t < s.T ? t : s.T) / (double)s.T;
    const float lr = (float)(0.5 * (double)s.lr0 * (1.0 + cos(3.14159265358979323846 * frac)));
    const double s1 = (double)(t + 1);
    const float bc1 = (float)(1.0 - pow((double)s.b1, s1));
    s.scal[0] = lr / bc1;
    s.scal[1] = (float)sqrt(1.0 - pow((double)s.b2, s1));
    s.scal[2] = 1.f - lr * s.wd;
    s.scal[3] = lr;
    *s.step = t + 1;
}

}  // namespace aadff
