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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

}  // namespace aadff
