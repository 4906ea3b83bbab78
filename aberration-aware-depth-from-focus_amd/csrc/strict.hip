// parity="strict": the reference's ray-surface arithmetic one IEEE float32 operation at a time, in the reference's order
// (deeplens/surfaces.py:391-830; the scalar specification is oracle/scalar_trace.py, checked op by op against the
// reference's tensor code).  One ray per lane, one launch per surface, ray state in global memory - the structure of the
// reference itself - because what has to be reproduced is not a formula but a rounding pattern:
//   * no fma contraction anywhere except the one place ATen contracts (the norm of F.normalize: x0*x0 -> fma -> fma);
//   * IEEE division and square root (hipcc's correctly rounded defaults);
//   * torch.sum(d * n, -1) as ((p0 + p1) + p2) with separately rounded products;
//   * the Newton loop runs a BATCH-WIDE number of iterations (`while (|ft| > 5e-5).any()`, surfaces.py:547): a counting
//     launch runs all ten iterations for every ray and records, per iteration, whether any ray of the batch was still above
//     the tolerance; the tracing launch then gives every ray exactly the reference's count;
//   * Python-float scalars (r^2, eta, eta^2, tolerances) enter as the float32 values the tensor ops round them to.
// What cannot be reproduced off the reference's own libraries: torch's CPU sqrt is MKL's vector sqrt, which is not
// correctly rounded for 0.7 % of arguments, so about 3.5 % of rays leave the lens one ulp away from the reference's
// (oracle/scalar_trace.py's test); sin / cos / atan of the sampling and of calc_fov stay on the host with torch itself
// (deeplens/optics.py, parity="strict").  ~20x the cost of the fused kernels; DESIGN.md section 2.
#include <cmath>
#include <cstdint>
#include "common.h"

#pragma clang fp contract(off)

namespace aadff {
namespace strict {

constexpr float kEps = 1e-9f, kTolLoose = 50e-6f, kTolTight = 10e-6f, kStep = 5.f;
constexpr int kMaxIter = 10;


__device__ __forceinline__ float powi(float x, int n) {
    // torch.pow(tensor, int): n = 2 is x*x, n = 3 is x*x*x; higher powers go through the vector pow of the reference's maths
    // library (<= 1 ulp): a correctly rounded value is the best stand-in (the terms are < 1e-3 of the sag)
    if (n == 2) return x * x;
    if (n == 3) return (x * x) * x;
    double p = (double)x;
    double r = p;
    for (int i = 1; i < n; ++i) r *= p;
    return (float)r;
}

struct Surf {                       // constants of one surface as the reference holds them
    float d, c, k, r_f32, r2, r2_shape, eta, eta2;
    int flat, spheric, k_gt_m1, n_ai, refract;
    float ai[AADFF_MAX_AI];
};

__device__ __forceinline__ float conic_a(const Surf& s, float r2) { return ((1.f + s.k) * r2) * (s.c * s.c); }

__device__ __forceinline__ float sag(const Surf& s, float r2) {                   // surfaces.py:787-809 (power form)
    float z = (r2 * s.c) / (1.f + sqrtf(1.f - conic_a(s, r2)));
    for (int j = 0; j < s.n_ai; ++j) z = z + (j == 0 ? s.ai[0] * r2 : s.ai[j] * powi(r2, j + 1));
    return z;
}

__device__ __forceinline__ float dsag(const Surf& s, float r2) {                  // surfaces.py:811-830
    const float sf = sqrtf(1.f - conic_a(s, r2));
    float g = (((1.f + sf) + (conic_a(s, r2) / 2.f) / sf) * s.c) / ((1.f + sf) * (1.f + sf));
    for (int j = 0; j < s.n_ai; ++j) {
        if (j == 0) g = g + s.ai[0];
        else if (j == 1) g = g + (2.f * s.ai[1]) * r2;
        else g = g + ((float)(j + 1) * s.ai[j]) * powi(r2, j);
    }
    return g;
}

__device__ __forceinline__ bool valid_strict(const Surf& s, float x, float y) {   // surfaces.py:724-732
    const float q = x * x + y * y;
    return s.k_gt_m1 ? (q < s.r2 && q < s.r2_shape) : (q < s.r2);
}
__device__ __forceinline__ bool valid_loose(const Surf& s, float x, float y) {    // surfaces.py:735-743
    const float q = x * x + y * y;
    return s.k_gt_m1 ? (q < s.r2_shape) : (q > 0.f);
}

struct R3 { float x, y, z; };

// one Newton residual + derivative (surfaces.py:549-553 / :572-576); STRICT selects the mask
template <bool STRICT>
__device__ __forceinline__ void residual(const Surf& s, R3 o, R3 d, bool alive, float t, float& ft, float& dfdt) {
    const float px = o.x + d.x * t, py = o.y + d.y * t, pz = o.z + d.z * t;
    const bool m = (STRICT ? valid_strict(s, px, py) : valid_loose(s, px, py)) && alive;
    const float mf = m ? 1.f : 0.f;
    const float xm = px * mf, ym = py * mf;
    const float r2 = xm * xm + ym * ym;
    ft = (sag(s, r2) + s.d) - pz;
    const float dr2dt = 2.f * ((d.x * d.x + d.y * d.y) * t + (d.x * o.x + d.y * o.y));
    dfdt = dsag(s, r2) * dr2dt - d.z;
}

__device__ __forceinline__ float clamp_step(float v) {                             // torch.clamp: NaN stays NaN
    return v != v ? v : fminf(fmaxf(v, -kStep), kStep);
}

// Counting launch: every ray runs all ten loose iterations from the vertex plane; bit j of `mask` says that some ray of
// the batch still had |ft| > 5e-5 in iteration j + 1 (dead rays take part with their masked residual, as in the reference).
__global__ __launch_bounds__(256) void newton_count_kernel(const float* __restrict__ o_in, const float* __restrict__ d_in,
                                                           const float* __restrict__ ra_in, int n, Surf s, unsigned* mask, unsigned* nan_mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned mine = 0, nans = 0;
    if (i < n) {
        const R3 o = {o_in[3 * i], o_in[3 * i + 1], o_in[3 * i + 2]}, d = {d_in[3 * i], d_in[3 * i + 1], d_in[3 * i + 2]};
        const bool alive = ra_in[i] > 0.f;
        float t = (s.d - o.z) / d.z;
        for (int it = 0; it < kMaxIter; ++it) {
            float ft, dfdt;
            residual<false>(s, o, d, alive, t, ft, dfdt);
            if (ft != ft) nans |= 1u << it;
            if (fabsf(ft) > kTolLoose) mine |= 1u << it;
            t = t - clamp_step(ft / (dfdt + kEps));
        }
    }
    for (int off = 32; off > 0; off >>= 1) { mine |= __shfl_xor((int)mine, off, 64); nans |= __shfl_xor((int)nans, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (mine) atomicOr(mask, mine);
        if (nans) atomicOr(nan_mask, nans);
    }
}

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {        // F.normalize: fused norm, three IEEE divisions
    const float n2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    const float den = fmaxf(sqrtf(n2), 1e-12f);
    x = x / den; y = y / den; z = z / den;
}

// One surface interaction with the batch's iteration count (surfaces.py:391-520).  n_iter < 0: read it from *mask (first
// iteration whose any-bit is clear, at most ten: the reference's loop condition evaluated on the counting launch's bits).
__global__ __launch_bounds__(256) void react_kernel(float* o_io, float* d_io, float* ra_io, int n, Surf s, int forward,
                                                    const unsigned* __restrict__ mask, const unsigned* __restrict__ nan_mask, int* nan_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && !s.flat) {
        // the reference exits on a NaN residual in an iteration it actually runs (surfaces.py:555-558)
        const unsigned m = *mask;
        int n_it = kMaxIter;
        for (int it = 0; it < kMaxIter; ++it)
            if (!((m >> it) & 1u)) { n_it = it + 1; break; }
        if (*nan_mask & ((1u << n_it) - 1u)) atomicOr(nan_flag, 1);
    }
    if (i >= n) return;
    R3 o = {o_io[3 * i], o_io[3 * i + 1], o_io[3 * i + 2]}, d = {d_io[3 * i], d_io[3 * i + 1], d_io[3 * i + 2]};
    float ra = ra_io[i];
    const bool alive = ra > 0.f;
    float px, py, pz;
    bool valid;
    if (s.flat) {                                                                  // stop / flat: surfaces.py:409-453
        const float t = (s.d - o.z) / d.z;
        px = o.x + t * d.x; py = o.y + t * d.y; pz = o.z + t * d.z;
        valid = (sqrtf(px * px + py * py) <= s.r_f32) && alive;
    } else {
        int n_iter = kMaxIter;
        {
            const unsigned m = *mask;                                              // loop runs while any |ft| > tol: it = first clear bit, +1 for the entry with ft = MAXT
            for (int it = 0; it < kMaxIter; ++it)
                if (!((m >> it) & 1u)) { n_iter = it + 1; break; }
        }
        const float t0 = (s.d - o.z) / d.z;
        float t = t0;
        for (int it = 0; it < n_iter; ++it) {
            float ft, dfdt;
            residual<false>(s, o, d, alive, t, ft, dfdt);
            t = t - clamp_step(ft / (dfdt + kEps));
        }
        const float t1 = t - t0;
        t = t0 + t1;                                                               // surfaces.py:565-569 (not an identity in float32)
        float ft, dfdt;
        residual<true>(s, o, d, alive, t, ft, dfdt);
        t = t - clamp_step(ft / (dfdt + kEps));
        px = o.x + t * d.x; py = o.y + t * d.y; pz = o.z + t * d.z;
        if (s.spheric) valid = (px * px + py * py <= s.r2) && (t >= 0.f) && alive;                         // Newton's own mask is discarded (:466)
        else valid = valid_strict(s, o.x + d.x * t, o.y + d.y * t) && (fabsf(ft) < kTolTight) && alive && (t > 0.f);
    }
    if (!valid) { px = o.x; py = o.y; pz = o.z; }
    ra = ra * (valid ? 1.f : 0.f);
    if (s.refract) {                                                               // surfaces.py:589-679
        float nx, ny, nz;
        if (s.flat) { nx = 0.f; ny = 0.f; nz = -1.f; }
        else if (s.spheric) {
            const float R = 1.f / s.c;
            if (s.c > 0.f) { nx = 2.f * px; ny = 2.f * py; nz = 2.f * pz - 2.f * (s.d + R); }
            else { nx = -2.f * px; ny = -2.f * py; nz = -2.f * pz + 2.f * (s.d + R); }
        } else {
            const float v = ra > 0.f ? 1.f : 0.f;
            const float xv = px * v, yv = py * v;
            const float g = dsag(s, xv * xv + yv * yv);
            nx = (g * 2.f) * xv; ny = (g * 2.f) * yv; nz = -1.f;
        }
        normalize3(nx, ny, nz);
        if (forward) { nx = -nx; ny = -ny; nz = -nz; }
        const float cosi = (d.x * nx + d.y * ny) + d.z * nz;
        const float c2 = cosi * cosi;
        const bool rv = (c2 > 0.1f) && (s.eta2 * (1.f - c2) < 1.f) && (ra > 0.f);
        const float sr = sqrtf(1.f - (s.eta2 * (1.f - c2)) * (rv ? 1.f : 0.f));
        const float ndx = sr * nx + s.eta * (d.x - cosi * nx);
        const float ndy = sr * ny + s.eta * (d.y - cosi * ny);
        const float ndz = sr * nz + s.eta * (d.z - cosi * nz);
        if (rv) { d.x = ndx; d.y = ndy; d.z = ndz; }
        ra = ra * (rv ? 1.f : 0.f);
    }
    o_io[3 * i] = px; o_io[3 * i + 1] = py; o_io[3 * i + 2] = pz;
    d_io[3 * i] = d.x; d_io[3 * i + 1] = d.y; d_io[3 * i + 2] = d.z;
    ra_io[i] = ra;
}

__global__ __launch_bounds__(256) void propagate_kernel(float* o_io, const float* __restrict__ d_in, int n, float z) {   // basics.py:255-273
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = (z - o_io[3 * i + 2]) / d_in[3 * i + 2];
    o_io[3 * i] = o_io[3 * i] + d_in[3 * i] * t;
    o_io[3 * i + 1] = o_io[3 * i + 1] + d_in[3 * i + 1] * t;
    o_io[3 * i + 2] = o_io[3 * i + 2] + d_in[3 * i + 2] * t;
}

}  // namespace strict
}  // namespace aadff

using namespace aadff;

extern "C" int aadff_trace_rays_strict(float* o, float* d, float* ra, int n, const aadff_surface_t* surf_host, int first, int last,
                                       int forward, int propagate, float z_sensor, unsigned* scratch, int* flags_or_null,
                                       aadff_stream_t stream) {
    AADFF_CHECK_ARG(o && d && ra && surf_host && scratch, "trace_rays_strict: NULL pointer");
    AADFF_CHECK_ARG(n >= 0 && first >= 0 && first <= last && last <= AADFF_MAX_SURF, "trace_rays_strict: bad range [%d,%d) or n=%d", first, last, n);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((n + 255) / 256), b(256);
    // scratch: AADFF_MAX_SURF any-masks + AADFF_MAX_SURF NaN-masks + one flag word, zeroed here (one memset per call, in stream order)
    AADFF_CHECK_HIP(hipMemsetAsync(scratch, 0, (2 * AADFF_MAX_SURF + 1) * sizeof(unsigned), st));
    int* flag = reinterpret_cast<int*>(scratch + 2 * AADFF_MAX_SURF);
    for (int k = 0; k < last - first; ++k) {
        const int i = forward ? first + k : last - 1 - k;
        const aadff_surface_t& h = surf_host[i];
        strict::Surf s{};
        s.d = h.d; s.c = h.c; s.k = h.k; s.r_f32 = h.r; s.r2 = h.r2; s.r2_shape = h.r2_shape;
        s.eta = forward ? h.eta_fwd : h.eta_bwd;
        s.eta2 = forward ? h.eta_fwd2 : h.eta_bwd2;
        s.flat = h.kind == AADFF_SURF_STOP;
        s.spheric = h.kind == AADFF_SURF_SPHERIC;
        s.k_gt_m1 = h.k_gt_m1;
        s.n_ai = h.n_ai;
        s.refract = forward ? h.refract_fwd : h.refract_bwd;
        if (!s.flat) s.refract = 1;
        for (int j = 0; j < AADFF_MAX_AI; ++j) s.ai[j] = h.ai[j];
        if (!s.flat)
            hipLaunchKernelGGL(strict::newton_count_kernel, g, b, 0, st, o, d, ra, n, s, scratch + i, scratch + AADFF_MAX_SURF + i);
        hipLaunchKernelGGL(strict::react_kernel, g, b, 0, st, o, d, ra, n, s, forward, scratch + i, scratch + AADFF_MAX_SURF + i, flag);
    }
    if (propagate) hipLaunchKernelGGL(strict::propagate_kernel, g, b, 0, st, o, d, n, z_sensor);
    if (flags_or_null) {
        // NaN in a Newton residual (the reference exits, surfaces.py:555-558): this call's flag word REPLACES the caller's word
        AADFF_CHECK_HIP(hipMemcpyAsync(flags_or_null, flag, sizeof(int), hipMemcpyDeviceToDevice, st));
    }
    AADFF_CHECK_LAUNCH();
    return 0;
}
