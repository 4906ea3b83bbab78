// Per-ray arithmetic of parity="strict" (shared by csrc/strict.hip - one launch per surface - and csrc/strict_fused.hip - one
// launch per level): the reference's ray-surface interaction one IEEE float32 operation at a time (deeplens/surfaces.py:391-830;
// specification oracle/scalar_trace.py).  Every translation unit that includes this is compiled without fma contraction.
#pragma once
#include <algorithm>
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

// the even-polynomial terms of the sag and of d sag / d r^2 (surfaces.py:799-809, 823-830), added to the conic part z / g
__device__ __forceinline__ float sag_poly(const Surf& s, float r2, float z) {
    if (s.n_ai > 0) {                                  // one uniform branch: spheres skip the polynomial (and its float64 powers) entirely
#pragma unroll
        for (int j = 0; j < AADFF_MAX_AI; ++j)         // constant indices: the coefficients stay in (scalar) registers
            if (j < s.n_ai) z = z + (j == 0 ? s.ai[0] * r2 : s.ai[j] * powi(r2, j + 1));
    }
    return z;
}

__device__ __forceinline__ float dsag_poly(const Surf& s, float r2, float g) {
    if (s.n_ai > 0) {
#pragma unroll
        for (int j = 0; j < AADFF_MAX_AI; ++j) {
            if (j >= s.n_ai) continue;
            if (j == 0) g = g + s.ai[0];
            else if (j == 1) g = g + (2.f * s.ai[1]) * r2;
            else g = g + ((float)(j + 1) * s.ai[j]) * powi(r2, j);
        }
    }
    return g;
}

__device__ __forceinline__ float sag(const Surf& s, float r2) {                   // surfaces.py:787-809 (power form)
    return sag_poly(s, r2, (r2 * s.c) / (1.f + sqrtf(1.f - conic_a(s, r2))));
}

__device__ __forceinline__ float dsag(const Surf& s, float r2) {                  // surfaces.py:811-830
    const float sf = sqrtf(1.f - conic_a(s, r2));
    return dsag_poly(s, r2, (((1.f + sf) + (conic_a(s, r2) / 2.f) / sf) * s.c) / ((1.f + sf) * (1.f + sf)));
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
// ten loose iterations from the vertex plane for ONE ray: bit j of `mine` = |ft| > 5e-5 in iteration j + 1, of `nans` = NaN residual
__device__ __forceinline__ float count_ray(const Surf& s, R3 o, R3 d, bool alive, unsigned& mine, unsigned& nans, int it0 = 0, int it1 = kMaxIter,
                                           float t_start = 0.f) {
    // iterations it0 .. it1 - 1 of the loose loop (it0 > 0: continued from t_start, the value after it0 iterations); returns t after them
    float t = it0 == 0 ? (s.d - o.z) / d.z : t_start;
    for (int it = it0; it < it1; ++it) {
        float ft, dfdt;
        residual<false>(s, o, d, alive, t, ft, dfdt);
        if (ft != ft) nans |= 1u << it;
        if (fabsf(ft) > kTolLoose) mine |= 1u << it;
        t = t - clamp_step(ft / (dfdt + kEps));
    }
    return t;
}

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {        // F.normalize: fused norm, three IEEE divisions
    const float n2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    const float den = fmaxf(sqrtf(n2), 1e-12f);
    x = x / den; y = y / den; z = z / den;
}

// iterations the reference's loop runs for a batch whose any-bits are `m`: the first iteration whose bit is clear, + 1 for
// the entry with ft = MAXT, at most ten (`while (|ft| > 5e-5).any() and it < 10`, surfaces.py:547)
__device__ __forceinline__ int iterations_of(unsigned m) {
    for (int it = 0; it < kMaxIter; ++it)
        if (!((m >> it) & 1u)) return it + 1;
    return kMaxIter;
}

// One surface interaction of ONE ray with the batch's iteration count (surfaces.py:391-520).
__device__ __forceinline__ void react_ray(const Surf& s, R3& o, R3& d, float& ra, int forward, int n_iter, bool have_t = false, float t_pre = 0.f) {
    const bool alive = ra > 0.f;
    float px, py, pz;
    bool valid;
    if (s.flat) {                                                                  // stop / flat: surfaces.py:409-453
        const float t = (s.d - o.z) / d.z;
        px = o.x + t * d.x; py = o.y + t * d.y; pz = o.z + t * d.z;
        valid = (sqrtf(px * px + py * py) <= s.r_f32) && alive;
    } else {
        const float t0 = (s.d - o.z) / d.z;
        float t = t0;
        if (have_t) t = t_pre;                                                     // the counting pass already holds t after n_iter iterations
        else
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
    o.x = px; o.y = py; o.z = pz;
}


__device__ __forceinline__ int ceil_log2(int x) { return x <= 2 ? 1 : 32 - __builtin_clz((unsigned)(x - 1)); }

template <typename P>
__host__ __device__ inline Surf make_surf_from(P h, int forward) {      // aadff_surface_t (host or device copy) -> Surf
    Surf s{};
    s.d = h->d; s.c = h->c; s.k = h->k; s.r_f32 = h->r; s.r2 = h->r2; s.r2_shape = h->r2_shape;
    s.eta = forward ? h->eta_fwd : h->eta_bwd;
    s.eta2 = forward ? h->eta_fwd2 : h->eta_bwd2;
    s.flat = h->kind == AADFF_SURF_STOP;
    s.spheric = h->kind == AADFF_SURF_SPHERIC;
    s.k_gt_m1 = h->k_gt_m1;
    s.n_ai = h->n_ai;
    s.refract = forward ? h->refract_fwd : h->refract_bwd;
    if (!s.flat) s.refract = 1;
    for (int j = 0; j < AADFF_MAX_AI; ++j) s.ai[j] = h->ai[j];
    return s;
}

}  // namespace strict
}  // namespace aadff
