// Ray tracing through the lens surfaces, PSF accumulation and refocus for gfx950.
//
// One thread per ray; the surface table is wave-uniform (scalar loads, SGPR operands), so every
// branch on the surface kind is uniform.  Spheres are intersected in closed form (the reference
// keeps Newton's root but discards its convergence mask for them, deeplens/surfaces.py:466); even
// aspheres run the reference's Newton iteration, whose loop exits per WAVE (`__any`) where the
// reference's exits per batch (`while (...).any()`, deeplens/surfaces.py:547) - extra steps on
// converged rays are the same arithmetic the reference performs.  Semantics: SURVEY.md Appendix
// A.1/A.2, citations per function below (paths relative to the reference repo).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <hip/hip_ext.h>
#include "common.h"

namespace aadff {

constexpr float kEps = 1e-9f;            // deeplens/basics.py:34
constexpr float kMaxT = 1e5f;            // deeplens/basics.py:32
constexpr int kNewtonMaxIter = 10;       // deeplens/surfaces.py:26
constexpr float kTolTight = 10e-6f;      // deeplens/surfaces.py:27
constexpr float kTolLoose = 50e-6f;      // deeplens/surfaces.py:28
constexpr float kStepBound = 5.f;        // deeplens/surfaces.py:29
// fused kernels: a Newton update shorter than aadff_surface_t::newton_step_tol (<= 10 um, set per surface by the host from
// the surface's largest profile curvature) ends the loop (see newton2)
[[maybe_unused]] constexpr float kTwoPiHi = 3.14159274101257324f;   // (float)np.pi (libm sin/cos builds)

struct Ray {
    float ox, oy, oz, dx, dy, dz, ra;
};

// ---- arithmetic primitives -------------------------------------------------------------
// Default: hardware reciprocal / sqrt / rsqrt (1 ulp) instead of the IEEE-exact sequences (11
// instructions per division, ~10 per sqrt).  Their error (~1e-7 relative on quantities of a few mm)
// is three orders below the fp32 noise the reference itself carries at the first surface (one ulp of
// t ~ 1500 mm is 1.2e-4 mm); -DAADFF_TRACE_IEEE restores the exact forms.
#ifdef AADFF_TRACE_IEEE
__device__ __forceinline__ float frcp(float x) { return 1.f / x; }
__device__ __forceinline__ float fsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return 1.f / sqrtf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a / b; }
#else
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#endif

// ---- even-asphere sag and d(sag)/d(r^2): deeplens/surfaces.py:787-830 ----
// sag = c r^2 / (1 + sf) + sum a_j r^(2j),  sf = sqrt(1 - (1+k) c^2 r^2).  The reference's expression for
// the conic part of d(sag)/d(r^2), (1 + sf + (1+k) r^2 c^2 / (2 sf)) c / (1 + sf)^2, is identically
// c / (2 sf); the closed form is used here (one reciprocal instead of two divisions).
__device__ __forceinline__ void sag_and_slope(const aadff_surface_t& s, float r2, float& sag, float& slope) {
    const float a = (1.f + s.k) * r2 * (s.c * s.c);
    const float sf = fsqrt(1.f - a);
    sag = r2 * s.c * frcp(1.f + sf);
    slope = 0.5f * s.c * frcp(sf);
    if (s.n_ai > 0) {
        // sum a_j r2^j and its derivative by Horner (the reference's power form, surfaces.py:799,823, differs
        // by rounding of terms that are < 1e-3 of the sag); derivative coefficients (j+1) a_j come from the table
        float ps, pd;
        if (s.n_ai <= 6) {
            ps = s.ai[5]; pd = s.dai[5];
#pragma unroll
            for (int j = 4; j >= 0; --j) { ps = ps * r2 + s.ai[j]; pd = pd * r2 + s.dai[j]; }
        } else {
            ps = s.ai[AADFF_MAX_AI - 1]; pd = s.dai[AADFF_MAX_AI - 1];
#pragma unroll
            for (int j = AADFF_MAX_AI - 2; j >= 0; --j) { ps = ps * r2 + s.ai[j]; pd = pd * r2 + s.dai[j]; }
        }
        sag += ps * r2;
        slope += pd;
    }
}

__device__ __forceinline__ float dsag_dr2(const aadff_surface_t& s, float r2) {
    float sag, slope;
    sag_and_slope(s, r2, sag, slope);
    return slope;
}

// ---- validity masks: deeplens/surfaces.py:724-743 ----
__device__ __forceinline__ bool valid_strict(const aadff_surface_t& s, float r2) {
    return s.k_gt_m1 ? (r2 < s.r2 && r2 < s.r2_shape) : (r2 < s.r2);
}
__device__ __forceinline__ bool valid_loose(const aadff_surface_t& s, float r2) {
    return s.k_gt_m1 ? (r2 < s.r2_shape) : (r2 > 0.f);
}

// ---- Newton intersection: deeplens/surfaces.py:523-586.  Called by alive lanes only. ----
template <bool STRICT>
__device__ __forceinline__ float newton_step(const aadff_surface_t& s, const Ray& r, float dxy2, float od, float& t) {
    const float px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
    float r2 = px * px + py * py;
    const bool m = STRICT ? valid_strict(s, r2) : valid_loose(s, r2);
    r2 = m ? r2 : 0.f;                                   // x*valid, y*valid
    float sag, slope;
    sag_and_slope(s, r2, sag, slope);
    const float ft = sag + s.d - pz;
    const float dr2dt = 2.f * (dxy2 * t + od);
    const float dfdt = slope * dr2dt - r.dz;
    float step = fdiv(ft, dfdt + kEps);
    step = fminf(fmaxf(step, -kStepBound), kStepBound);
    t -= step;
    return ft;
}

// Root of the ray with the CONIC part of the surface, measured from the vertex-plane point p0 = o + d t0:
//   c (1 + k dz^2) tau^2 + 2 beta tau + c rho^2 = 0,  beta = c (p0x dx + p0y dy) - dz,  rho^2 = p0x^2 + p0y^2
// (from c (1+k) z^2 - 2 z + c r^2 = 0 with z = dz tau); the root next to the vertex plane in cancellation-free
// form.  k = 0 is the sphere.  Returns false when the ray misses the conic.
__device__ __forceinline__ bool conic_root(const aadff_surface_t& s, const Ray& r, float t0, float& p0x, float& p0y, float& tau) {
    p0x = r.ox + r.dx * t0; p0y = r.oy + r.dy * t0;
    const float rho2 = p0x * p0x + p0y * p0y;
    const float beta = s.c * (p0x * r.dx + p0y * r.dy) - r.dz;
    const float A = s.c * (1.f + s.k * r.dz * r.dz);
    const float disc = beta * beta - A * (s.c * rho2);
    const float root = fsqrt(fmaxf(disc, 0.f));
    tau = fdiv(s.c * rho2, beta < 0.f ? (root - beta) : -(root + beta));
    return disc >= 0.f;
}

__device__ __forceinline__ void newton(const aadff_surface_t& s, const Ray& r, float& t_out, bool& valid_out, int& nan_flag) {
    const float dxy2 = r.dx * r.dx + r.dy * r.dy;
    const float od = r.dx * r.ox + r.dy * r.oy;
    const float t0 = fdiv(s.d - r.oz, r.dz);
    float t = t0;
#ifndef AADFF_NEWTON_PLANE_START
    // start from the conic root instead of the vertex plane (surfaces.py:543): Newton then only has to absorb
    // the polynomial departure (1-2 steps instead of 3-4) and reaches the same root within its tolerance
    {
        float p0x, p0y, tau;
        if (conic_root(s, r, t0, p0x, p0y, tau)) t = t0 + tau;
    }
#endif
    float ft = kMaxT;
    for (int it = 0; it < kNewtonMaxIter; ++it) {
        if (!__any(fabsf(ft) > kTolLoose)) break;
        ft = newton_step<false>(s, r, dxy2, od, t);
        if (ft != ft) nan_flag = 1;
    }
    const float t1 = t - t0;
    t = t0 + t1;                                         // surfaces.py:565-569 (not an fp32 identity)
    ft = newton_step<true>(s, r, dxy2, od, t);
    const float px = r.ox + r.dx * t, py = r.oy + r.dy * t;
    valid_out = valid_strict(s, px * px + py * py) && (fabsf(ft) < kTolTight) && (t > 0.f);
    t_out = t;
}

// ---- vector Snell refraction with the surface normal: deeplens/surfaces.py:589-679 ----
// KEEP = true: a ray that fails keeps its direction (the reference leaves dead rays untouched and the
// generic trace API exposes them); KEEP = false (fused PSF / refocus kernels): dead rays are never read
// again, so nothing is preserved.
// Sphere normal: the reference normalises +-2 (x, y, z - (d + R)); since |p - centre| = |R| on the sphere
// that unit vector is exactly (c x, c y, c (z - d) - 1) for either sign of c -> no normalisation needed.
// Refracted direction sr n + eta (d - cosi n) is evaluated as eta d + (sr - eta cosi) n.
template <bool KEEP>
__device__ __forceinline__ void refract(const aadff_surface_t& s, Ray& r, bool forward) {
    float nx, ny, nz;
    if (s.kind == AADFF_SURF_STOP) {
        nx = 0.f; ny = 0.f; nz = -1.f;
    } else if (s.kind == AADFF_SURF_SPHERIC) {
        nx = s.c * r.ox; ny = s.c * r.oy; nz = s.c * (r.oz - s.d) - 1.f;
    } else {
        const float g = dsag_dr2(s, r.ox * r.ox + r.oy * r.oy);
        nx = g * 2.f * r.ox; ny = g * 2.f * r.oy; nz = -1.f;
        const float inv = frsq(fmaxf(nx * nx + ny * ny + 1.f, 1e-24f));           // F.normalize
        nx *= inv; ny *= inv; nz = -inv;
    }
    // forward rays use -n (surfaces.py:654-655): fold the sign into cosi and the final combination
    const float sgn = forward ? -1.f : 1.f;
    const float eta = forward ? s.eta_fwd : s.eta_bwd;
    const float eta2 = forward ? s.eta_fwd2 : s.eta_bwd2;
    const float cosi = sgn * (r.dx * nx + r.dy * ny + r.dz * nz);
    const float sin2 = eta2 * (1.f - cosi * cosi);
    const bool valid = (cosi * cosi > 0.1f) && (sin2 < 1.f);
    const float g = sgn * (fsqrt(fmaxf(1.f - sin2, 0.f)) - eta * cosi);
    if (KEEP) {
        if (valid) {
            r.dx = eta * r.dx + g * nx; r.dy = eta * r.dy + g * ny; r.dz = eta * r.dz + g * nz;
        } else {
            r.ra = 0.f;
        }
    } else {
        r.dx = eta * r.dx + g * nx; r.dy = eta * r.dy + g * ny; r.dz = eta * r.dz + g * nz;
        r.ra = valid ? r.ra : 0.f;
    }
}

// ---- one surface interaction: deeplens/surfaces.py:391-520 (dead rays are skipped) ----
template <bool KEEP>
__device__ __forceinline__ void react(const aadff_surface_t& s, Ray& r, bool forward, int& nan_flag) {
    if (!(r.ra > 0.f)) return;
    float t, px, py, pz;
    bool valid;
    if (s.kind == AADFF_SURF_STOP) {
        t = fdiv(s.d - r.oz, r.dz);
        px = r.ox + t * r.dx; py = r.oy + t * r.dy; pz = r.oz + t * r.dz;
        valid = fsqrt(px * px + py * py) <= s.r;
#ifndef AADFF_SPHERE_NEWTON
    } else if (s.kind == AADFF_SURF_SPHERIC) {
        // Sphere (k = 0, no polynomial): closed-form root instead of the reference's Newton iteration
        // (deeplens/surfaces.py:456-487 keeps Newton's t but DISCARDS its convergence mask, so only the root
        // matters).  From the vertex-plane point p0 = o + d*t0 the sphere |p - (0,0,d+R)| = R reads
        //   tau^2 + 2 b tau + rho^2 = 0,  rho^2 = p0x^2 + p0y^2,  b = p0x dx + p0y dy - R dz,
        // whose root next to the vertex plane is, in cancellation-free form and scaled by c = 1/R,
        //   tau = c rho^2 / (-beta + sgn(-beta) sqrt(beta^2 - c^2 rho^2)),   beta = c (p0x dx + p0y dy) - dz.
        const float t0 = fdiv(s.d - r.oz, r.dz);
        float p0x, p0y, tau;
        const bool hit = conic_root(s, r, t0, p0x, p0y, tau);
        t = t0 + tau;
        px = p0x + r.dx * tau; py = p0y + r.dy * tau; pz = s.d + r.dz * tau;
        valid = hit && (px * px + py * py <= s.r2) && (t >= 0.f);
#endif
    } else {
        bool nvalid;
        newton(s, r, t, nvalid, nan_flag);
        px = r.ox + t * r.dx; py = r.oy + t * r.dy; pz = r.oz + t * r.dz;
        valid = s.kind == AADFF_SURF_SPHERIC ? ((px * px + py * py <= s.r2) && (t >= 0.f))   // Newton's own mask is discarded (:466)
                                             : nvalid;
    }
    if (KEEP) {
        if (!valid) { r.ra = 0.f; return; }
        r.ox = px; r.oy = py; r.oz = pz;
    } else {
        r.ox = px; r.oy = py; r.oz = pz;
        if (!valid) { r.ra = 0.f; return; }
    }
    if (s.kind != AADFF_SURF_STOP || (forward ? s.refract_fwd : s.refract_bwd)) refract<KEEP>(s, r, forward);
}

template <bool KEEP>
__device__ __forceinline__ void trace_range(const aadff_surface_t* __restrict__ surf, int first, int last, bool forward,
                                            Ray& r, int& nan_flag) {
    if (forward)
        for (int i = first; i < last; ++i) react<KEEP>(surf[i], r, true, nan_flag);
    else
        for (int i = last - 1; i >= first; --i) react<KEEP>(surf[i], r, false, nan_flag);
}

__device__ __forceinline__ void propagate_to(Ray& r, float z) {      // deeplens/basics.py:255-273 (all rays)
    const float t = fdiv(z - r.oz, r.dz);
    r.ox += r.dx * t; r.oy += r.dy * t; r.oz += r.dz * t;
}

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
    const float inv = frsq(fmaxf(x * x + y * y + z * z, 1e-24f));
    x *= inv; y *= inv; z *= inv;
}

// pupil / disc sample from two raw uniforms: deeplens/optics.py:480-485, surfaces.py:192-195
__device__ __forceinline__ void disc_sample(float u_theta, float u_r, float R2, float& x, float& y) {
    const float rr = fsqrt(u_r * R2);
#ifndef AADFF_LIBM_SINCOS
    // v_sin_f32 / v_cos_f32 take their argument in revolutions: cos(2*pi*u) in one instruction each.  The
    // reference evaluates cos(fp32(u*2*pi)); both forms differ from the exact angle by ~4e-7 rad, i.e. < 1e-5 mm
    // at the pupil rim (measured: rendered-image parity unchanged); -DAADFF_LIBM_SINCOS restores cosf/sinf.
    x = rr * __builtin_amdgcn_cosf(u_theta);
    y = rr * __builtin_amdgcn_sinf(u_theta);
#else
    const float theta = u_theta * 2.f * kTwoPiHi;
    x = rr * cosf(theta);
    y = rr * sinf(theta);
#endif
}

__device__ __forceinline__ Ray ray_to(float px, float py, float pz, float tx, float ty, float tz) {
    Ray r;
    r.ox = px; r.oy = py; r.oz = pz;
    r.dx = tx - px; r.dy = ty - py; r.dz = tz - pz;
    normalize3(r.dx, r.dy, r.dz);
    r.ra = 1.f;
    return r;
}

// ------------------------------------------------------------------------------------
// Two rays per lane (fused PSF kernel only).  Every quantity is a float2 whose components belong to two
// independent rays, so nearly all arithmetic issues as packed fp32 (v_pk_fma/mul/add_f32: two rays per
// 4-cycle instruction instead of one ray per ~3-cycle instruction); only the transcendentals, compares
// and selects stay per component.  Same formulas as the scalar core above, KEEP = false semantics (a dead
// ray's state is never read again), masks are int2 (-1 / 0).
// ------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

struct Ray2 {
    f2 ox, oy, oz, dx, dy, dz, ra;
    i2 alive;          // validity while tracing (a lane mask: no per-surface compare/select); ra = alive ? 1 : 0 afterwards
};
__device__ __forceinline__ f2 f2s(float a) { return (f2){a, a}; }
__device__ __forceinline__ f2 vsqrt(f2 a) { return (f2){fsqrt(a.x), fsqrt(a.y)}; }
__device__ __forceinline__ f2 vrcp(f2 a) { return (f2){frcp(a.x), frcp(a.y)}; }
__device__ __forceinline__ f2 vrsq(f2 a) { return (f2){frsq(a.x), frsq(a.y)}; }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ f2 vmin(f2 a, f2 b) { return __builtin_elementwise_min(a, b); }
// max(x, 0) as ONE v_max_f32 per component (the generic form first canonicalises x: two instructions)
__device__ __forceinline__ f2 vmax0(f2 a) {
    float x, y;
    asm("v_max_f32 %0, 0, %1" : "=v"(x) : "v"(a.x));
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(a.y));
    return (f2){x, y};
}
__device__ __forceinline__ f2 vabs(f2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ f2 vsel(i2 m, f2 a, f2 b) { return (f2){m.x ? a.x : b.x, m.y ? a.y : b.y}; }
__device__ __forceinline__ bool any2(i2 m) { return (m.x | m.y) != 0; }

// sum a_j r2^j (to be multiplied by r2) and its r2-derivative, by Horner
__device__ __forceinline__ void poly2(const aadff_surface_t& s, f2 r2, f2& ps, f2& pd) {
    if (s.n_ai <= 6) {
        ps = f2s(s.ai[5]); pd = f2s(s.dai[5]);
#pragma unroll
        for (int j = 4; j >= 0; --j) { ps = ps * r2 + s.ai[j]; pd = pd * r2 + s.dai[j]; }
    } else {
        ps = f2s(s.ai[AADFF_MAX_AI - 1]); pd = f2s(s.dai[AADFF_MAX_AI - 1]);
#pragma unroll
        for (int j = AADFF_MAX_AI - 2; j >= 0; --j) { ps = ps * r2 + s.ai[j]; pd = pd * r2 + s.dai[j]; }
    }
}
__device__ __forceinline__ void sag_and_slope2(const aadff_surface_t& s, f2 r2, f2& sag, f2& slope) {
    const f2 a = (1.f + s.k) * r2 * (s.c * s.c);
#ifndef AADFF_NO_RSQ_SLOPE
    // 1/sf from one v_rsq and sf = (1 - a) / sf: one transcendental less per ray than sqrt + rcp (callers mask r2 so
    // that 1 - a >= 1e-9, valid_loose2 / valid_strict2)
    const f2 om = 1.f - a;
    const f2 isf = vrsq(om);
    const f2 sf = om * isf;
    sag = r2 * s.c * vrcp(1.f + sf);
    slope = (0.5f * s.c) * isf;
#else
    const f2 sf = vsqrt(1.f - a);
    sag = r2 * s.c * vrcp(1.f + sf);
    slope = (0.5f * s.c) * vrcp(sf);
#endif
    if (s.n_ai > 0) {
        f2 ps, pd;
        poly2(s, r2, ps, pd);
        sag += ps * r2;
        slope += pd;
    }
}
__device__ __forceinline__ i2 valid_strict2(const aadff_surface_t& s, f2 r2) {
    const i2 a = r2 < s.r2;
    return s.k_gt_m1 ? (a & (r2 < s.r2_shape)) : a;
}
__device__ __forceinline__ i2 valid_loose2(const aadff_surface_t& s, f2 r2) {
    return s.k_gt_m1 ? (r2 < s.r2_shape) : (r2 > 0.f);
}
// Paired core, vertex-plane form: every surface first moves the ray IN PLACE to the vertex plane z = d (origin p0,
// parameter tau from there: no cancellation for far objects), the kind-specific code only produces tau, a validity
// mask and the normal, and the common tail advances and refracts the ray in place -- the loop-carried ray state is
// never defined inside a branch, which saves the ~12 register copies per surface the branchy form cost.
template <bool STRICT>
__device__ __forceinline__ f2 newton_step2(const aadff_surface_t& s, const Ray2& r, f2 dxy2, f2 od, f2& tau, f2* slope_out = nullptr,
                                           f2* step_out = nullptr) {
    const f2 px = r.ox + r.dx * tau, py = r.oy + r.dy * tau;
    f2 r2 = px * px + py * py;
    const i2 m = STRICT ? valid_strict2(s, r2) : valid_loose2(s, r2);
    r2 = vsel(m, r2, f2s(0.f));
    f2 sag, slope;
    sag_and_slope2(s, r2, sag, slope);
    const f2 ft = sag - r.dz * tau;                      // sag + d - p_z with p_z = d + dz tau
    const f2 dr2dt = 2.f * (dxy2 * tau + od);
    const f2 dfdt = slope * dr2dt - r.dz;
    f2 step = ft * vrcp(dfdt + kEps);
    step = vmin(vmax(step, f2s(-kStepBound)), f2s(kStepBound));
    tau -= step;
    if (slope_out) *slope_out = slope;
    if (step_out) *step_out = step;
    return ft;
}
// conic root from the vertex-plane point (r.ox, r.oy, d)
template <bool SPHERE = false>                                  // SPHERE: kind SPHERIC implies k == 0 (Aspheric.kind())
__device__ __forceinline__ i2 conic_root2(const aadff_surface_t& s, const Ray2& r, f2& tau) {
    const f2 rho2 = r.ox * r.ox + r.oy * r.oy;
    const f2 beta = s.c * (r.ox * r.dx + r.oy * r.dy) - r.dz;
    const f2 A = SPHERE ? f2s(s.c) : s.c * (1.f + s.k * r.dz * r.dz);
    const f2 disc = beta * beta - A * (s.c * rho2);
    const f2 root = vsqrt(vmax0(disc));
#ifndef AADFF_NO_SIGNXFER
    // -(beta + sign(beta) root): the sign transfer is one v_bfi per ray where the two-sided form costs a compare and a
    // select each (beta = -0 exactly would pick the other root; beta is c (p0 . d) - dz with dz ~ 1)
    const f2 sroot = (f2){__builtin_copysignf(root.x, beta.x), __builtin_copysignf(root.y, beta.y)};
    tau = (s.c * rho2) * vrcp(-(beta + sroot));
#else
    tau = (s.c * rho2) * vrcp(vsel(beta < 0.f, root - beta, -(root + beta)));
#endif
    return disc >= 0.f;
}
// slope_out: d sag / d r^2 of the last (strict) step, i.e. one converged Newton update (<= 5e-5 mm, typically 1e-7)
// before the hit point; the fused kernels reuse it for the surface normal instead of evaluating the asphere again
// (relative change of the slope over that update <= 2e-6: below the fp32 noise of the trace; -DAADFF_NORMAL_REEVAL
// restores the literal evaluation at the hit point, surfaces.py:589-630).  r is at the vertex plane; t0 = distance
// travelled to get there (the reference's t > 0 test is on t0 + tau).
__device__ __forceinline__ void newton2(const aadff_surface_t& s, const Ray2& r, i2 alive, f2 t0, f2& tau_out, i2& valid_out, int& nan_flag,
                                        f2* slope_out = nullptr) {
    const f2 dxy2 = r.dx * r.dx + r.dy * r.dy;
    const f2 od = r.dx * r.ox + r.dy * r.oy;
    f2 tau = f2s(0.f);
#ifndef AADFF_NEWTON_PLANE_START
    i2 hit0;
    {
        f2 tc;
        hit0 = conic_root2(s, r, tc);
        tau = vsel(hit0, tc, f2s(0.f));
    }
#endif
    // The reference leaves its loop when the residual BEFORE the last update is below 5e-5 mm, i.e. it spends one
    // whole evaluation confirming a converged point, then takes its extra (strict) step.  A Newton update of length
    // `step` leaves a residual of f''/(2 f') x step^2 (f'' <= the largest curvature kappa of the surface profile, f' ~ -dz);
    // the host sets newton_step_tol = min(1e-2, sqrt(4e-6 / kappa)) mm per surface (deeplens/surfaces.py: pack), so that
    // this residual stays <= 3e-6 mm, 3x under the strict step's |residual| < 1e-5 test, for ANY lens file (kappa = 0.03/mm
    // gives the 10 um that the shipped 50 mm lenses run with); the strict step — a full Newton update like any other —
    // then takes the point to ~1e-12 mm: the confirming evaluation is skipped without moving the hit (with 1 um the rim
    // rays of nearly every wave forced one more whole evaluation).
    [[maybe_unused]] const float step_tol = s.newton_step_tol;
    f2 ft = f2s(kMaxT), step = f2s(kMaxT);
    i2 nanm = {0, 0};                                    // NaN seen in ANY residual (the reference exits on the first, surfaces.py:555)
#if !defined(AADFF_NEWTON_LITERAL_EXIT) && !defined(AADFF_NEWTON_CHECK_EVERY_STEP)
    // The first evaluation always runs (the reference enters its loop with ft = MAXT) and the second does whenever any
    // ray of the batch started more than 5e-5 mm off the surface, which from the conic root of an asphere is every
    // batch: run both without the wave-wide exit test, then continue under it.  (Steps on converged rays are no-ops
    // at the 1e-9 level, which is also what the reference's batch-wide loop does to them.)  A NaN residual does NOT
    // survive the updates (the step clamp is v_max/v_min: maxnum/minnum drop a NaN operand, where torch.clamp propagates
    // it), so every residual is tested as it is produced - one compare and a scalar OR each - and the flag is raised once.
#if !defined(AADFF_NEWTON_PLANE_START) && !defined(AADFF_NEWTON_GENERIC_FIRST)
    if (s.n_ai > 0 && !__any(any2(alive & ~hit0))) {
        // Every live ray of the wave starts ON the conic (tau = its conic root), where the residual is the polynomial
        // part alone, sag_conic = dz tau, and sf = sqrt(1 - (1+k) c^2 r^2) = 1 - (1+k) c dz tau: the first Newton step
        // needs no square root and no conic sag (30 instead of 74 instructions per lane); same update as the generic
        // evaluation up to rounding.
        const f2 px = r.ox + r.dx * tau, py = r.oy + r.dy * tau;
        const f2 r2 = px * px + py * py;
        f2 ps, pd;
        poly2(s, r2, ps, pd);
        ft = ps * r2;
        nanm |= ft != ft;
        const f2 sf = 1.f - ((1.f + s.k) * s.c) * (r.dz * tau);
        const f2 slope = (0.5f * s.c) * vrcp(sf) + pd;
        const f2 dfdt = slope * (2.f * (dxy2 * tau + od)) - r.dz;
        step = ft * vrcp(dfdt + kEps);
        step = vmin(vmax(step, f2s(-kStepBound)), f2s(kStepBound));
        tau -= step;
    } else
#endif
    {
        ft = newton_step2<false>(s, r, dxy2, od, tau, nullptr, &step);
        nanm |= ft != ft;
    }
    ft = newton_step2<false>(s, r, dxy2, od, tau, nullptr, &step);
    nanm |= ft != ft;
    for (int it = 2; it < kNewtonMaxIter; ++it) {
        if (!__any(any2(alive & (vabs(ft) > kTolLoose) & (vabs(step) > step_tol)))) break;
        ft = newton_step2<false>(s, r, dxy2, od, tau, nullptr, &step);
        nanm |= ft != ft;
    }
#else
    for (int it = 0; it < kNewtonMaxIter; ++it) {
#ifndef AADFF_NEWTON_LITERAL_EXIT
        if (!__any(any2(alive & (vabs(ft) > kTolLoose) & (vabs(step) > step_tol)))) break;
#else
        if (!__any(any2(alive & (vabs(ft) > kTolLoose)))) break;
#endif
        ft = newton_step2<false>(s, r, dxy2, od, tau, nullptr, &step);
        nanm |= ft != ft;
    }
#endif
    ft = newton_step2<true>(s, r, dxy2, od, tau, slope_out);
    nanm |= ft != ft;
    if (any2(alive & nanm)) nan_flag = 1;
    const f2 px = r.ox + r.dx * tau, py = r.oy + r.dy * tau;
    valid_out = valid_strict2(s, px * px + py * py) & (vabs(ft) < kTolTight) & (t0 + tau > 0.f);
    tau_out = tau;
}
// vector Snell refraction at a surface with unit normal (nx, ny, nz) (surfaces.py:633-679)
__device__ __forceinline__ i2 refract_dir2(const aadff_surface_t& s, Ray2& r, bool forward, f2 nx, f2 ny, f2 nz) {
    const float sgn = forward ? -1.f : 1.f;
    const float eta = forward ? s.eta_fwd : s.eta_bwd;
    const float eta2 = forward ? s.eta_fwd2 : s.eta_bwd2;
    const f2 cosi = sgn * (r.dx * nx + r.dy * ny + r.dz * nz);
    const f2 cos2 = cosi * cosi;
    // cos^2 i > 0.1 and eta^2 (1 - cos^2 i) < 1  <=>  cos^2 i > max(0.1, 1 - 1/eta^2)  (host-computed per surface)
    const i2 valid = cos2 > (forward ? s.cos2_min_fwd : s.cos2_min_bwd);
#ifndef AADFF_NO_FOLDED_SNELL
    const f2 k2 = eta2 * cos2 + (1.f - eta2);           // 1 - eta^2 (1 - cos^2 i) as one fma with wave-uniform constants
#else
    const f2 k2 = 1.f - eta2 * (1.f - cos2);
#endif
    const f2 g = sgn * (vsqrt(vmax0(k2)) - eta * cosi);
    r.dx = eta * r.dx + g * nx; r.dy = eta * r.dy + g * ny; r.dz = eta * r.dz + g * nz;
    return valid;
}
__device__ __forceinline__ void react2(const aadff_surface_t& s, Ray2& r, bool forward, int& nan_flag) {
    const i2 alive = r.alive;
#ifdef AADFF_SURFACE_SKIP
    // wave-uniform skip only: a per-lane early return turns the whole surface body into a divergent region whose
    // results are merged back with ~12 v_mov per surface (dead lanes just compute values nobody reads)
    if (!__any(any2(alive))) return;
#endif
    // (no skip at all by default: vignetted rays are scattered over the lanes, a wave with no live ray is a rarity
    // before the compaction and impossible after it, and the test costs two vector instructions per surface)
    // to the vertex plane, in place
    const f2 t0 = (s.d - r.oz) * vrcp(r.dz);
    r.ox += r.dx * t0; r.oy += r.dy * t0;
    f2 tau, nx, ny, nz;
    i2 valid;
    if (s.kind == AADFF_SURF_SPHERIC) {
        const i2 hit = conic_root2<true>(s, r, tau);
        valid = hit & (t0 + tau >= 0.f);
        r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + r.dz * tau;
        valid &= (r.ox * r.ox + r.oy * r.oy) <= s.r2;
        nx = s.c * r.ox; ny = s.c * r.oy; nz = (s.c * r.dz) * tau - 1.f;      // c (z - d) - 1 with z - d = dz tau
    } else if (s.kind == AADFF_SURF_STOP) {
        r.oz = f2s(s.d);
        valid = (r.ox * r.ox + r.oy * r.oy) <= s.r * s.r;           // sqrt(x^2+y^2) <= r (surfaces.py:418)
        nx = f2s(0.f); ny = f2s(0.f); nz = f2s(-1.f);
    } else {
        f2 g;
#ifndef AADFF_NORMAL_REEVAL
        newton2(s, r, alive, t0, tau, valid, nan_flag, &g);
        r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + r.dz * tau;
#else
        newton2(s, r, alive, t0, tau, valid, nan_flag);
        r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + r.dz * tau;
        f2 sag;
        sag_and_slope2(s, r.ox * r.ox + r.oy * r.oy, sag, g);
#endif
        nx = g * 2.f * r.ox; ny = g * 2.f * r.oy;
        const f2 inv = vrsq(vmax(nx * nx + ny * ny + 1.f, f2s(1e-24f)));
        nx *= inv; ny *= inv; nz = -inv;
    }
    valid &= alive;
    if (s.kind != AADFF_SURF_STOP || (forward ? s.refract_fwd : s.refract_bwd)) valid &= refract_dir2(s, r, forward, nx, ny, nz);
    r.alive = valid;
}
// ---- run-structured forward trace (default in the fused kernels) ---------------------------------------------------
// Validity is carried as a MARGIN vm (alive <=> vm >= 0) that every test lowers with v_min3_f32 instead of a compare,
// a mask AND and a select each: r^2 <= R^2 becomes R^2 - r^2, disc >= 0 the discriminant itself, cos^2 i > thr becomes
// cos^2 i - nextup(thr).  (v_min3 drops NaN operands; a NaN can only come from a ray that an earlier margin already
// marked dead, and the splat window test rejects NaN positions.)
__device__ __forceinline__ f2 vmin3(f2 a, f2 b, f2 c) {
    return (f2){__builtin_fminf(__builtin_fminf(a.x, b.x), c.x), __builtin_fminf(__builtin_fminf(a.y, b.y), c.y)};
}
// One spherical surface, forward, unit directions.  Consecutive spheres run in a loop of their own (trace_part2) so the
// ray lives in the same registers from one sphere to the next; with the three surface kinds behind one switch the
// compiler merged the branches with 4-7 register copies per surface.
// The refraction needs no normal vector: with n = (c x, c y, c dz tau - 1) on the sphere, d.n = beta + c tau
// (beta = c (p0.d) - dz from the root), and  d' = eta d + g n  with  g = -(sqrt(1 - eta^2 (1 - (d.n)^2)) + eta d.n)
// is  (eta dx + (g c) x,  eta dy + (g c) y,  eta dz + (g c)(dz tau) - g)   (surfaces.py:589-679 restated).
typedef const __attribute__((address_space(4))) aadff_surface_t* csurf_t;
struct SphereP {                          // the scalars one sphere step needs
    float d, c, r2, eta, eta2, thr;
};
// (Measured and rejected: issuing the next entry's scalar loads by hand ahead of the arithmetic, with the ray registers
// tied to the asm statements to pin the order: 376 us against 332 us for the compiler's own s_load / s_waitcnt placement.)
__device__ __forceinline__ void sphere_step2(const SphereP& s, Ray2& r, f2& vm) {
    const f2 t0 = (s.d - r.oz) * vrcp(r.dz);
    r.ox += r.dx * t0; r.oy += r.dy * t0;                               // vertex plane
    const f2 rho2 = r.ox * r.ox + r.oy * r.oy;
    const f2 beta = s.c * (r.ox * r.dx + r.oy * r.dy) - r.dz;
    const f2 crho = s.c * rho2;
    const f2 disc = beta * beta - s.c * crho;
    const f2 root = vsqrt(vmax0(disc));
    const f2 sroot = (f2){__builtin_copysignf(root.x, beta.x), __builtin_copysignf(root.y, beta.y)};
    const f2 tau = crho * vrcp(-(beta + sroot));
    const f2 dzt = r.dz * tau;
    r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + dzt;
    vm = vmin3(vm, disc, t0 + tau);                                     // hit the sphere, t >= 0 (surfaces.py:466-470)
    const f2 dn = beta + s.c * tau;
    const f2 cos2 = dn * dn;
    vm = vmin3(vm, s.r2 - (r.ox * r.ox + r.oy * r.oy), cos2 - s.thr);   // inside the aperture; refraction valid (surfaces.py:660-666)
    const f2 sq = vsqrt(vmax0(s.eta2 * cos2 + (1.f - s.eta2)));
    const f2 g = -(sq + s.eta * dn);
    const f2 gc = g * s.c;
    r.dx = s.eta * r.dx + gc * r.ox;
    r.dy = s.eta * r.dy + gc * r.oy;
    r.dz = s.eta * r.dz + (gc * dzt - g);
}
#ifdef AADFF_SPHERE_FROM_POINT
// The same sphere met from where the ray IS (a point of the previous surface, a few mm away) instead of from the vertex
// plane: with p = o - (0,0,d) the quadratic  c t^2 + 2 t (c p.d - dz) + (c |p|^2 - 2 pz) = 0  has the same near root
// t = C / -(B + sign(B) sqrt(B^2 - c C)); no division by dz to reach the plane first - one transcendental and two packed
// instructions fewer per ray pair.  Not for a ray that starts metres away (the object point): c |p|^2 and B^2 then cancel
// catastrophically in fp32, so the first surface of a trace keeps the vertex-plane form.
__device__ __forceinline__ void sphere_step2_from_point(const SphereP& s, Ray2& r, f2& vm) {
    const f2 pz = r.oz - s.d;
    const f2 p2 = r.ox * r.ox + r.oy * r.oy + pz * pz;
    const f2 beta = s.c * (r.ox * r.dx + r.oy * r.dy + pz * r.dz) - r.dz;
    const f2 cc = s.c * p2 - 2.f * pz;
    const f2 disc = beta * beta - s.c * cc;
    const f2 root = vsqrt(vmax0(disc));
    const f2 sroot = (f2){__builtin_copysignf(root.x, beta.x), __builtin_copysignf(root.y, beta.y)};
    const f2 tau = cc * vrcp(-(beta + sroot));
    const f2 zt = pz + r.dz * tau;                                      // height above the vertex at the hit
    r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + zt;
    vm = vmin3(vm, disc, tau);
    const f2 dn = beta + s.c * tau;
    const f2 cos2 = dn * dn;
    vm = vmin3(vm, s.r2 - (r.ox * r.ox + r.oy * r.oy), cos2 - s.thr);
    const f2 sq = vsqrt(vmax0(s.eta2 * cos2 + (1.f - s.eta2)));
    const f2 g = -(sq + s.eta * dn);
    const f2 gc = g * s.c;
    r.dx = s.eta * r.dx + gc * r.ox;
    r.dy = s.eta * r.dy + gc * r.oy;
    r.dz = s.eta * r.dz + (gc * zt - g);
}
#endif
// stop / asphere, forward (the general code of react2 with the margin form of validity)
__device__ __forceinline__ void other_step2(const aadff_surface_t& s, Ray2& r, f2& vm, int& nan_flag) {
    const f2 t0 = (s.d - r.oz) * vrcp(r.dz);
    r.ox += r.dx * t0; r.oy += r.dy * t0;
    if (s.kind == AADFF_SURF_STOP) {
        r.oz = f2s(s.d);
        vm = vmin3(vm, s.r * s.r - (r.ox * r.ox + r.oy * r.oy), vm);    // sqrt(x^2+y^2) <= r (surfaces.py:418)
        if (s.refract_fwd) {
            const i2 v = refract_dir2(s, r, true, f2s(0.f), f2s(0.f), f2s(-1.f));
            vm = vsel(v, vm, f2s(-1.f));
        }
        return;
    }
    f2 tau, g;
    i2 valid;
    const i2 alive = vm >= 0.f;
    if (s.kind == AADFF_SURF_SPHERIC) {                                 // only with -DAADFF_SPHERE_NEWTON-style builds; kept general
        const i2 hit = conic_root2<true>(s, r, tau);
        valid = hit & (t0 + tau >= 0.f);
        r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + r.dz * tau;
        valid &= (r.ox * r.ox + r.oy * r.oy) <= s.r2;
        const i2 v = refract_dir2(s, r, true, s.c * r.ox, s.c * r.oy, (s.c * r.dz) * tau - 1.f);
        vm = vsel(valid & v, vm, f2s(-1.f));
        return;
    }
#ifndef AADFF_NORMAL_REEVAL
    newton2(s, r, alive, t0, tau, valid, nan_flag, &g);
    r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + r.dz * tau;
#else
    newton2(s, r, alive, t0, tau, valid, nan_flag);
    r.ox += r.dx * tau; r.oy += r.dy * tau; r.oz = s.d + r.dz * tau;
    f2 sag;
    sag_and_slope2(s, r.ox * r.ox + r.oy * r.oy, sag, g);
#endif
    f2 nx = g * 2.f * r.ox, ny = g * 2.f * r.oy;
    const f2 inv = vrsq(vmax(nx * nx + ny * ny + 1.f, f2s(1e-24f)));
    nx *= inv; ny *= inv;
    valid &= refract_dir2(s, r, true, nx, ny, -inv);
    vm = vsel(valid, vm, f2s(-1.f));
}
// surfaces [first, last) forward; r.alive in/out (r.ra not touched)
__device__ __forceinline__ void trace_part2(const aadff_surface_t* __restrict__ surf, int first, int last, Ray2& r, int& nan_flag) {
#ifdef AADFF_PSF_SWITCH_LOOP
    for (int i = first; i < last; ++i) react2(surf[i], r, true, nan_flag);
#else
    // the table is read through the constant address space: wave-uniform scalar loads (the kernel's own global stores
    // otherwise make the compiler fall back to per-lane vector loads inside the sphere loop)
    const csurf_t cs = (csurf_t)surf;
    f2 vm = vsel(r.alive, f2s(3e38f), f2s(-1.f));
    int i = first;
    while (i < last) {
        if (cs[i].kind == AADFF_SURF_SPHERIC) {
            do {
                SphereP cur;
                cur.d = cs[i].d; cur.c = cs[i].c; cur.r2 = cs[i].r2; cur.eta = cs[i].eta_fwd; cur.eta2 = cs[i].eta_fwd2;
                cur.thr = __builtin_bit_cast(float, __builtin_bit_cast(int, cs[i].cos2_min_fwd) + 1);
#ifdef AADFF_SPHERE_FROM_POINT
                if (i == 0) sphere_step2(cur, r, vm);
                else sphere_step2_from_point(cur, r, vm);
#else
                sphere_step2(cur, r, vm);
#endif
                ++i;
            } while (i < last && cs[i].kind == AADFF_SURF_SPHERIC);
        } else {
            other_step2(surf[i], r, vm, nan_flag);
            ++i;
        }
    }
    r.alive = vm >= 0.f;
#endif
}
// in: r.alive; out: r.alive and r.ra = alive ? 1 : 0
__device__ __forceinline__ void trace_forward2(const aadff_surface_t* __restrict__ surf, int n_surf, Ray2& r, int& nan_flag) {
    trace_part2(surf, 0, n_surf, r, nan_flag);
    r.ra = vsel(r.alive, f2s(1.f), f2s(0.f));
}
__device__ __forceinline__ void disc_sample2(f2 u_theta, f2 u_r, float R2, f2& x, f2& y) {
    const f2 rr = vsqrt(u_r * R2);
#ifndef AADFF_LIBM_SINCOS
    x = rr * (f2){__builtin_amdgcn_cosf(u_theta.x), __builtin_amdgcn_cosf(u_theta.y)};
    y = rr * (f2){__builtin_amdgcn_sinf(u_theta.x), __builtin_amdgcn_sinf(u_theta.y)};
#else
    const f2 theta = u_theta * 2.f * kTwoPiHi;
    x = rr * (f2){cosf(theta.x), cosf(theta.y)};
    y = rr * (f2){sinf(theta.x), sinf(theta.y)};
#endif
}
// rays from one object point to two pupil samples, then through the lens to the sensor plane
__device__ __forceinline__ Ray2 trace_pair_to_sensor(float px, float py, float pz, f2 tx, f2 ty, float tz, i2 active,
                                                     const aadff_surface_t* __restrict__ surf, int n_surf, float d_sensor,
                                                     int& nan_flag) {
    Ray2 r;
    r.ox = f2s(px); r.oy = f2s(py); r.oz = f2s(pz);
    r.dx = tx - px; r.dy = ty - py; r.dz = f2s(tz - pz);
    const f2 inv = vrsq(vmax(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz, f2s(1e-24f)));
    r.dx *= inv; r.dy *= inv; r.dz *= inv;
    r.alive = active;
    trace_forward2(surf, n_surf, r, nan_flag);
    const f2 t = (d_sensor - r.oz) * vrcp(r.dz);
    r.ox += r.dx * t; r.oy += r.dy * t; r.oz += r.dz * t;
    return r;
}

// ------------------------------------------------------------------------------------
// generic kernels
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trace_rays_kernel(const float* o_in, const float* d_in, const float* ra_in,
                                                          float* o_out, float* d_out, float* ra_out, int n,
                                                          const aadff_surface_t* __restrict__ surf, int first, int last,
                                                          int forward, const aadff_lens_state_t* __restrict__ state,
                                                          int* flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.ox = o_in[3 * i]; r.oy = o_in[3 * i + 1]; r.oz = o_in[3 * i + 2];
    r.dx = d_in[3 * i]; r.dy = d_in[3 * i + 1]; r.dz = d_in[3 * i + 2];
    r.ra = ra_in ? ra_in[i] : 1.f;
    int nan_flag = 0;
    trace_range<true>(surf, first, last, forward != 0, r, nan_flag);
    if (state) propagate_to(r, state->d_sensor);
    o_out[3 * i] = r.ox; o_out[3 * i + 1] = r.oy; o_out[3 * i + 2] = r.oz;
    d_out[3 * i] = r.dx; d_out[3 * i + 1] = r.dy; d_out[3 * i + 2] = r.dz;
    ra_out[i] = r.ra;
    if (nan_flag && flags) atomicOr(flags, 1);
}

__global__ __launch_bounds__(256) void trace_points_kernel(const float* __restrict__ pts, int N,
                                                            const float* __restrict__ u_theta,
                                                            const float* __restrict__ u_r, int spp, float pupil_z,
                                                            float pupil_r2, const aadff_surface_t* __restrict__ surf,
                                                            int n_surf, const aadff_lens_state_t* __restrict__ state,
                                                            float* o_out, float* d_out, float* ra_out) {
    // x: sample (fastest across the wave so a wave shares one object point), y: point
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (i >= spp) return;
    float x2, y2;
    disc_sample(u_theta[i], u_r[i], pupil_r2, x2, y2);
    Ray r = ray_to(pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], x2, y2, pupil_z);
    int nan_flag = 0;
    trace_range<true>(surf, 0, n_surf, true, r, nan_flag);
    propagate_to(r, state->d_sensor);
    const size_t e = (size_t)i * N + n;
    o_out[3 * e] = r.ox; o_out[3 * e + 1] = r.oy; o_out[3 * e + 2] = r.oz;
    d_out[3 * e] = r.dx; d_out[3 * e + 1] = r.dy; d_out[3 * e + 2] = r.dz;
    ra_out[e] = r.ra;
}

// ------------------------------------------------------------------------------------
// PSF histogram: deeplens/monte_carlo.py:9-121
// ------------------------------------------------------------------------------------
struct SplatGeom {
    float lo, hi, lim, den_row, den_col;   // all float64 -> fp32 once
    int ks;
};

__host__ __device__ inline SplatGeom make_splat_geom(float pixel_size, int ks) {
    const double ps = (double)pixel_size;
    const double lo = (-ks / 2.0 + 0.5) * ps, hi = (ks / 2.0 - 0.5) * ps;
    SplatGeom g;
    g.lo = (float)lo; g.hi = (float)hi;
    g.lim = (float)(hi - 0.01 * ps);
    g.den_row = (float)(lo - hi);
    g.den_col = (float)(hi - lo);
    g.ks = ks;
    return g;
}

// hit (ox,oy) on the sensor -> 4 bilinear taps into hist[ks*ks] (LDS), weight ra
__device__ __forceinline__ void splat_hit(float* hist, const SplatGeom& g, float ox, float oy, float ra, float cx, float cy) {
    float X = -ox - cx, Y = -oy - cy;                 // image flip, then centre
    const bool in = (fabsf(X) < g.lim) && (fabsf(Y) < g.lim) && (ra > 0.f);
    if (!in) return;                                   // zero-weight taps at the centre bin are skipped
    const float rowf = fdiv(Y - g.hi, g.den_row) * (float)(g.ks - 1);
    const float colf = fdiv(X - g.lo, g.den_col) * (float)(g.ks - 1);
    const float fr = floorf(rowf), fc = floorf(colf);
    const float wb = rowf - fr, wr = colf - fc;
    const int r0 = (int)fr, c0 = (int)fc;
    const int r1 = (int)floorf(rowf + 1.f), c1 = (int)floorf(colf + 1.f);
    const int ks = g.ks;
    atomicAdd(&hist[r0 * ks + c0], (1.f - wb) * (1.f - wr) * ra);
    atomicAdd(&hist[r0 * ks + c1], (1.f - wb) * wr * ra);
    atomicAdd(&hist[r1 * ks + c0], wb * (1.f - wr) * ra);
    atomicAdd(&hist[(r0 + 1) * ks + (c0 + 1)], wb * wr * ra);
}

// "Edge-exact" PSF grid (Lensgroup(parity="edge"), aadff_psf_points_edge): the only discontinuity of the histogram is the window
// test of deeplens/monte_carlo.py:37 - a hit within the float32 noise of the window's edge lands inside or outside depending on
// the LAST BIT of the trace, and one flipped border ray changes a cropped PSF by 1 / (rays inside).  The fast kernel therefore does
// not decide such rays: a live ray whose hit lies within `delta` of the edge (and not further outside than delta in the other
// coordinate) is not splatted but appended as (point << 16 | sample) to the list of its (focus state, wavelength) job;
// aadff_strict_edge_retrace (csrc/strict_fused.hip) re-traces exactly those rays in the reference's own float32 operation
// order, decides them and adds their taps, and aadff_psf_normalise divides.  Every other ray is the fast kernel's.
struct EdgeArgs {
    float delta;              // half-width of the undecided band around the window edge [mm]
    unsigned* count;          // [S*L] rays appended per job (zeroed by the caller)
    unsigned* list;           // [S*L][cap]
    int cap;
    float* raw;               // [S*L][N][ks*ks] unnormalised histograms (every element written)
    float* slope;             // [S*L][N][2] or NULL: mean direction tangents (dx/dz, dy/dz) of the valid chief rays at the sensor - what
                              // moves the centre when the sensor plane moves (the re-trace corrects the centre for the caller's exact
                              // d_sensor when this launch ran on provisional lens states, aadff_strict_edge_retrace)
};

// returns true when the hit was left to the re-trace (appended or, beyond cap, counted: the host sees count > cap)
__device__ __forceinline__ bool edge_defer(const SplatGeom& g, const EdgeArgs& e, float ox, float oy, bool alive, float cx, float cy, int job, int n,
                                           int sample) {
    const float ax = fabsf(-ox - cx), ay = fabsf(-oy - cy);
    const float out = g.lim + e.delta;
    const bool near = (fabsf(ax - g.lim) < e.delta) || (fabsf(ay - g.lim) < e.delta);
    if (!(alive && near && ax < out && ay < out)) return false;
    const unsigned slot = atomicAdd(e.count + job, 1u);
    if (slot < (unsigned)e.cap) e.list[(size_t)job * e.cap + slot] = ((unsigned)n << 16) | (unsigned)sample;
    return true;
}

__global__ __launch_bounds__(256) void psf_splat_kernel(const float* __restrict__ o, const float* __restrict__ ra,
                                                         const float* __restrict__ centre, int spp, int N,
                                                         SplatGeom g, float* psf_raw, float* psf) {
    extern __shared__ float hist[];                      // ks * ks floats (dynamic: ks up to AADFF_MAX_KS = 51)
    __shared__ float red[4];
    const int n = blockIdx.x, kk = g.ks * g.ks;
    for (int e = threadIdx.x; e < kk; e += blockDim.x) hist[e] = 0.f;
    __syncthreads();
    const float cx = centre[2 * n], cy = centre[2 * n + 1];
    for (int i = threadIdx.x; i < spp; i += blockDim.x) {
        const size_t e = (size_t)i * N + n;
        splat_hit(hist, g, o[3 * e], o[3 * e + 1], ra[e], cx, cy);
    }
    __syncthreads();
    float part = 0.f;
    for (int e = threadIdx.x; e < kk; e += blockDim.x) part += hist[e];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const float total = red[0] + red[1] + red[2] + red[3];
    for (int e = threadIdx.x; e < kk; e += blockDim.x) {
        if (psf_raw) psf_raw[(size_t)n * kk + e] = hist[e];
        psf[(size_t)n * kk + e] = hist[e] / total;        // 0/0 -> NaN as in deeplens/optics.py:978
    }
}

// ------------------------------------------------------------------------------------
// Fused PSF kernel: workgroup = (point n, wavelength l, focus state s).
//   phase 1  chief-ray centre (shrunk pupil, chief table): deeplens/optics.py:888-913
//   phase 2  main rays -> LDS histogram                    deeplens/optics.py:933-976
//   phase 3  normalise, write (optionally in psf_map tiling, optics.py:1025)
// ------------------------------------------------------------------------------------
// Upload riding on a psf_points launch (see aadff_stage_t in include/aadff.h).
struct StageArgs {
    const float4* src;
    float4* dst;
    long slice_n4;
    int first_slice, copy_wgs;
    unsigned* counters;
    unsigned target;
};

#ifndef AADFF_PSF_THREADS
#define AADFF_PSF_THREADS 512
#endif
#ifndef AADFF_STAGE_SPIN_MAX
#define AADFF_STAGE_SPIN_MAX (1 << 22)      // x s_sleep(16): ~0.1 s; -DAADFF_STAGE_SPIN_MAX=0 forces the late path (tests)
#endif
constexpr int kPsfThreads = AADFF_PSF_THREADS, kPsfWaves = kPsfThreads / 64;
constexpr int kCompactMax = 2048;         // rays per compaction chunk of the main pass (48 KB of LDS)
// sum of the histogram and the normalised write (deeplens/optics.py:978, psf_map tiling :1025): the tail of the fused kernel and the
// whole of psf_normalise_kernel - one summation order, so a PSF without deferred rays is bit for bit the fused kernel's
__device__ __forceinline__ void normalise_and_write(const float* hist, float* red, const SplatGeom& g, int map_grid, float* psf, int s, int l, int n, int N,
                                                    int L) {
    const int tid = threadIdx.x, kk = g.ks * g.ks;
    float part = 0.f;
    for (int e = tid; e < kk; e += kPsfThreads) part += hist[e];
    part = wave_sum(part);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < kPsfWaves; ++w) total += red[w];
    const int ks = g.ks;
    for (int e = tid; e < kk; e += kPsfThreads) {
        const float v = hist[e] / total;
        if (map_grid > 0) {
            const int gi = n / map_grid, gj = n - gi * map_grid, u = e / ks, w = e - u * ks;
            const int G = map_grid * ks;
            psf[((size_t)(s * L + l) * G + gi * ks + u) * G + gj * ks + w] = v;
        } else {
            psf[(((size_t)s * N + n) * L + l) * kk + e] = v;
        }
    }
}

template <bool EDGE>
__device__ __forceinline__ void psf_points_body(const float* __restrict__ points, int N, int L,
                                                          const aadff_surface_t* __restrict__ surf_main,
                                                          const aadff_surface_t* __restrict__ surf_chief,
                                                          aadff_lens_const_t lc,
                                                          const aadff_lens_state_t* __restrict__ states,
                                                          const float* __restrict__ u_main, int spp, long main_ss,
                                                          long main_sl, const float* __restrict__ u_chief,
                                                          int spp_chief, long chief_ss, long chief_sl, SplatGeom g, int centre_mode, int map_grid, float* psf,
                                                          float* centre_out, int* flags, StageArgs stage, const EdgeArgs& edge) {
    extern __shared__ float hist[];                      // ks * ks floats (dynamic: ks up to AADFF_MAX_KS = 51)
    __shared__ float red[(EDGE ? 5 : 3) * kPsfWaves];
    __shared__ int stage_late;
#if !defined(AADFF_PSF_SCALAR) && !defined(AADFF_PSF_NO_COMPACT)
    __shared__ float cbuf[6][kCompactMax];               // survivors of the first surfaces: origin and direction
    __shared__ unsigned short cidx[EDGE ? kCompactMax : 1];   // EDGE: and which sample each of them is
    __shared__ int c_count;
#endif
    const int n = blockIdx.x, l = blockIdx.y;
    int s = blockIdx.z;
    const int tid = threadIdx.x, kk = g.ks * g.ks;
    if (stage.src) {
        // Staged launch (aadff_psf_points_staged): the z = 0 plane of the grid is dispatched first; its first
        // stage.copy_wgs workgroups stream the uniform blocks of focus states >= first_slice from pinned host
        // memory into HBM and publish a per-state counter; PSF workgroups of those states wait on it.
        if (s == 0) {
            const int w = blockIdx.y * gridDim.x + blockIdx.x;
            if (w >= stage.copy_wgs) return;
            const int S = (int)gridDim.z - 1;
            const long stride = (long)stage.copy_wgs * kPsfThreads;
            for (int sl = stage.first_slice; sl < S; ++sl) {
                const float4* src = stage.src + (long)sl * stage.slice_n4;
                float4* dst = stage.dst + (long)sl * stage.slice_n4;
                for (long i = (long)w * kPsfThreads + tid; i < stage.slice_n4; i += stride) dst[i] = src[i];
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(stage.counters + sl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        s -= 1;
        if (s >= stage.first_slice) {
            // HIP promises no dispatch order: if the copy workgroups have not delivered this state's block within the
            // bound (they normally lead by tens of microseconds), do not trace stale samples — read this state's draws
            // straight from the mapped pinned block over PCIe (slow, still correct) and raise flag bit 3 so the host
            // learns that the overlap was lost.
            if (tid == 0) {
                int spins = 0, late = 0;
                while ((int)(__hip_atomic_load(stage.counters + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - stage.target) < 0) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > AADFF_STAGE_SPIN_MAX) { if (flags) atomicOr(flags, 8); late = 1; break; }     // never hang the queue
                }
                stage_late = late;
            }
            __syncthreads();
            if (stage_late) {
                u_main = reinterpret_cast<const float*>(stage.src) + (u_main - reinterpret_cast<const float*>(stage.dst));
                if (u_chief) u_chief = reinterpret_cast<const float*>(stage.src) + (u_chief - reinterpret_cast<const float*>(stage.dst));
            }
        }
    }
    // (coherent loads, common.h: fresh - the per-call API and the strict / edge paths re-upload lens states to a fixed address)
    struct { float d_sensor, tan_hfov; } st;
    st.d_sensor = fresh_uniform(&states[s].d_sensor);
    st.tan_hfov = fresh_uniform(&states[s].tan_hfov);
    // a focus state whose refocus found no valid ray has d_sensor = 0/0: the reference stops there with "sensor position is
    // negative." (deeplens/optics.py:1176); in a pipelined stack the condition travels in the flags word (bit 2)
    if (tid == 0 && !(st.d_sensor > 0.f) && flags) atomicOr(flags, 4);
    for (int e = tid; e < kk; e += kPsfThreads) hist[e] = 0.f;

    // object-space point: deeplens/optics.py:953-959 with calc_scale_pinhole (:1286-1290)
    const float* pt = points + ((size_t)s * N + n) * 3;
    const float xn = pt[0], yn = pt[1], depth = pt[2];
    const float scale = (-depth) * st.tan_hfov / lc.r_last;
    const float px = xn * scale * lc.sensor_w / 2.f;
    const float py = yn * scale * lc.sensor_h / 2.f;
    int nan_flag = 0;

    float cx, cy;
    if (centre_mode == 1) {
        const float* ut = u_chief + (size_t)s * chief_ss + (size_t)l * chief_sl;
        const float* ur = ut + spp_chief;
        float sx = 0.f, sy = 0.f, sw = 0.f;
        [[maybe_unused]] float tx = 0.f, ty = 0.f;       // EDGE: sums of the direction tangents of the valid chief rays
#ifndef AADFF_PSF_SCALAR
        for (int i = tid; i < spp_chief; i += 2 * kPsfThreads) {
            const int i1 = i + kPsfThreads;
            const i2 act = {-1, i1 < spp_chief ? -1 : 0};
            const int j1 = act.y ? i1 : i;
            f2 x2, y2;
            // (EDGE: the stack's uniforms are a block the host re-uploads to one address: coherent loads, common.h: fresh)
            disc_sample2(EDGE ? (f2){fresh(ut + i), fresh(ut + j1)} : (f2){ut[i], ut[j1]}, EDGE ? (f2){fresh(ur + i), fresh(ur + j1)} : (f2){ur[i], ur[j1]},
                         lc.enp_r2_shrunk, x2, y2);
            const Ray2 r = trace_pair_to_sensor(px, py, depth, x2, y2, lc.enp_z, act, surf_chief, lc.n_surf, st.d_sensor, nan_flag);
            const f2 wx = r.ox * r.ra, wy = r.oy * r.ra;
            sx += wx.x + wx.y; sy += wy.x + wy.y; sw += r.ra.x + r.ra.y;
            if (EDGE) {
                const f2 iz = vrcp(r.dz);
                const f2 ax = vsel(r.alive, r.dx * iz, f2s(0.f)), ay = vsel(r.alive, r.dy * iz, f2s(0.f));
                tx += ax.x + ax.y; ty += ay.x + ay.y;
            }
        }
#else
        for (int i = tid; i < spp_chief; i += kPsfThreads) {
            float x2, y2;
            disc_sample(ut[i], ur[i], lc.enp_r2_shrunk, x2, y2);
            Ray r = ray_to(px, py, depth, x2, y2, lc.enp_z);
            trace_range<false>(surf_chief, 0, lc.n_surf, true, r, nan_flag);
            propagate_to(r, st.d_sensor);
            sx += r.ox * r.ra; sy += r.oy * r.ra; sw += r.ra;
        }
#endif
        sx = wave_sum(sx); sy = wave_sum(sy); sw = wave_sum(sw);
        if (EDGE) { tx = wave_sum(tx); ty = wave_sum(ty); }
        if ((tid & 63) == 0) {
            red[(tid >> 6) * 3] = sx; red[(tid >> 6) * 3 + 1] = sy; red[(tid >> 6) * 3 + 2] = sw;
            if (EDGE) { red[3 * kPsfWaves + (tid >> 6) * 2] = tx; red[3 * kPsfWaves + (tid >> 6) * 2 + 1] = ty; }
        }
        __syncthreads();
        sx = sy = sw = 0.f;
#pragma unroll
        for (int w = 0; w < kPsfWaves; ++w) { sx += red[3 * w]; sy += red[3 * w + 1]; sw += red[3 * w + 2]; }
        cx = -(sx / (sw + kEps));
        cy = -(sy / (sw + kEps));
        if (EDGE && edge.slope && tid == 0) {
            tx = ty = 0.f;
            for (int w = 0; w < kPsfWaves; ++w) { tx += red[3 * kPsfWaves + 2 * w]; ty += red[3 * kPsfWaves + 2 * w + 1]; }
            float* so = edge.slope + ((size_t)(s * L + l) * N + n) * 2;
            so[0] = tx / (sw + kEps); so[1] = ty / (sw + kEps);
        }
        if (sw == 0.f && tid == 0 && flags) atomicOr(flags, 2);      // "No sampled rays is valid." (optics.py:901)
    } else {
        cx = xn * (lc.sensor_w / 2.f);                               // optics.py:972-974
        cy = yn * (lc.sensor_h / 2.f);
        __syncthreads();
    }
    if (centre_out && tid == 0) {
        float* co = centre_out + ((size_t)(s * L + l) * N + n) * 2;
        co[0] = cx; co[1] = cy;
    }

    const aadff_surface_t* tab = surf_main + (size_t)l * lc.n_surf;
#if !defined(AADFF_PSF_SCALAR) && !defined(AADFF_PSF_NO_COMPACT)
    int split = lc.n_surf / 2;                            // compaction point: two surfaces behind the stop
    for (int i = 0; i < lc.n_surf; ++i)
        if (tab[i].kind == AADFF_SURF_STOP) { split = min(i + 2, lc.n_surf); break; }
#endif
    const float* ut = u_main + (size_t)s * main_ss + (size_t)l * main_sl;
    const float* ur = ut + spp;
#ifndef AADFF_PSF_SCALAR
#ifndef AADFF_PSF_NO_COMPACT
    // Main pass in two phases with the survivors compacted in between: a third of the rays of an off-axis point die
    // at the apertures right behind the stop, scattered over all lanes, so no wave ever finishes early; after the
    // compaction the remaining surfaces (both aspheres of rf50mm) run on full waves only.  Chunks of kCompactMax rays.
    for (int c0 = 0; c0 < spp; c0 += kCompactMax) {
        const int cn = min(spp - c0, kCompactMax);
        if (tid == 0) c_count = 0;
        __syncthreads();
        for (int i = tid; i < cn; i += 2 * kPsfThreads) {
            const int i1 = i + kPsfThreads;
            const i2 act = {-1, i1 < cn ? -1 : 0};
            const int j1 = act.y ? i1 : i;
            f2 x2, y2;
            disc_sample2(EDGE ? (f2){fresh(ut + c0 + i), fresh(ut + c0 + j1)} : (f2){ut[c0 + i], ut[c0 + j1]},
                         EDGE ? (f2){fresh(ur + c0 + i), fresh(ur + c0 + j1)} : (f2){ur[c0 + i], ur[c0 + j1]}, lc.enp_r2, x2, y2);
            Ray2 r;
            r.ox = f2s(px); r.oy = f2s(py); r.oz = f2s(depth);
            r.dx = x2 - px; r.dy = y2 - py; r.dz = f2s(lc.enp_z - depth);
            const f2 inv = vrsq(vmax(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz, f2s(1e-24f)));
            r.dx *= inv; r.dy *= inv; r.dz *= inv;
            r.alive = act;
            trace_part2(tab, 0, split, r, nan_flag);
            // append the survivors: one LDS atomic per wave, slots by ballot prefix
            const unsigned long long bx = __ballot(r.alive.x != 0), by = __ballot(r.alive.y != 0);
            const int nx = __popcll(bx), ny = __popcll(by);
            int base = 0;
            if ((tid & 63) == 0) base = atomicAdd(&c_count, nx + ny);
            base = __builtin_amdgcn_readfirstlane(base);
            const unsigned long long below = (1ull << (tid & 63)) - 1ull;
            if (r.alive.x) {
                const int k = base + __popcll(bx & below);
                cbuf[0][k] = r.ox.x; cbuf[1][k] = r.oy.x; cbuf[2][k] = r.oz.x; cbuf[3][k] = r.dx.x; cbuf[4][k] = r.dy.x; cbuf[5][k] = r.dz.x;
                if (EDGE) cidx[k] = (unsigned short)(c0 + i);
            }
            if (r.alive.y) {
                const int k = base + nx + __popcll(by & below);
                cbuf[0][k] = r.ox.y; cbuf[1][k] = r.oy.y; cbuf[2][k] = r.oz.y; cbuf[3][k] = r.dx.y; cbuf[4][k] = r.dy.y; cbuf[5][k] = r.dz.y;
                if (EDGE) cidx[k] = (unsigned short)(c0 + j1);
            }
        }
        __syncthreads();
        const int ns = c_count;
        for (int q = tid; 2 * q < ns; q += kPsfThreads) {
            const bool two = 2 * q + 1 < ns;
            const int k0 = 2 * q, k1 = two ? k0 + 1 : k0;
            Ray2 r;
            r.ox = (f2){cbuf[0][k0], cbuf[0][k1]}; r.oy = (f2){cbuf[1][k0], cbuf[1][k1]}; r.oz = (f2){cbuf[2][k0], cbuf[2][k1]};
            r.dx = (f2){cbuf[3][k0], cbuf[3][k1]}; r.dy = (f2){cbuf[4][k0], cbuf[4][k1]}; r.dz = (f2){cbuf[5][k0], cbuf[5][k1]};
            r.alive = (i2){-1, two ? -1 : 0};
            trace_part2(tab, split, lc.n_surf, r, nan_flag);
            const f2 t = (st.d_sensor - r.oz) * vrcp(r.dz);
            r.ox += r.dx * t; r.oy += r.dy * t;
            if (!(EDGE && edge_defer(g, edge, r.ox.x, r.oy.x, r.alive.x != 0, cx, cy, s * L + l, n, cidx[EDGE ? k0 : 0])))
                splat_hit(hist, g, r.ox.x, r.oy.x, r.alive.x ? 1.f : 0.f, cx, cy);
            if (!(EDGE && edge_defer(g, edge, r.ox.y, r.oy.y, r.alive.y != 0, cx, cy, s * L + l, n, cidx[EDGE ? k1 : 0])))
                splat_hit(hist, g, r.ox.y, r.oy.y, r.alive.y ? 1.f : 0.f, cx, cy);
        }
        __syncthreads();                                  // cbuf / c_count are reused by the next chunk
    }
#else
    for (int i = tid; i < spp; i += 2 * kPsfThreads) {
        const int i1 = i + kPsfThreads;
        const i2 act = {-1, i1 < spp ? -1 : 0};
        const int j1 = act.y ? i1 : i;
        f2 x2, y2;
        disc_sample2((f2){ut[i], ut[j1]}, (f2){ur[i], ur[j1]}, lc.enp_r2, x2, y2);
        const Ray2 r = trace_pair_to_sensor(px, py, depth, x2, y2, lc.enp_z, act, tab, lc.n_surf, st.d_sensor, nan_flag);
        if (!(EDGE && edge_defer(g, edge, r.ox.x, r.oy.x, r.ra.x > 0.f, cx, cy, s * L + l, n, i))) splat_hit(hist, g, r.ox.x, r.oy.x, r.ra.x, cx, cy);
        if (!(EDGE && edge_defer(g, edge, r.ox.y, r.oy.y, r.ra.y > 0.f, cx, cy, s * L + l, n, j1))) splat_hit(hist, g, r.ox.y, r.oy.y, r.ra.y, cx, cy);
    }
#endif
#else
    for (int i = tid; i < spp; i += kPsfThreads) {
        float x2, y2;
        disc_sample(ut[i], ur[i], lc.enp_r2, x2, y2);
        Ray r = ray_to(px, py, depth, x2, y2, lc.enp_z);
        trace_range<false>(tab, 0, lc.n_surf, true, r, nan_flag);
        propagate_to(r, st.d_sensor);
        if (!(EDGE && edge_defer(g, edge, r.ox, r.oy, r.ra > 0.f, cx, cy, s * L + l, n, i))) splat_hit(hist, g, r.ox, r.oy, r.ra, cx, cy);
    }
#endif
    __syncthreads();
    if (EDGE) {                                          // unnormalised: the re-trace adds the deferred rays, aadff_psf_normalise divides
        float* raw = edge.raw + ((size_t)(s * L + l) * N + n) * kk;
        for (int e = tid; e < kk; e += kPsfThreads) raw[e] = hist[e];
    } else {
        normalise_and_write(hist, red, g, map_grid, psf, s, l, n, N, L);
    }
    if (nan_flag && flags) atomicOr(flags, 1);
}

// TIMED: the same code under a second name for the launches that carry aadff_time_next_launch's events (see conv.hip)
template <bool TIMED>
__global__ __launch_bounds__(kPsfThreads, 6) void psf_points_kernel(const float* __restrict__ points, int N, int L,
                                                          const aadff_surface_t* __restrict__ surf_main,
                                                          const aadff_surface_t* __restrict__ surf_chief,
                                                          aadff_lens_const_t lc,
                                                          const aadff_lens_state_t* __restrict__ states,
                                                          const float* __restrict__ u_main, int spp, long main_ss,
                                                          long main_sl, const float* __restrict__ u_chief,
                                                          int spp_chief, long chief_ss, long chief_sl, SplatGeom g, int centre_mode, int map_grid, float* psf,
                                                          float* centre_out, int* flags, StageArgs stage) {
    psf_points_body<false>(points, N, L, surf_main, surf_chief, lc, states, u_main, spp, main_ss, main_sl, u_chief, spp_chief, chief_ss, chief_sl, g,
                           centre_mode, map_grid, psf, centre_out, flags, stage, EdgeArgs{});
}

// the edge-exact form: band rays deferred to aadff_strict_edge_retrace, histograms left unnormalised in edge.raw
__global__ __launch_bounds__(kPsfThreads, 6) void psf_points_edge_kernel(const float* __restrict__ points, int N, int L,
                                                          const aadff_surface_t* __restrict__ surf_main,
                                                          const aadff_surface_t* __restrict__ surf_chief,
                                                          aadff_lens_const_t lc,
                                                          const aadff_lens_state_t* __restrict__ states,
                                                          const float* __restrict__ u_main, int spp, long main_ss,
                                                          long main_sl, const float* __restrict__ u_chief,
                                                          int spp_chief, long chief_ss, long chief_sl, SplatGeom g, int centre_mode,
                                                          float* centre_out, int* flags, EdgeArgs edge) {
    psf_points_body<true>(points, N, L, surf_main, surf_chief, lc, states, u_main, spp, main_ss, main_sl, u_chief, spp_chief, chief_ss, chief_sl, g,
                          centre_mode, 0, nullptr, centre_out, flags, StageArgs{}, edge);
}

// raw [S*L][N][ks*ks] -> normalised PSFs in either layout of aadff_psf_points (grid: N x L x S)
__global__ __launch_bounds__(kPsfThreads) void psf_normalise_kernel(const float* __restrict__ raw, int N, int L, SplatGeom g, int map_grid, float* psf) {
    extern __shared__ float hist[];
    __shared__ float red[3 * kPsfWaves];
    const int n = blockIdx.x, l = blockIdx.y, s = blockIdx.z, kk = g.ks * g.ks;
    const float* src = raw + ((size_t)(s * L + l) * N + n) * kk;
    for (int e = threadIdx.x; e < kk; e += kPsfThreads) hist[e] = src[e];
    __syncthreads();
    normalise_and_write(hist, red, g, map_grid, psf, s, l, n, N, L);
}

// ------------------------------------------------------------------------------------
// Refocus + post_computation: workgroup per focus state.
//   deeplens/optics.py:1155-1180 (refocus), :1187-1217 (calc_fov), :178-187, :1097-1102
// ------------------------------------------------------------------------------------
#ifndef AADFF_STAGE_COPY_WGS
#define AADFF_STAGE_COPY_WGS 14
#endif
constexpr int kStageWorkgroups = 64;     // upload workgroups (x 1024 threads) riding on the refocus launch (aadff_refocus_staged)
constexpr int kRefocusThreads = 1024;     // plain entry: 16 waves per focus state
constexpr int kRefocusSplit = 4;          // staged entry: 4 workgroups x 256 threads per focus state, last arriver finishes
struct RefocusScratch {                   // per focus state, zero-initialised once by the caller; self-resetting
    float part[kRefocusSplit][2];         // (sum, count) of each quarter of the rays
    unsigned arrived;
    unsigned nan_seen;                    // NaN in a Newton residual of a quarter that did not finish the state
    unsigned loaded, passed;              // [state 0 only] gate of the upload workgroups (see refocus_kernel)
    unsigned pad[4];
};
static_assert(sizeof(RefocusScratch) == 64, "aadff.h documents 64 bytes per focus state");

template <int NT, int SPLIT>
__global__ __launch_bounds__(NT) void refocus_kernel(const float* __restrict__ depth, const float* __restrict__ u,
                                                       int spp, long u_ss, const aadff_surface_t* __restrict__ surf,
                                                       aadff_lens_const_t lc, aadff_lens_state_t* states,
                                                       int do_refocus, int S, const float* __restrict__ stage_src,
                                                       float* __restrict__ stage_dst, long stage_n, RefocusScratch* scratch) {
    if ((int)blockIdx.x >= S * SPLIT) {
        // Upload workgroups: copy the pinned-host uniform block to HBM for the PSF kernel while the
        // focus workgroups (which read their own draws straight from the host block) trace.
        // The bulk copy would queue ~20 us of PCIe reads in front of the focus workgroups' own 16 KB of draws:
        // wait until every focus workgroup holds its draws (they have lower block ids: dispatched first).
        if constexpr (SPLIT > 1) {
            if (threadIdx.x == 0) {
                int spins = 0;
                while (__hip_atomic_load(&scratch->loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(S * SPLIT)) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 20)) break;                       // never hang the queue; only the overlap is lost
                }
                const unsigned old = __hip_atomic_fetch_add(&scratch->passed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == gridDim.x - S * SPLIT - 1) {                  // last one through resets the gate for the next launch
                    __hip_atomic_store(&scratch->loaded, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&scratch->passed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
        }
        const long n4 = stage_n >> 2;
        const int first = S * SPLIT;
        const long stride = (long)(gridDim.x - first) * NT;
        const float4* src4 = reinterpret_cast<const float4*>(stage_src);
        float4* dst4 = reinterpret_cast<float4*>(stage_dst);
        long i = (long)(blockIdx.x - first) * NT + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {          // four loads in flight per lane: PCIe latency
            const float4 a = src4[i], b = src4[i + stride], c = src4[i + 2 * stride], d = src4[i + 3 * stride];
            dst4[i] = a; dst4[i + stride] = b; dst4[i + 2 * stride] = c; dst4[i + 3 * stride] = d;
        }
        for (; i < n4; i += stride) dst4[i] = src4[i];
        if ((int)blockIdx.x == first && threadIdx.x < (stage_n & 3)) stage_dst[(n4 << 2) + threadIdx.x] = stage_src[(n4 << 2) + threadIdx.x];
        return;
    }
    __shared__ float red[2 * (NT / 64)];
    __shared__ float s_dsensor;
    __shared__ int s_count;
    __shared__ int s_last, s_flags0;
    const int s = blockIdx.x / SPLIT, part = blockIdx.x - s * SPLIT, tid = threadIdx.x;
    int nan_flag = 0;
    int flags = 0;
    if (do_refocus) {
        const float* ut = u + (size_t)s * u_ss;
        const float* ur = ut + spp;
        const float dep = depth[s];
        // this workgroup's rays: [begin, end); two rays per lane (packed arithmetic)
        const int chunk = (spp + SPLIT - 1) / SPLIT;
        const int begin = part * chunk, end = min(spp, begin + chunk);
        float sum = 0.f, cnt = 0.f;
        // first pair of draws up front; with the staged entry they come over PCIe and open the upload gate
        float pt0 = 0.f, pt1 = 0.f, pr0 = 0.f, pr1 = 0.f;
        if (begin + tid < end) {
            const int i = begin + tid, j1 = i + NT < end ? i + NT : i;
            pt0 = ut[i]; pt1 = ut[j1]; pr0 = ur[i]; pr1 = ur[j1];
        }
        if constexpr (SPLIT > 1) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pt0), "+v"(pt1), "+v"(pr0), "+v"(pr1));
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&scratch->loaded, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int i = begin + tid; i < end; i += 2 * NT) {
            const int i1 = i + NT;
            const i2 act = {-1, i1 < end ? -1 : 0};
            const int j1 = act.y ? i1 : i;
            f2 x2, y2;
            const bool firstit = i == begin + tid;
            disc_sample2((f2){firstit ? pt0 : ut[i], firstit ? pt1 : ut[j1]}, (f2){firstit ? pr0 : ur[i], firstit ? pr1 : ur[j1]}, lc.first_r2, x2, y2);
            Ray2 r;
            r.ox = x2; r.oy = y2; r.oz = f2s(lc.first_d);
            r.dx = x2; r.dy = y2; r.dz = f2s(lc.first_d - dep);               // o - (0,0,depth)
            const f2 inv = vrsq(vmax(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz, f2s(1e-24f)));
            r.dx *= inv; r.dy *= inv; r.dz *= inv;
            r.alive = act;
            trace_forward2(surf, lc.n_surf, r, nan_flag);
            f2 t = (r.dx * r.ox + r.dy * r.oy) * vrcp(r.dx * r.dx + r.dy * r.dy);
            t = t * r.ra;
            const f2 fd = r.oz - r.dz * t;
            const i2 ok = (r.ra > 0.f) & (fd == fd) & (fd > 0.f);
            sum += (ok.x ? fd.x : 0.f) + (ok.y ? fd.y : 0.f);
            cnt += (ok.x ? 1.f : 0.f) + (ok.y ? 1.f : 0.f);
        }
        sum = wave_sum(sum); cnt = wave_sum(cnt);
        if ((tid & 63) == 0) { red[(tid >> 6) * 2] = sum; red[(tid >> 6) * 2 + 1] = cnt; }
        const int s_nan = __syncthreads_or(nan_flag);
        if (tid == 0) {
            float ts = 0.f, tc = 0.f;
            for (int w = 0; w < NT / 64; ++w) { ts += red[2 * w]; tc += red[2 * w + 1]; }
            int last = 1;
            s_flags0 = 0;
            if constexpr (SPLIT > 1) {
                // publish the partial, count arrivals (the counter wraps to 0 by itself), the last one reduces in
                // fixed order (deterministic) and goes on to the field-of-view trace
                RefocusScratch* sc = scratch + s;
                sc->part[part][0] = ts; sc->part[part][1] = tc;
                if (s_nan) atomicOr(&sc->nan_seen, 1u);
                const unsigned old = __hip_atomic_fetch_add(&sc->arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                last = old == SPLIT - 1;
                if (last) {
                    __hip_atomic_store(&sc->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (atomicExch(&sc->nan_seen, 0u)) s_flags0 = 1;
                    ts = 0.f; tc = 0.f;
                    for (int p = 0; p < SPLIT; ++p) {
                        ts += __hip_atomic_load(&sc->part[p][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tc += __hip_atomic_load(&sc->part[p][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            s_last = last;
            s_dsensor = ts / tc;
            s_count = (int)tc;
        }
        __syncthreads();
        if (!s_last) return;
        flags = s_flags0;
    } else {
        if (tid == 0) { s_dsensor = states[s].d_sensor; s_count = states[s].n_focus_rays; }
        __syncthreads();
    }
    const float d_sensor = s_dsensor;

    // calc_fov: M=100 rays from the sensor corner through the shrunk exit pupil, traced backward
    constexpr int M = 100;
    float tsum = 0.f, wsum = 0.f;
    if (tid < M) {
        const float start = -lc.exp_r_shrunk, end = lc.exp_r_shrunk;
        const float step = (end - start) / (float)(M - 1);
        const float x2 = tid < M / 2 ? start + step * (float)tid : end - step * (float)(M - 1 - tid);   // torch.linspace
        Ray r = ray_to(lc.r_last, 0.f, d_sensor, x2, 0.f, lc.exp_z);
        trace_range<false>(surf, 0, lc.n_surf, false, r, nan_flag);
        tsum = (r.dx / r.dz) * r.ra;
        wsum = r.ra;
    }
    tsum = wave_sum(tsum); wsum = wave_sum(wsum);
    __syncthreads();
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = tsum; red[(tid >> 6) * 2 + 1] = wsum; }
    __syncthreads();
    const int any_nan = __syncthreads_or(nan_flag);
    if (tid == 0) {
        if (any_nan) flags |= 1;
        float ts = 0.f, tw = 0.f;
        for (int w = 0; w < NT / 64; ++w) { ts += red[2 * w]; tw += red[2 * w + 1]; }
        float hfov = atanf(ts / tw);
        if (hfov != hfov) { hfov = 0.5f; flags |= 2; }
        const double th = tan((double)hfov);
        const double foclen = (double)lc.r_last / th;
        aadff_lens_state_t o;
        o.d_sensor = d_sensor;
        o.hfov = hfov;
        o.tan_hfov = (float)th;
        o.foclen = (float)foclen;
        o.fnum = (float)(foclen / (double)lc.enp_r / 2.0);
        o.n_focus_rays = s_count;
        o.flags = flags;
        o.pad = 0;
        states[s] = o;
    }
}

}  // namespace aadff

using namespace aadff;

extern "C" {

int aadff_trace_rays(const float* o_in, const float* d_in, const float* ra_in, float* o_out, float* d_out,
                     float* ra_out, int n, const aadff_surface_t* surf, int first, int last, int forward,
                     const aadff_lens_state_t* state_or_null, int* flags_or_null, aadff_stream_t stream) {
    AADFF_CHECK_ARG(o_in && d_in && o_out && d_out && ra_out && surf, "trace_rays: NULL pointer");
    AADFF_CHECK_ARG(n >= 0 && first >= 0 && first <= last && last <= AADFF_MAX_SURF, "trace_rays: bad range [%d,%d) or n=%d", first, last, n);
    if (n == 0) return 0;
    hipLaunchKernelGGL(trace_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, o_in, d_in, ra_in,
                       o_out, d_out, ra_out, n, surf, first, last, forward, state_or_null, flags_or_null);
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_trace_points(const float* points_obj, int N, const float* u_theta, const float* u_r, int spp, float pupil_z,
                       float pupil_r, const aadff_surface_t* surf, int n_surf, const aadff_lens_state_t* state,
                       float* o_out, float* d_out, float* ra_out, aadff_stream_t stream) {
    AADFF_CHECK_ARG(points_obj && u_theta && u_r && surf && state && o_out && d_out && ra_out, "trace_points: NULL pointer");
    AADFF_CHECK_ARG(N > 0 && N <= 65535 && spp > 0 && n_surf > 0 && n_surf <= AADFF_MAX_SURF, "trace_points: bad sizes N=%d spp=%d n_surf=%d", N, spp, n_surf);
    const float r2 = (float)((double)pupil_r * (double)pupil_r);
    hipLaunchKernelGGL(trace_points_kernel, dim3((spp + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, points_obj, N,
                       u_theta, u_r, spp, pupil_z, r2, surf, n_surf, state, o_out, d_out, ra_out);
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_psf_splat(const float* o, const float* ra, const float* centre, int spp, int N, float pixel_size, int ks,
                    float* psf_raw_or_null, float* psf, aadff_stream_t stream) {
    AADFF_CHECK_ARG(o && ra && centre && psf, "psf_splat: NULL pointer");
    AADFF_CHECK_ARG(spp > 0 && N > 0 && ks >= 1 && ks <= AADFF_MAX_KS, "psf_splat: bad sizes spp=%d N=%d ks=%d", spp, N, ks);
    hipLaunchKernelGGL(psf_splat_kernel, dim3(N), dim3(256), (size_t)ks * ks * sizeof(float), (hipStream_t)stream, o, ra, centre, spp, N,
                       make_splat_geom(pixel_size, ks), psf_raw_or_null, psf);
    AADFF_CHECK_LAUNCH();
    return 0;
}

static int psf_points_launch(const float* points, int S, int N, int L, const aadff_surface_t* surf_main,
                     const aadff_surface_t* surf_chief, aadff_lens_const_t lc, const aadff_lens_state_t* states,
                     const float* u_main, int spp, long main_stride_s, long main_stride_l, const float* u_chief,
                     int spp_chief, long chief_stride_s, long chief_stride_l, int ks, int centre_mode,
                     int map_layout, float* psf, float* centre_out_or_null, int* flags_or_null,
                     const aadff_stage_t* stage, aadff_stream_t stream) {
    AADFF_CHECK_ARG(points && surf_main && states && u_main && psf, "psf_points: NULL pointer");
    AADFF_CHECK_ARG(centre_mode == 0 || (surf_chief && u_chief && spp_chief > 0), "psf_points: chief-ray centre needs surf_chief/u_chief");
    AADFF_CHECK_ARG(S > 0 && S <= 65535 && N > 0 && L > 0 && L <= 65535 && spp > 0, "psf_points: bad sizes S=%d N=%d L=%d spp=%d", S, N, L, spp);
    AADFF_CHECK_ARG(ks >= 1 && ks <= AADFF_MAX_KS, "psf_points: ks %d outside [1,%d]", ks, AADFF_MAX_KS);
    AADFF_CHECK_ARG(lc.n_surf > 0 && lc.n_surf <= AADFF_MAX_SURF, "psf_points: n_surf %d", lc.n_surf);
    int map_grid = 0;
    if (map_layout) {
        map_grid = (int)lround(std::sqrt((double)N));
        AADFF_CHECK_ARG(map_grid * map_grid == N, "psf_points: psf_map layout needs N = g*g, got %d", N);
    }
    StageArgs sa{};
    if (stage && stage->first_slice < S) {
        AADFF_CHECK_ARG(stage->src_host && stage->dst_dev && stage->counters, "psf_points_staged: NULL pointer in aadff_stage_t");
        AADFF_CHECK_ARG(stage->first_slice >= 0 && stage->slice_stride > 0 && (stage->slice_stride & 3) == 0,
                        "psf_points_staged: slice_stride %ld must be a positive multiple of 4", stage->slice_stride);
        AADFF_CHECK_ARG((((uintptr_t)stage->src_host | (uintptr_t)stage->dst_dev) & 15) == 0, "psf_points_staged: blocks must be 16-byte aligned");
        AADFF_CHECK_ARG(S < 65535, "psf_points_staged: S=%d", S);
        void* mapped = nullptr;
        if (hipHostGetDevicePointer(&mapped, const_cast<float*>(stage->src_host), 0) != hipSuccess || !mapped) {
            (void)hipGetLastError();
            AADFF_CHECK_ARG(false, "psf_points_staged: src_host is not pinned (device-mapped) host memory");
        }
        sa.src = reinterpret_cast<const float4*>(mapped);
        sa.dst = reinterpret_cast<float4*>(stage->dst_dev);
        sa.slice_n4 = stage->slice_stride >> 2;
        sa.first_slice = stage->first_slice;
        sa.copy_wgs = (int)std::min<long>(std::min<long>(AADFF_STAGE_COPY_WGS, (long)N * L), (sa.slice_n4 + kPsfThreads - 1) / kPsfThreads);
        sa.counters = stage->counters;
        sa.target = stage->generation * (unsigned)sa.copy_wgs;
        // the late path re-bases u_main / u_chief from dst_dev onto src_host: both must point into the staged block
        const float* lo = stage->dst_dev;
        const float* hi = stage->dst_dev + (long)S * stage->slice_stride;
        AADFF_CHECK_ARG(u_main >= lo && u_main < hi && (!u_chief || (u_chief >= lo && u_chief < hi)),
                        "psf_points_staged: u_main/u_chief must point into dst_dev[0 .. S*slice_stride)");
    }
    hipEvent_t ev0 = g_time_start, ev1 = g_time_stop;                    // aadff_time_next_launch
    g_time_start = g_time_stop = nullptr;
    if (ev0)
        hipExtLaunchKernelGGL(psf_points_kernel<true>, dim3(N, L, sa.src ? S + 1 : S), dim3(kPsfThreads), (size_t)ks * ks * sizeof(float), (hipStream_t)stream,
                              ev0, ev1, 0, points, N, L, surf_main, surf_chief, lc, states, u_main, spp, main_stride_s, main_stride_l, u_chief,
                              spp_chief, chief_stride_s, chief_stride_l, make_splat_geom(lc.pixel_size, ks), centre_mode, map_grid, psf,
                              centre_out_or_null, flags_or_null, sa);
    else
        hipLaunchKernelGGL(psf_points_kernel<false>, dim3(N, L, sa.src ? S + 1 : S), dim3(kPsfThreads), (size_t)ks * ks * sizeof(float), (hipStream_t)stream, points, N, L, surf_main,
                           surf_chief, lc, states, u_main, spp, main_stride_s, main_stride_l, u_chief, spp_chief, chief_stride_s,
                           chief_stride_l, make_splat_geom(lc.pixel_size, ks),
                           centre_mode, map_grid, psf, centre_out_or_null, flags_or_null, sa);
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_psf_points(const float* points, int S, int N, int L, const aadff_surface_t* surf_main,
                     const aadff_surface_t* surf_chief, aadff_lens_const_t lc, const aadff_lens_state_t* states,
                     const float* u_main, int spp, long main_stride_s, long main_stride_l, const float* u_chief,
                     int spp_chief, long chief_stride_s, long chief_stride_l, int ks, int centre_mode,
                     int map_layout, float* psf, float* centre_out_or_null, int* flags_or_null,
                     aadff_stream_t stream) {
    return psf_points_launch(points, S, N, L, surf_main, surf_chief, lc, states, u_main, spp, main_stride_s, main_stride_l,
                             u_chief, spp_chief, chief_stride_s, chief_stride_l, ks, centre_mode, map_layout, psf,
                             centre_out_or_null, flags_or_null, nullptr, stream);
}

int aadff_psf_points_staged(const float* points, int S, int N, int L, const aadff_surface_t* surf_main,
                            const aadff_surface_t* surf_chief, aadff_lens_const_t lc, const aadff_lens_state_t* states,
                            const float* u_main, int spp, long main_stride_s, long main_stride_l, const float* u_chief,
                            int spp_chief, long chief_stride_s, long chief_stride_l, int ks, int centre_mode,
                            int map_layout, float* psf, float* centre_out_or_null, int* flags_or_null,
                            const aadff_stage_t* stage, aadff_stream_t stream) {
    AADFF_CHECK_ARG(stage, "psf_points_staged: NULL stage");
    return psf_points_launch(points, S, N, L, surf_main, surf_chief, lc, states, u_main, spp, main_stride_s, main_stride_l,
                             u_chief, spp_chief, chief_stride_s, chief_stride_l, ks, centre_mode, map_layout, psf,
                             centre_out_or_null, flags_or_null, stage, stream);
}

int aadff_psf_points_edge(const float* points, int S, int N, int L, const aadff_surface_t* surf_main,
                          const aadff_surface_t* surf_chief, aadff_lens_const_t lc, const aadff_lens_state_t* states,
                          const float* u_main, int spp, long main_stride_s, long main_stride_l, const float* u_chief,
                          int spp_chief, long chief_stride_s, long chief_stride_l, int ks, float delta_mm, float* raw,
                          float* centre_out, float* slope_out_or_null, unsigned* edge_count, unsigned* edge_list, int edge_cap,
                          int* flags_or_null, aadff_stream_t stream) {
#if defined(AADFF_PSF_SCALAR) || defined(AADFF_PSF_NO_COMPACT) || defined(AADFF_PSF_SWITCH_LOOP)
    set_error("psf_points_edge: not part of the measurement builds of the trace kernels");
    return AADFF_EUNSUPPORTED;
#else
    AADFF_CHECK_ARG(points && surf_main && surf_chief && states && u_main && u_chief && raw && centre_out && edge_count && edge_list,
                    "psf_points_edge: NULL pointer");
    AADFF_CHECK_ARG(S > 0 && S <= 65535 && N > 0 && N <= 65535 && L > 0 && L <= 65535 && spp > 0 && spp <= 65536 && spp_chief > 0,
                    "psf_points_edge: bad sizes S=%d N=%d L=%d spp=%d spp_chief=%d (point and sample ids are 16 bits each)", S, N, L, spp, spp_chief);
    AADFF_CHECK_ARG(ks >= 1 && ks <= AADFF_MAX_KS, "psf_points_edge: ks %d outside [1,%d]", ks, AADFF_MAX_KS);
    AADFF_CHECK_ARG(lc.n_surf > 0 && lc.n_surf <= AADFF_MAX_SURF, "psf_points_edge: n_surf %d", lc.n_surf);
    AADFF_CHECK_ARG(delta_mm >= 0.f && edge_cap >= 1, "psf_points_edge: delta %g mm, capacity %d", (double)delta_mm, edge_cap);
    hipStream_t st = (hipStream_t)stream;
    AADFF_CHECK_HIP(hipMemsetAsync(edge_count, 0, (size_t)S * L * sizeof(unsigned), st));
    EdgeArgs e{};
    e.delta = delta_mm; e.count = edge_count; e.list = edge_list; e.cap = edge_cap; e.raw = raw; e.slope = slope_out_or_null;
    hipLaunchKernelGGL(psf_points_edge_kernel, dim3(N, L, S), dim3(kPsfThreads), (size_t)ks * ks * sizeof(float), st, points, N, L, surf_main, surf_chief, lc,
                       states, u_main, spp, main_stride_s, main_stride_l, u_chief, spp_chief, chief_stride_s, chief_stride_l,
                       make_splat_geom(lc.pixel_size, ks), 1, centre_out, flags_or_null, e);
    AADFF_CHECK_LAUNCH();
    return 0;
#endif
}

int aadff_psf_normalise(const float* raw, int S, int N, int L, float pixel_size, int ks, int map_layout, float* psf, aadff_stream_t stream) {
    AADFF_CHECK_ARG(raw && psf, "psf_normalise: NULL pointer");
    AADFF_CHECK_ARG(S > 0 && S <= 65535 && N > 0 && L > 0 && L <= 65535 && ks >= 1 && ks <= AADFF_MAX_KS, "psf_normalise: bad sizes S=%d N=%d L=%d ks=%d", S, N, L, ks);
    int map_grid = 0;
    if (map_layout) {
        map_grid = (int)lround(std::sqrt((double)N));
        AADFF_CHECK_ARG(map_grid * map_grid == N, "psf_normalise: psf_map layout needs N = g*g, got %d", N);
    }
    hipLaunchKernelGGL(psf_normalise_kernel, dim3(N, L, S), dim3(kPsfThreads), (size_t)ks * ks * sizeof(float), (hipStream_t)stream, raw, N, L,
                       make_splat_geom(pixel_size, ks), map_grid, psf);
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_refocus(const float* depth, int S, const float* u, int spp, long u_stride_s,
                  const aadff_surface_t* surf_green, aadff_lens_const_t lc, aadff_lens_state_t* states,
                  aadff_stream_t stream) {
    AADFF_CHECK_ARG(depth && u && surf_green && states, "refocus: NULL pointer");
    AADFF_CHECK_ARG(S > 0 && spp > 0 && lc.n_surf > 0 && lc.n_surf <= AADFF_MAX_SURF, "refocus: bad sizes S=%d spp=%d", S, spp);
    hipLaunchKernelGGL((refocus_kernel<kRefocusThreads, 1>), dim3(S), dim3(kRefocusThreads), 0, (hipStream_t)stream, depth, u, spp, u_stride_s, surf_green, lc, states, 1, S,
                       (const float*)nullptr, (float*)nullptr, 0L, (RefocusScratch*)nullptr);
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_refocus_staged(const float* depth, int S, const float* u_host, float* u_dev, long n_u, int spp,
                         long u_stride_s, const aadff_surface_t* surf_green, aadff_lens_const_t lc,
                         aadff_lens_state_t* states, void* scratch, aadff_stream_t stream) {
    AADFF_CHECK_ARG(depth && u_host && u_dev && surf_green && states && scratch, "refocus_staged: NULL pointer");
    AADFF_CHECK_ARG(S > 0 && spp > 0 && lc.n_surf > 0 && lc.n_surf <= AADFF_MAX_SURF, "refocus_staged: bad sizes S=%d spp=%d", S, spp);
    AADFF_CHECK_ARG(n_u >= 0, "refocus_staged: n_u %ld", n_u);
    AADFF_CHECK_ARG((((uintptr_t)u_host | (uintptr_t)u_dev) & 15) == 0, "refocus_staged: u_host/u_dev must be 16-byte aligned");
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, const_cast<float*>(u_host), 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        AADFF_CHECK_ARG(false, "refocus_staged: u_host is not pinned (device-mapped) host memory");
    }
    constexpr int NT = 256;
    const long n4 = n_u >> 2;
    const int copy_wgs = (int)std::min<long>(kStageWorkgroups * (kRefocusThreads / NT), std::max<long>(1, (n4 + NT - 1) / NT));
    hipLaunchKernelGGL((refocus_kernel<NT, kRefocusSplit>), dim3(S * kRefocusSplit + copy_wgs), dim3(NT), 0, (hipStream_t)stream, depth,
                       (const float*)mapped, spp, u_stride_s, surf_green, lc, states, 1, S, (const float*)mapped, u_dev, n_u,
                       reinterpret_cast<RefocusScratch*>(scratch));
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_post_computation(int S, const aadff_surface_t* surf_green, aadff_lens_const_t lc, aadff_lens_state_t* states,
                           aadff_stream_t stream) {
    AADFF_CHECK_ARG(surf_green && states, "post_computation: NULL pointer");
    AADFF_CHECK_ARG(S > 0 && lc.n_surf > 0 && lc.n_surf <= AADFF_MAX_SURF, "post_computation: bad sizes S=%d", S);
    hipLaunchKernelGGL((refocus_kernel<kRefocusThreads, 1>), dim3(S), dim3(kRefocusThreads), 0, (hipStream_t)stream, (const float*)nullptr,
                       (const float*)nullptr, 0, 0L, surf_green, lc, states, 0, S, (const float*)nullptr, (float*)nullptr, 0L,
                       (RefocusScratch*)nullptr);
    AADFF_CHECK_LAUNCH();
    return 0;
}

// The address at which the GPU sees a block of pinned host memory (hipHostMalloc / torch pin_memory), or an error when the block
// is not device-mapped: the per-call API hands such addresses to aadff_refocus (2 x 2048 draws and the depth read over PCIe by one
// workgroup) and as `points` to aadff_psf_points_staged instead of queueing a copy in front of every launch.
int aadff_host_device_pointer(const void* host, void** dev_out) {
    AADFF_CHECK_ARG(host && dev_out, "host_device_pointer: NULL pointer");
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, const_cast<void*>(host), 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        AADFF_CHECK_ARG(false, "host_device_pointer: not pinned (device-mapped) host memory");
    }
    *dev_out = mapped;
    return 0;
}

__global__ void publish_flags_kernel(const int* __restrict__ flags, int* mirror) {
    if (threadIdx.x == 0) __hip_atomic_store(mirror, __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int aadff_publish_flags(const int* flags_dev, int* mirror_host, aadff_stream_t stream) {
    AADFF_CHECK_ARG(flags_dev && mirror_host, "publish_flags: NULL pointer");
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, mirror_host, 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        AADFF_CHECK_ARG(false, "publish_flags: mirror_host is not pinned (device-mapped) host memory");
    }
    hipLaunchKernelGGL(publish_flags_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flags_dev, reinterpret_cast<int*>(mapped));
    AADFF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
