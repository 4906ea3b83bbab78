// csrc/strict_math.h for TWO rays per lane: the same float32 operations in the same order, every value a float2 whose halves
// belong to two independent rays, so that the multiplications and additions issue as v_pk_mul_f32 / v_pk_add_f32 and the
// refinement steps of the divisions as v_pk_fma_f32 (the strict trace is VALU-issue bound: 2.4 G wave-instructions per psf_map
// level, profiles/r05_*).  Nothing is contracted, reassociated or approximated:
//   * a * b + c stays a rounded product and a rounded sum (fp contract off);
//   * div2 is the correctly rounded quotient: the reciprocal-refinement sequence the compiler itself emits for an IEEE float32
//     division (v_rcp_f32, two Newton steps on the reciprocal, quotient, two residual corrections, v_div_fixup_f32 for zeros /
//     infinities / NaNs), with the refinement as packed FMAs and WITHOUT the v_div_scale_f32 pre-scaling, whose only job is to keep
//     the intermediate terms normal when a numerator is below 2^-104 or the exponents of numerator and denominator are more than
//     96 apart.  The operands of this trace are millimetres and direction cosines (1e-12 .. 1e4) or exact zeros; tests/ compares
//     whole levels bit for bit with the scalar form and `aadff_selftest_strict_ops` sweeps the operand ranges;
//   * sqrt2 is the correctly rounded root: v_sqrt_f32 and the compiler's own two-neighbour correction with packed residuals.
// Aspheric polynomial terms (float64 powers, 2 of rf50mm's 12 surfaces) go through the scalar functions half by half.
#pragma once
#include "strict_math.h"

#pragma clang fp contract(off)

namespace aadff {
namespace strict {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 f2s(float a) { return (f2){a, a}; }
__device__ __forceinline__ f2 sel(i2 m, f2 a, f2 b) { return (f2){m.x ? a.x : b.x, m.y ? a.y : b.y}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 abs2(f2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ f2 sqrt2_ieee(f2 a) { return (f2){sqrtf(a.x), sqrtf(a.y)}; }
// Correctly rounded square root of both halves for arguments that are 0, >= 2^-96, negative (-> NaN), +inf or NaN: v_sqrt_f32
// (1 ulp), then the two neighbours s -+ 1 ulp are tested with exact residuals x - s_dn * s <= 0 -> s_dn,  x - s_up * s > 0 -> s_up - the
// algorithm the compiler's sqrtf expands to, minus its pre-scaling of arguments below 2^-96, with the residuals as packed FMAs.
// Every square root of the trace qualifies (1 - (1 + k) r^2 c^2 is 0 or >= 2^-24; squared norms; 1 - eta^2 (1 - cos^2)); the
// stop's sqrt(x^2 + y^2) <= r compares against millimetres, so it uses the general form.
__device__ __forceinline__ f2 sqrt2(f2 x) {
    const f2 s = (f2){__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const f2 sd = (f2){__uint_as_float(__float_as_uint(s.x) - 1u), __uint_as_float(__float_as_uint(s.y) - 1u)};
    const f2 su = (f2){__uint_as_float(__float_as_uint(s.x) + 1u), __uint_as_float(__float_as_uint(s.y) + 1u)};
    const f2 rd = __builtin_elementwise_fma(-sd, s, x), ru = __builtin_elementwise_fma(-su, s, x);
    f2 r = (f2){rd.x <= 0.f ? sd.x : s.x, rd.y <= 0.f ? sd.y : s.y};
    r = (f2){ru.x > 0.f ? su.x : r.x, ru.y > 0.f ? su.y : r.y};
    // 0 and +inf are their own roots (their neighbours are not numbers the tests above can rank)
    const i2 keep = (x == 0.f) | (x == __builtin_inff());
    return (f2){keep.x ? x.x : r.x, keep.y ? x.y : r.y};
}

// The division in two halves: the refined reciprocal of the denominator (from any approximation r0 with a relative error of a
// few 2^-23: one Newton step), then quotient, two residual corrections and the fix-up.  Several quotients over ONE denominator
// share the reciprocal (the three components of a normal), and the reciprocal of d * d starts from the square of the refined
// reciprocal of d instead of a second pair of quarter-rate v_rcp_f32 (sag and d sag / d r^2 divide by 1 + sf and (1 + sf)^2):
// the quotient is the correctly rounded one either way - `aadff_selftest_strict_ops` op 2 checks that form for EVERY float
// 1 + sf in [1, 2] against the compiler's division.
__device__ __forceinline__ f2 recip_refine2(f2 d, f2 r0) {
    const f2 e = fma2(-d, r0, f2s(1.f));
    return fma2(e, r0, r0);
}
__device__ __forceinline__ f2 recip2(f2 d) { return recip_refine2(d, (f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}); }
__device__ __forceinline__ f2 div2_r(f2 n, f2 d, f2 r) {
    f2 q = n * r;
    const f2 e2 = fma2(-d, q, n);
    q = fma2(e2, r, q);
    const f2 e3 = fma2(-d, q, n);
    q = fma2(e3, r, q);
    return (f2){__builtin_amdgcn_div_fixupf(q.x, d.x, n.x), __builtin_amdgcn_div_fixupf(q.y, d.y, n.y)};
}
__device__ __forceinline__ f2 div2(f2 n, f2 d) { return div2_r(n, d, recip2(d)); }

struct R32 { f2 x, y, z; };

__device__ __forceinline__ f2 conic_a2(const Surf& s, f2 r2) { return ((1.f + s.k) * r2) * (s.c * s.c); }

// sag and d sag / d r^2 (surfaces.py:787-830) sharing sf = sqrt(1 - (1 + k) r^2 c^2); the polynomial terms (float64 powers) half by half
__device__ __forceinline__ void sag_dsag2(const Surf& s, f2 r2, f2& z, f2& g) {
    const f2 a = conic_a2(s, r2);
    const f2 sf = sqrt2(1.f - a);
    const f2 opsf = 1.f + sf;
    const f2 ro = recip2(opsf);
    z = div2_r(r2 * s.c, opsf, ro);
    const f2 o2 = opsf * opsf;
    g = div2_r((opsf + div2(a * 0.5f, sf)) * s.c, o2, recip_refine2(o2, ro * ro));
    if (s.n_ai > 0) {                                    // uniform
        z = (f2){sag_poly(s, r2.x, z.x), sag_poly(s, r2.y, z.y)};
        g = (f2){dsag_poly(s, r2.x, g.x), dsag_poly(s, r2.y, g.y)};
    }
}

// validity masks as 1.0 / 0.0 factors (the reference multiplies by them): surfaces.py:724-743.  KGT = k > -1 (uniform per surface)
template <bool KGT>
__device__ __forceinline__ f2 valid_strict_f(const Surf& s, f2 x, f2 y) {
    const f2 q = x * x + y * y;
    if (KGT) return (f2){(q.x < s.r2 && q.x < s.r2_shape) ? 1.f : 0.f, (q.y < s.r2 && q.y < s.r2_shape) ? 1.f : 0.f};
    return (f2){q.x < s.r2 ? 1.f : 0.f, q.y < s.r2 ? 1.f : 0.f};
}
template <bool KGT>
__device__ __forceinline__ f2 valid_loose_f(const Surf& s, f2 x, f2 y) {
    const f2 q = x * x + y * y;
    if (KGT) return (f2){q.x < s.r2_shape ? 1.f : 0.f, q.y < s.r2_shape ? 1.f : 0.f};
    return (f2){q.x > 0.f ? 1.f : 0.f, q.y > 0.f ? 1.f : 0.f};
}

// alive: 1.0 / 0.0 per half
template <bool STRICT, bool KGT>
__device__ __forceinline__ void residual2(const Surf& s, const R32& o, const R32& d, f2 alive, f2 t, f2& ft, f2& dfdt) {
    const f2 px = o.x + d.x * t, py = o.y + d.y * t, pz = o.z + d.z * t;
    const f2 mf = (STRICT ? valid_strict_f<KGT>(s, px, py) : valid_loose_f<KGT>(s, px, py)) * alive;      // m = mask & alive
    const f2 xm = px * mf, ym = py * mf;
    const f2 r2 = xm * xm + ym * ym;
    f2 z, g;
    sag_dsag2(s, r2, z, g);
    ft = (z + s.d) - pz;
    const f2 dr2dt = 2.f * ((d.x * d.x + d.y * d.y) * t + (d.x * o.x + d.y * o.y));
    dfdt = g * dr2dt - d.z;
}

__device__ __forceinline__ f2 clamp_step2(f2 v) { return (f2){clamp_step(v.x), clamp_step(v.y)}; }

__device__ __forceinline__ void normalize32(f2& x, f2& y, f2& z) {
    const f2 n2 = fma2(z, z, fma2(y, y, x * x));
    const f2 sq = sqrt2(n2);
    const f2 den = (f2){fmaxf(sq.x, 1e-12f), fmaxf(sq.y, 1e-12f)};
    const f2 rd = recip2(den);
    x = div2_r(x, den, rd); y = div2_r(y, den, rd); z = div2_r(z, den, rd);
}

__device__ __forceinline__ unsigned fbits(float x) { return __float_as_uint(x); }

// n iterations of the loose loop (deeplens/surfaces.py:547-563) from t0 for the two rays of a lane: returns t after n iterations, ORs
// bit j - 1 into `mine` / `nans` when |ft| > 5e-5 / ft is NaN in iteration j (1 <= j <= n) for either ray.  Evaluates only until both
// iterates repeat with a period p <= 3 (t_j == t_{j-p}): then t_{m+p} = t_m and iteration m + p + 1 repeats iteration m + 1 for every
// m >= j - p, so the remaining iterates and bits follow from the last p.  (Period 3 is not exotic: about 1 % of the live rays end up
// walking three neighbouring floats, tools/strict_cycle_stats.py - half the waves would otherwise run all ten iterations for one
// such ray.)  A half that became periodic earlier just keeps walking its cycle - t_j == t_{j-p} then holds at every later j as
// well - so nothing is frozen and both periods are read off at the common exit.  At j = 1 (2) the histories still hold t_0, so a
// "period 2 (3)" match there is the fixed point and is taken as p = 1.
__device__ __forceinline__ float cycle_finish(int p, int j, int n, unsigned& bm, unsigned& bn, float tj, float tj1, float tj2) {
    const unsigned all = (1u << n) - 1u;
    const int dn = n - j;
    const unsigned rep = p == 1 ? 0x3ffu : (p == 2 ? 0x155u : 0x249u);
    const unsigned wm = (1u << p) - 1u;
    bm |= ((((bm >> (j - p)) & wm) * rep) << j) & all;
    bn |= ((((bn >> (j - p)) & wm) * rep) << j) & all;
    const int r = p == 1 ? 0 : (p == 2 ? (dn & 1) : dn - 3 * ((dn * 11) >> 5));
    return r == 0 ? tj : (p - r == 1 ? tj1 : tj2);
}

// t_prev (optional): t after n - 1 iterations as well - what the loop would have returned under a count one lower (the two-variant
// chief pass of fused_psf_kernel, csrc/strict_fused.hip).
template <bool KGT, bool WITH_PREV = false>
__device__ __forceinline__ f2 loose_cycle2(const Surf& s, const R32& o, const R32& d, f2 alive, int n, f2 t0, unsigned& mine, unsigned& nans, f2* t_prev = nullptr) {
    f2 t = t0, h2 = t0, h3 = t0, tn = t0;
    unsigned bmx = 0, bmy = 0, bnx = 0, bny = 0;
    int j = 0, px = 0, py = 0;
    while (true) {
        ++j;
        f2 ft, dfdt;
        residual2<false, KGT>(s, o, d, alive, t, ft, dfdt);
        const f2 af = abs2(ft);
        bmx |= (af.x > kTolLoose ? 1u : 0u) << (j - 1);
        bmy |= (af.y > kTolLoose ? 1u : 0u) << (j - 1);
        bnx |= (ft.x != ft.x ? 1u : 0u) << (j - 1);
        bny |= (ft.y != ft.y ? 1u : 0u) << (j - 1);
        tn = t - clamp_step2(div2(ft, dfdt + kEps));
        if (j >= n) break;
        px = fbits(tn.x) == fbits(t.x) ? 1 : (fbits(tn.x) == fbits(h2.x) ? 2 : (fbits(tn.x) == fbits(h3.x) && j >= 3 ? 3 : 0));
        py = fbits(tn.y) == fbits(t.y) ? 1 : (fbits(tn.y) == fbits(h2.y) ? 2 : (fbits(tn.y) == fbits(h3.y) && j >= 3 ? 3 : 0));
        if (px && py) break;
        h3 = h2; h2 = t; t = tn;
    }
    f2 tfin = tn;
    if (WITH_PREV) {
        *t_prev = t;                                     // the loop ran all n iterations: t entered the last one
        if (j < n) {                                     // periodic from iteration j <= n - 1 on: read t_{n-1} off the cycle (bits untouched)
            unsigned b0 = bmx, b1 = bnx, b2 = bmy, b3 = bny;
            t_prev->x = cycle_finish(px, j, n - 1, b0, b1, tn.x, t.x, h2.x);
            t_prev->y = cycle_finish(py, j, n - 1, b2, b3, tn.y, t.y, h2.y);
        }
    }
    if (j < n) {                                         // both periodic at iteration j (periods px, py)
        tfin.x = cycle_finish(px, j, n, bmx, bnx, tn.x, t.x, h2.x);
        tfin.y = cycle_finish(py, j, n, bmy, bny, tn.y, t.y, h2.y);
    }
    mine |= bmx | bmy; nans |= bnx | bny;
    return tfin;
}

// react_ray for two rays, the loose iterate given (t_loose: t after the batch's count of loose iterations; unused for flat surfaces)
template <bool KGT>
__device__ __forceinline__ void react_ray2(const Surf& s, R32& o, R32& d, f2& ra, int forward, f2 t0, f2 t_loose) {
    const i2 alive = ra > 0.f;
    f2 px, py, pz;
    i2 valid;
    if (s.flat) {                                                                  // stop / flat: surfaces.py:409-453
        const f2 t = t0;
        px = o.x + t * d.x; py = o.y + t * d.y; pz = o.z + t * d.z;
        valid = (sqrt2_ieee(px * px + py * py) <= s.r_f32) & alive;
    } else {
        const f2 t1 = t_loose - t0;
        f2 t = t0 + t1;                                                            // surfaces.py:565-569 (not an identity in float32)
        f2 ft, dfdt;
        residual2<true, KGT>(s, o, d, sel(alive, f2s(1.f), f2s(0.f)), t, ft, dfdt);
        t = t - clamp_step2(div2(ft, dfdt + kEps));
        px = o.x + t * d.x; py = o.y + t * d.y; pz = o.z + t * d.z;
        if (s.spheric) valid = ((px * px + py * py) <= s.r2) & (t >= 0.f) & alive;                         // Newton's own mask is discarded (:466)
        else valid = (valid_strict_f<KGT>(s, o.x + d.x * t, o.y + d.y * t) > 0.f) & (abs2(ft) < kTolTight) & alive & (t > 0.f);
    }
    px = sel(valid, px, o.x); py = sel(valid, py, o.y); pz = sel(valid, pz, o.z);
    ra = ra * sel(valid, f2s(1.f), f2s(0.f));
    if (s.refract) {                                                               // surfaces.py:589-679
        f2 nx, ny, nz;
        if (s.flat) { nx = f2s(0.f); ny = f2s(0.f); nz = f2s(-1.f); }
        else if (s.spheric) {
            const float R = 1.f / s.c;
            if (s.c > 0.f) { nx = 2.f * px; ny = 2.f * py; nz = 2.f * pz - 2.f * (s.d + R); }
            else { nx = -2.f * px; ny = -2.f * py; nz = -2.f * pz + 2.f * (s.d + R); }
        } else {
            const f2 v = sel(ra > 0.f, f2s(1.f), f2s(0.f));
            const f2 xv = px * v, yv = py * v;
            const f2 r2 = xv * xv + yv * yv;
            const f2 g = (f2){dsag(s, r2.x), dsag(s, r2.y)};
            nx = (g * 2.f) * xv; ny = (g * 2.f) * yv; nz = f2s(-1.f);
        }
        normalize32(nx, ny, nz);
        if (forward) { nx = -nx; ny = -ny; nz = -nz; }
        const f2 cosi = (d.x * nx + d.y * ny) + d.z * nz;
        const f2 c2 = cosi * cosi;
        const i2 rv = (c2 > 0.1f) & ((s.eta2 * (1.f - c2)) < 1.f) & (ra > 0.f);
        const f2 sr = sqrt2(1.f - (s.eta2 * (1.f - c2)) * sel(rv, f2s(1.f), f2s(0.f)));
        const f2 ndx = sr * nx + s.eta * (d.x - cosi * nx);
        const f2 ndy = sr * ny + s.eta * (d.y - cosi * ny);
        const f2 ndz = sr * nz + s.eta * (d.z - cosi * nz);
        d.x = sel(rv, ndx, d.x); d.y = sel(rv, ndy, d.y); d.z = sel(rv, ndz, d.z);
        ra = ra * sel(rv, f2s(1.f), f2s(0.f));
    }
    o.x = px; o.y = py; o.z = pz;
}

}  // namespace strict
}  // namespace aadff
