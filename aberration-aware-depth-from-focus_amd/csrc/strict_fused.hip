// parity="strict" at production speed (round 5): every level of a strict focal stack in ONE launch, with the batch-wide Newton
// iteration counts of deeplens/surfaces.py:547 (`while (|ft| > 5e-5).any() and it < 10`) SPECULATED and verified after the fact.
//
// The count is a property of (reference call, surface): how many loose iterations the whole batch runs.  csrc/strict.hip finds it
// with a counting pass in front of every surface (one launch pair per surface, the ray state through HBM in between).  It is stable
// from stack to stack (tools/strict_iterations.py), so the caller keeps a table of predicted counts and this file traces a ray
// through ALL surfaces in registers, running exactly the predicted number of iterations at each - in the per-ray arithmetic of
// csrc/strict_math.h, bit for bit - while it ORs together the per-iteration any-bits it sees on the way.  The host then checks, per
// (batch, surface) with prediction n: "bits 0..n-2 set and (bit n-1 clear or n = 10)" <=> the reference's loop ran exactly n
// iterations; a batch that fails is replayed through aadff_trace_rays_strict_batched and its table row corrected
// (aadff/strict_stack.py).  A misprediction at surface i makes everything behind it meaningless, so the check is in surface order.
//
// Second saving, also bit-exact: with o, d and the mask fixed, t_{j+1} = f(t_j) is a deterministic float32 map, so once an iterate
// REPEATS (t_j == t_{j-1}: fixed point; t_j == t_{j-2}: two-cycle) every later iterate and every later any-bit is known without
// evaluating it.  The reference makes all ten iterations at nine of rf50mm's eleven curved surfaces for the full-pupil batches
// (dead rays frozen far from a surface keep |ft| above the tolerance for ever), 96 residual evaluations per ray; a ray's iterate is
// periodic after 3.0 evaluations on average (tools/strict_cycle_stats.py), a wave's slowest after 5.
#include "strict_math.h"

#pragma clang fp contract(off)

namespace aadff {
namespace strict {

typedef const __attribute__((address_space(4))) aadff_surface_t* csurf_t;      // wave-uniform reads -> scalar loads
typedef const __attribute__((address_space(4))) int* cpred_t;

__device__ __forceinline__ unsigned fbits(float x) { return __float_as_uint(x); }

// n iterations of the loose loop (deeplens/surfaces.py:547-563) from t0 for ONE ray: returns t after n iterations, ORs bit j - 1 into
// `mine` / `nans` when |ft| > 5e-5 / ft is NaN in iteration j (1 <= j <= n).  Evaluates only until the iterate repeats with period
// p <= 3 (t_j == t_{j-p}): then t_{m+p} = t_m and iteration m + p + 1 repeats iteration m + 1 for every m >= j - p.  (Period 3 is
// not exotic: about 1 % of the live rays end up walking three neighbouring floats, tools/strict_cycle_stats.py - half the waves
// would otherwise run all ten iterations for one such ray.)
__device__ __forceinline__ float loose_cycle(const Surf& s, R3 o, R3 d, bool alive, int n, float t0, unsigned& mine, unsigned& nans) {
    float t = t0, h2 = t0, h3 = t0, tn = t0;            // t = t_{j-1}, h2 = t_{j-2}, h3 = t_{j-3}; tn = t_j
    unsigned bm = 0, bn = 0;
    int j = 0, p = 0;
    // the loop body is the residual and three compares; what a detected cycle implies is worked out ONCE behind the loop
    while (true) {
        ++j;
        float ft, dfdt;
        residual<false>(s, o, d, alive, t, ft, dfdt);
        bm |= (fabsf(ft) > kTolLoose ? 1u : 0u) << (j - 1);
        bn |= (ft != ft ? 1u : 0u) << (j - 1);
        tn = t - clamp_step(ft / (dfdt + kEps));
        if (j >= n) break;
        // at j = 1 (2) the histories still hold t_0, so a "period 2 (3)" match there is the fixed point and is taken as p = 1
        p = fbits(tn) == fbits(t) ? 1 : (fbits(tn) == fbits(h2) ? 2 : (fbits(tn) == fbits(h3) && j >= 3 ? 3 : 0));
        if (p) break;
        h3 = h2; h2 = t; t = tn;
    }
    float tfin = tn;
    if (p) {
        // bit indices j .. n-1 repeat the window of the last p iterations (bit indices j-p .. j-1)
        const unsigned all = (1u << n) - 1u;
        const unsigned rep = p == 1 ? 0x3ffu : (p == 2 ? 0x155u : 0x249u);
        const unsigned wm = (1u << p) - 1u;
        bm |= ((((bm >> (j - p)) & wm) * rep) << j) & all;
        bn |= ((((bn >> (j - p)) & wm) * rep) << j) & all;
        const int dn = n - j;                            // r = dn mod p without an integer division (dn <= 9)
        const int r = p == 1 ? 0 : (p == 2 ? (dn & 1) : dn - 3 * ((dn * 11) >> 5));      // t_n = t_{j-p+r}: r = 0 -> t_j (= t_{j-p}), then t_{j-p+1}, ...
        tfin = r == 0 ? tn : (p - r == 1 ? t : h2);
    }
    mine |= bm; nans |= bn;
    return tfin;
}

__device__ __forceinline__ Surf load_surf(csurf_t h, int forward) { return make_surf_from(h, forward); }

// One ray through surfaces [first, last) in travel order with the predicted counts pred[surface] (1..10); any-bits / NaN-bits of
// curved surface i are ORed into sink(i, bits) as (nan << 16 | any).
template <typename Sink>
__device__ __forceinline__ void trace_ray_fused(csurf_t tab, int first, int last, int forward, cpred_t pred, R3& o, R3& d, float& ra, Sink sink) {
    const int nsteps = last - first;
    for (int k = 0; k < nsteps; ++k) {
        const int i = forward ? first + k : last - 1 - k;
        const Surf s = load_surf(tab + i, forward);
        if (s.flat) {
            react_ray(s, o, d, ra, forward, 0);
        } else {
            int n = pred[i];
            n = n < 1 ? 1 : (n > kMaxIter ? kMaxIter : n);
            unsigned mine = 0, nans = 0;
            const float t0 = (s.d - o.z) / d.z;
            const float t = loose_cycle(s, o, d, ra > 0.f, n, t0, mine, nans);
            sink(i, mine | (nans << 16));
            react_ray(s, o, d, ra, forward, n, true, t);
        }
    }
}

// workgroup-wide bit collection: [MAX_SURF] words (nan << 16 | any) in LDS.  The OR saturates after the first rays, so a lane only
// touches the word when it has something new.
struct LdsSink {
    unsigned* w;
    __device__ __forceinline__ void operator()(int i, unsigned v) const {
        if (v & ~w[i]) atomicOr(&w[i], v);
    }
};

__device__ __forceinline__ void flush_bits(const unsigned* lds_w, unsigned* g_any, unsigned* g_nan) {
    for (int i = threadIdx.x; i < AADFF_MAX_SURF; i += blockDim.x) {
        const unsigned v = lds_w[i];
        if (v & 0xffffu) atomicOr(g_any + i, v & 0xffffu);
        if (v >> 16) atomicOr(g_nan + i, v >> 16);
    }
}

struct __attribute__((packed, aligned(4))) f3p { float x, y, z; };

// ---- flat form: B batches of n rays, one ray per thread (refocus and field-of-view levels; any caller-built batch) ---------------
// bits: [B][2][AADFF_MAX_SURF] (any, nan), zeroed by the caller.
template <bool FROM_POINTS>
__global__ __launch_bounds__(256) void fused_flat_kernel(float* o_io, float* d_io, float* ra_io, int n, const int* __restrict__ batch_table,
                                                         const aadff_surface_t* tables, int n_surf, const float* __restrict__ points,
                                                         const int* __restrict__ point_set, const float* __restrict__ pupil, int N, int first, int last,
                                                         int forward, const float* __restrict__ z_sensor, const int* pred, unsigned* bits,
                                                         int origin_at_pupil, int out_mode, float* __restrict__ out0, float* __restrict__ out1,
                                                         const int* __restrict__ pupil_set) {
    __shared__ unsigned w[AADFF_MAX_SURF];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    for (int k = threadIdx.x; k < AADFF_MAX_SURF; k += blockDim.x) w[k] = 0u;
    __syncthreads();
    if (i < n) {
        const size_t idx = (size_t)b * n + i;
        R3 o, d;
        float ra;
        if constexpr (FROM_POINTS) {
            const int sample = i / N, pt = i - sample * N;
            const f3p po = *reinterpret_cast<const f3p*>(points + ((size_t)point_set[b] * N + pt) * 3);
            const f3p pp = *reinterpret_cast<const f3p*>(pupil + ((size_t)(pupil_set ? pupil_set[b] : b) * (n / N) + sample) * 3);
            o = {po.x, po.y, po.z};
            d = {pp.x - po.x, pp.y - po.y, pp.z - po.z};
            normalize3(d.x, d.y, d.z);
            if (origin_at_pupil) o = {pp.x, pp.y, pp.z};                         // refocus: rays leave the aperture points (optics.py:1166-1170)
            ra = 1.f;
        } else {
            const f3p a = *reinterpret_cast<const f3p*>(o_io + idx * 3), c = *reinterpret_cast<const f3p*>(d_io + idx * 3);
            o = {a.x, a.y, a.z}; d = {c.x, c.y, c.z};
            ra = ra_io[idx];
        }
        const csurf_t tab = (csurf_t)(tables + (size_t)batch_table[b] * n_surf);
        trace_ray_fused(tab, first, last, forward, (cpred_t)(pred + (size_t)b * AADFF_MAX_SURF), o, d, ra, LdsSink{w});
        if (z_sensor) {                                                          // Ray.propagate_to, basics.py:255-273
            const float t = (z_sensor[b] - o.z) / d.z;
            o.x = o.x + d.x * t; o.y = o.y + d.y * t; o.z = o.z + d.z * t;
        }
        if (out_mode == 1) {                                                     // refocus: where the ray crosses the axis (optics.py:1171-1174)
            float tt = (d.x * o.x + d.y * o.y) / (d.x * d.x + d.y * d.y);
            tt = tt * ra;
            out0[idx] = o.z - d.z * tt;
            out1[idx] = ra;
        } else if (out_mode == 2) {                                              // calc_fov: tan of the ray's angle (optics.py:1205)
            out0[idx] = d.x / d.z;
            out1[idx] = ra;
        } else {
            *reinterpret_cast<f3p*>(o_io + idx * 3) = (f3p){o.x, o.y, o.z};
            *reinterpret_cast<f3p*>(d_io + idx * 3) = (f3p){d.x, d.y, d.z};
            ra_io[idx] = ra;
        }
    }
    __syncthreads();
    flush_bits(w, bits + (size_t)b * 2 * AADFF_MAX_SURF, bits + ((size_t)b * 2 + 1) * AADFF_MAX_SURF);
}

// ---- level 3: psf_map (deeplens/optics.py:888-1026) for B = S x L batches in one launch ----------------------------------------
// Workgroup = (object point pt, batch b).  Phase 1: the spp_c chief rays of the point (shrunk pupil, the batch's chief table) are
// built (sample_from_points + Ray.__init__), traced and propagated to the sensor; their weighted hits stay in LDS and the centre
// -(sum o ra) / (sum ra + 1e-9) is summed in the ORDER of ATen's CPU sum(0) (see centroid_kernel in csrc/strict.hip; the same
// association of float32 additions).  Phase 2: the spp main rays -> bilinear histogram in LDS (forward_integral,
// deeplens/monte_carlo.py:9-121, IEEE divisions) -> normalised (optics.py:978) -> written in the psf_map tiling (optics.py:1025).
// No ray state goes through HBM.
struct PsfArgs {
    const float* points;            // [P][N][3] object points
    const int* job_batch;           // [J] or NULL (job j renders batch j)
    const int* point_set;           // [B]
    const aadff_surface_t* tables;  // [n_tables][n_surf] (device)
    const int* table_main;          // [B]
    const int* table_chief;         // [B]
    const float* z_sensor;          // [B]
    const float* pupil_main;        // [B][spp][3]
    const float* pupil_chief;       // [B][spp_c][3]
    const int* pred;      // [B][2][AADFF_MAX_SURF]: chief, main
    float* psf;                     // map_grid 0: [B][N][ks][ks]; g: [B][g*ks][g*ks]
    float* centre;                  // [B][N][2]
    unsigned* bits;                 // [J][2 phases][2: any, nan][AADFF_MAX_SURF]
    int* any_valid;                 // [J]
    int N, n_surf, spp, spp_c, ks, map_grid;
    float lo, hi, lim, den_row, den_col;   // histogram geometry (float64 -> fp32 once, host)
};

// kPsfThreads = 256: 6 workgroups per CU, what a whole level needs; 1024: a replay of a few batches is a few hundred workgroups on
// 256 CUs - one per CU, each a serial chain of 16 rays per thread - so they get four times the threads instead.
template <int kPsfThreads>
__global__ __launch_bounds__(kPsfThreads) void fused_psf_kernel(PsfArgs a) {
    extern __shared__ float lds[];                       // rows [3][spp_c] | block sums [2][stride] | hist [ks*ks]
    __shared__ unsigned w[2][AADFF_MAX_SURF];
    __shared__ float red[kPsfThreads / 64], cxy[2];
    __shared__ int s_valid;
    const int pt = blockIdx.x, job = blockIdx.y, b = a.job_batch ? a.job_batch[job] : job, tid = threadIdx.x;
    const int spp_c = a.spp_c, N = a.N, ks = a.ks, kk = ks * ks;
    float* rows = lds;
    // block-sum scratch of the cascade: at most 4 * (spp_c / 4 / 16 + 1) <= spp_c / 16 + 4 words per component
    const int bs_stride = spp_c / 16 + 8;
    float* bsum = rows + 3 * (size_t)spp_c;
    float* hist = bsum + 2 * bs_stride;
    for (int k = tid; k < 2 * AADFF_MAX_SURF; k += kPsfThreads) (&w[0][0])[k] = 0u;
    for (int e = tid; e < kk; e += kPsfThreads) hist[e] = 0.f;
    if (tid == 0) s_valid = 0;
    __syncthreads();

    const f3p po = *reinterpret_cast<const f3p*>(a.points + ((size_t)a.point_set[b] * N + pt) * 3);
    const float zs = a.z_sensor[b];

    // ---- phase 1: chief rays
    {
        const csurf_t tab = (csurf_t)(a.tables + (size_t)a.table_chief[b] * a.n_surf);
        const cpred_t pred = (cpred_t)(a.pred + ((size_t)job * 2 + 0) * AADFF_MAX_SURF);
        bool valid = false;
        for (int smp = tid; smp < spp_c; smp += kPsfThreads) {
            const f3p pp = *reinterpret_cast<const f3p*>(a.pupil_chief + ((size_t)b * spp_c + smp) * 3);
            R3 o = {po.x, po.y, po.z}, d = {pp.x - po.x, pp.y - po.y, pp.z - po.z};
            normalize3(d.x, d.y, d.z);
            float ra = 1.f;
            trace_ray_fused(tab, 0, a.n_surf, 1, pred, o, d, ra, LdsSink{w[0]});
            const float t = (zs - o.z) / d.z;
            o.x = o.x + d.x * t; o.y = o.y + d.y * t;
            rows[smp] = o.x * ra; rows[spp_c + smp] = o.y * ra; rows[2 * spp_c + smp] = ra;
            valid |= ra == 1.f;
        }
        if (__any(valid) && (tid & 63) == 0) atomicOr(&s_valid, 1);
    }
    __syncthreads();
    // ---- centre in ATen's summation order (the program of centroid_kernel, csrc/strict.hip, on the rows in LDS)
    {
        const int vec_cols = (N * 3 / 32) * 32;
        int nsub[2], nb[2], rem[2], step[2], lp[2];
        for (int k = 0; k < 2; ++k) {
            nsub[k] = pt * 3 + k >= vec_cols ? 4 : 1;
            const int sub = spp_c / nsub[k];
            lp[k] = max(4, ceil_log2(sub) / 4);
            step[k] = 1 << lp[k];
            nb[k] = sub / step[k];
            rem[k] = sub - nb[k] * step[k];
        }
        for (int k = 0; k < 2; ++k) {
            const float* v = rows + (size_t)k * spp_c;
            const int total = nsub[k] * (nb[k] + 1);
            for (int id = tid; id < total; id += kPsfThreads) {
                const int q = id / (nb[k] + 1), kb = id - q * (nb[k] + 1);
                const int first = q + nsub[k] * kb * step[k], count = kb < nb[k] ? step[k] : rem[k];
                float acc = 0.f;
                for (int j = 0; j < count; ++j) acc = acc + v[first + nsub[k] * j];
                bsum[k * bs_stride + id] = acc;
            }
        }
        float wsum = 0.f;                                  // 0 / 1 weights: exact in any order
        for (int row = tid; row < spp_c; row += kPsfThreads) wsum += rows[2 * spp_c + row];
        wsum = wave_sum(wsum);
        if ((tid & 63) == 0) red[tid >> 6] = wsum;
        __syncthreads();
        if (tid < 2) {
            const int k = tid;
            const float* v = rows + (size_t)k * spp_c;
            float wt = 0.f;
            for (int i = 0; i < kPsfThreads / 64; ++i) wt += red[i];
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < nsub[k]; ++q) {
                const float* bs = bsum + k * bs_stride + q * (nb[k] + 1);
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                const int lmask = step[k] - 1;
                int i = 0;
                for (int kb = 0; kb < nb[k]; ++kb) {
                    acc[0] = acc[0] + bs[kb];
                    i += step[k];
                    for (int j = 1; j < 4; ++j) {
                        acc[j] = acc[j] + acc[j - 1];
                        acc[j - 1] = 0.f;
                        if ((i & (lmask << (j * lp[k]))) != 0) break;
                    }
                }
                if (rem[k]) acc[0] = acc[0] + bs[nb[k]];
                for (int j = 1; j < 4; ++j) acc[0] = acc[0] + acc[j];
                p[q] = acc[0];
            }
            float total = p[0];
            if (nsub[k] == 4) {
                for (int i = 4 * (spp_c / 4); i < spp_c; ++i) p[0] = p[0] + v[i];
                total = ((p[0] + p[1]) + p[2]) + p[3];
            }
            const float c = -(total / (wt + kEps));
            cxy[k] = c;
            a.centre[((size_t)b * N + pt) * 2 + k] = c;
        }
        if (tid == 0 && s_valid) atomicOr(a.any_valid + job, 1);
    }
    __syncthreads();
    // ---- phase 2: main rays -> histogram
    {
        const csurf_t tab = (csurf_t)(a.tables + (size_t)a.table_main[b] * a.n_surf);
        const cpred_t pred = (cpred_t)(a.pred + ((size_t)job * 2 + 1) * AADFF_MAX_SURF);
        const float cx = cxy[0], cy = cxy[1];
        const float km1 = (float)(ks - 1);
        for (int smp = tid; smp < a.spp; smp += kPsfThreads) {
            const f3p pp = *reinterpret_cast<const f3p*>(a.pupil_main + ((size_t)b * a.spp + smp) * 3);
            R3 o = {po.x, po.y, po.z}, d = {pp.x - po.x, pp.y - po.y, pp.z - po.z};
            normalize3(d.x, d.y, d.z);
            float ra = 1.f;
            trace_ray_fused(tab, 0, a.n_surf, 1, pred, o, d, ra, LdsSink{w[1]});
            const float t = (zs - o.z) / d.z;
            o.x = o.x + d.x * t; o.y = o.y + d.y * t;
            // forward_integral: flip, centre, window test (monte_carlo.py:24-38); a ray outside deposits zero weights: skipped
            const float X = -o.x - cx, Y = -o.y - cy;
            if ((fabsf(X) < a.lim) && (fabsf(Y) < a.lim) && (ra > 0.f)) {
                const float rowf = ((Y - a.hi) / a.den_row) * km1, colf = ((X - a.lo) / a.den_col) * km1;   // monte_carlo.py:86-92
                const float fr = floorf(rowf), fc = floorf(colf);
                const float wb = rowf - fr, wr = colf - fc;
                const int r0 = (int)fr, c0 = (int)fc;
                const int r1 = (int)floorf(rowf + 1.f), c1 = (int)floorf(colf + 1.f);
                atomicAdd(&hist[r0 * ks + c0], ((1.f - wb) * (1.f - wr)) * ra);
                atomicAdd(&hist[r0 * ks + c1], ((1.f - wb) * wr) * ra);
                atomicAdd(&hist[r1 * ks + c0], (wb * (1.f - wr)) * ra);
                atomicAdd(&hist[(r0 + 1) * ks + (c0 + 1)], (wb * wr) * ra);
            }
        }
    }
    __syncthreads();
    // ---- normalise (optics.py:978; 0/0 -> NaN like the reference) and write
    float part = 0.f;
    for (int e = tid; e < kk; e += kPsfThreads) part += hist[e];
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    float total = 0.f;
    for (int i = 0; i < kPsfThreads / 64; ++i) total += red[i];
    if (a.map_grid > 0) {
        const int g = a.map_grid, gy = pt / g, gx = pt - gy * g;
        float* dst = a.psf + (size_t)b * g * ks * g * ks + (size_t)gy * ks * g * ks + (size_t)gx * ks;
        for (int e = tid; e < kk; e += kPsfThreads) {
            const int r = e / ks, c = e - r * ks;
            dst[(size_t)r * g * ks + c] = hist[e] / total;
        }
    } else {
        float* dst = a.psf + ((size_t)b * N + pt) * kk;
        for (int e = tid; e < kk; e += kPsfThreads) dst[e] = hist[e] / total;
    }
    unsigned* gb = a.bits + (size_t)job * 4 * AADFF_MAX_SURF;
    flush_bits(w[0], gb, gb + AADFF_MAX_SURF);
    flush_bits(w[1], gb + 2 * AADFF_MAX_SURF, gb + 3 * AADFF_MAX_SURF);
}

}  // namespace strict
}  // namespace aadff

using namespace aadff;

extern "C" int aadff_trace_rays_strict_fused(float* o, float* d, float* ra, int n, int B, const aadff_surface_t* tables_dev, int n_tables, int n_surf,
                                             const int* batch_table, const float* points_or_null, const int* point_set, const float* pupil, int N,
                                             int first, int last, int forward, const float* z_sensor_or_null, const int* pred,
                                             unsigned* bits, int origin_at_pupil, int out_mode, float* out0, float* out1, const int* pupil_set_or_null,
                                             aadff_stream_t stream) {
    AADFF_CHECK_ARG(tables_dev && batch_table && pred && bits, "trace_rays_strict_fused: NULL pointer");
    AADFF_CHECK_ARG(out_mode >= 0 && out_mode <= 2, "trace_rays_strict_fused: out_mode %d", out_mode);
    AADFF_CHECK_ARG(out_mode == 0 ? (o && d && ra) : (out0 && out1 && points_or_null), "trace_rays_strict_fused: output pointers of out_mode %d", out_mode);
    AADFF_CHECK_ARG(!origin_at_pupil || points_or_null, "trace_rays_strict_fused: origin_at_pupil needs points");
    AADFF_CHECK_ARG(n >= 0 && B >= 1 && B <= 65535, "trace_rays_strict_fused: n=%d B=%d", n, B);
    AADFF_CHECK_ARG(n_tables >= 1, "trace_rays_strict_fused: %d tables", n_tables);
    AADFF_CHECK_ARG(first >= 0 && first <= last && last <= n_surf && n_surf <= AADFF_MAX_SURF, "trace_rays_strict_fused: bad range [%d,%d) of %d", first, last, n_surf);
    AADFF_CHECK_ARG(!points_or_null || (point_set && pupil && N >= 1 && n % N == 0), "trace_rays_strict_fused: points need point_set, pupil and n %% N == 0");
    hipStream_t st = (hipStream_t)stream;
    AADFF_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)B * 2 * AADFF_MAX_SURF * sizeof(unsigned), st));
    if (n == 0 || first == last) return 0;
    const dim3 g((n + 255) / 256, B), blk(256);
    if (points_or_null)
        hipLaunchKernelGGL(strict::fused_flat_kernel<true>, g, blk, 0, st, o, d, ra, n, batch_table, tables_dev, n_surf, points_or_null, point_set, pupil, N,
                           first, last, forward, z_sensor_or_null, pred, bits, origin_at_pupil, out_mode, out0, out1, pupil_set_or_null);
    else
        hipLaunchKernelGGL(strict::fused_flat_kernel<false>, g, blk, 0, st, o, d, ra, n, batch_table, tables_dev, n_surf, (const float*)nullptr,
                           (const int*)nullptr, (const float*)nullptr, 1, first, last, forward, z_sensor_or_null, pred, bits, 0, 0, (float*)nullptr, (float*)nullptr, (const int*)nullptr);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_strict_psf_points(const float* points, int N, int B, const int* job_batch_or_null, const int* point_set, const aadff_surface_t* tables_dev, int n_tables,
                                       int n_surf, const int* table_main, const int* table_chief, const float* z_sensor, const float* pupil_main,
                                       int spp, const float* pupil_chief, int spp_chief, const int* pred, float pixel_size, int ks,
                                       int map_grid, float* psf, float* centre, unsigned* bits, int* any_valid, aadff_stream_t stream) {
    AADFF_CHECK_ARG(points && point_set && tables_dev && table_main && table_chief && z_sensor && pupil_main && pupil_chief && pred && psf && centre &&
                    bits && any_valid, "strict_psf_points: NULL pointer");
    AADFF_CHECK_ARG(N >= 1 && B >= 1 && B <= 65535 && spp >= 1 && spp_chief >= 1 && spp_chief <= 65536, "strict_psf_points: N=%d B=%d spp=%d spp_chief=%d", N, B,
                    spp, spp_chief);
    AADFF_CHECK_ARG(n_tables >= 1 && n_surf >= 1 && n_surf <= AADFF_MAX_SURF, "strict_psf_points: n_tables=%d n_surf=%d", n_tables, n_surf);
    AADFF_CHECK_ARG(ks >= 1 && ks <= AADFF_MAX_KS && (ks & 1), "strict_psf_points: ks=%d", ks);
    AADFF_CHECK_ARG(map_grid == 0 || map_grid * map_grid == N, "strict_psf_points: map layout needs N = grid^2 (N=%d grid=%d)", N, map_grid);
    const size_t lds = ((size_t)3 * spp_chief + 2 * (spp_chief / 16 + 8) + (size_t)ks * ks) * sizeof(float);
    if (lds > 64 * 1024 - 1024) {
        set_error("strict_psf_points: spp_chief=%d with ks=%d needs %zu bytes of LDS", spp_chief, ks, lds);
        return AADFF_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    AADFF_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)B * 4 * AADFF_MAX_SURF * sizeof(unsigned), st));
    AADFF_CHECK_HIP(hipMemsetAsync(any_valid, 0, (size_t)B * sizeof(int), st));
    strict::PsfArgs a{};
    a.points = points; a.job_batch = job_batch_or_null; a.point_set = point_set; a.tables = tables_dev; a.table_main = table_main; a.table_chief = table_chief; a.z_sensor = z_sensor;
    a.pupil_main = pupil_main; a.pupil_chief = pupil_chief; a.pred = pred; a.psf = psf; a.centre = centre; a.bits = bits; a.any_valid = any_valid;
    a.N = N; a.n_surf = n_surf; a.spp = spp; a.spp_c = spp_chief; a.ks = ks; a.map_grid = map_grid;
    const double ps = (double)pixel_size;                                        // monte_carlo.py:24: Python floats, rounded once
    const double lo = (-ks / 2.0 + 0.5) * ps, hi = (ks / 2.0 - 0.5) * ps;
    a.lo = (float)lo; a.hi = (float)hi; a.lim = (float)(hi - 0.01 * ps); a.den_row = (float)(lo - hi); a.den_col = (float)(hi - lo);
    if ((long)N * B >= 1024) hipLaunchKernelGGL(strict::fused_psf_kernel<256>, dim3(N, B), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(strict::fused_psf_kernel<1024>, dim3(N, B), dim3(1024), lds, st, a);
    AADFF_CHECK_LAUNCH();
    return 0;
}
