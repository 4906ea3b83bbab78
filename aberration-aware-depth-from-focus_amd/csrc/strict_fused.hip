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
#include <atomic>
#include "strict_math2.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace aadff {
namespace strict {

typedef const __attribute__((address_space(4))) aadff_surface_t* csurf_t;      // wave-uniform reads -> scalar loads
// (the predicted counts are a per-launch block: kernels copy their row into LDS with coherent loads - common.h: fresh - and hand that in)

__device__ __forceinline__ Surf load_surf(csurf_t h, int forward) { return make_surf_from(h, forward); }

// Two rays (the halves of a lane's float2 values, csrc/strict_math2.h) through surfaces [first, last) in travel order with the
// predicted counts pred[surface] (1..10); any-bits / NaN-bits of curved surface i are ORed into sink(i, bits) as (nan << 16 | any).
// ALT (the two-variant chief pass of fused_psf_kernel): at surface s_alt the loop also reports t one iteration earlier; *differs is
// set when that iterate is another float for a live ray of the pair - under a count one lower THIS pair would go on differently.
template <typename Sink, bool ALT = false>
__device__ __forceinline__ void trace_ray_fused2(csurf_t tab, int first, int last, int forward, const int* pred, R32& o, R32& d, f2& ra, Sink sink,
                                                 int s_alt = -1, bool* differs = nullptr) {
    const int nsteps = last - first;
    for (int k = 0; k < nsteps; ++k) {
        const int i = forward ? first + k : last - 1 - k;
        const Surf s = load_surf(tab + i, forward);
        const f2 t0 = div2(s.d - o.z, d.z);
        if (s.flat) {
            react_ray2<true>(s, o, d, ra, forward, t0, t0);
        } else {
            int n = pred[i];
            n = n < 1 ? 1 : (n > kMaxIter ? kMaxIter : n);
            unsigned mine = 0, nans = 0;
            const f2 alive = sel(ra > 0.f, f2s(1.f), f2s(0.f));
            f2 t;
            if (ALT && i == s_alt) {                     // uniform
                f2 tp;
                t = s.k_gt_m1 ? loose_cycle2<true, true>(s, o, d, alive, n, t0, mine, nans, &tp) : loose_cycle2<false, true>(s, o, d, alive, n, t0, mine, nans, &tp);
                *differs = (alive.x > 0.f && fbits(tp.x) != fbits(t.x)) || (alive.y > 0.f && fbits(tp.y) != fbits(t.y));
                sink(i, mine | (nans << 16));
                if (s.k_gt_m1) react_ray2<true>(s, o, d, ra, forward, t0, t);
                else react_ray2<false>(s, o, d, ra, forward, t0, t);
            } else if (s.k_gt_m1) {                      // uniform: the usual case (k > -1) and the rest as two instances of the loop
                t = loose_cycle2<true>(s, o, d, alive, n, t0, mine, nans);
                sink(i, mine | (nans << 16));
                react_ray2<true>(s, o, d, ra, forward, t0, t);
            } else {
                t = loose_cycle2<false>(s, o, d, alive, n, t0, mine, nans);
                sink(i, mine | (nans << 16));
                react_ray2<false>(s, o, d, ra, forward, t0, t);
            }
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

// the two-variant chief pass: bits of surfaces BEHIND s_alt go to `wm` for a pair that goes on differently under the lower count
struct AltSink {
    unsigned* wc;                   // wm = wc + skip (one LDS array: an offset, not a second pointer the compiler would park in scratch)
    int skip;
    const bool* differs;
    int s_alt;
    __device__ __forceinline__ void operator()(int i, unsigned v) const {
        unsigned* w = wc + ((i > s_alt && *differs) ? skip : 0);
        if (v & ~w[i]) atomicOr(&w[i], v);
    }
};

__device__ __forceinline__ void flush_bits2(const unsigned* w0, const unsigned* w1, unsigned* g_any, unsigned* g_nan) {     // the OR of two collections
    for (int i = threadIdx.x; i < AADFF_MAX_SURF; i += blockDim.x) {
        const unsigned v = w0[i] | w1[i];
        if (v & 0xffffu) atomicOr(g_any + i, v & 0xffffu);
        if (v >> 16) atomicOr(g_nan + i, v >> 16);
    }
}

__device__ __forceinline__ void flush_bits(const unsigned* lds_w, unsigned* g_any, unsigned* g_nan) {
    for (int i = threadIdx.x; i < AADFF_MAX_SURF; i += blockDim.x) {
        const unsigned v = lds_w[i];
        if (v & 0xffffu) atomicOr(g_any + i, v & 0xffffu);
        if (v >> 16) atomicOr(g_nan + i, v >> 16);
    }
}

struct __attribute__((packed, aligned(4))) f3p { float x, y, z; };
// a pupil / object point out of a block the host re-uploads to the same address before every launch: coherent loads (common.h: fresh)
__device__ __forceinline__ f3p fresh3(const float* p) { return (f3p){fresh(p), fresh(p + 1), fresh(p + 2)}; }

// ---- flat form: B batches of n rays, two rays per thread (refocus and field-of-view levels; any caller-built batch) --------------
// bits: [B][2][AADFF_MAX_SURF] (any, nan), zeroed by the caller.
template <bool FROM_POINTS>
__global__ __launch_bounds__(256) void fused_flat_kernel(float* o_io, float* d_io, float* ra_io, int n, const int* __restrict__ batch_table,
                                                         const aadff_surface_t* tables, int n_surf, const float* __restrict__ points,
                                                         const int* __restrict__ point_set, const float* __restrict__ pupil, int N, int first, int last,
                                                         int forward, const float* __restrict__ z_sensor, const int* pred, unsigned* bits,
                                                         int origin_at_pupil, int out_mode, float* __restrict__ out0, float* __restrict__ out1,
                                                         const int* __restrict__ pupil_set) {
    __shared__ unsigned w[AADFF_MAX_SURF];
    __shared__ int s_pred[AADFF_MAX_SURF];
    const int b = blockIdx.y;
    const int i0 = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    for (int k = threadIdx.x; k < AADFF_MAX_SURF; k += blockDim.x) {
        w[k] = 0u;
        s_pred[k] = fresh(pred + (size_t)b * AADFF_MAX_SURF + k);      // per-launch parameters through coherent loads (common.h: fresh)
    }
    __syncthreads();
    if (i0 < n) {
        const int i1 = i0 + 1 < n ? i0 + 1 : i0;        // a lone last ray is traced in both halves (same bits), stored once
        const size_t base = (size_t)b * n;
        R32 o, d;
        f2 ra;
        if constexpr (FROM_POINTS) {
            const int sa = i0 / N, pa = i0 - sa * N, sb = i1 / N, pb = i1 - sb * N;
            const float* prow = pupil + (size_t)(pupil_set ? fresh_uniform(pupil_set + b) : b) * (n / N) * 3;
            const float* qrow = points + (size_t)fresh_uniform(point_set + b) * N * 3;
            const float *qa = qrow + (size_t)pa * 3, *qb = qrow + (size_t)pb * 3;
            const f3p poa = fresh3(qa), pob = fresh3(qb);
            const f3p ppa = fresh3(prow + (size_t)sa * 3), ppb = fresh3(prow + (size_t)sb * 3);
            o = {(f2){poa.x, pob.x}, (f2){poa.y, pob.y}, (f2){poa.z, pob.z}};
            const R32 pp = {(f2){ppa.x, ppb.x}, (f2){ppa.y, ppb.y}, (f2){ppa.z, ppb.z}};
            d = {pp.x - o.x, pp.y - o.y, pp.z - o.z};
            normalize32(d.x, d.y, d.z);
            if (origin_at_pupil) o = pp;                                         // refocus: rays leave the aperture points (optics.py:1166-1170)
            ra = f2s(1.f);
        } else {
            const f3p oa = *reinterpret_cast<const f3p*>(o_io + (base + i0) * 3), ob = *reinterpret_cast<const f3p*>(o_io + (base + i1) * 3);
            const f3p da = *reinterpret_cast<const f3p*>(d_io + (base + i0) * 3), db = *reinterpret_cast<const f3p*>(d_io + (base + i1) * 3);
            o = {(f2){oa.x, ob.x}, (f2){oa.y, ob.y}, (f2){oa.z, ob.z}};
            d = {(f2){da.x, db.x}, (f2){da.y, db.y}, (f2){da.z, db.z}};
            ra = (f2){ra_io[base + i0], ra_io[base + i1]};
        }
        const csurf_t tab = (csurf_t)(tables + (size_t)fresh_uniform(batch_table + b) * n_surf);
        trace_ray_fused2(tab, first, last, forward, s_pred, o, d, ra, LdsSink{w});
        if (z_sensor) {                                                          // Ray.propagate_to, basics.py:255-273
            const f2 t = div2(fresh_uniform(z_sensor + b) - o.z, d.z);
            o.x = o.x + d.x * t; o.y = o.y + d.y * t; o.z = o.z + d.z * t;
        }
        f2 v0, v1 = ra;
        // The two epilogue quotients use the compiler's full IEEE division (with v_div_scale_f32), not div2: their operands are
        // NOT confined to the range div2 is exact on - dx^2 + dy^2 of a nearly axial ray can be tiny or denormal (its reciprocal
        // overflows), dx can be below 2^-104.  One division per ray: no cost beside the trace.
        if (out_mode == 1) {                                                     // refocus: where the ray crosses the axis (optics.py:1171-1174)
            const f2 num = d.x * o.x + d.y * o.y, den = d.x * d.x + d.y * d.y;
            f2 tt = (f2){num.x / den.x, num.y / den.y};
            tt = tt * ra;
            v0 = o.z - d.z * tt;
        } else if (out_mode == 2) {                                              // calc_fov: tan of the ray's angle (optics.py:1205)
            v0 = (f2){d.x.x / d.z.x, d.x.y / d.z.y};
        }
        if (out_mode != 0) {
            out0[base + i0] = v0.x; out1[base + i0] = v1.x;
            if (i1 != i0) { out0[base + i1] = v0.y; out1[base + i1] = v1.y; }
        } else {
            *reinterpret_cast<f3p*>(o_io + (base + i0) * 3) = (f3p){o.x.x, o.y.x, o.z.x};
            *reinterpret_cast<f3p*>(d_io + (base + i0) * 3) = (f3p){d.x.x, d.y.x, d.z.x};
            ra_io[base + i0] = ra.x;
            if (i1 != i0) {
                *reinterpret_cast<f3p*>(o_io + (base + i1) * 3) = (f3p){o.x.y, o.y.y, o.z.y};
                *reinterpret_cast<f3p*>(d_io + (base + i1) * 3) = (f3p){d.x.y, d.y.y, d.z.y};
                ra_io[base + i1] = ra.y;
            }
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
    // two-variant jobs (aadff_strict_psf_points_alt): alt[j] = surface | n_lo << 8 of the chief row's one undecided count (the row holds
    // n_lo + 1 there), or -1; the second variant's results are indexed by the JOB
    const int* alt;                 // [J] or NULL
    float* psf_alt;                 // [J] x the per-batch layout of psf
    float* centre_alt;              // [J][N][2]
    unsigned* bits_alt;             // [J][2: any, nan][AADFF_MAX_SURF] chief bits of the lower-count variant
    int* any_valid_alt;             // [J]: 1 / 0 as any_valid for that variant, -1: its list overflowed (variant not available)
    int N, n_surf, spp, spp_c, ks, map_grid;
    float lo, hi, lim, den_row, den_col;   // histogram geometry (float64 -> fp32 once, host)
};


// kPsfThreads = 256: 6 workgroups per CU, what a whole level needs; 1024: a replay of a few batches is a few hundred workgroups on
// 256 CUs - one per CU, each a serial chain of 16 rays per thread - so they get four times the threads instead.
//
// Two-variant jobs.  The batch-wide count of the chief rays at ONE aspheric surface flips between n and n + 1 from draw to draw for a
// few batches (whether the slowest of 250 k rays is above 5e-5 after n iterations); mispredicting it cost a re-launch and a host round
// trip on every stack.  Such a job traces the chief rays under n + 1 - the any-bits of that run tell the host which count was the true
// one - and notes the pairs whose iterate after n iterations is another float (rays walking a 2- or 3-cycle, or still converging - up to
// a third of an off-axis point's rays at the 4 / 5 flip of rf50mm: the rest sit on their fixed point and go on identically either way).
// Those are re-traced under n, their rows
// patched, the centre summed again: centre_alt; the main rays - traced once - are binned against both centres: psf_alt.  The host
// takes the variant the bits confirm.  Any-bits behind the undecided surface are kept apart for the noted pairs, so that each variant
// reports exactly the bits a launch under its row would have reported.
template <int kPsfThreads>
__global__ __launch_bounds__(kPsfThreads) void fused_psf_kernel(PsfArgs a) {
    extern __shared__ float lds[];                       // rows [3][spp_c] | block sums [2][stride] | hist [ks*ks] | hist of the second variant [ks*ks] | noted samples [spp_c] u16
    __shared__ unsigned w[4][AADFF_MAX_SURF];            // chief (all / common), main, chief: noted pairs under n + 1, noted pairs under n
    __shared__ int s_pred[3][AADFF_MAX_SURF];            // chief, main, chief with the lower count
    __shared__ float red[kPsfThreads / 64], cxy[2], cxy_lo[2];
    __shared__ int s_valid;                              // bit 0: a valid chief ray (n + 1 variant), 1: one among the pairs not noted, 2: one among the re-traced
    __shared__ unsigned alt_cnt;
    const int pt = blockIdx.x, job = blockIdx.y, b = a.job_batch ? fresh_uniform(a.job_batch + job) : job, tid = threadIdx.x;
    const int spp_c = a.spp_c, N = a.N, ks = a.ks, kk = ks * ks;
    const int altw = a.alt ? fresh_uniform(a.alt + job) : -1;
    const bool two = altw >= 0;                          // uniform
    const int s_alt = two ? (altw & 0xff) : -1, n_lo = two ? (altw >> 8) : 0;
    float* rows = lds;
    // block-sum scratch of the cascade: at most 4 * (spp_c / 4 / 16 + 1) <= spp_c / 16 + 4 words per component
    const int bs_stride = spp_c / 16 + 8;
    float* bsum = rows + 3 * (size_t)spp_c;
    float* hist = bsum + 2 * bs_stride;
    float* hist2 = hist + kk;
    unsigned short* alt_list = reinterpret_cast<unsigned short*>(hist2 + kk);      // every chief sample fits: no overflow (spp_c <= 65535 checked by the entry)
    const unsigned kAltCap = (unsigned)spp_c;
    for (int k = tid; k < 2 * AADFF_MAX_SURF; k += kPsfThreads) {
        const int v = fresh(a.pred + (size_t)job * 2 * AADFF_MAX_SURF + k);
        (&s_pred[0][0])[k] = v;
        if (k < AADFF_MAX_SURF) s_pred[2][k] = k == s_alt ? n_lo : v;
    }
    for (int k = tid; k < 4 * AADFF_MAX_SURF; k += kPsfThreads) (&w[0][0])[k] = 0u;
    for (int e = tid; e < 2 * kk; e += kPsfThreads) hist[e] = 0.f;
    if (tid == 0) { s_valid = 0; alt_cnt = 0u; }
    __syncthreads();

    // per-launch parameters through coherent loads (common.h: fresh): the batch of this job, its object point and sensor plane, and both
    // count rows into LDS (the trace reads its count per surface from there)
    const float* pp0 = a.points + ((size_t)fresh_uniform(a.point_set + b) * N + pt) * 3;
    const struct { float x, y, z; } po = {fresh_uniform(pp0), fresh_uniform(pp0 + 1), fresh_uniform(pp0 + 2)};
    const float zs = fresh_uniform(a.z_sensor + b);
    const csurf_t tab_c = (csurf_t)(a.tables + (size_t)fresh_uniform(a.table_chief + b) * a.n_surf);

    // one pair of chief rays (samples smp, smp2) to its weighted sensor hits
    auto chief_pair = [&](int smp, int smp2, auto sink, auto alt_tag, const int* pred, bool* differs, f2& wx, f2& wy, f2& ra) {
        const f3p pa = fresh3(a.pupil_chief + ((size_t)b * spp_c + smp) * 3);
        const f3p pb = fresh3(a.pupil_chief + ((size_t)b * spp_c + smp2) * 3);
        R32 o = {f2s(po.x), f2s(po.y), f2s(po.z)};
        R32 d = {(f2){pa.x, pb.x} - o.x, (f2){pa.y, pb.y} - o.y, (f2){pa.z, pb.z} - o.z};
        normalize32(d.x, d.y, d.z);
        ra = f2s(1.f);
        trace_ray_fused2<decltype(sink), decltype(alt_tag)::value>(tab_c, 0, a.n_surf, 1, pred, o, d, ra, sink, s_alt, differs);
        const f2 t = div2(zs - o.z, d.z);
        o.x = o.x + d.x * t; o.y = o.y + d.y * t;
        wx = o.x * ra; wy = o.y * ra;
    };
    using No = std::false_type;
    using Yes = std::true_type;

    // ---- phase 1: chief rays
    {
        bool valid = false, valid_c = false;
        // two rays per lane: samples smp and smp + kPsfThreads (a lone last sample is traced twice: same bits, stored once)
        for (int smp = tid; smp < spp_c; smp += 2 * kPsfThreads) {
            const int smp2 = smp + kPsfThreads < spp_c ? smp + kPsfThreads : smp;
            f2 wx, wy, ra;
            bool differs = false;
            if (two) chief_pair(smp, smp2, AltSink{w[0], 2 * AADFF_MAX_SURF, &differs, s_alt}, Yes{}, s_pred[0], &differs, wx, wy, ra);
            else chief_pair(smp, smp2, LdsSink{w[0]}, No{}, s_pred[0], nullptr, wx, wy, ra);
            rows[smp] = wx.x; rows[spp_c + smp] = wy.x; rows[2 * spp_c + smp] = ra.x;
            if (smp2 != smp) { rows[smp2] = wx.y; rows[spp_c + smp2] = wy.y; rows[2 * spp_c + smp2] = ra.y; }
            const bool v = ra.x == 1.f || ra.y == 1.f;
            valid |= v;
            if (differs) {                               // both samples of the pair are noted (its bits were collected as one word)
                const unsigned at = atomicAdd(&alt_cnt, smp2 != smp ? 2u : 1u);
                if (at < kAltCap) alt_list[at] = (unsigned short)smp;
                if (smp2 != smp && at + 1 < kAltCap) alt_list[at + 1] = (unsigned short)smp2;
            } else {
                valid_c |= v;
            }
        }
        if (__any(valid) && (tid & 63) == 0) atomicOr(&s_valid, 1);
        if (two && __any(valid_c) && (tid & 63) == 0) atomicOr(&s_valid, 2);
    }
    __syncthreads();
    // ---- centre in ATen's summation order (the program of centroid_kernel, csrc/strict.hip, on the rows in LDS); holds barriers
    auto centroid = [&](float* c_lds, float* c_out) {
        const int vec_cols = (N * 3 / 32) * 32;
        // the cascade's geometry of component k (x or y column of this point): scalars, not arrays - k is the thread index further down
        auto geom = [&](int k, int& nsub, int& nb, int& rem, int& step, int& lp) {
            nsub = pt * 3 + k >= vec_cols ? 4 : 1;
            const int sub = spp_c / nsub;
            lp = max(4, ceil_log2(sub) / 4);
            step = 1 << lp;
            nb = sub / step;
            rem = sub - nb * step;
        };
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int nsub, nb, rem, step, lp;
            geom(k, nsub, nb, rem, step, lp);
            const float* v = rows + (size_t)k * spp_c;
            const int total = nsub * (nb + 1);
            for (int id = tid; id < total; id += kPsfThreads) {
                const int q = id / (nb + 1), kb = id - q * (nb + 1);
                const int first = q + nsub * kb * step, count = kb < nb ? step : rem;
                float acc = 0.f;
                for (int j = 0; j < count; ++j) acc = acc + v[first + nsub * j];
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
            int nsub, nb, rem, step, lp;
            geom(k, nsub, nb, rem, step, lp);
            const float* v = rows + (size_t)k * spp_c;
            float wt = 0.f;
            for (int i = 0; i < kPsfThreads / 64; ++i) wt += red[i];
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            for (int q = 0; q < nsub; ++q) {
                const float* bs = bsum + k * bs_stride + q * (nb + 1);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;     // the four levels of ATen's cascade
                const int lmask = step - 1;
                int i = 0;
                for (int kb = 0; kb < nb; ++kb) {
                    a0 = a0 + bs[kb];
                    i += step;
                    a1 = a1 + a0; a0 = 0.f;
                    if ((i & (lmask << lp)) == 0) {
                        a2 = a2 + a1; a1 = 0.f;
                        if ((i & (lmask << (2 * lp))) == 0) { a3 = a3 + a2; a2 = 0.f; }
                    }
                }
                if (rem) a0 = a0 + bs[nb];
                a0 = a0 + a1; a0 = a0 + a2; a0 = a0 + a3;
                if (q == 0) p0 = a0; else if (q == 1) p1 = a0; else if (q == 2) p2 = a0; else p3 = a0;
            }
            float total = p0;
            if (nsub == 4) {
                for (int i = 4 * (spp_c / 4); i < spp_c; ++i) p0 = p0 + v[i];
                total = ((p0 + p1) + p2) + p3;
            }
            const float c = -(total / (wt + kEps));
            c_lds[k] = c;
            c_out[k] = c;
        }
        __syncthreads();
    };
    centroid(cxy, a.centre + ((size_t)b * N + pt) * 2);
    if (tid == 0 && (s_valid & 1)) atomicOr(a.any_valid + job, 1);
    // ---- the second variant: the noted pairs again under the lower count, rows patched, centre summed again
    bool lo_ok = false;                                      // uniform
    if (two) {
        const unsigned have = alt_cnt;
        lo_ok = have <= kAltCap;
        if (lo_ok) {
            bool valid_l = false;
            for (int k2 = 2 * tid; k2 < (int)have; k2 += 2 * kPsfThreads) {
                const int smp = alt_list[k2], smp2 = k2 + 1 < (int)have ? alt_list[k2 + 1] : smp;
                f2 wx, wy, ra;
                chief_pair(smp, smp2, LdsSink{w[3]}, No{}, s_pred[2], nullptr, wx, wy, ra);
                rows[smp] = wx.x; rows[spp_c + smp] = wy.x; rows[2 * spp_c + smp] = ra.x;
                if (smp2 != smp) { rows[smp2] = wx.y; rows[spp_c + smp2] = wy.y; rows[2 * spp_c + smp2] = ra.y; }
                valid_l |= ra.x == 1.f || ra.y == 1.f;
            }
            if (__any(valid_l) && (tid & 63) == 0) atomicOr(&s_valid, 4);
            __syncthreads();
            centroid(cxy_lo, a.centre_alt + ((size_t)job * N + pt) * 2);
        }
        if (tid == 0) {                                      // every workgroup of the job adds its share: the sign bit (overflow) wins on the host
            if (!lo_ok) atomicOr(a.any_valid_alt + job, (int)0x80000000);
            else if (s_valid & 6) atomicOr(a.any_valid_alt + job, 1);
            // telemetry in the last (never a surface's) any-word: the longest list among the job's workgroups
            if (a.n_surf < AADFF_MAX_SURF) atomicMax(a.bits_alt + (size_t)job * 2 * AADFF_MAX_SURF + AADFF_MAX_SURF - 1, have);
        }
    }
    // ---- phase 2: main rays -> histogram(s)
    {
        const csurf_t tab = (csurf_t)(a.tables + (size_t)fresh_uniform(a.table_main + b) * a.n_surf);
        const int* pred = s_pred[1];
        const float km1 = (float)(ks - 1);
        auto splat = [&](float* h, float cx, float cy, float ox, float oy, float ra) {
            // forward_integral: flip, centre, window test (monte_carlo.py:24-38); a ray outside deposits zero weights: skipped
            const float X = -ox - cx, Y = -oy - cy;
            if ((fabsf(X) < a.lim) && (fabsf(Y) < a.lim) && (ra > 0.f)) {
                const float rowf = ((Y - a.hi) / a.den_row) * km1, colf = ((X - a.lo) / a.den_col) * km1;   // monte_carlo.py:86-92
                const float fr = floorf(rowf), fc = floorf(colf);
                const float wb = rowf - fr, wr = colf - fc;
                const int r0 = (int)fr, c0 = (int)fc;
                const int r1 = (int)floorf(rowf + 1.f), c1 = (int)floorf(colf + 1.f);
                atomicAdd(&h[r0 * ks + c0], ((1.f - wb) * (1.f - wr)) * ra);
                atomicAdd(&h[r0 * ks + c1], ((1.f - wb) * wr) * ra);
                atomicAdd(&h[r1 * ks + c0], (wb * (1.f - wr)) * ra);
                atomicAdd(&h[(r0 + 1) * ks + (c0 + 1)], (wb * wr) * ra);
            }
        };
        const float cx = cxy[0], cy = cxy[1], cx2 = lo_ok ? cxy_lo[0] : 0.f, cy2 = lo_ok ? cxy_lo[1] : 0.f;
        for (int smp = tid; smp < a.spp; smp += 2 * kPsfThreads) {
            const int smp2 = smp + kPsfThreads < a.spp ? smp + kPsfThreads : smp;
            const f3p pa = fresh3(a.pupil_main + ((size_t)b * a.spp + smp) * 3);
            const f3p pb = fresh3(a.pupil_main + ((size_t)b * a.spp + smp2) * 3);
            R32 o = {f2s(po.x), f2s(po.y), f2s(po.z)};
            R32 d = {(f2){pa.x, pb.x} - o.x, (f2){pa.y, pb.y} - o.y, (f2){pa.z, pb.z} - o.z};
            normalize32(d.x, d.y, d.z);
            f2 ra = f2s(1.f);
            trace_ray_fused2(tab, 0, a.n_surf, 1, pred, o, d, ra, LdsSink{w[1]});
            const f2 t = div2(zs - o.z, d.z);
            o.x = o.x + d.x * t; o.y = o.y + d.y * t;
            splat(hist, cx, cy, o.x.x, o.y.x, ra.x);
            if (smp2 != smp) splat(hist, cx, cy, o.x.y, o.y.y, ra.y);
            if (lo_ok) {
                splat(hist2, cx2, cy2, o.x.x, o.y.x, ra.x);
                if (smp2 != smp) splat(hist2, cx2, cy2, o.x.y, o.y.y, ra.y);
            }
        }
    }
    __syncthreads();
    // ---- normalise (optics.py:978; 0/0 -> NaN like the reference) and write
    auto write_psf = [&](const float* h, float* base) {      // holds a barrier
        float part = 0.f;
        for (int e = tid; e < kk; e += kPsfThreads) part += h[e];
        part = wave_sum(part);
        __syncthreads();                                     // (red is free again)
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        float total = 0.f;
        for (int i = 0; i < kPsfThreads / 64; ++i) total += red[i];
        if (a.map_grid > 0) {
            const int g = a.map_grid, gy = pt / g, gx = pt - gy * g;
            float* dst = base + (size_t)gy * ks * g * ks + (size_t)gx * ks;
            for (int e = tid; e < kk; e += kPsfThreads) {
                const int r = e / ks, c = e - r * ks;
                dst[(size_t)r * g * ks + c] = h[e] / total;
            }
        } else {
            float* dst = base + (size_t)pt * kk;
            for (int e = tid; e < kk; e += kPsfThreads) dst[e] = h[e] / total;
        }
    };
    const size_t per_batch = a.map_grid > 0 ? (size_t)a.map_grid * ks * a.map_grid * ks : (size_t)N * kk;
    write_psf(hist, a.psf + (size_t)b * per_batch);
    if (lo_ok) write_psf(hist2, a.psf_alt + (size_t)job * per_batch);
    unsigned* gb = a.bits + (size_t)job * 4 * AADFF_MAX_SURF;
    flush_bits2(w[0], w[2], gb, gb + AADFF_MAX_SURF);
    flush_bits(w[1], gb + 2 * AADFF_MAX_SURF, gb + 3 * AADFF_MAX_SURF);
    if (two) flush_bits2(w[0], w[3], a.bits_alt + (size_t)job * 2 * AADFF_MAX_SURF, a.bits_alt + ((size_t)job * 2 + 1) * AADFF_MAX_SURF);
}

// ---- edge-exact PSF grid: the deferred border rays of aadff_psf_points_edge in the reference's arithmetic ---------------------------
// The fast kernel (csrc/trace.hip) leaves every live ray whose hit lies within delta of the histogram's window edge
// (deeplens/monte_carlo.py:37) undecided and lists it per (focus state, wavelength) batch as (point << 16 | sample).  This kernel
// re-traces exactly those rays the way fused_psf_kernel's phase 2 traces ALL rays - ray built from the host-exact pupil point and the
// host-exact object point (sample_from_points + Ray.__init__), every surface in the float32 operation order of csrc/strict_math.h
// under the batch's Newton counts (the main row of the psf_map count table), propagate_to - and applies forward_integral's window
// test and bilinear taps to THAT hit, with the fast kernel's chief-ray centre (a centre error shifts the window by ~1e-7 mm:
// tools/edge_sim.py).  Taps are added to the fast kernel's unnormalised histograms; aadff_psf_normalise divides.
// Counts are taken from the table without verification (a few dozen rays per batch cannot confirm a batch-wide count): the
// main-batch rows do not change from draw to draw (tools/strict_count_stability.py: the entries that flip belong to the focus and
// chief batches), and a wrong count moves a hit by a fraction of an ulp.
struct EdgeRetraceArgs {
    const float* points;            // [P][N][3] object points (host arithmetic of optics.py:945-950)
    const int* point_set;           // [B]
    const aadff_surface_t* tables;  // [n_tables][n_surf] (device)
    const int* table_main;          // [B]
    const float* z_sensor;          // [B]
    const float* pupil_main;        // [B][spp][3]
    const int* pred;                // [B][2][AADFF_MAX_SURF]: row [b][1] = main
    const float* centre;            // [B][N][2] (the fast kernel's)
    const unsigned* count;          // [B]
    const unsigned* list;           // [B][cap]
    float* raw;                     // [B][N][ks*ks]
    int* flags;                     // bit 4: a batch's list overflowed its capacity (the caller falls back to the strict psf_map)
    // The edge launch may have run on PROVISIONAL lens states (the fast refocus kernel's, a few ulps from the reference's: it then
    // overlaps the strict refocus / calc_fov round trips instead of waiting for them).  Its centre belongs to that world; the
    // re-traced hit to the exact one.  The centre is moved over: the sensor plane shifted by dd = d_exact - d_prov moves every
    // chief hit by its direction tangent times dd (exactly linear), and the object height scales with tan(hfov) (the image height
    // with it, to first order - the correction is ~5e-6 mm, its own error a few per cent of that: tools/edge_sim.py shows a centre
    // off by 1e-7 mm changes nothing).  NULL: the launch ran on the exact states.
    const aadff_lens_state_t* states_prov;   // [P]
    const float* tan_exact;                  // [P] float(tan(hfov)) of the exact states
    const float* slope;                      // [B][N][2] mean direction tangents of the valid chief rays (aadff_psf_points_edge)
    int N, n_surf, spp, ks, cap;
    float lo, hi, lim, den_row, den_col;
};

struct NullSink {
    __device__ __forceinline__ void operator()(int, unsigned) const {}
};

__global__ __launch_bounds__(256) void edge_retrace_kernel(EdgeRetraceArgs a) {
    __shared__ int s_pred[AADFF_MAX_SURF];
    const int b = blockIdx.y;
    const unsigned have = fresh_uniform(a.count + b);
    if (have > (unsigned)a.cap && blockIdx.x == 0 && threadIdx.x == 0 && a.flags) atomicOr(a.flags, 16);
    const int cnt = (int)(have < (unsigned)a.cap ? have : (unsigned)a.cap);
    if (2 * (int)(blockIdx.x * blockDim.x) >= cnt) return;                       // the whole workgroup: nothing listed for it
    for (int k = threadIdx.x; k < AADFF_MAX_SURF; k += blockDim.x)                // per-launch parameters through coherent loads (common.h: fresh)
        s_pred[k] = fresh(a.pred + ((size_t)b * 2 + 1) * AADFF_MAX_SURF + k);
    __syncthreads();
    const int i0 = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i0 >= cnt) return;
    const int i1 = i0 + 1 < cnt ? i0 + 1 : i0;          // a lone last ray is traced in both halves, splatted once
    const unsigned e0 = a.list[(size_t)b * a.cap + i0], e1 = a.list[(size_t)b * a.cap + i1];
    const int pt0 = (int)(e0 >> 16), pt1 = (int)(e1 >> 16), s0 = (int)(e0 & 0xffffu), s1 = (int)(e1 & 0xffffu);
    const int ps = fresh_uniform(a.point_set + b);
    const float *qa = a.points + ((size_t)ps * a.N + pt0) * 3, *qb = a.points + ((size_t)ps * a.N + pt1) * 3;
    const f3p oa = fresh3(qa), ob = fresh3(qb);
    const f3p pa = fresh3(a.pupil_main + ((size_t)b * a.spp + s0) * 3);
    const f3p pb = fresh3(a.pupil_main + ((size_t)b * a.spp + s1) * 3);
    R32 o = {(f2){oa.x, ob.x}, (f2){oa.y, ob.y}, (f2){oa.z, ob.z}};
    R32 d = {(f2){pa.x, pb.x} - o.x, (f2){pa.y, pb.y} - o.y, (f2){pa.z, pb.z} - o.z};
    normalize32(d.x, d.y, d.z);
    f2 ra = f2s(1.f);
    const csurf_t tab = (csurf_t)(a.tables + (size_t)fresh_uniform(a.table_main + b) * a.n_surf);
    trace_ray_fused2(tab, 0, a.n_surf, 1, s_pred, o, d, ra, NullSink{});
    const float zs = fresh_uniform(a.z_sensor + b);
    const f2 t = div2(zs - o.z, d.z);
    o.x = o.x + d.x * t; o.y = o.y + d.y * t;
    const int ks = a.ks, kk = ks * ks;
    const float km1 = (float)(ks - 1);
    auto splat = [&](float ox, float oy, float w, int pt) {
        float cx = a.centre[((size_t)b * a.N + pt) * 2], cy = a.centre[((size_t)b * a.N + pt) * 2 + 1];
        if (a.states_prov) {
            const float dd = zs - fresh_uniform(&a.states_prov[ps].d_sensor), rt = fresh_uniform(a.tan_exact + ps) / fresh_uniform(&a.states_prov[ps].tan_hfov);
            cx = (cx - a.slope[((size_t)b * a.N + pt) * 2] * dd) * rt;
            cy = (cy - a.slope[((size_t)b * a.N + pt) * 2 + 1] * dd) * rt;
        }
        const float X = -ox - cx, Y = -oy - cy;                                                         // monte_carlo.py:24-38
        if ((fabsf(X) < a.lim) && (fabsf(Y) < a.lim) && (w > 0.f)) {
            float* hist = a.raw + ((size_t)b * a.N + pt) * kk;
            const float rowf = ((Y - a.hi) / a.den_row) * km1, colf = ((X - a.lo) / a.den_col) * km1;   // monte_carlo.py:86-92
            const float fr = floorf(rowf), fc = floorf(colf);
            const float wb = rowf - fr, wr = colf - fc;
            const int r0 = (int)fr, c0 = (int)fc;
            const int r1 = (int)floorf(rowf + 1.f), c1 = (int)floorf(colf + 1.f);
            atomicAdd(&hist[r0 * ks + c0], ((1.f - wb) * (1.f - wr)) * w);
            atomicAdd(&hist[r0 * ks + c1], ((1.f - wb) * wr) * w);
            atomicAdd(&hist[r1 * ks + c0], (wb * (1.f - wr)) * w);
            atomicAdd(&hist[(r0 + 1) * ks + (c0 + 1)], (wb * wr) * w);
        }
    };
    splat(o.x.x, o.y.x, ra.x, pt0);
    if (i1 != i0) splat(o.x.y, o.y.y, ra.y, pt1);
}

// self test of the packed primitives against the compiler's IEEE forms (tests/test_gpu_parity.py)
__global__ void selftest_ops_kernel(const float* __restrict__ num, const float* __restrict__ den, int n, int op, unsigned* __restrict__ mism) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int k = i + 1 < n ? i + 1 : i;
    const f2 a = {num[i], num[k]};
    f2 q;
    float r0, r1;
    if (op == 0) {
        const f2 b = {den[i], den[k]};
        q = div2(a, b);
        r0 = a.x / b.x; r1 = a.y / b.y;
    } else if (op == 2) {                                // n / (d * d) with the reciprocal of d * d grown from the one of d (sag_dsag2)
        const f2 b = {den[i], den[k]};
        const f2 b2 = b * b;
        const f2 rb = recip2(b);
        q = div2_r(a, b2, recip_refine2(b2, rb * rb));
        r0 = a.x / b2.x; r1 = a.y / b2.y;
    } else {
        q = sqrt2(a);
        r0 = sqrtf(a.x); r1 = sqrtf(a.y);
    }
    auto same = [](float u, float v) { return (u != u && v != v) || __float_as_uint(u) == __float_as_uint(v); };
    if (!same(q.x, r0)) { const unsigned c = atomicAdd(mism, 1u); if (c < 8) { mism[1 + 2 * c] = i; mism[2 + 2 * c] = __float_as_uint(q.x); } }
    if (k != i && !same(q.y, r1)) { const unsigned c = atomicAdd(mism, 1u); if (c < 8) { mism[1 + 2 * c] = k; mism[2 + 2 * c] = __float_as_uint(q.y); } }
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
    const dim3 g(((n + 1) / 2 + 255) / 256, B), blk(256);
    if (points_or_null)
        hipLaunchKernelGGL(strict::fused_flat_kernel<true>, g, blk, 0, st, o, d, ra, n, batch_table, tables_dev, n_surf, points_or_null, point_set, pupil, N,
                           first, last, forward, z_sensor_or_null, pred, bits, origin_at_pupil, out_mode, out0, out1, pupil_set_or_null);
    else
        hipLaunchKernelGGL(strict::fused_flat_kernel<false>, g, blk, 0, st, o, d, ra, n, batch_table, tables_dev, n_surf, (const float*)nullptr,
                           (const int*)nullptr, (const float*)nullptr, 1, first, last, forward, z_sensor_or_null, pred, bits, 0, 0, (float*)nullptr, (float*)nullptr, (const int*)nullptr);
    AADFF_CHECK_LAUNCH();
    return 0;
}

// Workgroup size of a psf_map launch with fewer than 1024 workgroups (a re-launch of a few mispredicted batches): 1024 threads
// shorten the latency chain on an idle chip; beside another stack's full launch (StrictPipeline) a 16-wave workgroup waits for a
// whole CU to drain, so the pipeline asks for the 256-thread form (measured: 5.7 -> 4.2 ms per stack at depth 3).
static std::atomic<int> g_replay_threads{1024};
extern "C" int aadff_strict_replay_threads(int threads) {
    AADFF_CHECK_ARG(threads == 256 || threads == 1024, "strict_replay_threads: %d (256 or 1024)", threads);
    g_replay_threads.store(threads, std::memory_order_relaxed);
    return 0;
}

extern "C" int aadff_strict_psf_points_alt(const float* points, int N, int B, const int* job_batch_or_null, const int* point_set, const aadff_surface_t* tables_dev,
                                           int n_tables, int n_surf, const int* table_main, const int* table_chief, const float* z_sensor,
                                           const float* pupil_main, int spp, const float* pupil_chief, int spp_chief, const int* pred, float pixel_size,
                                           int ks, int map_grid, float* psf, float* centre, unsigned* bits, int* any_valid, const int* alt_or_null,
                                           float* psf_alt, float* centre_alt, unsigned* bits_alt, int* any_valid_alt, aadff_stream_t stream) {
    AADFF_CHECK_ARG(points && point_set && tables_dev && table_main && table_chief && z_sensor && pupil_main && pupil_chief && pred && psf && centre &&
                    bits && any_valid, "strict_psf_points: NULL pointer");
    AADFF_CHECK_ARG(!alt_or_null || (psf_alt && centre_alt && bits_alt && any_valid_alt), "strict_psf_points_alt: two-variant jobs need the second set of outputs");
    AADFF_CHECK_ARG(N >= 1 && B >= 1 && B <= 65535 && spp >= 1 && spp_chief >= 1 && spp_chief <= 65536, "strict_psf_points: N=%d B=%d spp=%d spp_chief=%d", N, B,
                    spp, spp_chief);
    AADFF_CHECK_ARG(n_tables >= 1 && n_surf >= 1 && n_surf <= AADFF_MAX_SURF, "strict_psf_points: n_tables=%d n_surf=%d", n_tables, n_surf);
    AADFF_CHECK_ARG(ks >= 1 && ks <= AADFF_MAX_KS && (ks & 1), "strict_psf_points: ks=%d", ks);
    AADFF_CHECK_ARG(map_grid == 0 || map_grid * map_grid == N, "strict_psf_points: map layout needs N = grid^2 (N=%d grid=%d)", N, map_grid);
    AADFF_CHECK_ARG(!alt_or_null || spp_chief <= 65535, "strict_psf_points_alt: two-variant jobs index chief samples with 16 bits (spp_chief=%d)", spp_chief);
    const size_t lds = ((size_t)3 * spp_chief + 2 * (spp_chief / 16 + 8) + 2 * (size_t)ks * ks) * sizeof(float) + (((size_t)spp_chief * 2 + 3) & ~(size_t)3);
    if (lds > 64 * 1024 - 4096) {
        set_error("strict_psf_points: spp_chief=%d with ks=%d needs %zu bytes of LDS", spp_chief, ks, lds);
        return AADFF_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    AADFF_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)B * 4 * AADFF_MAX_SURF * sizeof(unsigned), st));
    AADFF_CHECK_HIP(hipMemsetAsync(any_valid, 0, (size_t)B * sizeof(int), st));
    if (alt_or_null) {
        AADFF_CHECK_HIP(hipMemsetAsync(bits_alt, 0, (size_t)B * 2 * AADFF_MAX_SURF * sizeof(unsigned), st));
        AADFF_CHECK_HIP(hipMemsetAsync(any_valid_alt, 0, (size_t)B * sizeof(int), st));
    }
    strict::PsfArgs a{};
    a.points = points; a.job_batch = job_batch_or_null; a.point_set = point_set; a.tables = tables_dev; a.table_main = table_main; a.table_chief = table_chief; a.z_sensor = z_sensor;
    a.pupil_main = pupil_main; a.pupil_chief = pupil_chief; a.pred = pred; a.psf = psf; a.centre = centre; a.bits = bits; a.any_valid = any_valid;
    a.alt = alt_or_null; a.psf_alt = psf_alt; a.centre_alt = centre_alt; a.bits_alt = bits_alt; a.any_valid_alt = any_valid_alt;
    a.N = N; a.n_surf = n_surf; a.spp = spp; a.spp_c = spp_chief; a.ks = ks; a.map_grid = map_grid;
    const double ps = (double)pixel_size;                                        // monte_carlo.py:24: Python floats, rounded once
    const double lo = (-ks / 2.0 + 0.5) * ps, hi = (ks / 2.0 - 0.5) * ps;
    a.lo = (float)lo; a.hi = (float)hi; a.lim = (float)(hi - 0.01 * ps); a.den_row = (float)(lo - hi); a.den_col = (float)(hi - lo);
    if ((long)N * B >= 1024 || g_replay_threads.load(std::memory_order_relaxed) == 256) hipLaunchKernelGGL(strict::fused_psf_kernel<256>, dim3(N, B), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(strict::fused_psf_kernel<1024>, dim3(N, B), dim3(1024), lds, st, a);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_strict_psf_points(const float* points, int N, int B, const int* job_batch_or_null, const int* point_set, const aadff_surface_t* tables_dev, int n_tables,
                                       int n_surf, const int* table_main, const int* table_chief, const float* z_sensor, const float* pupil_main,
                                       int spp, const float* pupil_chief, int spp_chief, const int* pred, float pixel_size, int ks,
                                       int map_grid, float* psf, float* centre, unsigned* bits, int* any_valid, aadff_stream_t stream) {
    return aadff_strict_psf_points_alt(points, N, B, job_batch_or_null, point_set, tables_dev, n_tables, n_surf, table_main, table_chief, z_sensor, pupil_main, spp,
                                       pupil_chief, spp_chief, pred, pixel_size, ks, map_grid, psf, centre, bits, any_valid, nullptr, nullptr, nullptr, nullptr,
                                       nullptr, stream);
}

extern "C" int aadff_strict_edge_retrace(const float* points, int N, int B, const int* point_set, const aadff_surface_t* tables_dev, int n_tables, int n_surf,
                                         const int* table_main, const float* z_sensor, const float* pupil_main, int spp, const int* pred,
                                         float pixel_size, int ks, const float* centre, const unsigned* edge_count, const unsigned* edge_list,
                                         int edge_cap, float* raw, int* flags_or_null, const aadff_lens_state_t* states_prov_or_null,
                                         const float* tan_exact, const float* slope, aadff_stream_t stream) {
    AADFF_CHECK_ARG(points && point_set && tables_dev && table_main && z_sensor && pupil_main && pred && centre && edge_count && edge_list && raw,
                    "strict_edge_retrace: NULL pointer");
    AADFF_CHECK_ARG(N >= 1 && N <= 65535 && B >= 1 && B <= 65535 && spp >= 1 && spp <= 65536, "strict_edge_retrace: N=%d B=%d spp=%d", N, B, spp);
    AADFF_CHECK_ARG(n_tables >= 1 && n_surf >= 1 && n_surf <= AADFF_MAX_SURF, "strict_edge_retrace: n_tables=%d n_surf=%d", n_tables, n_surf);
    AADFF_CHECK_ARG(ks >= 1 && ks <= AADFF_MAX_KS && (ks & 1), "strict_edge_retrace: ks=%d", ks);
    AADFF_CHECK_ARG(edge_cap >= 1, "strict_edge_retrace: capacity %d", edge_cap);
    AADFF_CHECK_ARG(!states_prov_or_null || (tan_exact && slope), "strict_edge_retrace: provisional states need tan_exact and slope");
    strict::EdgeRetraceArgs a{};
    a.points = points; a.point_set = point_set; a.tables = tables_dev; a.table_main = table_main; a.z_sensor = z_sensor; a.pupil_main = pupil_main;
    a.pred = pred; a.centre = centre; a.count = edge_count; a.list = edge_list; a.raw = raw; a.flags = flags_or_null;
    a.states_prov = states_prov_or_null; a.tan_exact = tan_exact; a.slope = slope;
    a.N = N; a.n_surf = n_surf; a.spp = spp; a.ks = ks; a.cap = edge_cap;
    const double ps = (double)pixel_size;                                        // monte_carlo.py:24: Python floats, rounded once
    const double lo = (-ks / 2.0 + 0.5) * ps, hi = (ks / 2.0 - 0.5) * ps;
    a.lo = (float)lo; a.hi = (float)hi; a.lim = (float)(hi - 0.01 * ps); a.den_row = (float)(lo - hi); a.den_col = (float)(hi - lo);
    hipLaunchKernelGGL(strict::edge_retrace_kernel, dim3((edge_cap + 511) / 512, B), dim3(256), 0, (hipStream_t)stream, a);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_selftest_strict_ops(const float* num, const float* den, int n, int op, unsigned* mismatches, aadff_stream_t stream) {
    AADFF_CHECK_ARG(num && (den || op == 1) && mismatches && n >= 0 && op >= 0 && op <= 2, "selftest_strict_ops: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    AADFF_CHECK_HIP(hipMemsetAsync(mismatches, 0, 17 * sizeof(unsigned), st));
    if (n == 0) return 0;
    hipLaunchKernelGGL(strict::selftest_ops_kernel, dim3(((n + 1) / 2 + 255) / 256), dim3(256), 0, st, num, den, n, op, mismatches);
    AADFF_CHECK_LAUNCH();
    return 0;
}
