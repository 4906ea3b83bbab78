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
#include "strict_math.h"

#pragma clang fp contract(off)

namespace aadff {
namespace strict {

__global__ __launch_bounds__(256) void newton_count_kernel(const float* __restrict__ o_in, const float* __restrict__ d_in,
                                                           const float* __restrict__ ra_in, int n, Surf s, unsigned* mask, unsigned* nan_mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned mine = 0, nans = 0;
    if (i < n) {
        const R3 o = {o_in[3 * i], o_in[3 * i + 1], o_in[3 * i + 2]}, d = {d_in[3 * i], d_in[3 * i + 1], d_in[3 * i + 2]};
        count_ray(s, o, d, ra_in[i] > 0.f, mine, nans);
    }
    for (int off = 32; off > 0; off >>= 1) { mine |= __shfl_xor((int)mine, off, 64); nans |= __shfl_xor((int)nans, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (mine) atomicOr(mask, mine);
        if (nans) atomicOr(nan_mask, nans);
    }
}

// n_iter is read from *mask (the counting launch's bits).
__global__ __launch_bounds__(256) void react_kernel(float* o_io, float* d_io, float* ra_io, int n, Surf s, int forward,
                                                    const unsigned* __restrict__ mask, const unsigned* __restrict__ nan_mask, int* nan_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_iter = s.flat ? 0 : iterations_of(*mask);
    // the reference exits on a NaN residual in an iteration it actually runs (surfaces.py:555-558)
    if (i == 0 && !s.flat && (*nan_mask & ((1u << n_iter) - 1u))) atomicOr(nan_flag, 1);
    if (i >= n) return;
    R3 o = {o_io[3 * i], o_io[3 * i + 1], o_io[3 * i + 2]}, d = {d_io[3 * i], d_io[3 * i + 1], d_io[3 * i + 2]};
    float ra = ra_io[i];
    react_ray(s, o, d, ra, forward, n_iter);
    o_io[3 * i] = o.x; o_io[3 * i + 1] = o.y; o_io[3 * i + 2] = o.z;
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


// ------------------------------------------------------------------------------------------------------------------------
// BATCHED form (round 4): B independent Newton batches of n rays each in ONE launch per surface (a focal stack is S refocus
// batches, then S field-of-view batches, then S x 3 x 2 PSF batches of spp x N rays - 720 launches per 10-slice stack as
// single calls).  Batch b: surface table batch_table[b] (its wavelength), sensor plane z_sensor[b], any-bits / NaN-bits words
// masks[b][surface].  Per-ray arithmetic = count_ray / react_ray above, unchanged.  Two fusions keep the ray state's round
// trips through memory at one read + one write per surface: the kernel that applies surface i also runs counting iterations
// of surface i + 1 on the state it still holds in registers, and the first kernel builds the rays (sample_from_points +
// Ray.__init__, deeplens/optics.py:482-491, basics.py:216-244: o2 - o, F.normalize) before it counts.
// Counting in two instalments: the reference's loop stops at the FIRST iteration in which no ray of the batch is above the
// tolerance, so only the any-bits up to that iteration matter.  The fused count runs kFirstIters iterations; for the batches whose
// first kFirstIters bits are all set `batched_count_more_kernel` CONTINUES to ten from where the fused count stopped (the others
// leave at once).  The iterate is handed on through `tbuf` (one float per ray): the fused count leaves t after kFirstIters
// iterations there (and t after kFirstIters - 1 in its second half), the continuation overwrites it with t after ten - and the
// launch that applies the surface takes it from there instead of iterating again whenever the batch's count is 3, 4 or ten (the same function on the same state: the same
// bits).  That matters: the chief-ray batches need 3-5 iterations per surface, but the full-pupil batches of psf_map run all TEN
// at nine of the eleven curved surfaces of rf50mm (rays outside a surface's clear aperture never converge and keep the batch-wide
// loop going, tools/strict_iterations.py) - 11 residual evaluations per ray and surface instead of 25.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kMaxTables = 4, kFirstIters = 4;
struct SurfSet { Surf s[kMaxTables]; };
struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };

__device__ __forceinline__ void publish_bits(unsigned mine, unsigned nans, unsigned* mask, unsigned* nan_mask) {
    __shared__ unsigned sm[2];
    if (threadIdx.x == 0) { sm[0] = 0u; sm[1] = 0u; }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) { mine |= __shfl_xor((int)mine, off, 64); nans |= __shfl_xor((int)nans, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (mine) atomicOr(&sm[0], mine);
        if (nans) atomicOr(&sm[1], nans);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sm[0]) atomicOr(mask, sm[0]);
        if (sm[1]) atomicOr(nan_mask, sm[1]);
    }
}

// First kernel of a batched trace: (FROM_POINTS) ray i = (sample i / N, point i % N) of batch b from object point
// points[point_set[b]][i % N] through pupil point pupil[b][i / N], direction L2-normalised, ra = 1, written to o / d / ra;
// then the counting iterations of the first surface.
template <bool FROM_POINTS>
__global__ __launch_bounds__(256) void batched_begin_kernel(float* o_io, float* d_io, float* ra_io, int n, const int* __restrict__ batch_table,
                                                            SurfSet first, int first_idx, const float* __restrict__ points,
                                                            const int* __restrict__ point_set, const float* __restrict__ pupil, int N,
                                                            unsigned* masks, int B, float* __restrict__ tbuf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const Surf& s = first.s[batch_table[b]];
    unsigned mine = 0, nans = 0;
    if (i < n) {
        const size_t idx = (size_t)b * n + i;
        R3 o, d;
        float ra;
        if constexpr (FROM_POINTS) {
            const int sample = i / N, pt = i - sample * N;
            const f3u po = *reinterpret_cast<const f3u*>(points + ((size_t)point_set[b] * N + pt) * 3);
            const f3u pp = *reinterpret_cast<const f3u*>(pupil + ((size_t)b * (n / N) + sample) * 3);
            o = {po.x, po.y, po.z};
            d = {pp.x - po.x, pp.y - po.y, pp.z - po.z};
            normalize3(d.x, d.y, d.z);
            ra = 1.f;
            *reinterpret_cast<f3u*>(o_io + idx * 3) = (f3u){o.x, o.y, o.z};
            *reinterpret_cast<f3u*>(d_io + idx * 3) = (f3u){d.x, d.y, d.z};
            ra_io[idx] = ra;
        } else {
            const f3u a = *reinterpret_cast<const f3u*>(o_io + idx * 3), c = *reinterpret_cast<const f3u*>(d_io + idx * 3);
            o = {a.x, a.y, a.z}; d = {c.x, c.y, c.z};
            ra = ra_io[idx];
        }
        if (!s.flat) {
            const float t3 = count_ray(s, o, d, ra > 0.f, mine, nans, 0, kFirstIters - 1);
            const float t4 = count_ray(s, o, d, ra > 0.f, mine, nans, kFirstIters - 1, kFirstIters, t3);
            if (tbuf) { tbuf[idx] = t4; tbuf[(size_t)B * n + idx] = t3; }
        }
    }
    publish_bits(mine, nans, masks + (size_t)b * AADFF_MAX_SURF + first_idx, masks + (size_t)(B + b) * AADFF_MAX_SURF + first_idx);
}

// Surface cur_idx for every batch (iteration count from the batch's any-bits), then - state still in registers - either the
// counting iterations of surface nxt_idx or, behind the last surface, Ray.propagate_to(z_sensor[b]) (basics.py:255-273).
__global__ __launch_bounds__(256) void batched_step_kernel(float* o_io, float* d_io, float* ra_io, int n, const int* __restrict__ batch_table,
                                                           SurfSet cur, int cur_idx, SurfSet nxt, int nxt_idx, int forward,
                                                           const float* __restrict__ z_sensor, unsigned* masks, int B, float* __restrict__ tbuf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const int tb = batch_table[b];
    const Surf& s = cur.s[tb];
    const int n_iter = s.flat ? 0 : iterations_of(masks[(size_t)b * AADFF_MAX_SURF + cur_idx]);
    if (i == 0 && !s.flat && (masks[(size_t)(B + b) * AADFF_MAX_SURF + cur_idx] & ((1u << n_iter) - 1u)))
        atomicOr(reinterpret_cast<int*>(masks + (size_t)2 * B * AADFF_MAX_SURF), 1);          // NaN residual in an iteration the reference runs
    unsigned mine = 0, nans = 0;
    if (i < n) {
        const size_t idx = (size_t)b * n + i;
        const f3u a = *reinterpret_cast<const f3u*>(o_io + idx * 3), c = *reinterpret_cast<const f3u*>(d_io + idx * 3);
        R3 o = {a.x, a.y, a.z}, d = {c.x, c.y, c.z};
        float ra = ra_io[idx];
        // the iterate the counting pass left: t after 4 or 10 iterations in the first half of tbuf, after 3 in the second
        const bool have_t = tbuf && !s.flat && (n_iter == kFirstIters || n_iter == kMaxIter || n_iter == kFirstIters - 1);
        react_ray(s, o, d, ra, forward, n_iter, have_t, have_t ? tbuf[(n_iter == kFirstIters - 1 ? (size_t)B * n : (size_t)0) + idx] : 0.f);
        if (nxt_idx < 0 && z_sensor) {
            const float t = (z_sensor[b] - o.z) / d.z;
            o.x = o.x + d.x * t; o.y = o.y + d.y * t; o.z = o.z + d.z * t;
        }
        *reinterpret_cast<f3u*>(o_io + idx * 3) = (f3u){o.x, o.y, o.z};
        *reinterpret_cast<f3u*>(d_io + idx * 3) = (f3u){d.x, d.y, d.z};
        ra_io[idx] = ra;
        if (nxt_idx >= 0 && !nxt.s[tb].flat) {
            const float t3 = count_ray(nxt.s[tb], o, d, ra > 0.f, mine, nans, 0, kFirstIters - 1);
            const float t4 = count_ray(nxt.s[tb], o, d, ra > 0.f, mine, nans, kFirstIters - 1, kFirstIters, t3);
            if (tbuf) { tbuf[idx] = t4; tbuf[(size_t)B * n + idx] = t3; }
        }
    }
    if (nxt_idx >= 0)
        publish_bits(mine, nans, masks + (size_t)b * AADFF_MAX_SURF + nxt_idx, masks + (size_t)(B + b) * AADFF_MAX_SURF + nxt_idx);
}

// Second instalment of the count: batches that were still above the tolerance in all of the first kFirstIters iterations.
__global__ __launch_bounds__(256) void batched_count_more_kernel(const float* __restrict__ o_in, const float* __restrict__ d_in, const float* __restrict__ ra_in,
                                                                 int n, const int* __restrict__ batch_table, SurfSet cur, int cur_idx, unsigned* masks, int B,
                                                                 float* __restrict__ tbuf) {
    // a fixed, small grid walking the batch (grid-stride): the launch mostly finds nothing to do, and 1.7 M workgroups that
    // leave at once cost 60 us per launch in dispatch alone (rocprofv3, round 4)
    const int b = blockIdx.y;
    const Surf& s = cur.s[batch_table[b]];
    constexpr unsigned first = (1u << kFirstIters) - 1u;
    if (s.flat || (masks[(size_t)b * AADFF_MAX_SURF + cur_idx] & first) != first) return;      // block-uniform
    unsigned mine = 0, nans = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t idx = (size_t)b * n + i;
        const f3u a = *reinterpret_cast<const f3u*>(o_in + idx * 3), c = *reinterpret_cast<const f3u*>(d_in + idx * 3);
        if (tbuf) tbuf[idx] = count_ray(s, {a.x, a.y, a.z}, {c.x, c.y, c.z}, ra_in[idx] > 0.f, mine, nans, kFirstIters, kMaxIter, tbuf[idx]);
        else count_ray(s, {a.x, a.y, a.z}, {c.x, c.y, c.z}, ra_in[idx] > 0.f, mine, nans);
    }
    publish_bits(mine, nans, masks + (size_t)b * AADFF_MAX_SURF + cur_idx, masks + (size_t)(B + b) * AADFF_MAX_SURF + cur_idx);
}

// Chief-ray centre of every (batch, point): -(sum_s o_xy ra) / (sum_s ra + 1e-9)  (psf_center, deeplens/optics.py:902-904), with
// the sums in the ORDER OF ATen's CPU `sum(0)` of a contiguous [spp, N, 3] tensor (SumKernel.cpp cascade_sum, outer reduction):
// columns below the last multiple of 32 (of the N*3 columns): rows accumulated in blocks of 2^p (p = max(4, ceil_log2(spp) / 4))
// through four accumulator levels; the remaining columns: four interleaved partial sums (row % 4), each by the same cascade over
// spp / 4 rows, combined ((p0 + p1) + p2) + p3.  Thread-count independent in ATen (its column split is rounded to 128 bytes);
// checked bit for bit against torch on the CPU in tests/test_oracle_golden.py (oracle/aten_sum.py holds the same program).

template <typename F>
__device__ __forceinline__ float cascade_sum(F val, int size, int stride, int first) {        // rows first, first + stride, ... (size of them)
    const int lp = max(4, ceil_log2(size) / 4), step = 1 << lp, lmask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + step <= size;) {
        for (int j = 0; j < step; ++j, ++i) acc[0] = acc[0] + val(first + i * stride);
        for (int j = 1; j < 4; ++j) {
            acc[j] = acc[j] + acc[j - 1];
            acc[j - 1] = 0.f;
            if ((i & (lmask << (j * lp))) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] = acc[0] + val(first + i * stride);
    for (int j = 1; j < 4; ++j) acc[0] = acc[0] + acc[j];
    return acc[0];
}

template <typename F>
__device__ __forceinline__ float aten_column_sum(F val, int size, bool tail_column) {
    if (!tail_column) return cascade_sum(val, size, 1, 0);
    const int n4 = size / 4;
    float p[4];
    for (int k = 0; k < 4; ++k) p[k] = cascade_sum(val, n4, 4, k);
    for (int i = 4 * n4; i < size; ++i) p[0] = p[0] + val(i);
    return ((p[0] + p[1]) + p[2]) + p[3];
}

// One wave per (batch, point).  The cascade is a fixed association of the rows: blocks of 2^p consecutive rows summed from
// zero, block sums folded through the levels.  Lanes sum whole blocks (each in the kernel's own row order), then one lane per
// component folds the block sums in the cascade's order - the same float32 additions, 64 rows in flight instead of one.

__global__ __launch_bounds__(64) void centroid_kernel(const float* __restrict__ o, const float* __restrict__ ra, int spp, int N, float* __restrict__ centre,
                                                      int* __restrict__ any_valid) {
    extern __shared__ float sm[];                        // [2][stride]
    const int pt = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float* ob = o + (size_t)b * spp * N * 3;
    const float* rb = ra + (size_t)b * spp * N;
    const int vec_cols = (N * 3 / 32) * 32;              // columns that go through the vectorised cascade
    auto val = [&](int row, int k) { return ob[((size_t)row * N + pt) * 3 + k] * rb[(size_t)row * N + pt]; };
    int nsub[2], nb[2], rem[2], step[2], lp[2], stride = 0;
    for (int k = 0; k < 2; ++k) {
        nsub[k] = pt * 3 + k >= vec_cols ? 4 : 1;
        const int sub = spp / nsub[k];
        lp[k] = max(4, ceil_log2(sub) / 4);
        step[k] = 1 << lp[k];
        nb[k] = sub / step[k];
        rem[k] = sub - nb[k] * step[k];
        stride = max(stride, nsub[k] * (nb[k] + 1));
    }
    for (int k = 0; k < 2; ++k) {
        const int total = nsub[k] * (nb[k] + 1);
        for (int id = lane; id < total; id += 64) {
            const int q = id / (nb[k] + 1), kb = id - q * (nb[k] + 1);
            const int first = q + nsub[k] * kb * step[k], count = kb < nb[k] ? step[k] : rem[k];
            float acc = 0.f;
            for (int j = 0; j < count; ++j) acc = acc + val(first + nsub[k] * j, k);
            sm[k * stride + id] = acc;
        }
    }
    float w = 0.f;                                       // 0 / 1 weights: exact in any order
    bool valid = false;
    for (int row = lane; row < spp; row += 64) { const float r = rb[(size_t)row * N + pt]; w += r; valid |= r == 1.f; }
    w = wave_sum(w);
    __syncthreads();
    if (lane < 2) {
        const int k = lane;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nsub[k]; ++q) {
            const float* bs = sm + k * stride + q * (nb[k] + 1);
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
            for (int i = 4 * (spp / 4); i < spp; ++i) p[0] = p[0] + val(i, k);
            total = ((p[0] + p[1]) + p[2]) + p[3];
        }
        centre[((size_t)b * N + pt) * 2 + k] = -(total / (w + kEps));
    }
    if (__any(valid) && lane == 0) atomicOr(any_valid + b, 1);
}

}  // namespace strict
}  // namespace aadff

using namespace aadff;

static strict::Surf make_surf(const aadff_surface_t& h, int forward) { return strict::make_surf_from(&h, forward); }

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
        const strict::Surf s = make_surf(surf_host[i], forward);
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

extern "C" int aadff_trace_rays_strict_batched(float* o, float* d, float* ra, int n, int B, const aadff_surface_t* tables_host, int n_tables,
                                               int n_surf, const int* batch_table, const float* points_or_null, const int* point_set,
                                               const float* pupil, int N, int first, int last, int forward, const float* z_sensor_or_null,
                                               unsigned* scratch, float* tbuf_or_null, int* flags_or_null, aadff_stream_t stream) {
    AADFF_CHECK_ARG(o && d && ra && tables_host && batch_table && scratch, "trace_rays_strict_batched: NULL pointer");
    AADFF_CHECK_ARG(n >= 0 && B >= 1 && B <= 65535, "trace_rays_strict_batched: n=%d B=%d", n, B);
    AADFF_CHECK_ARG(n_tables >= 1 && n_tables <= strict::kMaxTables, "trace_rays_strict_batched: %d tables (1..%d)", n_tables, strict::kMaxTables);
    AADFF_CHECK_ARG(first >= 0 && first <= last && last <= n_surf && n_surf <= AADFF_MAX_SURF, "trace_rays_strict_batched: bad range [%d,%d) of %d", first, last, n_surf);
    AADFF_CHECK_ARG(!points_or_null || (point_set && pupil && N >= 1 && n % N == 0), "trace_rays_strict_batched: points need point_set, pupil and n %% N == 0");
    if (n == 0 || first == last) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((n + 255) / 256, B), blk(256);
    const dim3 gm(std::min((n + 255) / 256, std::max(1, 4096 / B)), B);        // count_more: ~4096 workgroups in all
    // scratch: [B][AADFF_MAX_SURF] any-bits, [B][AADFF_MAX_SURF] NaN-bits, one flag word; zeroed here in stream order
    const size_t words = (size_t)2 * B * AADFF_MAX_SURF + 1;
    AADFF_CHECK_HIP(hipMemsetAsync(scratch, 0, words * sizeof(unsigned), st));
    auto set_of = [&](int i) {
        strict::SurfSet ss{};
        for (int t = 0; t < n_tables; ++t) ss.s[t] = make_surf(tables_host[(size_t)t * n_surf + i], forward);
        return ss;
    };
    const int nsteps = last - first;
    auto surf_at = [&](int k) { return forward ? first + k : last - 1 - k; };
    strict::SurfSet cur = set_of(surf_at(0));
    if (points_or_null)
        hipLaunchKernelGGL(strict::batched_begin_kernel<true>, g, blk, 0, st, o, d, ra, n, batch_table, cur, surf_at(0), points_or_null, point_set, pupil, N, scratch, B, tbuf_or_null);
    else
        hipLaunchKernelGGL(strict::batched_begin_kernel<false>, g, blk, 0, st, o, d, ra, n, batch_table, cur, surf_at(0), (const float*)nullptr, (const int*)nullptr,
                           (const float*)nullptr, 1, scratch, B, tbuf_or_null);
    hipLaunchKernelGGL(strict::batched_count_more_kernel, gm, blk, 0, st, o, d, ra, n, batch_table, cur, surf_at(0), scratch, B, tbuf_or_null);
    for (int k = 0; k < nsteps; ++k) {
        const bool has_next = k + 1 < nsteps;
        strict::SurfSet nxt = has_next ? set_of(surf_at(k + 1)) : cur;
        hipLaunchKernelGGL(strict::batched_step_kernel, g, blk, 0, st, o, d, ra, n, batch_table, cur, surf_at(k), nxt, has_next ? surf_at(k + 1) : -1, forward,
                           z_sensor_or_null, scratch, B, tbuf_or_null);
        if (has_next) hipLaunchKernelGGL(strict::batched_count_more_kernel, gm, blk, 0, st, o, d, ra, n, batch_table, nxt, surf_at(k + 1), scratch, B, tbuf_or_null);
        cur = nxt;
    }
    if (flags_or_null)
        AADFF_CHECK_HIP(hipMemcpyAsync(flags_or_null, scratch + (size_t)2 * B * AADFF_MAX_SURF, sizeof(int), hipMemcpyDeviceToDevice, st));
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_strict_centroid(const float* o, const float* ra, int spp, int N, int B, float* centre, int* any_valid, aadff_stream_t stream) {
    AADFF_CHECK_ARG(o && ra && centre && any_valid, "strict_centroid: NULL pointer");
    AADFF_CHECK_ARG(spp >= 1 && N >= 1 && B >= 1 && B <= 65535, "strict_centroid: spp=%d N=%d B=%d", spp, N, B);
    hipStream_t st = (hipStream_t)stream;
    AADFF_CHECK_HIP(hipMemsetAsync(any_valid, 0, (size_t)B * sizeof(int), st));
    AADFF_CHECK_ARG(spp <= 65536, "strict_centroid: spp %d above 65536", spp);
    const int sub = spp / 4 > 0 ? spp / 4 : 1;
    const size_t lds = (size_t)2 * (4 * (sub / 16 + 1) > spp / 16 + 1 ? 4 * (sub / 16 + 1) : spp / 16 + 1) * sizeof(float);
    hipLaunchKernelGGL(strict::centroid_kernel, dim3(N, B), dim3(64), lds, st, o, ra, spp, N, centre, any_valid);
    AADFF_CHECK_LAUNCH();
    return 0;
}
