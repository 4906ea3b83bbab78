// Host driver of the two SHORT LEVELS of a strict / edge focal stack (refocus and calc_fov for all slices: deeplens/optics.py:1155-1217)
// and of the edge-exact psf_map level - the per-stack host work of aadff/strict_stack.py between two GPU waits, as one call each.
//
// A strict / edge stack is host-bound: its GPU work is 0.6 ms, the Python between its waits was 1.2 ms (profiles/r06_*_edge_bench.txt).
// Nothing here decides anything new: the same parameter blocks, the same launches (the C entry points of this library), the same
// count check ("bits 0..n-2 set and (bit n-1 clear or n = 10)", csrc/strict_fused.hip) and the same host arithmetic (numpy's mean,
// psf_diff's object points in float32) as the Python form, which stays as the specification, the fallback (any status != 0: a
// batch no candidate row confirms, a NaN residual) and the reference the tests compare this against bit for bit.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "aadff.h"
#include "common.h"

using namespace aadff;

namespace {
constexpr int MS = AADFF_MAX_SURF, kFocusRays = 2048, kMaxIter = 10;

// "the reference's loop ran exactly n iterations at every curved surface the rays cross" for one job; nan_run: a NaN residual in an
// iteration the reference runs (it exits there, deeplens/surfaces.py:555-558)
bool counts_hold(const unsigned* any, const unsigned* nan, const int* pred, const unsigned char* curved, int n_surf, bool* nan_run) {
    bool ok = true;
    for (int i = 0; i < n_surf; ++i) {
        if (!curved[i]) continue;
        const unsigned n = (unsigned)pred[i], m = any[i];
        const unsigned low = (1u << (n - 1)) - 1u;
        if ((m & low) != low || (n < (unsigned)kMaxIter && ((m >> (n - 1)) & 1u))) ok = false;
        if (nan[i] & ((1u << n) - 1u)) *nan_run = true;
    }
    return ok;
}

// first candidate job of every batch whose counts hold; returns 0, 1 (a batch without one) or 2 (NaN in a confirmed run)
int choose_jobs(const aadff_levels_t* p, const int* h_par, int G, int J, const unsigned* bits, int* chosen) {
    for (int b = 0; b < p->S; ++b) chosen[b] = -1;
    bool nan_any = false;
    for (int j = 0; j < J; ++j) {
        const int b = h_par[G + j];
        if (b < 0 || b >= p->S || chosen[b] >= 0) continue;
        bool nan_run = false;
        if (counts_hold(bits + (size_t)j * 2 * MS, bits + ((size_t)j * 2 + 1) * MS, h_par + G + p->jobs_max + (size_t)j * MS, p->curved, p->n_surf, &nan_run)) {
            chosen[b] = j;
            nan_any |= nan_run;
        }
    }
    for (int b = 0; b < p->S; ++b)
        if (chosen[b] < 0) return 1;
    return nan_any ? 2 : 0;
}

int submit(int* h_par, int* d_par, long n_up, int* h_res, int* d_res, long n_down, hipStream_t st, void* event, int rc_launch) {
    if (rc_launch != 0) return rc_launch;
    AADFF_CHECK_HIP(hipMemcpyAsync(h_res, d_res, (size_t)n_down * 4, hipMemcpyDeviceToHost, st));
    if (event) AADFF_CHECK_HIP(hipEventRecord((hipEvent_t)event, st));
    (void)h_par; (void)d_par; (void)n_up;
    return 0;
}
}  // namespace

extern "C" int aadff_levels_focus_submit(const aadff_levels_t* p, const float* u_host, const long* off_focus, float pi_f, float R2, float z_first,
                                         const float* focus, const void* cos_fn, const void* sin_fn, const void* sqrt_fn, int width,
                                         aadff_stream_t stream, void* event_or_null) {
    AADFF_CHECK_ARG(p && u_host && off_focus && focus, "levels_focus_submit: NULL pointer");
    AADFF_CHECK_ARG(p->S >= 1 && p->J1 >= p->S && p->J1 <= p->jobs_max && p->n_surf >= 1 && p->n_surf <= MS, "levels_focus_submit: S=%d J=%d", p->S, p->J1);
    hipStream_t st = (hipStream_t)stream;
    const int S = p->S, J = p->J1, G = S * 3;
    // the aperture points of the first surface in the reference's host arithmetic (surfaces.py:188-199), rays leave them away from (0, 0, focus)
    long off_r[4096];
    AADFF_CHECK_ARG(S <= 4096, "levels_focus_submit: S=%d", S);
    for (int k = 0; k < S; ++k) off_r[k] = off_focus[k] + kFocusRays;
    int rc = aadff_host_pupil_points(u_host, S, off_focus, off_r, kFocusRays, pi_f, R2, z_first, p->h_pupil, cos_fn, sin_fn, sqrt_fn, width);
    if (rc != 0) return rc;
    AADFF_CHECK_HIP(hipMemcpyAsync(p->d_pupil, p->h_pupil, (size_t)S * kFocusRays * 3 * 4, hipMemcpyHostToDevice, st));
    float* t = reinterpret_cast<float*>(p->h_par1);
    for (int k = 0; k < S; ++k) { t[3 * k] = 0.f; t[3 * k + 1] = 0.f; t[3 * k + 2] = focus[k]; }
    const long n_up = G + p->jobs_max + (long)J * MS;
    AADFF_CHECK_HIP(hipMemcpyAsync(p->d_par1, p->h_par1, (size_t)n_up * 4, hipMemcpyHostToDevice, st));
    float* res = reinterpret_cast<float*>(p->d_res1);
    rc = aadff_trace_rays_strict_fused(nullptr, nullptr, nullptr, kFocusRays, J, p->tables_dev, p->n_tables, p->n_surf, p->bt_green,
                                       reinterpret_cast<const float*>(p->d_par1), p->d_par1 + G, p->d_pupil, 1, 0, p->n_surf, 1, nullptr,
                                       p->d_par1 + G + p->jobs_max, reinterpret_cast<unsigned*>(p->d_res1 + 2L * J * kFocusRays), 1, 1, res,
                                       res + (long)J * kFocusRays, p->d_par1 + G, stream);
    return submit(p->h_par1, p->d_par1, n_up, p->h_res1, p->d_res1, 2L * J * kFocusRays + (long)J * 2 * MS, st, event_or_null, rc);
}

extern "C" int aadff_levels_focus_finish(const aadff_levels_t* p, int* chosen, float* d_sensor, float* scratch) {
    AADFF_CHECK_ARG(p && chosen && d_sensor && scratch, "levels_focus_finish: NULL pointer");
    const int S = p->S, J = p->J1, G = S * 3;
    const int status = choose_jobs(p, p->h_par1, G, J, reinterpret_cast<const unsigned*>(p->h_res1 + 2L * J * kFocusRays), chosen);
    if (status != 0) return status;
    const float* out = reinterpret_cast<const float*>(p->h_res1);
    for (int b = 0; b < S; ++b) {                                          // np.mean of the countable crossing distances (optics.py:1175-1178)
        const int j = chosen[b];
        const int rc = aadff_host_masked_mean_f32(out + (long)j * kFocusRays, out + ((long)J + j) * kFocusRays, 1, kFocusRays, scratch, d_sensor + b);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int aadff_levels_fov_submit(const aadff_levels_t* p, const float* d_sensor, float r_last, int forward, aadff_stream_t stream,
                                       void* event_or_null) {
    AADFF_CHECK_ARG(p && d_sensor, "levels_fov_submit: NULL pointer");
    AADFF_CHECK_ARG(p->J2 >= p->S && p->J2 <= p->jobs_max && p->fov_rays >= 1, "levels_fov_submit: J=%d M=%d", p->J2, p->fov_rays);
    hipStream_t st = (hipStream_t)stream;
    const int S = p->S, J = p->J2, M = p->fov_rays, G = S * 3 + M * 3;
    float* o1 = reinterpret_cast<float*>(p->h_par2);                       // the sensor corner of every slice (optics.py:1198): the pupil points sit behind
    for (int k = 0; k < S; ++k) { o1[3 * k] = r_last; o1[3 * k + 1] = 0.f; o1[3 * k + 2] = d_sensor[k]; }
    const long n_up = G + p->jobs_max + (long)J * MS;
    AADFF_CHECK_HIP(hipMemcpyAsync(p->d_par2, p->h_par2, (size_t)n_up * 4, hipMemcpyHostToDevice, st));
    float* res = reinterpret_cast<float*>(p->d_res2);
    const int rc = aadff_trace_rays_strict_fused(nullptr, nullptr, nullptr, M, J, p->tables_dev, p->n_tables, p->n_surf, p->bt_green,
                                                 reinterpret_cast<const float*>(p->d_par2), p->d_par2 + G, reinterpret_cast<const float*>(p->d_par2 + S * 3), 1, 0,
                                                 p->n_surf, forward, nullptr, p->d_par2 + G + p->jobs_max, reinterpret_cast<unsigned*>(p->d_res2 + 2L * J * M), 0, 2,
                                                 res, res + (long)J * M, p->zeros, stream);
    return submit(p->h_par2, p->d_par2, n_up, p->h_res2, p->d_res2, 2L * J * M + (long)J * 2 * MS, st, event_or_null, rc);
}

extern "C" int aadff_levels_fov_finish(const aadff_levels_t* p, int* chosen, float* tan_fov, float* ra) {
    AADFF_CHECK_ARG(p && chosen && tan_fov && ra, "levels_fov_finish: NULL pointer");
    const int S = p->S, J = p->J2, M = p->fov_rays, G = S * 3 + M * 3;
    const int status = choose_jobs(p, p->h_par2, G, J, reinterpret_cast<const unsigned*>(p->h_res2 + 2L * J * M), chosen);
    if (status != 0) return status;
    const float* out = reinterpret_cast<const float*>(p->h_res2);
    for (int b = 0; b < S; ++b) {
        std::memcpy(tan_fov + (long)b * M, out + (long)chosen[b] * M, (size_t)M * 4);
        std::memcpy(ra + (long)b * M, out + ((long)J + chosen[b]) * M, (size_t)M * 4);
    }
    return 0;
}

// ---- edge-exact psf_map level (aadff_psf_points_edge -> aadff_strict_edge_retrace -> aadff_psf_normalise) ---------------------------
extern "C" int aadff_edge_provisional(const aadff_edge_stack_t* e, const float* focus, float* centre, aadff_stream_t stream) {
    AADFF_CHECK_ARG(e && focus && centre, "edge_provisional: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int S = e->S, B = e->S * e->L;
    AADFF_CHECK_HIP(hipMemcpyAsync(e->d_u, e->h_u, (size_t)S * e->per * 4, hipMemcpyHostToDevice, st));
    for (int k = 0; k < S; ++k) e->h_focus[k] = focus[k];
    AADFF_CHECK_HIP(hipMemcpyAsync(e->d_focus, e->h_focus, (size_t)S * 4, hipMemcpyHostToDevice, st));
    AADFF_CHECK_HIP(hipMemsetAsync(e->count + B, 0, 4, st));
    const aadff_surface_t* green = e->tables_dev + (size_t)e->t_green * e->n_surf;
    int rc = aadff_refocus(e->d_focus, S, e->d_u, kFocusRays, e->per, green, e->lc, reinterpret_cast<aadff_lens_state_t*>(e->states_prov), stream);
    if (rc != 0) return rc;
    return aadff_psf_points_edge(e->d_pts, S, e->N, e->L, e->tables_dev, green, e->lc, reinterpret_cast<const aadff_lens_state_t*>(e->states_prov),
                                 e->d_u + e->o_main, e->spp, e->per, e->per_l, e->d_u + e->o_main + 2L * e->spp, kFocusRays, e->per, e->per_l, e->ks, e->delta,
                                 e->raw, centre, e->slope, e->count, e->list, e->cap, reinterpret_cast<int*>(e->count + B), stream);
}

extern "C" int aadff_edge_finish(const aadff_edge_stack_t* e, const float* pts_norm, const double* hfov, const float* d_sensor, float r_last,
                                 float sensor_w, float sensor_h, const float* centre, float* maps, aadff_stream_t stream, void* event_uploaded,
                                 void* event_done) {
    AADFF_CHECK_ARG(e && pts_norm && hfov && d_sensor && centre && maps, "edge_finish: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int S = e->S, L = e->L, N = e->N, B = S * L;
    float* h = reinterpret_cast<float*>(e->h_par3);
    for (int k = 0; k < S; ++k)
        for (int l = 0; l < L; ++l) h[k * L + l] = d_sensor[k];
    // psf_diff's object points (optics.py:945-950): float32 tensor-times-scalar operations, the scalars rounded to float32 once
    float* pobj = h + B;
    for (int k = 0; k < S; ++k) {
        const float th = (float)std::tan(hfov[k]);
        e->h_focus[S + k] = th;
        for (int n = 0; n < N; ++n) {
            const float x = pts_norm[3 * n], y = pts_norm[3 * n + 1], z = pts_norm[3 * n + 2];
            const float scale = ((-z) * th) / r_last;
            float* o = pobj + ((size_t)k * N + n) * 3;
            o[0] = ((x * scale) * sensor_w) / 2.f;
            o[1] = ((y * scale) * sensor_h) / 2.f;
            o[2] = z;
        }
    }
    AADFF_CHECK_HIP(hipMemcpyAsync(e->d_focus + S, e->h_focus + S, (size_t)S * 4, hipMemcpyHostToDevice, st));
    AADFF_CHECK_HIP(hipMemcpyAsync(e->d_par3, e->h_par3, ((size_t)B + (size_t)S * N * 3 + (size_t)B * 2 * MS) * 4, hipMemcpyHostToDevice, st));
    AADFF_CHECK_HIP(hipMemcpyAsync(e->d_pupil_main, e->h_pupil_main, (size_t)e->n_pm * 4, hipMemcpyHostToDevice, st));
    if (event_uploaded) AADFF_CHECK_HIP(hipEventRecord((hipEvent_t)event_uploaded, st));
    const float* par = reinterpret_cast<const float*>(e->d_par3);
    int rc = aadff_strict_edge_retrace(par + B, N, B, e->pset, e->tables_dev, e->n_tables, e->n_surf, e->bt_main, par, e->d_pupil_main, e->spp,
                                       e->d_par3 + B + S * N * 3, e->pixel_size, e->ks, centre, e->count, e->list, e->cap, e->raw,
                                       reinterpret_cast<int*>(e->count + B), reinterpret_cast<const aadff_lens_state_t*>(e->states_prov), e->d_focus + S,
                                       e->slope, stream);
    if (rc != 0) return rc;
    rc = aadff_psf_normalise(e->raw, S, N, L, e->pixel_size, e->ks, 1, maps, stream);
    if (rc != 0) return rc;
    AADFF_CHECK_HIP(hipMemcpyAsync(e->h_back, e->count, ((size_t)B + 1) * 4, hipMemcpyDeviceToHost, st));
    if (event_done) AADFF_CHECK_HIP(hipEventRecord((hipEvent_t)event_done, st));
    return 0;
}
