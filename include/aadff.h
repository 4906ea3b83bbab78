/*
 * aadff.h — C ABI of the MI355X-native focal-stack rendering hot path.
 *
 * The reference (singer-yang/Aberration-Aware-Depth-from-Focus) has no FFI: its hot
 * path is a chain of stock torch ops behind plain Python functions.
 * This header is the boundary a maintainer would bind from those Python functions
 * (ctypes stub: INTEGRATION.md).  Every entry point cites the reference code it
 * replaces (paths relative to the reference repo).
 *
 * Conventions
 *  - all pointers are DEVICE pointers to contiguous fp32 (or the struct types below)
 *    unless the name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *  - inputs are never written, outputs are caller-allocated;
 *  - return 0 on success, a negative AADFF_E* for argument errors, a positive
 *    hipError_t for runtime errors; aadff_last_error() gives the message.
 */
#ifndef AADFF_H_
#define AADFF_H_

#ifdef __cplusplus
extern "C" {
#endif

#define AADFF_ABI_VERSION 9

#define AADFF_EINVAL      (-1)   /* bad shape / size / NULL pointer                  */
#define AADFF_EUNSUPPORTED (-2)  /* parameter outside what the kernels were built for */

#define AADFF_MAX_GRID   64      /* PSF grid per side (reference uses 7..11)          */
#define AADFF_MAX_KS     51      /* PSF kernel size, odd (psf_map default: optics.py:1006) */
#define AADFF_MAX_SURF   32      /* surfaces per lens                                 */
#define AADFF_MAX_AI     8       /* even-asphere coefficients a2..a16                 */

typedef void* aadff_stream_t;

/* Surface kinds = the three branches of Aspheric.ray_reaction, deeplens/surfaces.py:409,456,491 */
enum { AADFF_SURF_STOP = 0, AADFF_SURF_SPHERIC = 1, AADFF_SURF_ASPHERIC = 2 };

/* One surface at ONE wavelength.  Host code fills it (deeplens/optics.py: LensTable);
 * values that the reference forms in float64 Python arithmetic and then feeds to fp32
 * tensor ops are rounded to fp32 exactly once, here.  136 bytes. */
typedef struct aadff_surface {
    float d;            /* vertex z [mm]                            surfaces.py:11-14 */
    float c;            /* curvature 1/roc                          surfaces.py:304   */
    float k;            /* conic                                    surfaces.py:305   */
    float r;            /* semi-diameter                            surfaces.py:16    */
    float r2;           /* (float)(r*r), r*r in double              surfaces.py:466,727 */
    float r2_shape;     /* fp32 (1-1e-9)/c^2/(1+k); +inf if unused  surfaces.py:727,738 */
    float eta_fwd;      /* n1/n2 (float64 -> fp32)                  surfaces.py:400-402 */
    float eta_fwd2;     /* (n1/n2)^2 squared in float64 -> fp32     surfaces.py:658   */
    float eta_bwd;      /* n2/n1                                    surfaces.py:403-405 */
    float eta_bwd2;
    int   kind;         /* AADFF_SURF_*                                               */
    int   n_ai;         /* number of even-asphere coefficients (0..AADFF_MAX_AI)      */
    int   refract_fwd;  /* 0 when kind==STOP and eta_fwd==1 (air-air stop skips it)   surfaces.py:449 */
    int   refract_bwd;
    int   k_gt_m1;      /* k > -1 selects the shape-domain test     surfaces.py:727,738 */
    float ai[AADFF_MAX_AI];   /* a2, a4, ... (coefficient of r^(2(j+1)))       surfaces.py:799 */
    float dai[AADFF_MAX_AI];  /* (j+1) * ai[j] in fp32: derivative coefficients surfaces.py:823 */
    float cos2_min_fwd; /* max(0.1, 1 - 1/eta_fwd^2): the two refraction validity tests of surfaces.py:660-663   */
    float cos2_min_bwd; /* (cos^2 i > 0.1 and eta^2 (1 - cos^2 i) < 1) as ONE threshold on cos^2 i (fused kernels) */
    float newton_step_tol; /* fused kernels: a Newton update shorter than this [mm] ends the loose loop before the strict step
                              (the reference spends one more evaluation confirming |residual| <= 5e-5, surfaces.py:547).
                              Must satisfy kappa * tol^2 <= 4e-6 with kappa the largest |d^2 sag / d r^2| over the aperture, so
                              that the residual the strict step sees stays 3x under its 1e-5 test; <= 1e-2.  0 = never skip. */
} aadff_surface_t;

/* Per-focus-setting lens state; lives on the device so a whole stack is rendered
 * without a host round trip.  Written by aadff_refocus / aadff_post_computation.
 * Mirrors the attributes Lensgroup.refocus mutates (deeplens/optics.py:1155-1187). */
typedef struct aadff_lens_state {
    float d_sensor;     /* sensor z [mm]                                              */
    float hfov;         /* half diagonal field of view [rad]   optics.py:1187-1217    */
    float tan_hfov;     /* (float)tan((double)hfov)            optics.py:1289         */
    float foclen;       /* r_last / tan(hfov)                  optics.py:1097-1102    */
    float fnum;         /* foclen / (2*entrance pupil radius)  optics.py:186-187      */
    int   n_focus_rays; /* rays that contributed to d_sensor (0 => refocus failed; optics.py:1176) */
    int   flags;        /* bit0: NaN seen in a Newton residual (reference would exit(0), surfaces.py:555);
                           bit1: hfov was NaN and replaced by 0.5 (optics.py:1210-1212) */
    int   pad;
} aadff_lens_state_t;

/* Lens-constant sensor/pupil description (host-side scalars, passed by value). */
typedef struct aadff_lens_const {
    int   n_surf;
    float r_last;          /* half sensor diagonal [mm]                   optics.py:2068 */
    float sensor_w, sensor_h;  /* sensor_size[1], sensor_size[0] [mm]     optics.py:169  */
    float pixel_size;      /* sensor_size[0]/H [mm]                       optics.py:175  */
    float enp_z, enp_r;    /* entrance pupil (z, radius)                  optics.py:1320-1403 */
    float enp_r2;          /* (float)(enp_r^2), squared in double         optics.py:481  */
    float enp_r2_shrunk;   /* (float)((0.5*enp_r)^2)                      optics.py:1400,481 */
    float exp_z;           /* exit pupil z                                               */
    float exp_r_shrunk;    /* 0.5 * exit pupil radius (calc_fov)          optics.py:1196 */
    float first_d;         /* first surface vertex z (refocus sampling)   surfaces.py:188-199 */
    float first_r2;        /* (float)(first surface r^2)                  surfaces.py:193 */
} aadff_lens_const_t;

int         aadff_abi_version(void);
const char* aadff_last_error(void);
/* number of CUs etc. of the current device, for host-side launch heuristics */
int         aadff_device_info(int* n_cu, int* lds_bytes, char* arch, int arch_len);

/* ------------------------------------------------------------------ image space */

/* Spatially-varying blur with a g x g PSF grid.  Replaces render_psf_map,
 * deeplens/render_psf.py:31-73.  img [B,C,H,W], psf_map [C,g*ks,g*ks], out [B,C,H,W].
 * Arithmetic contract of the three convolution entries (ks <= 11 on the matrix cores): every fp32 operand is carried as an
 * fp16 hi + lo pair after a power-of-two pre-scale (per image tile / per PSF), products hi*hi + hi*lo + lo*hi accumulate in
 * fp32: >= 21-22 significand bits per operand, <= 2e-6 abs from the reference's fp32 conv2d on every golden case (tested;
 * seven decades of dynamic range inside one tile stay within 1e-6 of the tile maximum).  Non-finite input: an inf / NaN pixel
 * makes the whole tile that staged it (<= 34 x 108 pixels with halo) NaN, where F.conv2d poisons only the ks x ks support;
 * everything the reference poisons is non-finite here too and nothing farther than 128 pixels from the bad pixel is affected
 * (tested).  A PSF patch that is NaN (no ray inside its window, optics.py:978) turns exactly its image patch into NaN, as in
 * the reference. */
int aadff_render_psf_map(const float* img, const float* psf_map, float* out,
                         int B, int C, int H, int W, int grid, int ks, aadff_stream_t stream);

/* Stack-fused form: one image tile staged once for S PSF maps.  Replaces the slice loop
 * of 2_aber_aware_dff_aif.py:104-114 over render_psf_map + torch.stack(dim=2).
 * img [B,C,H,W], psf_maps [S,C,g*ks,g*ks], out [B,C,S,H,W]. */
int aadff_render_psf_map_stack(const float* img, const float* psf_maps, float* out,
                               int B, int C, int S, int H, int W, int grid, int ks,
                               aadff_stream_t stream);
/* The same with the destination of every slice given by strides (elements): the plane of (b, c, s) starts at
 * out + (b*C + c)*stride_bc + s*stride_s.  stride_bc = S*H*W, stride_s = H*W is the contiguous stack above;
 * stride_bc = H*W, stride_s = C*H*W (B = 1) writes the slices as consecutive [C,H,W] units - the layout of a sharded
 * run's all-gather buffer (BASELINE.json config 3), so a rank renders straight into it.  Planes must not
 * overlap.  Same reference counterpart as aadff_render_psf_map_stack. */
int aadff_render_psf_map_stack_strided(const float* img, const float* psf_maps, float* out,
                                       long stride_bc, long stride_s,
                                       int B, int C, int S, int H, int W, int grid, int ks,
                                       aadff_stream_t stream);
/* M1-layered stack (SURVEY.md 8(d) "M1-layered": the depth MAP quantised into L layers, one ray-traced PSF map per (slice, layer)):
 * out[b][c][s][y][x] = render_psf_map(img, psf_maps[s * L + layer_idx[b][y][x]])[b][c][y][x], i.e. the per-pixel selection among the L
 * candidates of slice s, computed in ONE launch from one staged image band per pass and written once (the composition renders
 * S * L full slices and gathers: L x the output bytes plus a pass over them).  Composes render_psf_map, deeplens/render_psf.py:31-73,
 * over the slice loop of 2_aber_aware_dff_aif.py:104-114.  img [B,C,H,W], psf_maps [S*L,C,g*ks,g*ks], layer_idx [B,H,W] uint8 (values
 * < L), out [B,C,S,H,W]; ks 11 (other sizes: AADFF_EUNSUPPORTED, compose).  Arithmetic contract as aadff_render_psf_map. */
int aadff_render_psf_map_stack_layered(const float* img, const float* psf_maps, const unsigned char* layer_idx, float* out, int B, int C, int S,
                                       int L, int H, int W, int grid, int ks, aadff_stream_t stream);

/* Measurement aid (no reference counterpart): arm the NEXT aadff_render_psf_map_stack (slice-batched kernel: ks 11, S >= 3)
 * or aadff_psf_points / aadff_psf_points_staged call made by this host thread so that its kernel is launched with the two
 * HIP events (hipEvent_t, timing enabled) attached to the dispatch (hipExtLaunchKernelGGL): hipEventElapsedTime then gives
 * the kernel's own begin-to-end time - what rocprofv3 reports - instead of the bracket of two stream events, which adds the
 * dispatch gaps (~2-4 us).  NULL, NULL disarms.  Other launches ignore it. */
int aadff_time_next_launch(void* start_event, void* stop_event);

/* One PSF for the whole image.  Replaces render_psf, deeplens/render_psf.py:12-28.
 * psf [C,ks,ks]. */
int aadff_render_psf(const float* img, const float* psf, float* out,
                     int B, int C, int H, int W, int ks, aadff_stream_t stream);

/* Per-pixel PSF gather (no flip, replicate padding, same PSF for every channel).
 * Replaces local_psf_render, deeplens/render_psf.py:76-107.  psf [B,H,W,ks,ks]. */
int aadff_local_psf_render(const float* img, const float* psf, float* out,
                           int B, int C, int H, int W, int ks, aadff_stream_t stream);

/* Thin-lens baseline renderer with the PSF evaluated in the kernel.  Replaces ThinLens.coc + ThinLens.render (4-D
 * branch), deeplens/psfnet.py:503-512,549-570: per pixel the circle of confusion of `depth` for focus distance
 * `foc_dist[b]`, a Gaussian of sigma = coc/2 pixels cut off at radius coc/2, L1-normalised, gathered over the
 * replicate-padded image (deeplens/render_psf.py:76-107).  No [N,H,W,ks,ks] tensor is materialised.
 *   img [B,C,H,W] (C <= 4), depth [B,1,H,W] (mm), foc_dist [B] (mm), out [B,C,H,W]; ks in {3,5,...,13}
 *   negate_or_null: device int, non-zero = negate depth and foc_dist first (the reference's whole-tensor test
 *                   `(depth < 0).any()`, psfnet.py:505, evaluated by the caller on the device: no host sync)
 *   foc_len_over_fnum, foc_len, inv_pixel_size = 1/ps, d_min, d_max: the lens constants of psfnet.py:491-500
 *   (depth is clamped to [d_min, d_max]; coc in pixels is clamped to >= 0.1). */
int aadff_thinlens_render(const float* img, const float* depth, const float* foc_dist, const int* negate_or_null,
                          float* out, int B, int C, int H, int W, int ks, float foc_len_over_fnum, float foc_len,
                          float inv_pixel_size, float d_min, float d_max, aadff_stream_t stream);

/* ------------------------------------------------------------------ ray tracing */

/* Generic trace of n rays through surfaces [first,last) in travel order (reverse when
 * !forward), optionally followed by propagation to z = state->d_sensor.  Replaces
 * Lensgroup.trace / trace2sensor, deeplens/optics.py:598-714 + Ray.propagate_to,
 * deeplens/basics.py:255-273.  o,d [n,3], ra [n]; in and out may alias. */
int aadff_trace_rays(const float* o_in, const float* d_in, const float* ra_in,
                     float* o_out, float* d_out, float* ra_out, int n,
                     const aadff_surface_t* surf, int first, int last, int forward,
                     const aadff_lens_state_t* state_or_null, int* flags_or_null,
                     aadff_stream_t stream);

/* parity="strict": the same trace in the reference's own operation order, one IEEE float32 operation at a time (no fma
 * contraction, IEEE division / sqrt, F.normalize's fused norm, the BATCH-WIDE Newton iteration count of
 * deeplens/surfaces.py:547: all n rays of the call are one batch, as in one reference call), one launch pair per
 * surface; csrc/strict.hip, specification oracle/scalar_trace.py.  In place on o,d [n,3], ra [n] (device).  `surf_host` is
 * the HOST copy of the packed table of ONE wavelength (n_surf records).  propagate != 0 adds Ray.propagate_to(z_sensor).
 * scratch: 2*AADFF_MAX_SURF + 1 device words (zeroed by the call).  flags_or_null (device): set to 1 on a NaN residual
 * in an iteration the reference would have run.  Replaces Aspheric.ray_reaction / Lensgroup.trace,
 * deeplens/surfaces.py:391-830, deeplens/optics.py:598-714, for Lensgroup(parity="strict").  A verification path: what it
 * can and cannot reproduce is in DESIGN.md section 2. */
int aadff_trace_rays_strict(float* o, float* d, float* ra, int n, const aadff_surface_t* surf_host, int first, int last,
                            int forward, int propagate, float z_sensor, unsigned* scratch, int* flags_or_null,
                            aadff_stream_t stream);

/* The same strict trace for B independent Newton batches of n rays each in ONE launch per surface (ABI v6): what a whole focal
 * stack of Lensgroup(parity="strict") runs on - the S refocus traces (deeplens/optics.py:1155-1180), the S field-of-view traces
 * (:1187-1217) and the S x 3 x 2 traces of psf_map (main + chief rays per wavelength, :888-1026) are three calls instead of
 * 72 S.  o, d [B,n,3], ra [B,n] (device), in place.  tables_host: n_tables (<= 4) packed tables of n_surf records (HOST);
 * batch_table [B] (device): which table (wavelength) batch b traces with.  Every batch keeps its own iteration counts
 * (`while (|ft| > 5e-5).any()` is per reference call, deeplens/surfaces.py:547).
 * points_or_null != NULL: the rays are BUILT first, as sample_from_points + Ray.__init__ do (deeplens/optics.py:482-491,
 * deeplens/basics.py:216-244): ray i of batch b = (sample i / N, point i % N): o = points[point_set[b]][i % N] ([P,N,3] object
 * points, device), d = F.normalize(pupil[b][i / N] - o) (pupil [B, n / N, 3], device), ra = 1; o / d / ra are outputs then.
 * z_sensor_or_null [B] (device): Ray.propagate_to(z_sensor[b]) behind the last surface (trace2sensor).
 * scratch: 2*B*AADFF_MAX_SURF + 1 device words (zeroed by the call); flags_or_null as aadff_trace_rays_strict (any batch).
 * tbuf_or_null: 2*B*n floats (device, scratch): the Newton iterates handed from the counting pass of a surface to the launch that
 * applies it, which then does not iterate again when the batch's count is 3, 4 or 10 (same function, same state: same bits); NULL
 * recomputes. */
int aadff_trace_rays_strict_batched(float* o, float* d, float* ra, int n, int B, const aadff_surface_t* tables_host, int n_tables,
                                    int n_surf, const int* batch_table, const float* points_or_null, const int* point_set,
                                    const float* pupil, int N, int first, int last, int forward, const float* z_sensor_or_null,
                                    unsigned* scratch, float* tbuf_or_null, int* flags_or_null, aadff_stream_t stream);

/* The strict trace of B batches in ONE launch for ALL surfaces, with SPECULATED batch-wide Newton counts (ABI v7; csrc/strict_fused.hip).
 * Same rays, same per-ray arithmetic and same result as aadff_trace_rays_strict_batched WHEN pred[b][i] - the number of loose
 * iterations the reference's `while (|ft| > 5e-5).any() and it < 10` loop (deeplens/surfaces.py:547) runs for batch b at surface i -
 * is right for every curved surface the batch crosses.  The call cannot know that; it reports what it saw:
 *   bits [B][2][AADFF_MAX_SURF] (device, zeroed by the call): [b][0][i] bit j = some ray of batch b had |ft| > 5e-5 in loose
 *   iteration j + 1 at surface i (j < pred[b][i]); [b][1][i] the same for a NaN residual (the reference exits, surfaces.py:555-558).
 * The caller checks, in the order the surfaces are crossed: n = pred[b][i] is the reference's count <=> bits 0..n-2 are set and
 * (bit n-1 is clear or n == 10).  At the first surface where that fails the batch's result (and its later bits) are meaningless:
 * replay the batch through aadff_trace_rays_strict_batched, whose any-bits give the true counts (aadff/strict_stack.py keeps the
 * table).  Rays whose Newton iterate becomes periodic (a fixed point or a two-cycle of the float32 map t -> t - clamp(ft / dfdt))
 * are not iterated further: the remaining iterates and any-bits follow from the cycle, bit for bit.
 * tables_dev: n_tables packed tables of n_surf records on the DEVICE; pred [B][AADFF_MAX_SURF] int32 (device), values 1..10 (flat
 * surfaces ignored); other arguments as aadff_trace_rays_strict_batched.  With points: origin_at_pupil != 0 starts the rays AT the
 * pupil points (refocus: rays leave the first surface's aperture away from the axis point, deeplens/optics.py:1166-1170).
 * out_mode 0: o / d / ra in place.  out_mode 1 (needs points; o, d, ra may be NULL): out0[b][i] = the z at which the ray crosses
 * the axis in the least-squares sense, o.z - d.z * ((d.x o.x + d.y o.y) / (d.x^2 + d.y^2)) * ra, out1[b][i] = ra - refocus's per-ray
 * arithmetic (optics.py:1171-1174, element-wise IEEE float32).  out_mode 2: out0 = d.x / d.z (calc_fov, optics.py:1205), out1 = ra.
 * pupil_set_or_null [B]: batch b takes its pupil points from row pupil_set[b] of `pupil` (NULL: row b) - the same rays traced
 * under several candidate count rows are several batches sharing one row.
 * Replaces Lensgroup.trace / trace2sensor,
 * deeplens/optics.py:598-714, for Lensgroup(parity="strict"). */
int aadff_trace_rays_strict_fused(float* o, float* d, float* ra, int n, int B, const aadff_surface_t* tables_dev, int n_tables, int n_surf,
                                  const int* batch_table, const float* points_or_null, const int* point_set, const float* pupil, int N,
                                  int first, int last, int forward, const float* z_sensor_or_null, const int* pred, unsigned* bits,
                                  int origin_at_pupil, int out_mode, float* out0, float* out1, const int* pupil_set_or_null,
                                  aadff_stream_t stream);

/* psf_map of a strict-parity lens for B = S x L (focus state, wavelength) batches in ONE launch (ABI v7): per batch and object point
 * the chief rays (shrunk pupil) -> centre in ATen's summation order (as aadff_strict_centroid), the main rays -> bilinear histogram
 * (forward_integral, deeplens/monte_carlo.py:9-121, IEEE float32 divisions) -> normalised PSF; no ray state goes through memory.
 * Replaces Lensgroup.psf_diff / psf_rgb / psf_map, deeplens/optics.py:888-1026, for Lensgroup(parity="strict"), with the batch-wide
 * Newton counts speculated as in aadff_trace_rays_strict_fused:
 *   points [P][N][3] object points (optics.py:945-950), point_set [B]; tables_dev [n_tables][n_surf] (device); table_main / table_chief
 *   [B]: the table each batch's main / chief rays trace with; z_sensor [B]; pupil_main [B][spp][3], pupil_chief [B][spp_chief][3]:
 *   pupil points (optics.py:480-486, computed by the caller with the reference's own host calls);
 *   pred [B][2][AADFF_MAX_SURF] int32: predicted counts of the chief ([b][0]) and main ([b][1]) batch;
 *   psf: map_grid = 0: [B][N][ks][ks]; map_grid = g (N = g*g): [B][g*ks][g*ks] in the psf_map tiling (optics.py:1025);
 *   centre [B][N][2]; bits [B][2][2][AADFF_MAX_SURF] ([b][phase][any | nan][surface], zeroed by the call); any_valid [B]: 1 where a chief
 *   ray of the batch is valid (the assert of optics.py:901).  All pointers are device pointers.
 * job_batch_or_null [B]: the call renders the B JOBS j = 0..B-1, job j being batch job_batch[j] of the arrays above (NULL: batch j):
 *   inputs, psf and centre are indexed by the batch, pred / bits / any_valid by the job - a caller replays the few batches whose
 *   prediction failed with corrected counts, overwriting exactly their PSFs. */
int aadff_strict_psf_points(const float* points, int N, int B, const int* job_batch_or_null, const int* point_set, const aadff_surface_t* tables_dev, int n_tables,
                            int n_surf, const int* table_main, const int* table_chief, const float* z_sensor, const float* pupil_main,
                            int spp, const float* pupil_chief, int spp_chief, const int* pred, float pixel_size, int ks, int map_grid,
                            float* psf, float* centre, unsigned* bits, int* any_valid, aadff_stream_t stream);

/* aadff_strict_psf_points with TWO-VARIANT jobs (ABI v9): a batch whose chief-ray count at ONE surface is known to flip between n and
 * n + 1 from draw to draw (whether the slowest ray of the batch is still above 5e-5 after n iterations, deeplens/surfaces.py:547) is
 * rendered under both counts in the same launch instead of being re-launched when the guess was wrong.
 *   alt_or_null [B] int32 (device): per job, surface | n << 8 with pred[j][0][surface] = n + 1, or -1 (an ordinary job).
 *   The job's ordinary outputs (psf, centre, bits, any_valid) are those of the row as given (count n + 1); the bits of that run show
 *   which count the reference's loop would have stopped at.  The variant under n: psf_alt [B] x the per-batch layout of psf and
 *   centre_alt [B][N][2], indexed by the JOB; bits_alt [B][2][AADFF_MAX_SURF] (any | nan) the chief bits a launch under that row
 *   would have reported (its last any-word, never a surface's: the largest number of re-traced chief rays among the job's object
 *   points); any_valid_alt [B]: > 0 a chief ray of that variant is valid, 0 none, < 0 the variant is not available.  All zeroed by
 *   the call.  Only the chief rays whose iterate after n iterations differs from the one after n + 1 are traced twice (rf50mm at its
 *   4 / 5 flip: up to a third of an off-axis point's 2048); the main rays are traced once and binned against both centres.
 * alt_or_null = NULL is aadff_strict_psf_points. */
int aadff_strict_psf_points_alt(const float* points, int N, int B, const int* job_batch_or_null, const int* point_set, const aadff_surface_t* tables_dev,
                                int n_tables, int n_surf, const int* table_main, const int* table_chief, const float* z_sensor,
                                const float* pupil_main, int spp, const float* pupil_chief, int spp_chief, const int* pred, float pixel_size,
                                int ks, int map_grid, float* psf, float* centre, unsigned* bits, int* any_valid, const int* alt_or_null,
                                float* psf_alt, float* centre_alt, unsigned* bits_alt, int* any_valid_alt, aadff_stream_t stream);

/* Self test (no reference counterpart) of the packed float32 primitives the two-rays-per-lane strict kernels use (csrc/strict_math2.h)
 * against the compiler's IEEE forms, bit for bit: op 0: num[i] / den[i] (reciprocal refinement with packed FMAs + v_div_fixup_f32,
 * no v_div_scale_f32 pre-scaling); op 1: sqrt(num[i]) (v_sqrt_f32 + two-neighbour correction, no pre-scaling of arguments below
 * 2^-96); op 2: num[i] / (den[i] * den[i]) with the reciprocal of the squared denominator grown from the refined reciprocal of
 * den[i] (how sag / d sag share 1 / (1 + sf), deeplens/surfaces.py:787-830) against the compiler's division by the rounded square.
 * n values (device).  mismatches: 17 device words = count, then (index, bits of the packed result) of the first 8. */
int aadff_selftest_strict_ops(const float* num, const float* den, int n, int op, unsigned* mismatches, aadff_stream_t stream);

/* Chief-ray PSF centres of B batches: centre[b][p] = -(sum_s o_xy[b,s,p] ra[b,s,p]) / (sum_s ra[b,s,p] + 1e-9), o [B,spp,N,3],
 * ra [B,spp,N] (device) -> centre [B,N,2]; any_valid [B] = 1 where some ray of the batch has ra == 1 (the reference asserts it:
 * "No sampled rays is valid.", deeplens/optics.py:901).  The sums run in the ORDER of ATen's CPU `tensor.sum(0)` (cascade
 * summation of an outer reduction), so the centre has the bits the reference's psf_center (deeplens/optics.py:888-913) computes
 * on the host; oracle/aten_sum.py is the specification.  For Lensgroup(parity="strict"). */
int aadff_strict_centroid(const float* o, const float* ra, int spp, int N, int B, float* centre, int* any_valid, aadff_stream_t stream);

/* Rays from object points through the entrance pupil to the sensor.  Replaces
 * sample_from_points + trace2sensor, deeplens/optics.py:457-491,635-661.
 * points_obj [N,3] (object space, mm); u_theta,u_r [spp] raw uniform draws
 * (theta = u*2*pi, r = sqrt(u*R^2)); outputs o,d [spp,N,3], ra [spp,N]. */
int aadff_trace_points(const float* points_obj, int N, const float* u_theta, const float* u_r, int spp,
                       float pupil_z, float pupil_r, const aadff_surface_t* surf, int n_surf,
                       const aadff_lens_state_t* state, float* o_out, float* d_out, float* ra_out,
                       aadff_stream_t stream);

/* Bilinear splat of sensor hits into ks x ks histograms + normalisation.  Replaces
 * forward_integral / assign_points_to_pixels, deeplens/monte_carlo.py:9-121 and the
 * division of deeplens/optics.py:978.  o [spp,N,3], ra [spp,N], centre [N,2];
 * psf_raw_or_null / psf [N,ks,ks]. */
int aadff_psf_splat(const float* o, const float* ra, const float* centre, int spp, int N,
                    float pixel_size, int ks, float* psf_raw_or_null, float* psf,
                    aadff_stream_t stream);

/* Fused PSF of N normalised points for S focus states and L wavelengths in ONE launch:
 * object-space mapping, chief-ray centre pass (shrunk pupil, `surf_chief` table), main
 * pass, splat, normalise.  Replaces Lensgroup.psf_diff / psf_rgb / psf_map,
 * deeplens/optics.py:888-1026.
 *   points      [S,N,3]  normalised (x,y in [-1,1], z = depth mm < 0)
 *   surf_main   [L][n_surf], surf_chief [n_surf]
 *   states      [S]
 *   u_main      raw uniforms: theta row at u_main + s*main_stride_s + l*main_stride_l, r row spp
 *               floats after it (dense [S,L,2,spp] has strides 2*L*spp, 2*spp)
 *   u_chief     same with spp_chief and its own strides (any layout that keeps the host
 *               generator's draw order can be consumed without a re-layout copy)
 *   centre_mode 1: chief-ray centre (center=True); 0: ideal perspective centre (optics.py:970-975)
 *   map_layout  0: psf [S,N,L,ks,ks] ; 1: psf_map layout [S,L,g*ks,g*ks] with N = g*g (optics.py:1025)
 *   centre_out_or_null [S,L,N,2]
 *   flags_or_null: see aadff_publish_flags.  A point whose rays all miss the ks x ks window gets a 0/0 = NaN PSF, as in
 *   the reference (optics.py:978, monte_carlo.py:37).
 */
int aadff_psf_points(const float* points, int S, int N, int L,
                     const aadff_surface_t* surf_main, const aadff_surface_t* surf_chief,
                     aadff_lens_const_t lc, const aadff_lens_state_t* states,
                     const float* u_main, int spp, long main_stride_s, long main_stride_l,
                     const float* u_chief, int spp_chief, long chief_stride_s, long chief_stride_l,
                     int ks, int centre_mode, int map_layout, float* psf, float* centre_out_or_null,
                     int* flags_or_null, aadff_stream_t stream);

/* Upload of the uniform blocks of focus states [first_slice, S) folded into an
 * aadff_psf_points launch: a few leading workgroups copy src_host (PINNED host memory,
 * [S][slice_stride] floats, the layout u_main/u_chief index into) to dst_dev and bump
 * counters[s]; the PSF workgroups of those states wait for counters[s] to reach
 * generation * (number of copy workgroups); HIP guarantees no dispatch order, so the wait is bounded and a
 * workgroup whose block is late reads its draws from src_host directly (flag bit 3 reports the lost overlap).  Blocks of states < first_slice must already be
 * in dst_dev (aadff_refocus_staged with n_u = first_slice*slice_stride puts them there, hidden
 * behind the focus traces).  counters: [S] device words zeroed once; generation = 1, 2, 3 ...
 * for successive launches on the same counters (same S, N, L, slice_stride each time). */
typedef struct aadff_stage {
    const float* src_host;
    float*       dst_dev;
    long         slice_stride;
    int          first_slice;
    unsigned     generation;
    unsigned*    counters;
} aadff_stage_t;

int aadff_psf_points_staged(const float* points, int S, int N, int L,
                     const aadff_surface_t* surf_main, const aadff_surface_t* surf_chief,
                     aadff_lens_const_t lc, const aadff_lens_state_t* states,
                     const float* u_main, int spp, long main_stride_s, long main_stride_l,
                     const float* u_chief, int spp_chief, long chief_stride_s, long chief_stride_l,
                     int ks, int centre_mode, int map_layout, float* psf, float* centre_out_or_null,
                     int* flags_or_null, const aadff_stage_t* stage, aadff_stream_t stream);

/* "Edge-exact" PSF grid (ABI v8; Lensgroup(parity="edge")): aadff_psf_points with the one decision the float32 noise of a trace
 * can flip taken out of the fast kernel.  The histogram's only discontinuity is the window test of deeplens/monte_carlo.py:37
 * (`|s| < R - 0.01 ps`): a hit within the noise of that edge lands inside or outside depending on the last bit of the trace
 * (deeplens/surfaces.py:523-586: the batch-wide Newton loop), and a flipped border ray changes a cropped PSF by 1 / (rays inside).
 * Three calls on one stream replace Lensgroup.psf_map, deeplens/optics.py:888-1026, for S focus states whose d_sensor / hfov the
 * caller computed in the reference's arithmetic (aadff_trace_rays_strict_fused levels, aadff/strict_stack.py):
 *   1. aadff_psf_points_edge: arguments as aadff_psf_points (chief-ray centres, centre_mode 1).  A live ray whose hit lies within
 *      delta_mm of the window edge is NOT splatted but appended as (point << 16 | sample) to edge_list[s*L + l][edge_cap], its job's
 *      edge_count[s*L + l] incremented (zeroed by the call; a count above edge_cap means rays were dropped).  raw [S*L][N][ks*ks]
 *      receives the UNNORMALISED histograms of all other rays, centre_out [S,L,N,2] the centres.
 *   2. aadff_strict_edge_retrace (csrc/strict_fused.hip): re-traces the listed rays of the B = S*L batches in the reference's float32
 *      operation order - arguments as aadff_strict_psf_points (object points and pupil points from the reference's host arithmetic,
 *      pred[b][1] = the main batch's Newton counts) - applies forward_integral's window test and bilinear taps to that hit with
 *      `centre` = step 1's centres, and adds the taps to raw.  flags bit 4: a list overflowed.
 *      PROVISIONAL STATES: step 1 may run on lens states that are only close to the exact ones (the fast refocus kernel's, a few
 *      ulps off) so that it overlaps the strict refocus / calc_fov round trips; it then also writes slope_out [S,L,N,2], the mean
 *      direction tangents of the valid chief rays, and step 2 - given states_prov [P] (what step 1 ran on), tan_exact [P] =
 *      float(tan(hfov)) and z_sensor of the exact states - moves each centre into the exact world before the window test:
 *      c = (c - slope * (d_exact - d_prov)) * tan_exact / tan_prov.  states_prov_or_null = NULL: step 1 ran on the exact states.
 *   3. aadff_psf_normalise: raw -> psf in either layout of aadff_psf_points (the division of optics.py:978; same summation order as
 *      aadff_psf_points: a PSF none of whose rays was deferred is what aadff_psf_points writes for the same states, up to the
 *      order of the float atomics of its LDS histogram, which no two launches share). */
int aadff_psf_points_edge(const float* points, int S, int N, int L,
                          const aadff_surface_t* surf_main, const aadff_surface_t* surf_chief,
                          aadff_lens_const_t lc, const aadff_lens_state_t* states,
                          const float* u_main, int spp, long main_stride_s, long main_stride_l,
                          const float* u_chief, int spp_chief, long chief_stride_s, long chief_stride_l,
                          int ks, float delta_mm, float* raw, float* centre_out, float* slope_out_or_null,
                          unsigned* edge_count, unsigned* edge_list, int edge_cap,
                          int* flags_or_null, aadff_stream_t stream);
int aadff_strict_edge_retrace(const float* points, int N, int B, const int* point_set, const aadff_surface_t* tables_dev, int n_tables,
                              int n_surf, const int* table_main, const float* z_sensor, const float* pupil_main, int spp,
                              const int* pred, float pixel_size, int ks, const float* centre, const unsigned* edge_count,
                              const unsigned* edge_list, int edge_cap, float* raw, int* flags_or_null,
                              const aadff_lens_state_t* states_prov_or_null, const float* tan_exact, const float* slope,
                              aadff_stream_t stream);
int aadff_psf_normalise(const float* raw, int S, int N, int L, float pixel_size, int ks, int map_layout, float* psf,
                        aadff_stream_t stream);

/* Fused PSF-surrogate network: P rows of (x, y, z, foc_z) -> MLP (Linear+ReLU ..., Linear+Sigmoid) -> L1-normalise
 * -> mode 0: psf_out[P][n_out] (PSFNet.pred, deeplens/psfnet.py:375-390, psfnet_arch.py:24-47), or
 *    mode 1: out[N][C][H][W] = per-pixel PSF gather over img (PSFNet.render + local_psf_render,
 *            deeplens/psfnet.py:393-441, deeplens/render_psf.py:76-107; P = N*H*W, ks*ks = n_out).
 *            out_slices = S > 0: a whole focal stack in one launch (2_aber_aware_dff_aif.py:104-114): inp holds
 *            P = B*S*H*W rows ordered [b][slice][y][x], img is [B][C][H][W], out is [B][C][S][H][W].
 * Layer l maps in_features[l] -> out_features[l] (widths <= 256, in_features[0] == 4, n_out <= 128).
 * wpack: the weights as exact fp16 (hi, lo) pairs in MFMA fragment order, per layer
 *   [out tile 16][k-step 32][plane hi|lo][lane 64][8 halves]: lane (m = lane&15, kg = lane>>4) holds
 *   weight[16*tile + m][32*step + 8*kg + 0..7], hi = fp16(w), lo = fp16(w - hi), zero-padded;
 * bias: per layer out_features padded to 16, concatenated.  (aadff/psfnet_pack.py builds both.) */
#define AADFF_PSFNET_MAX_LAYERS 16
int aadff_psfnet_forward(const float* inp, long P, const void* wpack, const float* bias, int n_layers,
                         const int* in_features, const int* out_features, int mode, float* psf_out,
                         const float* img, float* out, int C, int H, int W, int ks, int out_slices, int precision,
                         int* flags_or_null, aadff_stream_t stream);
/* precision (both psfnet entries): 0 = fp32-equivalent (fp16 hi/lo operand split, three MFMAs per product: <= 2e-7 from
 * torch fp32); 1 = fp16 single pass (opt-in: operands rounded to fp16, one MFMA per product, PSFs ~5e-4 relative).
 * flags_or_null (both psfnet entries): bit 4 is ORed in when a hidden activation exceeded 65504, the largest value the
 * fp16 hi half of the split operand can carry — the affected outputs are then inf/NaN garbage and the caller must not
 * use them (the shipped rf50mm checkpoint peaks at 41, tests/golden/g11_ckpt_activation_range.json). */

/* PSFNet.render for a whole RGB-D focal stack with the network input generated in the kernel
 * (deeplens/psfnet.py:393-441 per slice, 2_aber_aware_dff_aif.py:104-114 for the stack): row (n, s, y, x) =
 * (xs[x], ys[y], clamp((depth[n][y][x] - d_min) * inv_range, 0, 1), foc_z[n][s]); out [N,C,S,H,W] (S = 1: [N,C,H,W]).
 * xs = linspace(-1, 1, W), ys = linspace(1, -1, H) (psfnet.py:427-431), foc_z = depth2z(foc_dist) (psfnet.py:447-450),
 * inv_range = 1 / (d_max - d_min) in fp32. */
int aadff_psfnet_render_rgbd(const float* depth, const float* xs, const float* ys, const float* foc_z, float d_min,
                             float inv_range, long N, int S, const void* wpack, const float* bias, int n_layers,
                             const int* in_features, const int* out_features, const float* img, float* out, int C, int H,
                             int W, int ks, int precision, int* flags_or_null, aadff_stream_t stream);

/* Refocus S lens states in one launch: trace spp rays from (0,0,depth[s]) (green table),
 * least-squares axis crossing -> d_sensor, then hfov/foclen/fnum.  Replaces
 * Lensgroup.refocus + post_computation + calc_fov + calc_efl,
 * deeplens/optics.py:1155-1217,178-187,1097-1102.  depth [S] (mm, <0); u: theta row at
 * u + s*u_stride_s, r row spp floats after it (dense [S,2,spp]: stride 2*spp). */
int aadff_refocus(const float* depth, int S, const float* u, int spp, long u_stride_s,
                  const aadff_surface_t* surf_green, aadff_lens_const_t lc,
                  aadff_lens_state_t* states, aadff_stream_t stream);

/* aadff_refocus with the host->device upload of the step's uniform block folded into the same
 * launch: u_host is PINNED host memory (hipHostMalloc / torch pin_memory) in the layout
 * aadff_psf_points will read; the S focus workgroups read their 2*spp draws
 * (u_host + s*u_stride_s) over PCIe while extra workgroups copy its first n_u floats to u_dev
 * (the rest can ride on the PSF launch, aadff_psf_points_staged).
 * Saves the separate hipMemcpyAsync and its queue gap in front of a focal stack
 * (the torch.rand draws of deeplens/optics.py:480-481,1166 stay on the host, Appendix B).
 * Both pointers 16-byte aligned.  The host block may be reused once this launch completed.
 * The rays of a focus state are traced by 4 workgroups; `scratch` (64 bytes per focus state,
 * device memory, zeroed once by the caller, not shared by concurrent launches) carries their
 * partial sums to the one that finishes the state (fixed summation order: deterministic). */
int aadff_refocus_staged(const float* depth, int S, const float* u_host, float* u_dev, long n_u, int spp,
                         long u_stride_s, const aadff_surface_t* surf_green, aadff_lens_const_t lc,
                         aadff_lens_state_t* states, void* scratch, aadff_stream_t stream);

/* The device-visible address of a block of PINNED host memory (error when it is not device-mapped).  The per-call mirror of
 * Lensgroup.refocus / psf_map (deeplens/optics.py:1155-1180, :888-1026) passes such addresses to aadff_refocus (draws and depth
 * read over PCIe by the one workgroup) and as `points` of aadff_psf_points_staged, so that no copy is queued in front of a launch. */
int aadff_host_device_pointer(const void* host, void** dev_out);

/* hfov/foclen/fnum for states whose d_sensor is already set (lens load, or a caller
 * that assigns d_sensor).  Replaces post_computation, deeplens/optics.py:178-187. */
int aadff_post_computation(int S, const aadff_surface_t* surf_green, aadff_lens_const_t lc,
                           aadff_lens_state_t* states, aadff_stream_t stream);

/* Copy the flags word the PSF / refocus kernels OR into (bit 0: NaN in a Newton residual — the reference exits,
 * deeplens/surfaces.py:555-558; bit 1: no valid chief ray, the assert of deeplens/optics.py:901; bit 2: a focus state
 * without a positive sensor position, i.e. a refocus that found no valid ray - "sensor position is negative.",
 * deeplens/optics.py:1176; bit 3: a staged upload arrived late and the samples were read over PCIe instead) to a PINNED host word from inside the stream:
 * the host can then poll the reference's error conditions of pipelined stacks without a device synchronisation
 * (it reads the mirror after an event it waits on anyway).  One 64-thread launch. */
int aadff_publish_flags(const int* flags_dev, int* mirror_host, aadff_stream_t stream);

/* One optimisation step of the PSF-network fit on a FLAT fp32 parameter buffer: torch.optim.AdamW (betas, eps, decoupled
 * weight decay) with the learning rate of CosineAnnealingLR(T_max = t_max, eta_min = 0) evaluated in the kernel from the
 * device step counter (`*step_dev` = completed steps; incremented by this call), so the launch pair is HIP-graph
 * capturable with no host-side scalar.  Replaces the optimiser half of deeplens/psfnet.py:85-108 (AdamW + scheduler.step()).
 *   grad: fp32 [n], or bf16 [n] with grad_is_bf16 != 0 (gradients of the bf16 parameter copy)
 *   param_bf16_or_null: bf16 [n] copy of the updated parameters (round to nearest even) for the bf16 leg
 *   scratch4: 4 device floats (the step's scalars, written by a one-thread launch in front of the update) */
int aadff_adamw_step(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                     void* param_bf16_or_null, long n, int* step_dev, float* scratch4, float lr0, int t_max, float beta1,
                     float beta2, float eps, float weight_decay, aadff_stream_t stream);

/* Two fused pieces of the fit step's backward (HIP-graph capturable, fp32 or bf16 tensors):
 * aadff_relu_bwd_bias: dz = dy * (y > 0) and db = column sums of dz for one hidden layer ([M,N] row-major) — the
 *   ReLU backward + bias gradient of deeplens/psfnet_arch.py:24-41 under autograd;
 * aadff_psfnet_head_loss_grad: pred = L1-normalised sigmoid(z) (psfnet_arch.py:41-47), and dz = d/dz of
 *   nn.MSELoss()(pred, target) (deeplens/psfnet.py:94-106) for z [B,N], N <= 128; pred and target are fp32. */
int aadff_relu_bwd_bias(const void* dy, const void* y, void* dz, void* db, int M, int N, int is_bf16, aadff_stream_t stream);
int aadff_psfnet_head_loss_grad(const void* z, const float* target, float* pred, void* dz, int B, int N, int is_bf16,
                                aadff_stream_t stream);

/* The fit step of the PSF network as hand-written bf16 MFMA kernels (csrc/mlp_train.hip; used by aadff/mlp_fit.py), i.e. the
 * autograd graph of deeplens/psfnet.py:94-106 over deeplens/psfnet_arch.py:24-47 without torch autograd:
 * aadff_fit_gemm_nt: out[b][a] = sum_c A[a][c] B[b][c] for bf16 A [na][lda], B [nb][ldb], contraction nc <= 256, with
 *   epilogue 0 forward (+ bias[a], bf16 out [nb][ld_out] and optional transposed copy outT [na][ld_outT]),
 *            1 forward + ReLU,
 *            2 dX (multiply by mask[b][a] > 0 — the forward output —, bf16 out / outT, dbias[a] += column sums, fp32 atomics),
 *            3 dW (fp32 out [nb][ld_out]);
 *   leading dimensions multiples of 8 (operands) / 4 (outputs), padding zero, na a multiple of 4;
 * aadff_fit_layer_bwd: the backward of one layer in ONE launch: dW [n][k] fp32 = dzT [n][:batch] . xT_prev [k][:batch], and,
 *   when wT != NULL, the dX epilogue for the layer below: dz_prev [batch][k] = (dz [batch][:n] . wT [k][:n]) masked by
 *   x_prev > 0, its transposed copy dzT_prev and dbias_prev[k] += column sums;
 * aadff_fit_input: fp32 batch [B][K] -> bf16 x [B][ld_x] and xT [K][ld_xT];
 * aadff_fit_head: sigmoid + L1 normalise + MSE gradient for logits z [B][ld_z] bf16 -> pred fp32 [B][N], dz / dzT bf16,
 *   dbias += column sums; when step_dev != NULL it also prepares the optimiser step (scalars into scratch4 as in
 *   aadff_adamw_step, step counter += 1) for the aadff_fit_adamw that follows in the same stream;
 * aadff_fit_adamw: AdamW on flat fp32 parameters with fp32 gradients (zeroed after use), refreshing the bf16 operand
 *   copies param_bf16[dst[i]] and, where dst_t[i] >= 0, param_bf16[dst_t[i]] (the transposed weights). */
/* The same step in THREE launches (aadff_fit_chain = 2, then aadff_fit_adamw): a workgroup takes 16 rows of the batch through
 * input cast, every Linear(+ReLU), the head and the whole dX chain with the activations in LDS (rows only meet in dW), then
 * one launch computes dW of all layers.  `aadff_fit_net` describes the network and its buffers:
 *   param_bf16:   W_l at off_w and W_l^T at off_wt in MFMA FRAGMENT ORDER - a matrix M [rows][cols] (W: n x k, W^T: k x n) is stored as
 *                 [row / 16][col / 32][lane = ((col % 32) / 8) * 16 + row % 16][col % 8], zero padded to whole 16-row tiles and
 *                 32-column steps, so that a wave-instruction reads 1 KiB contiguously; bias_l [up4(n)] at off_b (bf16; offsets in
 *                 elements, multiples of 8; refreshed by aadff_fit_adamw through its destination maps; ld_k / ld_n are not used
 *                 by this entry);
 *   scratch_bf16: X_l^T [k_l][ld_batch] at off_xt[l] (l = 0..L-1) and dZ_l^T [n_{l-1}][ld_batch] at off_dzt[l] (l = 1..L), written
 *                 by the chain kernel, read by the dW kernel (zero padded columns batch..ld_batch);
 *   grad (fp32):  dW_l [n][k] at off_gw (stored), db_l [n] at off_gb (ADDED by atomics: must be zero on entry; aadff_fit_adamw
 *                 leaves it zero);
 *   inp [B][k_0], target / pred [B][n_last] fp32.   Widths <= 256, input widths multiples of 4, n_last <= 128, B <= 256.
 * When step_dev != NULL the chain also prepares the optimiser step as aadff_fit_head does. */
#define AADFF_FIT_MAX_LAYERS 16
typedef struct aadff_fit_net {
    int n_layers, batch, ld_batch;
    int k[AADFF_FIT_MAX_LAYERS], n[AADFF_FIT_MAX_LAYERS];
    int ld_k[AADFF_FIT_MAX_LAYERS], ld_n[AADFF_FIT_MAX_LAYERS];
    int off_w[AADFF_FIT_MAX_LAYERS], off_wt[AADFF_FIT_MAX_LAYERS], off_b[AADFF_FIT_MAX_LAYERS];
    int off_xt[AADFF_FIT_MAX_LAYERS + 1], off_dzt[AADFF_FIT_MAX_LAYERS + 1];
    int off_gw[AADFF_FIT_MAX_LAYERS], off_gb[AADFF_FIT_MAX_LAYERS];
    const void* param_bf16;
    void* scratch_bf16;
    float* grad;
    const float* inp;
    const float* target;
    float* pred;
} aadff_fit_net;
int aadff_fit_chain(const aadff_fit_net* net, int* step_dev, float* scratch4, float lr0, int t_max, float beta1, float beta2,
                    float weight_decay, aadff_stream_t stream);
int aadff_fit_gemm_nt(const void* A, int lda, int na, const void* B, int ldb, int nb, int nc, int epilogue, void* out,
                      int ld_out, void* outT, int ld_outT, const void* bias, const void* mask, int ld_mask, float* dbias,
                      aadff_stream_t stream);
int aadff_fit_layer_bwd(const void* xT_prev, int ld_xT, int k, const void* dzT, int ld_dzT, int n, int batch, float* dW,
                        const void* wT, int ld_wT, const void* dz, int ld_dz, const void* x_prev, int ld_x, void* dz_prev,
                        int ld_dzp, void* dzT_prev, int ld_dzTp, float* dbias_prev, aadff_stream_t stream);
int aadff_fit_input(const float* inp, void* x, int ld_x, void* xT, int ld_xT, int B, int K, aadff_stream_t stream);
int aadff_fit_head(const void* z, int ld_z, const float* target, float* pred, void* dz, int ld_dz, void* dzT, int ld_dzT,
                   float* dbias, int B, int N, int* step_dev, float* scratch4, float lr0, int t_max, float beta1, float beta2,
                   float weight_decay, aadff_stream_t stream);
int aadff_fit_adamw(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, const int* dst,
                    const int* dst_t, long n, const float* scal4, float beta1, float beta2, float eps, aadff_stream_t stream);

/* ------------------------------------------------------------------ host helper */

/* HOST routine (no GPU work): the next n float32 uniforms of torch's CPU generator, bit-identical
 * to torch.rand(n), produced from / written back to the byte state of torch.get_rng_state()
 * (5056 bytes).  Replaces the host draws torch.rand(spp) of deeplens/optics.py:480-481 and
 * deeplens/surfaces.py:192-193 at ~7x the speed while leaving torch's generator exactly where the
 * reference's call sequence would.  out_host may be pinned memory. */
int aadff_host_mt19937_uniform_f32(unsigned char* torch_state_host, long state_bytes, long n, float* out_host);

/* HOST routine: advance the same byte state past n float32 draws without producing them (the generator's regenerations
 * only) - the state torch.rand(n) would leave behind.  A rank of the sharded configuration that owns some
 * slices of a scene skips the other slices' draws of the reference's per-scene stream (deeplens/optics.py:480-481 in the
 * call order of a stack) instead of producing and dropping them. */
int aadff_host_mt19937_discard(unsigned char* torch_state_host, long state_bytes, long n);

/* HOST routine: the [n_rows][row_len] block of draws that starts at the generator's position, produced in TWO passes - phase 0: the
 * first `head` draws of every row (the rest skipped, a 2504-byte generator snapshot per row kept in `snapshots`), the state left
 * behind the whole block as torch.rand(n_rows * row_len) would; phase 1: the remaining row_len - head draws of every row from the
 * snapshots.  Same block as one pass, bit for bit.  For a focal stack the heads are the focus draws (torch.rand(2048) twice per
 * slice, deeplens/surfaces.py:192-193 via optics.py:1166): the refocus launch can start after phase 0 while phase 1 fills the PSF
 * rows (deeplens/optics.py:480-481) behind it. */
int aadff_host_mt19937_rows(unsigned char* torch_state_host, long state_bytes, int n_rows, long row_len, long head, float* out_host,
                            unsigned char* snapshots_host, int phase);

/* HOST routine (strict / edge parity): out[k] = np.mean of the values of row k whose weight is > 0 and which are > 0 (and not NaN) -
 * refocus's `focus_d[ra > 0]`, `focus_d[~isnan & (focus_d > 0)]`, `np.mean` (deeplens/optics.py:1175-1178) for all slices of a stack in
 * numpy's own arithmetic: pairwise float32 summation in numpy's blocking, the division in float64, rounded to float32; NaN for a row
 * without a countable value.  values / weights [rows][n], scratch [n], out [rows]; all host memory. */
int aadff_host_masked_mean_f32(const float* values, const float* weights, long rows, long n, float* scratch, float* out);

/* HOST DRIVER of a strict / edge focal stack (ABI v8, csrc/stack_host.cpp): the per-stack host work between two GPU waits as one call
 * each.  A strict / edge stack is host-bound (0.6 ms of GPU work, 1.2 ms of Python between its waits); these calls do exactly what
 * aadff/strict_stack.py does there - same parameter blocks, same launches (the entry points above), same count check, same host
 * arithmetic - and return a status instead of deciding anything new: 0 = done; 1 = some batch was confirmed by none of its
 * candidate count rows; 2 = a NaN residual in a run the reference makes (the caller repeats the level the slow way, which corrects the
 * table / raises the reference's error).  Replaces the host side of Lensgroup.refocus + calc_fov (deeplens/optics.py:1155-1217) for S
 * slices and of Lensgroup.psf_map (:888-1026) of an edge-parity lens.
 * aadff_levels_t: blocks in the layout of aadff/strict_stack.py (_Stage): level 1 parameters [S*3 axis points | jobs_max job -> batch |
 * J*MAX_SURF predicted rows], results [J*2048 crossing z | J*2048 ra | J*2*MAX_SURF any / nan bits]; level 2 parameters [S*3 sensor
 * corners | M*3 pupil points | job -> batch | rows], results [J*M tan | J*M ra | bits].  The caller has written job -> batch, the rows
 * and the M pupil points; pinned host blocks (h_*) and their device twins (d_*). */
typedef struct aadff_levels {
    int S, n_surf, n_tables, jobs_max, fov_rays, J1, J2, pad;
    const aadff_surface_t* tables_dev;
    const int* bt_green;                 /* [>= jobs_max] table index of the focus wavelength */
    const int* zeros;                    /* [>= jobs_max] */
    int *h_par1, *d_par1, *h_res1, *d_res1;
    int *h_par2, *d_par2, *h_res2, *d_res2;
    float *h_pupil, *d_pupil;            /* [S][2048][3] aperture points of the focus rays */
    unsigned char curved[AADFF_MAX_SURF];
} aadff_levels_t;
/* level 1: aperture points of the first surface from the stack's uniforms (as aadff_host_pupil_points: theta row of slice k at
 * u_host + off_focus[k], radius row 2048 floats behind), upload, launch, download, event (a hipEvent_t, or NULL) */
int aadff_levels_focus_submit(const aadff_levels_t* p, const float* u_host, const long* off_focus, float pi_f, float R2, float z_first,
                              const float* focus, const void* cos_fn, const void* sin_fn, const void* sqrt_fn, int width,
                              aadff_stream_t stream, void* event_or_null);
/* after the event: chosen[S] = the confirmed job of every slice, d_sensor[S] = np.mean of its countable crossing distances
 * (aadff_host_masked_mean_f32; NaN for a slice without one: the reference's "sensor position is negative."); scratch [2048] */
int aadff_levels_focus_finish(const aadff_levels_t* p, int* chosen, float* d_sensor, float* scratch);
/* level 2: the sensor corners (r_last, 0, d_sensor[k]) into the block, upload, launch (forward = 0: the usual backward trace), download, event */
int aadff_levels_fov_submit(const aadff_levels_t* p, const float* d_sensor, float r_last, int forward, aadff_stream_t stream, void* event_or_null);
/* after the event: chosen[S], tan_fov [S][M] and ra [S][M] of the confirmed jobs (the caller takes torch's own sum and atan of them) */
int aadff_levels_fov_finish(const aadff_levels_t* p, int* chosen, float* tan_fov, float* ra);

/* the edge-exact psf_map level of a stack (see aadff_psf_points_edge): everything the three launches need, owned by the caller */
typedef struct aadff_edge_stack {
    int S, L, N, spp, ks, n_surf, n_tables, t_green, cap, pad;
    long per, per_l, o_main, n_pm;       /* floats per slice / per wavelength of the uniform block, offset of the main rows, floats of main pupil points */
    float delta, pixel_size;
    aadff_lens_const_t lc;
    const aadff_surface_t* tables_dev;
    float *h_u, *d_u;                    /* [S*per] the stack's uniforms */
    float *h_focus, *d_focus;            /* [2 S]: focus distances | float(tan(hfov)) of the exact states */
    const float* d_pts;                  /* [S][N][3] normalised field points */
    void* states_prov;                   /* aadff_lens_state_t[S] (device): written by the provisional refocus */
    float *raw, *slope;
    unsigned *count, *list;              /* [S*L + 1] (last word: flags), [S*L][cap] */
    int* h_back;                         /* [S*L + 1] pinned: counts and flags come back here */
    int *h_par3, *d_par3;                /* [S*L z_sensor | S*N*3 object points | S*L*2*MAX_SURF count rows (written by the caller)] */
    const int *pset, *bt_main;           /* [S*L] */
    float *h_pupil_main, *d_pupil_main;  /* [S*L][spp][3] */
} aadff_edge_stack_t;
/* provisional pass: uniforms and focus distances up, fast refocus -> states_prov, aadff_psf_points_edge on them (centre [S*L][N][2]) */
int aadff_edge_provisional(const aadff_edge_stack_t* e, const float* focus, float* centre, aadff_stream_t stream);
/* after levels 1 and 2: object points of the exact states (psf_diff's float32 arithmetic, deeplens/optics.py:945-950) from pts_norm
 * [N][3], hfov [S] (float64, as the reference's Python floats) and d_sensor [S]; uploads (event_uploaded: the pinned blocks may be
 * reused), re-trace with the centres moved into the exact world, normalise into maps [S][L][g ks][g ks], counts / flags to h_back,
 * event_done */
int aadff_edge_finish(const aadff_edge_stack_t* e, const float* pts_norm, const double* hfov, const float* d_sensor, float r_last,
                      float sensor_w, float sensor_h, const float* centre, float* maps, aadff_stream_t stream, void* event_uploaded,
                      void* event_done);

/* Workgroup size (256 or 1024, default 1024) of an aadff_strict_psf_points launch with fewer than 1024 workgroups, i.e. a re-launch
 * of the few batches whose speculated Newton counts (deeplens/surfaces.py:547) were off: 1024 threads shorten it on an idle GPU,
 * 256 get scheduled beside another stack's full launch (aadff.strict_stack.StrictPipeline).  Process-wide. */
int aadff_strict_replay_threads(int threads);

/* HOST routine (strict-parity mode): rows of pupil / aperture points from host uniforms in the reference's float32 operations -
 * theta = (u * 2) * pi, r = sqrt(u' * R^2), (r cos theta, r sin theta, z): deeplens/optics.py:480-486 (sample_point_source's pupil
 * sampling) and deeplens/surfaces.py:188-199 (the aperture points of refocus).  cos_fn / sin_fn / sqrt_fn: addresses of the vector routines
 * torch's CPU kernels use, resolved by the caller in the libtorch_cpu.so of its process - kind 1: MKL's vmsCos / vmsSin / vmsSqrt
 * (torch built with MKL: ATen/cpu/vml.h), kind 16: Sleef_cosf16_u10 / Sleef_sinf16_u10, kind 8: Sleef_cosf8_u10 / Sleef_sinf8_u10
 * (sqrt_fn unused: IEEE root) - same routines, same bits, without the ~70 small tensor operations per stack.  Row i: n theta uniforms at u + theta_off[i], n radius uniforms at u + r_off[i] -> out[i][n][3]. */
int aadff_host_pupil_points(const float* u_host, long n_rows, const long* theta_off, const long* r_off, long n, float pi_f, float R2,
                            float z, float* out_host, const void* cos_fn, const void* sin_fn, const void* sqrt_fn, int kind);

#ifdef __cplusplus
}
#endif
#endif /* AADFF_H_ */
