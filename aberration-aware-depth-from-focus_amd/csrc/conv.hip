// Image-space PSF application for gfx950 (MI355X): patch-wise PSF-grid convolution
// (render_psf_map / render_psf, single slice and stack-fused) and per-pixel PSF gather
// (local_psf_render).  Semantics follow deeplens/render_psf.py of the reference; see
// include/aadff.h for the per-entry citations and DESIGN.md for the kernel design.
#include <cmath>
#include <cstdarg>
#include <mutex>
#include "common.h"

namespace aadff {

// ------------------------------------------------------------------------------------
// error string shared by the whole library
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------
// Patch bounds: Python `int(i / grid * n)` in float64 (deeplens/render_psf.py:65-66).
// Passed by value in the kernarg segment -> read with scalar loads.
// ------------------------------------------------------------------------------------
struct PatchBounds {
    int hb[AADFF_MAX_GRID + 1];
    int wb[AADFF_MAX_GRID + 1];
};

static void fill_bounds(int* b, int grid, int n) {
    for (int i = 0; i <= grid; ++i) b[i] = (int)((double)i / (double)grid * (double)n);
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    // torch 'reflect' padding (no edge repeat), then clamped: the clamp only matters for
    // masked-out lanes of ragged tiles.
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// ------------------------------------------------------------------------------------
// Fast path: one wave per 32x32 output tile of ONE patch and ONE channel plane.
//   lane = (k = lane>>4, q = lane&15) owns outputs rows 8k..8k+7, cols 2q,2q+1
//   -> 16 accumulators; input rows slide through 12 registers read as ds_read_b64;
//   PSF taps are wave-uniform (one patch per tile) -> SGPR operands of v_fma_f32.
// LDS pitch P with P % 8 == 4: the two thread-rows of a half-wave are 8 tile rows apart,
// 8*P*4 B = 128 (mod 256) -> the 32 lanes of a ds_read_b64 group hit 64 distinct banks.
// Taps are consumed in groups of UG PSF rows so that UG*KS weights stay in SGPRs while
// each input row is re-read only ceil(KS/UG) times.
// ------------------------------------------------------------------------------------
constexpr int TW = 32, TH = 32, RR = 8;

template <int KS>
struct ConvCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int TWP = TW + KS - 1;
    static constexpr int THP = TH + KS - 1;
    static constexpr int PITCH = ((TWP - 4 + 7) / 8) * 8 + 4;
    static constexpr int NIN = KS + 1;                       // input regs per row (2 cols + KS-1 halo)
    static constexpr int UG = KS <= 7 ? KS : (KS <= 13 ? 4 : 3);   // PSF rows per SGPR group
};

template <int KS>
__global__ __launch_bounds__(64) void conv_psf_map_kernel(const float* __restrict__ img,
                                                           const float* __restrict__ psf,
                                                           float* __restrict__ out, int C, int S, int H,
                                                           int W, int grid, int ntx, int nty,
                                                           PatchBounds pb) {
    using Cfg = ConvCfg<KS>;
    constexpr int P = Cfg::PITCH;
    __shared__ __attribute__((aligned(16))) float tile[Cfg::THP * P];

    const int lane = threadIdx.x;
    const int pj = blockIdx.x / ntx, tx = blockIdx.x - pj * ntx;
    const int pi = blockIdx.y / nty, ty = blockIdx.y - pi * nty;
    const int bc = blockIdx.z;
    const int c = bc % C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TW, y0 = pb.hb[pi] + ty * TH;
    if (x0 >= x_hi || y0 >= y_hi) return;

    // ---- stage the reflect-padded input tile once (shared by all S slices) ----
    const float* plane = img + (size_t)bc * H * W;
    for (int e = lane; e < Cfg::THP * Cfg::TWP; e += kWave) {
        const int r = e / Cfg::TWP, cc = e - r * Cfg::TWP;
        const int yy = reflect_idx(y0 - Cfg::PAD + r, H);
        const int xx = reflect_idx(x0 - Cfg::PAD + cc, W);
        tile[r * P + cc] = plane[(size_t)yy * W + xx];
    }
    __syncthreads();

    const int q = lane & 15, k = lane >> 4;
    const int G = grid * KS;
    const float* trow = &tile[(RR * k) * P + 2 * q];

    for (int s = 0; s < S; ++s) {
        // PSF block of this patch; conv2d is a correlation with the FLIPPED PSF
        // (render_psf.py:62): w(u,v) = psf[KS-1-u][KS-1-v].
        const float* wp = psf + ((size_t)(s * C + c) * G + pi * KS) * G + pj * KS;
        float acc[RR][2];
#pragma unroll
        for (int r = 0; r < RR; ++r) acc[r][0] = acc[r][1] = 0.f;

#pragma unroll
        for (int u0 = 0; u0 < KS; u0 += Cfg::UG) {
            float w[Cfg::UG][KS];
#pragma unroll
            for (int ug = 0; ug < Cfg::UG; ++ug)
#pragma unroll
                for (int v = 0; v < KS; ++v)
                    w[ug][v] = (u0 + ug < KS) ? wp[(size_t)(KS - 1 - (u0 + ug)) * G + (KS - 1 - v)] : 0.f;
#pragma unroll
            for (int ir = u0; ir < u0 + Cfg::UG - 1 + RR; ++ir) {
                if (ir >= KS - 1 + RR) continue;
                float in[Cfg::NIN];
                const float2* rp = reinterpret_cast<const float2*>(trow + ir * P);
#pragma unroll
                for (int h = 0; h < Cfg::NIN / 2; ++h) {
                    const float2 t2 = rp[h];
                    in[2 * h] = t2.x;
                    in[2 * h + 1] = t2.y;
                }
#pragma unroll
                for (int ug = 0; ug < Cfg::UG; ++ug) {
                    const int u = u0 + ug, r = ir - u;
                    if (u >= KS || r < 0 || r >= RR) continue;
#pragma unroll
                    for (int v = 0; v < KS; ++v) {
                        acc[r][0] = fmaf(w[ug][v], in[v], acc[r][0]);
                        acc[r][1] = fmaf(w[ug][v], in[v + 1], acc[r][1]);
                    }
                }
            }
        }

        float* oplane = out + ((size_t)bc * S + s) * H * W;
        const int x = x0 + 2 * q;
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const int y = y0 + RR * k + r;
            if (y < y_hi) {
                if (x < x_hi) oplane[(size_t)y * W + x] = acc[r][0];
                if (x + 1 < x_hi) oplane[(size_t)y * W + x + 1] = acc[r][1];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Generic path (any odd ks <= AADFF_MAX_KS): same tiling, runtime loops, PSF taps staged
// flipped in LDS and read as broadcasts.  Correctness path for unusual kernel sizes.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_psf_map_generic_kernel(const float* __restrict__ img,
                                                                    const float* __restrict__ psf,
                                                                    float* __restrict__ out, int C, int S,
                                                                    int H, int W, int grid, int ks, int ntx,
                                                                    int nty, PatchBounds pb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pad = ks / 2, twp = TW + ks - 1, thp = TH + ks - 1;
    float* tile = smem;
    float* wl = smem + thp * twp;
    const int tid = threadIdx.x;
    const int pj = blockIdx.x / ntx, tx = blockIdx.x - pj * ntx;
    const int pi = blockIdx.y / nty, ty = blockIdx.y - pi * nty;
    const int bc = blockIdx.z, c = bc % C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TW, y0 = pb.hb[pi] + ty * TH;
    if (x0 >= x_hi || y0 >= y_hi) return;
    const float* plane = img + (size_t)bc * H * W;
    for (int e = tid; e < thp * twp; e += 256) {
        const int r = e / twp, cc = e - r * twp;
        tile[e] = plane[(size_t)reflect_idx(y0 - pad + r, H) * W + reflect_idx(x0 - pad + cc, W)];
    }
    const int G = grid * ks;
    const int lx = tid & 31, ly = tid >> 5;   // 32 x 8 threads, 4 rows each
    for (int s = 0; s < S; ++s) {
        __syncthreads();
        const float* wp = psf + ((size_t)(s * C + c) * G + pi * ks) * G + pj * ks;
        for (int e = tid; e < ks * ks; e += 256) {
            const int u = e / ks, v = e - u * ks;
            wl[e] = wp[(size_t)(ks - 1 - u) * G + (ks - 1 - v)];
        }
        __syncthreads();
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < ks; ++u)
            for (int v = 0; v < ks; ++v) {
                const float wv = wl[u * ks + v];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(wv, tile[(ly * 4 + r + u) * twp + lx + v], acc[r]);
            }
        float* oplane = out + ((size_t)bc * S + s) * H * W;
        const int x = x0 + lx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + ly * 4 + r;
            if (y < y_hi && x < x_hi) oplane[(size_t)y * W + x] = acc[r];
        }
    }
}

template <int KS>
static void launch_fast(const float* img, const float* psf, float* out, int B, int C, int S, int H, int W,
                        int grid, int ntx, int nty, const PatchBounds& pb, hipStream_t st) {
    dim3 g(ntx * grid, nty * grid, B * C);
    hipLaunchKernelGGL(conv_psf_map_kernel<KS>, g, dim3(64), 0, st, img, psf, out, C, S, H, W, grid, ntx, nty, pb);
}

static int conv_dispatch(const float* img, const float* psf, float* out, int B, int C, int S, int H, int W,
                         int grid, int ks, hipStream_t st) {
    AADFF_CHECK_ARG(img && psf && out, "render_psf_map: NULL pointer");
    AADFF_CHECK_ARG(B > 0 && C > 0 && S > 0 && H > 0 && W > 0, "render_psf_map: empty tensor (B=%d C=%d S=%d H=%d W=%d)", B, C, S, H, W);
    AADFF_CHECK_ARG(grid >= 1 && grid <= AADFF_MAX_GRID, "render_psf_map: grid %d outside [1,%d]", grid, AADFF_MAX_GRID);
    AADFF_CHECK_ARG(ks % 2 == 1, "PSF kernel size should be odd");
    AADFF_CHECK_ARG(ks >= 1 && ks <= AADFF_MAX_KS, "render_psf_map: ks %d outside [1,%d]", ks, AADFF_MAX_KS);
    AADFF_CHECK_ARG(ks / 2 < H && ks / 2 < W, "render_psf_map: reflect padding %d needs H,W > pad", ks / 2);
    AADFF_CHECK_ARG(grid <= H && grid <= W, "render_psf_map: grid %d larger than image %dx%d", grid, H, W);
    AADFF_CHECK_ARG((size_t)B * C <= 65535, "render_psf_map: B*C too large");

    PatchBounds pb;
    std::memset(&pb, 0, sizeof(pb));
    fill_bounds(pb.hb, grid, H);
    fill_bounds(pb.wb, grid, W);
    int mh = 0, mw = 0;
    for (int i = 0; i < grid; ++i) {
        mh = std::max(mh, pb.hb[i + 1] - pb.hb[i]);
        mw = std::max(mw, pb.wb[i + 1] - pb.wb[i]);
    }
    const int ntx = (mw + TW - 1) / TW, nty = (mh + TH - 1) / TH;
    switch (ks) {
#define AADFF_CASE(K) case K: launch_fast<K>(img, psf, out, B, C, S, H, W, grid, ntx, nty, pb, st); break;
        AADFF_CASE(3) AADFF_CASE(5) AADFF_CASE(7) AADFF_CASE(9) AADFF_CASE(11) AADFF_CASE(13)
        AADFF_CASE(15) AADFF_CASE(21)
#undef AADFF_CASE
        default: {
            dim3 g(ntx * grid, nty * grid, B * C);
            const size_t lds = ((size_t)(TH + ks - 1) * (TW + ks - 1) + (size_t)ks * ks) * sizeof(float);
            hipLaunchKernelGGL(conv_psf_map_generic_kernel, g, dim3(256), lds, st, img, psf, out, C, S, H, W, grid,
                               ks, ntx, nty, pb);
        }
    }
    AADFF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------
// Per-pixel PSF gather (local_psf_render): HBM-bound on the PSF tensor (ks*ks*4 B/pixel).
// One wave per run of NPX consecutive pixels of one image row:
//   * the NPX*ks*ks PSF floats are contiguous in [B,H,W,ks,ks] -> fully coalesced stream
//     into LDS; lane l then walks its own PSF at stride ks*ks (odd -> conflict-free);
//   * the C x ks x (NPX+ks-1) replicate-clamped image window sits in LDS as well;
//   * no flip (render_psf.py:99-105 multiplies unfold() patches with the kernel as is).
// ------------------------------------------------------------------------------------
template <int NPX>
__global__ __launch_bounds__(64) void local_psf_kernel(const float* __restrict__ img, const float* __restrict__ psf,
                                                        float* __restrict__ out, int C, int H, int W, int ks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int kk = ks * ks, pad = ks / 2, tw = NPX + ks - 1;
    float* wl = smem;                 // [NPX][kk]
    float* tl = smem + NPX * kk;      // [C][ks][tw]
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * NPX, y = blockIdx.y, b = blockIdx.z;
    const int npx = min(NPX, W - x0);

    const float* pp = psf + ((size_t)(b * H + y) * W + x0) * kk;
    for (int e = lane; e < npx * kk; e += kWave) wl[e] = pp[e];
    for (int e = lane; e < C * ks * tw; e += kWave) {
        const int cc = e / (ks * tw), rem = e - cc * ks * tw;
        const int u = rem / tw, xx = rem - u * tw;
        const int yy = min(max(y - pad + u, 0), H - 1);
        const int xs = min(max(x0 - pad + xx, 0), W - 1);
        tl[e] = img[((size_t)(b * C + cc) * H + yy) * W + xs];
    }
    __syncthreads();
    if (lane < npx) {
        const float* wr = wl + lane * kk;
        for (int cc = 0; cc < C; ++cc) {
            const float* tr = tl + cc * ks * tw + lane;
            float acc = 0.f;
            for (int u = 0; u < ks; ++u)
                for (int v = 0; v < ks; ++v) acc = fmaf(tr[u * tw + v], wr[u * ks + v], acc);
            out[((size_t)(b * C + cc) * H + y) * W + x0 + lane] = acc;
        }
    }
}

}  // namespace aadff

using namespace aadff;

extern "C" {

int aadff_abi_version(void) { return AADFF_ABI_VERSION; }
const char* aadff_last_error(void) { return g_err; }

int aadff_device_info(int* n_cu, int* lds_bytes, char* arch, int arch_len) {
    int dev = 0;
    AADFF_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    AADFF_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (arch && arch_len > 0) {
        std::strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}

int aadff_render_psf_map(const float* img, const float* psf_map, float* out, int B, int C, int H, int W, int grid,
                         int ks, aadff_stream_t stream) {
    return conv_dispatch(img, psf_map, out, B, C, 1, H, W, grid, ks, (hipStream_t)stream);
}

int aadff_render_psf_map_stack(const float* img, const float* psf_maps, float* out, int B, int C, int S, int H,
                               int W, int grid, int ks, aadff_stream_t stream) {
    return conv_dispatch(img, psf_maps, out, B, C, S, H, W, grid, ks, (hipStream_t)stream);
}

int aadff_render_psf(const float* img, const float* psf, float* out, int B, int C, int H, int W, int ks,
                     aadff_stream_t stream) {
    // a 1x1 PSF grid: the [C,ks,ks] PSF is its own map (render_psf.py:12-28 vs :31-73)
    return conv_dispatch(img, psf, out, B, C, 1, H, W, 1, ks, (hipStream_t)stream);
}

int aadff_local_psf_render(const float* img, const float* psf, float* out, int B, int C, int H, int W, int ks,
                           aadff_stream_t stream) {
    AADFF_CHECK_ARG(img && psf && out, "local_psf_render: NULL pointer");
    AADFF_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "local_psf_render: empty tensor");
    AADFF_CHECK_ARG(ks % 2 == 1 && ks >= 1 && ks <= AADFF_MAX_KS, "local_psf_render: ks %d must be odd and <= %d", ks, AADFF_MAX_KS);
    AADFF_CHECK_ARG(H <= 65535 && B <= 65535, "local_psf_render: H or B too large for the launch grid");
    hipStream_t st = (hipStream_t)stream;
    const int npx = ks <= 15 ? 64 : 16;
    const size_t lds = ((size_t)npx * ks * ks + (size_t)C * ks * (npx + ks - 1)) * sizeof(float);
    AADFF_CHECK_ARG(lds <= 160 * 1024, "local_psf_render: C=%d ks=%d needs %zu B of LDS", C, ks, lds);
    dim3 g((W + npx - 1) / npx, H, B);
    if (npx == 64) {
        if (lds > 64 * 1024)
            AADFF_CHECK_HIP(hipFuncSetAttribute((const void*)local_psf_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(local_psf_kernel<64>, g, dim3(64), lds, st, img, psf, out, C, H, W, ks);
    } else {
        if (lds > 64 * 1024)
            AADFF_CHECK_HIP(hipFuncSetAttribute((const void*)local_psf_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(local_psf_kernel<16>, g, dim3(64), lds, st, img, psf, out, C, H, W, ks);
    }
    AADFF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
