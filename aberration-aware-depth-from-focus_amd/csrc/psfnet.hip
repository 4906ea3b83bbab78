// Fused PSF-surrogate network for gfx950 (MI355X): per-pixel (x, y, z, foc_z) -> MLP -> sigmoid -> L1-normalise
// -> either the PSFs themselves (PSFNet.pred) or the per-pixel PSF gather over the image (PSFNet.render), in ONE
// kernel.  Reference: deeplens/psfnet_arch.py:24-47 (MLP 4 -> 64 -> 256 -> 8 x 256 -> ks^2, ReLU, Sigmoid,
// F.normalize(p=1)), deeplens/psfnet.py:393-441 (render), deeplens/render_psf.py:76-107 (local_psf_render:
// replicate padding, no flip).  1.14 MFLOP per pixel against 28 bytes of HBM traffic: the activations never leave
// the CU (the unfused path writes and re-reads 484 B/pixel of PSFs plus every layer's activations).
//
// Workgroup = 128 pixels x 8 waves.  Activations live in LDS as two fp16 planes (hi, lo) [128][264]; every layer is
// the GEMM  out^T[feat][px] = sum_k W[feat][k] act[px][k]  on v_mfma_f32_16x16x32_f16 with the exact fp16 hi/lo
// operand split of conv.hip (hi*hi + hi*lo + lo*hi, fp32 accumulate: every product exact, dropped term 2^-22).
// A operand = weights, pre-packed by the host in fragment order (16 B per lane, streamed from L2, one k-step ahead);
// B operand = activations (ds_read_b128, row pitch 264 halves: conflict-free); D^T puts 4 consecutive features of
// one pixel in a lane, so bias + ReLU + split + one 8-byte LDS store per plane write the next layer's input.
// Wave w owns feature tiles w, w + 8 (16 features each) for all 128 pixels: 64 accumulator registers.
#include <cmath>
#include "common.h"

namespace aadff {
namespace pn {

constexpr int TP = 128;                 // pixels per workgroup
constexpr int NWV = 8, NTH = 64 * NWV;
constexpr int AP = 264;                 // activation row pitch in halves (256 + 8: ds_read_b128 conflict-free)
constexpr int MAXL = AADFF_PSFNET_MAX_LAYERS;

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

struct Layers {
    int n;                              // number of Linear layers
    int kpad[MAXL];                     // input features padded to 32
    int npad[MAXL];                     // output features padded to 16
    int woff[MAXL];                     // offset of the layer's packed weights, in uint4 (16 B) units
    int boff[MAXL];                     // offset of the layer's (padded) bias, in floats
};

__device__ __forceinline__ void split4(float4v v, half4v& h, half4v& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (_Float16)v[i];
        l[i] = (_Float16)(v[i] - (float)h[i]);
    }
}

// mode 0: psf_out[P][nout] (normalised PSFs); mode 1: out[N][C][H][W] = per-pixel PSF gather over img
__global__ __launch_bounds__(NTH) void psfnet_fused_kernel(const float* __restrict__ inp, long P, const uint4v* __restrict__ wpack,
                                                           const float* __restrict__ bias, Layers L, int nout, int mode,
                                                           float* __restrict__ psf_out, const float* __restrict__ img,
                                                           float* __restrict__ out, int C, int H, int W, int ks) {
    __shared__ __attribute__((aligned(16))) _Float16 act[2][TP * AP];          // 135 168 B; reused for the fp32 PSFs at the end
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, lo4 = lane & 15;
    const long p0 = (long)blockIdx.x * TP;

    // ---- layer-0 input: features 0..3, zero-padded to 32 ----
    for (int e = tid; e < TP * 8; e += NTH) {                                   // 8 groups of 4 halves per pixel and plane
        const int px = e >> 3, g4 = e & 7;
        float4v v = {0.f, 0.f, 0.f, 0.f};
        if (g4 == 0 && p0 + px < P) v = *reinterpret_cast<const float4v*>(inp + (p0 + px) * 4);
        half4v h, l;
        split4(v, h, l);
        *reinterpret_cast<half4v*>(&act[0][px * AP + 4 * g4]) = h;
        *reinterpret_cast<half4v*>(&act[1][px * AP + 4 * g4]) = l;
    }
    __syncthreads();

    float4v acc[2][8];
#pragma unroll 1
    for (int l = 0; l < L.n; ++l) {
        const int nks = L.kpad[l] >> 5, ntile = L.npad[l] >> 4;
        const bool t0 = wave < ntile, t1 = wave + NWV < ntile;                  // this wave's feature tiles: wave, wave + 8
        const bool last = l == L.n - 1;
        if (t0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[j][p] = (float4v){0.f, 0.f, 0.f, 0.f};
            // packed weights: [tile][k-step][plane][lane] x 16 B
            const uint4v* w0 = wpack + L.woff[l] + ((size_t)wave * nks * 2) * 64 + lane;
            const uint4v* w1 = wpack + L.woff[l] + ((size_t)(wave + NWV) * nks * 2) * 64 + lane;
            uint4v a0h = w0[0], a0l = w0[64], a1h = a0h, a1l = a0l;
            if (t1) { a1h = w1[0]; a1l = w1[64]; }
#pragma unroll 1
            for (int s = 0; s < nks; ++s) {
                const uint4v c0h = a0h, c0l = a0l, c1h = a1h, c1l = a1l;
                if (s + 1 < nks) {                                               // next k-step's weights (L2 latency)
                    a0h = w0[(s + 1) * 128]; a0l = w0[(s + 1) * 128 + 64];
                    if (t1) { a1h = w1[(s + 1) * 128]; a1l = w1[(s + 1) * 128 + 64]; }
                }
                const half8v th0 = __builtin_bit_cast(half8v, c0h), tl0 = __builtin_bit_cast(half8v, c0l);
                const half8v th1 = __builtin_bit_cast(half8v, c1h), tl1 = __builtin_bit_cast(half8v, c1l);
                const int boffs = lo4 * AP + 32 * s + 8 * kg;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const half8v bh = *reinterpret_cast<const half8v*>(&act[0][16 * p * AP + boffs]);
                    const half8v bl = *reinterpret_cast<const half8v*>(&act[1][16 * p * AP + boffs]);
                    acc[0][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th0, bh, acc[0][p], 0, 0, 0);
                    acc[0][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th0, bl, acc[0][p], 0, 0, 0);
                    acc[0][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl0, bh, acc[0][p], 0, 0, 0);
                    if (t1) {
                        acc[1][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th1, bh, acc[1][p], 0, 0, 0);
                        acc[1][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th1, bl, acc[1][p], 0, 0, 0);
                        acc[1][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl1, bh, acc[1][p], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                                        // every wave is done reading this layer's input
        if (!last) {
            // D^T[feat = 16 tile + 4 kg + i][px = 16 p + lo4]: bias, ReLU, split, 8-byte stores
            if (t0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && !t1) break;
                    const int f0 = 16 * (wave + NWV * j) + 4 * kg;
                    const float4v b = *reinterpret_cast<const float4v*>(bias + L.boff[l] + f0);
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        float4v v = acc[j][p] + b;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                        half4v h, lo;
                        split4(v, h, lo);
                        const int o = (16 * p + lo4) * AP + f0;
                        *reinterpret_cast<half4v*>(&act[0][o]) = h;
                        *reinterpret_cast<half4v*>(&act[1][o]) = lo;
                    }
                }
            }
            // zero the K padding of the next layer (features npad .. kpad_next)
            const int kn = L.kpad[l + 1], nn = L.npad[l];
            if (kn > nn) {
                const int gw = (kn - nn) >> 2;
                for (int e = tid; e < TP * gw; e += NTH) {
                    const int px = e / gw, g4 = e - px * gw;
                    *reinterpret_cast<half4v*>(&act[0][px * AP + nn + 4 * g4]) = (half4v){0, 0, 0, 0};
                    *reinterpret_cast<half4v*>(&act[1][px * AP + nn + 4 * g4]) = (half4v){0, 0, 0, 0};
                }
            }
        } else {
            // last layer: bias + sigmoid, fp32, into the (now free) activation buffer as psf[px][PP]
            float* psf = reinterpret_cast<float*>(&act[0][0]);
            constexpr int PP = 132;                                             // floats per pixel row (>= 128, 16-B aligned rows)
            if (t0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && !t1) break;
                    const int f0 = 16 * (wave + NWV * j) + 4 * kg;
                    const float4v b = *reinterpret_cast<const float4v*>(bias + L.boff[l] + f0);
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        float4v v = acc[j][p] + b;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = 1.f / (1.f + expf(-v[i]));
                        *reinterpret_cast<float4v*>(&psf[(16 * p + lo4) * PP + f0]) = v;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: 4 threads per pixel; L1 normalisation (F.normalize eps 1e-12), then PSFs or gather ----
    {
        const float* psf = reinterpret_cast<const float*>(&act[0][0]);
        constexpr int PP = 132;
        const int px = tid >> 2, q = tid & 3;
        const long gp = p0 + px;
        const float* row = psf + px * PP;
        float sum = 0.f;
        for (int t = q; t < nout; t += 4) sum += row[t];
        sum += __shfl_xor(sum, 1, kWave);
        sum += __shfl_xor(sum, 2, kWave);
        const float inv = 1.f / fmaxf(sum, 1e-12f);
        if (gp < P) {
            if (mode == 0) {
                float* o = psf_out + gp * nout;
                for (int t = q; t < nout; t += 4) o[t] = row[t] * inv;
            } else {
                const long hw = (long)H * W;
                const long nimg = gp / hw;
                const int rem = (int)(gp - nimg * hw);
                const int y = rem / W, x = rem - y * W;
                const int pad = ks >> 1;
                for (int c = 0; c < C; ++c) {
                    const float* plane = img + (nimg * C + c) * hw;
                    float a = 0.f;
                    for (int t = q; t < nout; t += 4) {
                        const int u = t / ks, v = t - u * ks;
                        const int yy = min(max(y + u - pad, 0), H - 1), xx = min(max(x + v - pad, 0), W - 1);
                        a = fmaf(row[t], plane[(size_t)yy * W + xx], a);
                    }
                    a += __shfl_xor(a, 1, kWave);
                    a += __shfl_xor(a, 2, kWave);
                    if (q == 0) out[(nimg * C + c) * hw + rem] = a * inv;
                }
            }
        }
    }
}

}  // namespace pn
}  // namespace aadff

using namespace aadff;

extern "C" {

int aadff_psfnet_forward(const float* inp, long P, const void* wpack, const float* bias, int n_layers,
                         const int* in_features, const int* out_features, int mode, float* psf_out,
                         const float* img, float* out, int C, int H, int W, int ks, aadff_stream_t stream) {
    AADFF_CHECK_ARG(inp && wpack && bias && in_features && out_features, "psfnet_forward: NULL pointer");
    AADFF_CHECK_ARG(n_layers >= 1 && n_layers <= AADFF_PSFNET_MAX_LAYERS, "psfnet_forward: %d layers outside [1,%d]", n_layers, AADFF_PSFNET_MAX_LAYERS);
    AADFF_CHECK_ARG(P >= 0 && P < (1L << 40), "psfnet_forward: bad P");
    AADFF_CHECK_ARG(mode == 0 ? psf_out != nullptr : (img && out && C > 0 && H > 0 && W > 0), "psfnet_forward: missing output/image for mode %d", mode);
    pn::Layers L;
    std::memset(&L, 0, sizeof(L));
    L.n = n_layers;
    int woff = 0, boff = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int k = in_features[l], n = out_features[l];
        AADFF_CHECK_ARG(k >= 1 && k <= 256 && n >= 1 && n <= 256, "psfnet_forward: layer %d is %d -> %d, widths above 256 are not supported", l, k, n);
        AADFF_CHECK_ARG(l == 0 ? k == 4 : k == out_features[l - 1], "psfnet_forward: layer %d input width %d does not chain", l, k);
        L.kpad[l] = (k + 31) / 32 * 32;
        L.npad[l] = (n + 15) / 16 * 16;
        L.woff[l] = woff;
        L.boff[l] = boff;
        woff += (L.npad[l] / 16) * (L.kpad[l] / 32) * 2 * 64;              // uint4 units
        boff += L.npad[l];
    }
    const int nout = out_features[n_layers - 1];
    AADFF_CHECK_ARG(nout <= 128, "psfnet_forward: %d outputs (ks^2 <= 128)", nout);
    AADFF_CHECK_ARG(mode == 0 || ks * ks == nout, "psfnet_forward: ks %d does not match %d outputs", ks, nout);
    if (P == 0) return 0;
    AADFF_CHECK_ARG(mode == 0 || P % ((long)H * W) == 0, "psfnet_forward: P is not a whole number of images");
    const long nwg = (P + pn::TP - 1) / pn::TP;
    AADFF_CHECK_ARG(nwg < (1L << 31), "psfnet_forward: too many pixels");
    hipLaunchKernelGGL(pn::psfnet_fused_kernel, dim3((unsigned)nwg), dim3(pn::NTH), 0, (hipStream_t)stream, inp, P,
                       reinterpret_cast<const pn::uint4v*>(wpack), bias, L, nout, mode, psf_out, img, out, C, H, W, ks);
    AADFF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
