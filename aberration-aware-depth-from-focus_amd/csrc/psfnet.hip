// Fused PSF-surrogate network for gfx950 (MI355X): per-pixel (x, y, z, foc_z) -> MLP -> sigmoid -> L1-normalise
// -> either the PSFs themselves (PSFNet.pred) or the per-pixel PSF gather over the image (PSFNet.render), in ONE
// kernel.  Reference: deeplens/psfnet_arch.py:24-47 (MLP 4 -> 64 -> 256 -> 8 x 256 -> ks^2, ReLU, Sigmoid,
// F.normalize(p=1)), deeplens/psfnet.py:393-441 (render), deeplens/render_psf.py:76-107 (local_psf_render:
// replicate padding, no flip).  1.14 MFLOP per pixel against 28 bytes of HBM traffic: the activations never leave
// the CU (the unfused path writes and re-reads 484 B/pixel of PSFs plus every layer's activations).
//
// Workgroup = 128 pixels x 8 waves.  Activations live in LDS as two fp16 planes (hi, lo) [128][256]; every layer is
// the GEMM  out^T[feat][px] = sum_k W[feat][k] act[px][k]  on v_mfma_f32_16x16x32_f16 with the exact fp16 hi/lo
// operand split of conv.hip (hi*hi + hi*lo + lo*hi, fp32 accumulate: every product exact, dropped term 2^-22).
// A operand = weights, pre-packed by the host in fragment order (16 B per lane, streamed from L2, one k-step ahead);
// B operand = activations (ds_read_b128, 16-byte slots XOR-swizzled by the pixel: conflict-free); D^T puts 4 consecutive features of
// one pixel in a lane, so bias + ReLU + split + one 8-byte LDS store per plane write the next layer's input.
// Wave w owns feature tiles w, w + 8 (16 features each) for all 128 pixels: 64 accumulator registers.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace aadff {
namespace pn {

#ifdef AADFF_PN_TRACE
// Timeline instrumentation (tools/m2_timeline.py, build libaadff_pntrace.so): thread 0 of the first 4096 workgroups stamps the
// 100 MHz real-time counter at its start, after the input stage and, per layer, after its k-loop, after the barrier behind it,
// after the write-back and after the second barrier; then at the end of the epilogue.
__device__ unsigned long long* g_pn_trace = nullptr;
#define AADFF_PN_STAMP(slot) do { if (g_pn_trace && threadIdx.x == 0 && blockIdx.x < 4096) g_pn_trace[(size_t)blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AADFF_PN_STAMP(slot) do {} while (0)
#endif

constexpr int NWV = 8, NTH = 64 * NWV;
constexpr int AP = 256;                 // activation row pitch in halves; 16-byte slots XOR-swizzled by the pixel (swz)
constexpr int MAXL = AADFF_PSFNET_MAX_LAYERS;

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// In-kernel layer-0 input (PSFNet.render, deeplens/psfnet.py:424-437): row (n, slice, y, x) of the network input is
// (xs[x], ys[y], clamp((depth[n][y][x] - d_min) * inv_range, 0, 1), foc_z[n][slice]) -- the [N,S,H,W,4] coordinate tensor
// of the reference (168 MB for a 1024^2 x 10 stack) is never written.  depth == nullptr: rows come from `inp`.
struct Coord {
    const float* depth;     // [N][H][W] mm (< 0)
    const float* xs;        // [W]  torch.linspace(-1, 1, W)
    const float* ys;        // [H]  torch.linspace(1, -1, H)
    const float* foc_z;     // [N][S]  depth2z(foc_dist)
    float d_min, inv_range; // depth2z: (depth - d_min) * (1 / (d_max - d_min)), the form ATen evaluates tensor / scalar in
};

struct Layers {
    int n;                              // number of Linear layers
    int kpad[MAXL];                     // input features padded to 32
    int npad[MAXL];                     // output features padded to 16
    int woff[MAXL];                     // offset of the layer's packed weights, in uint4 (16 B) units
    int boff[MAXL];                     // offset of the layer's (padded) bias, in floats
};

// LDS offset (halves) of feature `f` of pixel `px`: the 16-byte slot index is XORed with px & 15.  The MFMA B-fragment
// read (lane = pixel px & 15, k-group kg: slot 4 s + kg) then touches every bank once per ds_read_b128 lane group
// (unswizzled with a padded pitch it is 2-way: SQ_LDS_BANK_CONFLICT was 49 % of the LDS cycles), and the write-back
// (lane = pixel, 4 features = half a slot) is 2-way instead of 4-way.
__device__ __forceinline__ int swz(int px, int f) { return px * AP + ((((f >> 3) ^ px) & 15) << 3 | (f & ~127) | (f & 7)); }

// x = hi + lo with hi = fp16(x) (RNE) and lo = fp16(x - hi): v_cvt_pk_f16_f32 for two hi halves, then one
// v_fma_mix{lo,hi}_f16 per lo half (fp16 hi * -1 + fp32 x, rounded once to fp16) -- the compiler emits convert-back,
// subtract and convert instead (20 instead of 10 VALU per four values of the layer write-back).
__device__ __forceinline__ void split4(float4v v, half4v& h, half4v& l) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v h01 = {(_Float16)v[0], (_Float16)v[1]}, h23 = {(_Float16)v[2], (_Float16)v[3]};
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l01) : "v"(__builtin_bit_cast(unsigned, h01)), "v"(v[0]), "v"(v[1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l23) : "v"(__builtin_bit_cast(unsigned, h23)), "v"(v[2]), "v"(v[3]));
    const half2v q01 = __builtin_bit_cast(half2v, l01), q23 = __builtin_bit_cast(half2v, l23);
    h = (half4v){h01[0], h01[1], h23[0], h23[1]};
    l = (half4v){q01[0], q01[1], q23[0], q23[1]};
}

// mode 0: psf_out[P][nout] (normalised PSFs); mode 1: out[N][C][H][W] = per-pixel PSF gather over img
// TP = pixels per workgroup (128: one workgroup per CU; 64: two, which overlap each other's write-back phases at
// twice the weight traffic from L2)
// SINGLE: fp16 single-pass mode (opt-in, `mlp_precision="fp16"`): operands are the fp16 roundings themselves, one MFMA per
// product instead of three, one activation plane (34 KB of LDS instead of 64 KB).  PSFs then carry ~5e-4 relative error —
// the level of torch's bf16 autocast and far below the surrogate's own fit error — at about a third of the matrix work.
template <int TP, bool SINGLE>
__global__ __launch_bounds__(NTH, TP == 128 ? 2 : 4) void psfnet_fused_kernel(const float* __restrict__ inp, long P, const uint4v* __restrict__ wpack,
                                                           const float* __restrict__ bias, Layers L, int nout, int mode,
                                                           float* __restrict__ psf_out, const float* __restrict__ img,
                                                           float* __restrict__ out, int C, int H, int W, int ks, int out_slices,
                                                           Coord coord, int* __restrict__ flags) {
    constexpr int PLANE = TP * AP;                                              // halves per activation plane
    // end of the kernel: fp32 PSFs [TP][132] and, behind them, the image window of the gather [3][11][TP + 10] (EPI_FLOATS)
    // The window only exists where it is free: in the fp32-equivalent mode the two activation planes (64 KB at TP = 64) dominate; in
    // the fp16 single-pass mode it would raise the workgroup's LDS from 33.8 to 43.6 KB and cost a resident workgroup per CU
    constexpr bool WINDOW = !SINGLE;
    constexpr int EPI_FLOATS = TP * 132 + (WINDOW ? 3 * 11 * (TP + 10) : 0);
    constexpr int ACT_HALVES = (SINGLE ? 1 : 2) * PLANE > EPI_FLOATS * 2 ? (SINGLE ? 1 : 2) * PLANE : EPI_FLOATS * 2;
    __shared__ __attribute__((aligned(16))) _Float16 act_raw[ACT_HALVES];      // [hi | lo] planes; reused for the fp32 PSFs [TP][132] at the end
    _Float16* const act[2] = {act_raw, act_raw + (SINGLE ? 0 : PLANE)};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, lo4 = lane & 15;
    const long p0 = (long)blockIdx.x * TP;
    AADFF_PN_STAMP(0);
    // gather mode, <= 3 channels, all TP pixels of the workgroup inside one row of one image (workgroup-uniform): the epilogue
    // stages the image window in LDS
    const bool row_tile = WINDOW && mode == 1 && C <= 3 && p0 + TP <= P && (int)((p0 % ((long)H * W)) % W) + TP <= W;

    // ---- layer-0 input: features 0..3, zero-padded to 32 ----
    for (int e = tid; e < TP * 8; e += NTH) {                                   // 8 groups of 4 halves per pixel and plane
        const int px = e >> 3, g4 = e & 7;
        float4v v = {0.f, 0.f, 0.f, 0.f};
        if (g4 == 0 && p0 + px < P) {
            if (coord.depth) {
                const long gp = p0 + px, hw = (long)H * W;
                const long ns = gp / hw;                                         // (n, slice) = n * S + slice
                const int rem = (int)(gp - ns * hw);
                const int yy = rem / W, xx = rem - yy * W;
                const long n = out_slices > 0 ? ns / out_slices : ns;
                const float z = fminf(fmaxf((coord.depth[n * hw + rem] - coord.d_min) * coord.inv_range, 0.f), 1.f);
                v = (float4v){coord.xs[xx], coord.ys[yy], z, coord.foc_z[ns]};
            } else {
                v = *reinterpret_cast<const float4v*>(inp + (p0 + px) * 4);
            }
        }
        half4v h, l;
        split4(v, h, l);
        *reinterpret_cast<half4v*>(&act[0][swz(px, 4 * g4)]) = h;
        if (!SINGLE) *reinterpret_cast<half4v*>(&act[1][swz(px, 4 * g4)]) = l;
    }
    __syncthreads();
    AADFF_PN_STAMP(1);

    constexpr int NPT = TP / 16;                                                // pixel tiles
    float4v acc[2][NPT];
    float amax = 0.f;                  // largest hidden activation this lane has split: above 65504 its fp16 hi half is inf
#pragma unroll 1
    for (int l = 0; l < L.n; ++l) {
        const int nks = L.kpad[l] >> 5, ntile = L.npad[l] >> 4;
        const bool t0 = wave < ntile, t1 = wave + NWV < ntile;                  // this wave's feature tiles: wave, wave + 8
        const bool last = l == L.n - 1;
        // accumulators start from the bias: D^T rows 4 kg + i of tile t are features 16 t + 4 kg + i
        auto kloop = [&](auto ntc) {
            constexpr int NTL = decltype(ntc)::value;                           // feature tiles of this wave in this layer: 1 or 2
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const float4v b = *reinterpret_cast<const float4v*>(bias + L.boff[l] + 16 * (wave + NWV * j) + 4 * kg);
#pragma unroll
                for (int p = 0; p < NPT; ++p) acc[j][p] = b;
            }
            // packed weights: [tile][k-step][plane][lane] x 16 B, streamed from L2 two k-steps ahead
            const uint4v* wq[2];
            uint4v ah[2], al[2], nh[2], nl[2];
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                wq[j] = wpack + L.woff[l] + ((size_t)(wave + NWV * j) * nks * 2) * 64 + lane;
                ah[j] = wq[j][0];
                if (!SINGLE) al[j] = wq[j][64];
                const int s1 = nks > 1 ? 1 : 0;
                nh[j] = wq[j][s1 * 128];
                if (!SINGLE) nl[j] = wq[j][s1 * 128 + 64];
            }
#pragma unroll 1
            for (int s = 0; s < nks; ++s) {
                half8v th[2], tl[2];
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    th[j] = __builtin_bit_cast(half8v, ah[j]);
                    ah[j] = nh[j];
                    if (!SINGLE) { tl[j] = __builtin_bit_cast(half8v, al[j]); al[j] = nl[j]; }
                }
                {
                    const int s2 = s + 2 < nks ? s + 2 : nks - 1;              // clamped: the tail re-reads the last step
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        nh[j] = wq[j][s2 * 128];
                        if (!SINGLE) nl[j] = wq[j][s2 * 128 + 64];
                    }
                }
                const int boffs = swz(lo4, 32 * s + 8 * kg);                    // (16 p + lo4) & 15 == lo4
#pragma unroll
                for (int p = 0; p < NPT; ++p) {
                    const half8v bh = *reinterpret_cast<const half8v*>(&act[0][16 * p * AP + boffs]);
                    if constexpr (SINGLE) {
#pragma unroll
                        for (int j = 0; j < NTL; ++j) acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[j], bh, acc[j][p], 0, 0, 0);
                    } else {
                        const half8v bl = *reinterpret_cast<const half8v*>(&act[1][16 * p * AP + boffs]);
#pragma unroll
                        for (int j = 0; j < NTL; ++j) {
                            acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[j], bh, acc[j][p], 0, 0, 0);
                            acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[j], bl, acc[j][p], 0, 0, 0);
                            acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl[j], bh, acc[j][p], 0, 0, 0);
                        }
                    }
                }
            }
        };
        if (t1) kloop(std::integral_constant<int, 2>{});
        else if (t0) kloop(std::integral_constant<int, 1>{});
        AADFF_PN_STAMP(2 + 4 * l);
        __syncthreads();                                                        // every wave is done reading this layer's input
        AADFF_PN_STAMP(3 + 4 * l);
        if (!last) {
            // D^T[feat = 16 tile + 4 kg + i][px = 16 p + lo4]: bias, ReLU, split, 8-byte stores
            if (t0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && !t1) break;
                    const int f0 = 16 * (wave + NWV * j) + 4 * kg;
#pragma unroll
                    for (int p = 0; p < NPT; ++p) {
                        float4v v = acc[j][p];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                        amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));      // two v_max3_f32
                        const int o = swz(16 * p + lo4, f0);
                        if constexpr (SINGLE) {
                            *reinterpret_cast<half4v*>(&act[0][o]) = (half4v){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        } else {
                            half4v h, lo;
                            split4(v, h, lo);
                            *reinterpret_cast<half4v*>(&act[0][o]) = h;
                            *reinterpret_cast<half4v*>(&act[1][o]) = lo;
                        }
                    }
                }
            }
            // zero the K padding of the next layer (features npad .. kpad_next)
            const int kn = L.kpad[l + 1], nn = L.npad[l];
            if (kn > nn) {
                const int gw = (kn - nn) >> 2;
                for (int e = tid; e < TP * gw; e += NTH) {
                    const int px = e / gw, g4 = e - px * gw;
                    *reinterpret_cast<half4v*>(&act[0][swz(px, nn + 4 * g4)]) = (half4v){0, 0, 0, 0};
                    if (!SINGLE) *reinterpret_cast<half4v*>(&act[1][swz(px, nn + 4 * g4)]) = (half4v){0, 0, 0, 0};
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
#pragma unroll
                    for (int p = 0; p < NPT; ++p) {
                        float4v v = acc[j][p];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = 1.f / (1.f + expf(-v[i]));
                        *reinterpret_cast<float4v*>(&psf[(16 * p + lo4) * PP + f0]) = v;
                    }
                }
            }
        }
        AADFF_PN_STAMP(4 + 4 * l);
        __syncthreads();
        AADFF_PN_STAMP(5 + 4 * l);
    }

    // A hidden activation beyond the fp16 range was split into (inf, -inf/NaN): the outputs of this workgroup are garbage.
    // Say so (flag bit 4) instead of returning it silently; !(amax <= max) also catches a NaN that survived the max chain.
    if (flags && !(amax <= 65504.f)) atomicOr(flags, 16);

    // ---- epilogue: 4 threads per pixel; L1 normalisation (F.normalize eps 1e-12), then PSFs or gather ----
    {
        const float* psf = reinterpret_cast<const float*>(&act[0][0]);
        constexpr int PP = 132;
        constexpr int TPP = NTH / TP;                                           // threads per pixel: 4 or 8
        const int px = tid / TPP, q = tid % TPP;
        const long gp = p0 + px;
        const float* row = psf + px * PP;
        float sum = 0.f;
        for (int t = q; t < nout; t += TPP) sum += row[t];
#pragma unroll
        for (int m = 1; m < TPP; m <<= 1) sum += __shfl_xor(sum, m, kWave);
        const float inv = 1.f / fmaxf(sum, 1e-12f);
        if (gp < P) {
            if (mode == 0) {
                float* o = psf_out + gp * nout;
                for (int t = q; t < nout; t += TPP) o[t] = row[t] * inv;
            } else if (!row_tile) {
                const long hw = (long)H * W;
                const long nimg = gp / hw;                  // image index; with out_slices = S it is (b, slice): b * S + slice
                const int rem = (int)(gp - nimg * hw);
                const long bimg = out_slices > 0 ? nimg / out_slices : nimg;
                const long sl = out_slices > 0 ? nimg - bimg * out_slices : 0;
                const long oslices = out_slices > 0 ? out_slices : 1;
                const int y = rem / W, x = rem - y * W;
                const int pad = ks >> 1;
                // thread q of the pixel takes tap columns q, q + TPP, ... of every tap row: clamped columns once
                constexpr int MAXV = (11 + TPP - 1) / TPP;                      // ks <= 11 (n_out <= 128)
                int xv[MAXV];
#pragma unroll
                for (int k = 0; k < MAXV; ++k) xv[k] = min(max(x + q + k * TPP - pad, 0), W - 1);
                for (int c = 0; c < C; ++c) {
                    const float* plane = img + (bimg * C + c) * hw;
                    float a = 0.f;
                    for (int u = 0; u < ks; ++u) {
                        const float* ir = plane + (size_t)min(max(y + u - pad, 0), H - 1) * W;
                        const float* pr = row + u * ks;
#pragma unroll
                        for (int k = 0; k < MAXV; ++k)
                            if (q + k * TPP < ks) a = fmaf(pr[q + k * TPP], ir[xv[k]], a);
                    }
#pragma unroll
                    for (int m = 1; m < TPP; m <<= 1) a += __shfl_xor(a, m, kWave);
                    if (q == 0) out[((bimg * C + c) * oslices + sl) * hw + rem] = a * inv;
                }
            }
        }
        if (row_tile) {
            // Round 4: the TP pixels of the workgroup are consecutive pixels of ONE image row, so their ks x (TP + ks - 1) x C
            // image window is staged in LDS once (coalesced rows, replicate-clamped) instead of every thread fetching its taps'
            // pixels from global memory (TP x ks^2 x C scattered loads: the epilogue went from 11.8 to 9.1 us of a workgroup's
            // 86 us, tools/m2_timeline.py; fetched into registers at the kernel's start and parked there: 6.8 us, but the five
            // extra registers cost the fp16 mode a wave per SIMD and the stack rate did not move - not kept).
            float* win = reinterpret_cast<float*>(&act[0][0]) + TP * PP;
            const long hw = (long)H * W;
            const long nimg = p0 / hw;
            const int rem0 = (int)(p0 - nimg * hw);
            const long bimg = out_slices > 0 ? nimg / out_slices : nimg;
            const long sl = out_slices > 0 ? nimg - bimg * out_slices : 0;
            const long oslices = out_slices > 0 ? out_slices : 1;
            const int y = rem0 / W, xs0 = rem0 - y * W;
            const int pad = ks >> 1, ww = TP + ks - 1;
            for (int e = tid; e < C * ks * ww; e += NTH) {
                const int cu = e / ww, col = e - cu * ww;
                const int c = cu / ks, u = cu - c * ks;
                const int yy = min(max(y + u - pad, 0), H - 1), xx = min(max(xs0 + col - pad, 0), W - 1);
                win[e] = img[(bimg * C + c) * hw + (size_t)yy * W + xx];
            }
            __syncthreads();
            float a[3] = {0.f, 0.f, 0.f};
            for (int t = q; t < nout; t += TPP) {
                const int u = t / ks, v = t - u * ks;
                const float wv = row[t];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < C) a[c] = fmaf(wv, win[(c * ks + u) * ww + px + v], a[c]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int m = 1; m < TPP; m <<= 1) a[c] += __shfl_xor(a[c], m, kWave);
                if (c < C && q == 0) out[((bimg * C + c) * oslices + sl) * hw + rem0 + px] = a[c] * inv;
            }
        }
    }
    AADFF_PN_STAMP(60);
}

}  // namespace pn
}  // namespace aadff

using namespace aadff;

extern "C" {

static int psfnet_launch(const float* inp, long P, const void* wpack, const float* bias, int n_layers,
                         const int* in_features, const int* out_features, int mode, float* psf_out,
                         const float* img, float* out, int C, int H, int W, int ks, int out_slices, pn::Coord coord,
                         int precision, int* flags_or_null, aadff_stream_t stream) {
    AADFF_CHECK_ARG(precision == 0 || precision == 1, "psfnet_forward: precision %d (0 = fp32-equivalent split, 1 = fp16 single pass)", precision);
    AADFF_CHECK_ARG((inp || coord.depth) && wpack && bias && in_features && out_features, "psfnet_forward: NULL pointer");
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
    AADFF_CHECK_ARG(out_slices >= 0 && (mode == 0 || out_slices == 0 || (P / ((long)H * W)) % out_slices == 0),
                    "psfnet_forward: %ld images are not a whole number of %d-slice stacks", P / std::max(1L, (long)H * W), out_slices);
    int tp = 64;                       // measured at 1024^2: 64 -> 3.09 ms, 128 -> 3.46 ms
    if (const char* e = getenv("AADFF_PSFNET_TP")) tp = atoi(e) == 128 ? 128 : 64;
    const long nwg = (P + tp - 1) / tp;
    AADFF_CHECK_ARG(nwg < (1L << 31), "psfnet_forward: too many pixels");
#define AADFF_PN_LAUNCH(TPV, SV) hipLaunchKernelGGL((pn::psfnet_fused_kernel<TPV, SV>), dim3((unsigned)nwg), dim3(pn::NTH), 0, (hipStream_t)stream, inp, P, \
        reinterpret_cast<const pn::uint4v*>(wpack), bias, L, nout, mode, psf_out, img, out, C, H, W, ks, out_slices, coord, flags_or_null)
    if (tp == 128) { if (precision) AADFF_PN_LAUNCH(128, true); else AADFF_PN_LAUNCH(128, false); }
    else { if (precision) AADFF_PN_LAUNCH(64, true); else AADFF_PN_LAUNCH(64, false); }
#undef AADFF_PN_LAUNCH
    AADFF_CHECK_LAUNCH();
    return 0;
}

#ifdef AADFF_PN_TRACE
int aadff_pn_trace_buffer(unsigned long long* dev_buf) {      // not part of the ABI: instrumentation builds only
    AADFF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(aadff::pn::g_pn_trace), &dev_buf, sizeof(dev_buf)));
    return 0;
}
#endif

int aadff_psfnet_forward(const float* inp, long P, const void* wpack, const float* bias, int n_layers,
                         const int* in_features, const int* out_features, int mode, float* psf_out,
                         const float* img, float* out, int C, int H, int W, int ks, int out_slices, int precision,
                         int* flags_or_null, aadff_stream_t stream) {
    AADFF_CHECK_ARG(inp, "psfnet_forward: NULL input rows");
    return psfnet_launch(inp, P, wpack, bias, n_layers, in_features, out_features, mode, psf_out, img, out, C, H, W, ks,
                         out_slices, pn::Coord{}, precision, flags_or_null, stream);
}

int aadff_psfnet_render_rgbd(const float* depth, const float* xs, const float* ys, const float* foc_z, float d_min,
                             float inv_range, long N, int S, const void* wpack, const float* bias, int n_layers,
                             const int* in_features, const int* out_features, const float* img, float* out, int C, int H,
                             int W, int ks, int precision, int* flags_or_null, aadff_stream_t stream) {
    AADFF_CHECK_ARG(depth && xs && ys && foc_z, "psfnet_render_rgbd: NULL pointer");
    AADFF_CHECK_ARG(N >= 0 && S >= 1 && H > 0 && W > 0, "psfnet_render_rgbd: bad sizes N=%ld S=%d", N, S);
    const pn::Coord c{depth, xs, ys, foc_z, d_min, inv_range};
    return psfnet_launch(nullptr, N * S * (long)H * W, wpack, bias, n_layers, in_features, out_features, 1, nullptr, img, out,
                         C, H, W, ks, S, c, precision, flags_or_null, stream);
}

}  // extern "C"
