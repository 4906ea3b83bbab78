// bf16 GEMM with fused epilogues for the PSF-network fit step (reference: the autograd graph of deeplens/psfnet.py:94-106
// over deeplens/psfnet_arch.py:24-47: Linear + ReLU chain, MSE loss).
//
// The fit step is 33 GEMMs of at most 256 x 256 x 256 on a batch of 128 rows: through hipBLASLt each is a launch of a
// 256 x 256-tile kernel on ONE workgroup (8-10 us), plus separate ReLU / mask / bias-gradient kernels.  Here ONE kernel form
//     out[b][a] = sum_c A[a][c] B[b][c]          (both operands contiguous along the contraction: "NT")
// covers all three GEMMs of a layer when every activation / gradient / weight is also kept transposed:
//     forward   Y [px][n] = relu(X [px][:] . W [n][:] + bias[n])      A = W,    B = X        -> Y and Y^T
//     dX        dZ[px][k] = (dZ'[px][:] . W^T[k][:]) * (Y[px][k] > 0) A = W^T,  B = dZ'      -> dZ, dZ^T, db += column sums
//     dW        dW[n][k]  = dZ^T[n][:] . X^T[k][:]                    A = X^T,  B = dZ^T     -> fp32 gradient
// v_mfma_f32_16x16x32_bf16, fp32 accumulate.  A wave owns 16 A-rows x 32 B-rows and keeps ALL its operand fragments
// (contraction <= 256: 8 k-steps x 3 x 16 B) in flight at once: the problem is latency, not bandwidth.  The accumulator
// lane layout (4 consecutive A-rows for one B-row) makes out[b][a..a+3] one 8-byte (bf16) or 16-byte (fp32) store.
// Leading dimensions are multiples of 8 elements and the padding is zero.
#include <algorithm>
#include <cstdint>
#include "common.h"

namespace aadff {
namespace fit {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

enum { EPI_FWD = 0, EPI_FWD_RELU = 1, EPI_DX = 2, EPI_DW = 3 };

struct GemmArgs {
    const uint16_t* A; int lda, na;          // [na][lda] bf16
    const uint16_t* B; int ldb, nb;          // [nb][ldb] bf16
    int nc;                                  // contraction length (<= 256)
    void* out; int ld_out;                   // [nb][ld_out]: bf16 (FWD, DX) or fp32 (DW)
    uint16_t* outT; int ld_outT;             // [na][ld_outT] bf16 transposed copy (FWD, DX) or null
    const uint16_t* bias;                    // FWD: bf16 [na]
    const uint16_t* mask; int ld_mask;       // DX: forward output of the previous layer [nb][ld_mask] (> 0 passes)
    float* dbias;                            // DX: fp32 [na], += column sums (atomics)
};

template <int EPI, int KS>
__device__ __forceinline__ void gemm_nt_tile(const GemmArgs& g, int bx, int by) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 4, lo4 = lane & 15;
    const int a0 = bx * 32 + (wave & 1) * 16, b0 = by * 64 + (wave >> 1) * 32;
    if (a0 >= g.na || b0 >= g.nb) return;
    // rows past the end are clamped (their results are never stored); columns past the leading dimension read column 0
    // and are zeroed, so every load is unconditional and all 3 KS of them are in flight together
    const uint16_t* pa = g.A + (size_t)min(a0 + lo4, g.na - 1) * g.lda;
    const uint16_t* pb0 = g.B + (size_t)min(b0 + lo4, g.nb - 1) * g.ldb;
    const uint16_t* pb1 = g.B + (size_t)min(b0 + 16 + lo4, g.nb - 1) * g.ldb;
    uint4v af[KS], bf0[KS], bf1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * kg;
        const int ca = c + 8 <= g.lda ? c : 0, cb = c + 8 <= g.ldb ? c : 0;
        af[s] = *reinterpret_cast<const uint4v*>(pa + ca);
        bf0[s] = *reinterpret_cast<const uint4v*>(pb0 + cb);
        bf1[s] = *reinterpret_cast<const uint4v*>(pb1 + cb);
    }
    float4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * kg;
        const unsigned keep = c + 8 <= g.lda ? 0xffffffffu : 0u;            // a zero A fragment zeroes the product
        const uint4v am = {af[s][0] & keep, af[s][1] & keep, af[s][2] & keep, af[s][3] & keep};
        const bf16x8 a = __builtin_bit_cast(bf16x8, am);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, bf0[s]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, bf1[s]), acc1, 0, 0, 0);
    }
    // lane holds out[b = b0 + 16 t + lo4][a = a0 + 4 kg + i], i = 0..3
    const int a = a0 + 4 * kg;
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float4v v = t ? acc1 : acc0;
        const int b = b0 + 16 * t + lo4;
        const bool inb = b < g.nb && a < g.na;                                  // na is a multiple of 4 in every use
        if constexpr (EPI == EPI_FWD || EPI == EPI_FWD_RELU) {
            if (inb) {
                uint16_t h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = v[i] + bf2f(g.bias[a + i]);
                    if (EPI == EPI_FWD_RELU) x = fmaxf(x, 0.f);
                    h[i] = f2bf(x);
                    if (g.outT) g.outT[(size_t)(a + i) * g.ld_outT + b] = h[i];
                }
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(g.out) + (size_t)b * g.ld_out + a) =
                    make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            }
        } else if constexpr (EPI == EPI_DX) {
            if (inb) {
                const uint2 m = *reinterpret_cast<const uint2*>(g.mask + (size_t)b * g.ld_mask + a);
                const uint16_t mk[4] = {(uint16_t)(m.x & 0xffffu), (uint16_t)(m.x >> 16), (uint16_t)(m.y & 0xffffu), (uint16_t)(m.y >> 16)};
                uint16_t h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = bf2f(mk[i]) > 0.f ? v[i] : 0.f;
                    h[i] = f2bf(x);
                    colsum[i] += bf2f(h[i]);                                    // the bias gradient sums what dW will see
                    if (g.outT) g.outT[(size_t)(a + i) * g.ld_outT + b] = h[i];
                }
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(g.out) + (size_t)b * g.ld_out + a) =
                    make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            }
        } else {
            if (inb) *reinterpret_cast<float4v*>(static_cast<float*>(g.out) + (size_t)b * g.ld_out + a) = v;
        }
    }
    if constexpr (EPI == EPI_DX) {
        if (g.dbias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = colsum[i];
                s += __shfl_xor(s, 1, kWave); s += __shfl_xor(s, 2, kWave); s += __shfl_xor(s, 4, kWave); s += __shfl_xor(s, 8, kWave);
                if (lo4 == 0 && a + i < g.na) unsafeAtomicAdd(g.dbias + a + i, s);
            }
        }
    }
}

template <int EPI, int KS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) { gemm_nt_tile<EPI, KS>(g, blockIdx.x, blockIdx.y); }

// Backward of one layer in one launch: blockIdx.z = 0 computes dW (fp32), 1 computes dX of the layer below (they share dZ).
template <int KSW, int KSX>
__global__ __launch_bounds__(256) void layer_bwd_kernel(GemmArgs dw, GemmArgs dx) {
    if (blockIdx.z == 0) gemm_nt_tile<EPI_DW, KSW>(dw, blockIdx.x, blockIdx.y);
    else gemm_nt_tile<EPI_DX, KSX>(dx, blockIdx.x, blockIdx.y);
}

// Network input of a batch: fp32 [B][K] -> bf16 X [B][ld] and X^T [K][ldT].
__global__ __launch_bounds__(256) void fit_input_kernel(const float* __restrict__ inp, uint16_t* __restrict__ x, int ld,
                                                        uint16_t* __restrict__ xT, int ldT, int B, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int r = i / K, c = i - r * K;
    const uint16_t h = f2bf(inp[i]);
    x[(size_t)r * ld + c] = h;
    xT[(size_t)c * ldT + r] = h;
}


// ------------------------------------------------------------------------------------
// The whole forward + loss + dX chain of the fit step in ONE launch.  A row of the batch never meets another row before
// dW, so a workgroup takes 16 rows (the MFMA N extent) through all layers: activations stay in LDS (they are the ReLU
// masks of the backward pass), the weights stream from L2 as MFMA A fragments (W for the forward, W^T for dX), and the
// only things written to HBM are what dW needs - every X_l^T and dZ_l^T - plus the bias gradients (one atomic per
// column and workgroup).  22 dependent launches of ~6 us become one kernel of ~22 layer passes.  Eight waves split the
// feature tiles of a layer, two tiles per wave and pass with all their weight fragments in flight together.
// LDS: X_l [16][up32(width_l) + 8] bf16 for l = 0..L (the +8 halves shift consecutive rows by one 16-byte slot:
// the ds_read_b128 of 16 rows x one k-group touches every bank once), two dZ buffers [16][up32(max width) + 8].
// ------------------------------------------------------------------------------------
constexpr int kChainThreads = 512, kChainWaves = 8, kChainRows = 16;
__host__ __device__ inline int chain_pitch(int w) { return (w + 31) / 32 * 32 + 8; }
__host__ __device__ inline int chain_width(const aadff_fit_net& a, int l) { return l == 0 ? a.k[0] : a.n[l - 1]; }

// The A operand of one layer pass for this wave: weight rows of feature tiles `wave` and `wave + 8` (widths <= 256: at most
// 16 tiles), all 8 k-steps.  Loaded for the NEXT pass before the epilogue stores of the current one are issued: gfx9 counts
// loads and stores in one in-order counter, so loads issued after the stores could not be waited for without the stores.
struct ChainFrag { uint4v a0[8], a1[8]; uint2 bias0, bias1; };     // + the forward pass's bias values of the two tiles

// M is stored in MFMA fragment order: [tile][k-step][lane] x 16 bytes (zero padded to whole tiles / k-steps), so one
// wave-instruction reads 1 KiB of contiguous memory: a workgroup streams every weight once per step, and one CU takes in
// 66 GB/s this way against 35 GB/s from the rows of a row-major matrix (tools/cu_stream_probe.hip, cold caches).
__device__ __forceinline__ void chain_load(ChainFrag& f, const uint16_t* M, int ntiles, int ksteps, int wave, int lane) {
    const uint4v* m0 = reinterpret_cast<const uint4v*>(M) + (size_t)min(wave, ntiles - 1) * ksteps * 64 + lane;
    const uint4v* m1 = reinterpret_cast<const uint4v*>(M) + (size_t)min(wave + kChainWaves, ntiles - 1) * ksteps * 64 + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int ss = min(s, ksteps - 1);                               // steps past the contraction re-read the last one; never used
        f.a0[s] = m0[ss * 64];
        f.a1[s] = m1[ss * 64];
    }
}

// acc{0,1} = A{0,1}[feat][c] . B[row][c] over the contraction, B rows from LDS (pitch pb).
__device__ __forceinline__ void chain_mma(const ChainFrag& f, const uint16_t* Bl, int pb, int ksteps, int lo4, int kg,
                                          float4v& acc0, float4v& acc1) {
    acc0 = acc1 = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < ksteps) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(Bl + lo4 * pb + 32 * s + 8 * kg));
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.a0[s]), b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.a1[s]), b, acc1, 0, 0, 0);
        }
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for every global store in flight
// (the X^T / dZ^T rows nobody reads in this kernel), once per layer.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(kChainThreads) void fit_chain_kernel(aadff_fit_net a, AdamwSchedule sch) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 4, lo4 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.n_layers, B = a.batch, r0 = blockIdx.x * kChainRows, ldb = a.ld_batch;
    const uint16_t* p16 = static_cast<const uint16_t*>(a.param_bf16);
    uint16_t* scr = static_cast<uint16_t*>(a.scratch_bf16);
    // pass p = 0..L-1: forward layer p (A = W_p); pass p = L..2L-2: dX of layer i = 2L-1-p (A = W_i^T).  The weights depend on
    // nothing computed here, so the A fragments are fetched TWO passes ahead into two register sets (one L2/HBM round trip
    // is longer than one pass).
    const int P = 2 * L - 1;
    auto fetch = [&](int q, ChainFrag& f) {
        if (q < L) {
            const int n4 = (a.n[q] + 3) & ~3;
            chain_load(f, p16 + a.off_w[q], (a.n[q] + 15) >> 4, (a.k[q] + 31) >> 5, wave, lane);
            // fetched with the weights: a load issued later (in the epilogue) could only be waited for together with every
            // prefetch issued before it (one in-order counter)
            const uint16_t* bias = p16 + a.off_b[q];
            f.bias0 = *reinterpret_cast<const uint2*>(bias + min(wave * 16 + 4 * kg, n4 - 4));
            f.bias1 = *reinterpret_cast<const uint2*>(bias + min((wave + kChainWaves) * 16 + 4 * kg, n4 - 4));
        } else if (q < P) { const int i = 2 * L - 1 - q; chain_load(f, p16 + a.off_wt[i], (a.k[i] + 15) >> 4, (a.n[i] + 31) >> 5, wave, lane); }
    };
    ChainFrag f0, f1;
    fetch(0, f0);
    fetch(1, f1);
    int xtotal = 0, maxw = 0;
    for (int l = 0; l <= L; ++l) {
        const int w = chain_width(a, l);
        xtotal += kChainRows * chain_pitch(w);
        maxw = w > maxw ? w : maxw;
    }
    const int PZ = chain_pitch(maxw);
    uint16_t* dzbuf[2] = {lds + xtotal, lds + xtotal + kChainRows * PZ};
    {   // zero everything once: the padding columns are contraction inputs
        const int n16 = (xtotal + 2 * kChainRows * PZ) / 8;
        uint4v* z = reinterpret_cast<uint4v*>(lds);
        for (int i = tid; i < n16; i += kChainThreads) z[i] = (uint4v){0u, 0u, 0u, 0u};
    }
    lds_barrier();
    {   // network input of the 16 rows: fp32 -> bf16 into X_0 and X_0^T
        const int K0 = a.k[0], p0 = chain_pitch(K0);
        uint16_t* xt0 = scr + a.off_xt[0];
        for (int i = tid; i < kChainRows * K0; i += kChainThreads) {
            const int row = i / K0, c = i - row * K0, g = r0 + row;
            if (g < B) {
                const uint16_t h = f2bf(a.inp[(size_t)g * K0 + c]);
                lds[row * p0 + c] = h;
                xt0[(size_t)c * ldb + g] = h;
            }
        }
    }
    // ---- forward
    int xoff = 0;
    auto forward = [&](int l, ChainFrag& f) {
        lds_barrier();
        const int K = a.k[l], N = a.n[l], N4 = (N + 3) & ~3;
        const int pl = chain_pitch(K), pn = chain_pitch(N);
        uint16_t* Xn = lds + xoff + kChainRows * pl;
        uint16_t* xt = scr + a.off_xt[l + 1 < L ? l + 1 : 0];
        const bool last = l == L - 1;
        float4v acc[2];
        chain_mma(f, lds + xoff, pl, (K + 31) >> 5, lo4, kg, acc[0], acc[1]);
        const uint2 bias2[2] = {f.bias0, f.bias1};
        fetch(l + 2, f);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int feat = (wave + t * kChainWaves) * 16 + 4 * kg;
            if (feat < N4) {
                const uint2 bb = bias2[t];
                const uint16_t bh[4] = {(uint16_t)(bb.x & 0xffffu), (uint16_t)(bb.x >> 16), (uint16_t)(bb.y & 0xffffu), (uint16_t)(bb.y >> 16)};
                uint16_t h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = acc[t][i] + bf2f(bh[i]);
                    if (!last) x = fmaxf(x, 0.f);
                    h[i] = f2bf(x);
                }
                *reinterpret_cast<uint2*>(Xn + lo4 * pn + feat) =
                    make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                if (!last && r0 + lo4 < B) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xt[(size_t)(feat + i) * ldb + r0 + lo4] = h[i];
                }
            }
        }
        xoff += kChainRows * pl;
    };
    for (int l = 0; l < L; l += 2) {
        forward(l, f0);
        if (l + 1 < L) forward(l + 1, f1);
    }
    lds_barrier();
    // ---- head: sigmoid, L1 normalise, d MSE / dz (the arithmetic of fit_head_kernel), two rows per wave
    const int NL = a.n[L - 1], pL = chain_pitch(NL);
    {
        const uint16_t* Z = lds + xoff;
        uint16_t* dzt = scr + a.off_dzt[L];
        if (blockIdx.x == 0 && tid == 0 && sch.step) adamw_prepare(sch);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, g = r0 + row;
            const bool valid = g < B;
            float sg[2], tg[2], S = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = lane + 64 * k;
                sg[k] = 0.f; tg[k] = 0.f;
                if (c < NL) {
                    sg[k] = bf2f(f2bf(1.f / (1.f + __expf(-bf2f(Z[row * pL + c])))));
                    tg[k] = valid ? a.target[(size_t)g * NL + c] : 0.f;
                    S += sg[k];
                }
            }
            S = fmaxf(wave_sum(S), 1e-12f);
            const float inv = 1.f / S, scale = 2.f / ((float)B * (float)NL);
            float gr[2], pr[2], dot = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                pr[k] = sg[k] * inv;
                gr[k] = scale * (pr[k] - tg[k]);
                dot += gr[k] * pr[k];
            }
            dot = wave_sum(dot);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = lane + 64 * k;
                if (c < NL) {
                    const uint16_t h = valid ? f2bf((gr[k] - dot) * inv * sg[k] * (1.f - sg[k])) : (uint16_t)0;
                    dzbuf[0][row * PZ + c] = h;
                    if (valid) {
                        a.pred[(size_t)g * NL + c] = pr[k];
                        dzt[(size_t)c * ldb + g] = h;
                    }
                }
            }
        }
    }
    lds_barrier();
    for (int c = tid; c < NL; c += kChainThreads) {           // bias gradient of the last layer: one atomic per column and workgroup
        float s = 0.f;
#pragma unroll
        for (int row = 0; row < kChainRows; ++row) s += bf2f(dzbuf[0][row * PZ + c]);
        unsafeAtomicAdd(a.grad + a.off_gb[L - 1] + c, s);
    }
    // ---- dX chain: dZ_{i-1} = (dZ_i . W_i) * (X_i > 0) for i = L-1 .. 1 (X_i = input of layer i, in LDS since the forward)
    int cur = 0;
    auto backward = [&](int q, ChainFrag& f) {
        const int i = 2 * L - 1 - q;
        const int K = a.k[i], N = a.n[i];
        const int pi = chain_pitch(K);
        xoff -= kChainRows * pi;                                  // X_i
        const uint16_t* Xi = lds + xoff;
        uint16_t* dZn = dzbuf[cur ^ 1];
        uint16_t* dzt = scr + a.off_dzt[i];
        float* gb = a.grad + a.off_gb[i - 1];
        float4v acc[2];
        chain_mma(f, dzbuf[cur], PZ, (N + 31) >> 5, lo4, kg, acc[0], acc[1]);
        fetch(q + 2, f);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kk = (wave + t * kChainWaves) * 16 + 4 * kg;
            const bool in = kk < K;                              // K is a multiple of 4
            float cs[4] = {0.f, 0.f, 0.f, 0.f};
            if (in) {
                const uint2 m = *reinterpret_cast<const uint2*>(Xi + lo4 * pi + kk);
                const uint16_t mk[4] = {(uint16_t)(m.x & 0xffffu), (uint16_t)(m.x >> 16), (uint16_t)(m.y & 0xffffu), (uint16_t)(m.y >> 16)};
                uint16_t h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[q] = f2bf(bf2f(mk[q]) > 0.f ? acc[t][q] : 0.f);
                    cs[q] = bf2f(h[q]);
                }
                *reinterpret_cast<uint2*>(dZn + lo4 * PZ + kk) =
                    make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                if (r0 + lo4 < B) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dzt[(size_t)(kk + q) * ldb + r0 + lo4] = h[q];
                }
            }
            if ((wave + t * kChainWaves) * 16 < K) {              // wave-uniform: this tile exists
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s = cs[q];
                    s += __shfl_xor(s, 1, kWave); s += __shfl_xor(s, 2, kWave); s += __shfl_xor(s, 4, kWave); s += __shfl_xor(s, 8, kWave);
                    if (lo4 == 0 && in) unsafeAtomicAdd(gb + kk + q, s);
                }
            }
        }
        cur ^= 1;
        lds_barrier();
    };
    // pass q uses register set q & 1 (the forward ended on set (L-1) & 1)
    for (int q = L; q < P; ++q) {
        if (q & 1) backward(q, f1);
        else backward(q, f0);
    }
}

// dW of every layer in one launch: blockIdx.z = layer, dW_l [n][k] = dZ_{l+1}^T [n][:B] . X_l^T [k][:B]  (fp32, plain stores).
template <int KSW>
__global__ __launch_bounds__(256) void fit_dw_all_kernel(aadff_fit_net a) {
    const int l = blockIdx.z;
    const uint16_t* scr = static_cast<const uint16_t*>(a.scratch_bf16);
    GemmArgs g{scr + a.off_xt[l], a.ld_batch, a.k[l], scr + a.off_dzt[l + 1], a.ld_batch, a.n[l], a.batch,
               a.grad + a.off_gw[l], a.k[l], nullptr, 0, nullptr, nullptr, 0, nullptr};
    gemm_nt_tile<EPI_DW, KSW>(g, blockIdx.x, blockIdx.y);
}

// Head of the network + loss gradient for the fused fit step: per row  s = sigmoid(z), pred = s / max(sum s, 1e-12),
// dz = d/dz mean((pred - target)^2)  (see aadff_psfnet_head_loss_grad), written as dZ [B][ld] AND dZ^T [n][ldT] for the
// dW / dX GEMMs, with the bias gradient (column sums of the bf16-rounded dz) added into `dbias` by atomics.
__global__ __launch_bounds__(64) void fit_head_kernel(const uint16_t* __restrict__ z, int ld_z, const float* __restrict__ target,
                                                      float* __restrict__ pred, uint16_t* __restrict__ dz, int ld_dz,
                                                      uint16_t* __restrict__ dzT, int ld_dzT, float* __restrict__ dbias, int B, int N,
                                                      AdamwSchedule sch) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (sch.step && r == 0 && lane == 0) adamw_prepare(sch);     // the optimiser kernel runs later in the same stream
    float s[2], t[2], S = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        s[k] = 0.f; t[k] = 0.f;
        if (c < N) {
            s[k] = bf2f(f2bf(1.f / (1.f + __expf(-bf2f(z[(size_t)r * ld_z + c])))));
            t[k] = target[(size_t)r * N + c];
            S += s[k];
        }
    }
    S = fmaxf(wave_sum(S), 1e-12f);
    const float inv = 1.f / S, scale = 2.f / ((float)B * (float)N);
    float g[2], p[2], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        p[k] = s[k] * inv;
        g[k] = scale * (p[k] - t[k]);
        dot += g[k] * p[k];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        if (c < N) {
            pred[(size_t)r * N + c] = p[k];
            const uint16_t h = f2bf((g[k] - dot) * inv * s[k] * (1.f - s[k]));
            dz[(size_t)r * ld_dz + c] = h;
            dzT[(size_t)c * ld_dzT + r] = h;
            unsafeAtomicAdd(dbias + c, bf2f(h));
        }
    }
}

// AdamW + cosine schedule on the flat fp32 parameters (see optim.hip) for the fused fit step: fp32 gradients that are
// ZEROED after use (the bias gradients are accumulated by atomics), and the bf16 copies the GEMMs read — row-major W and
// transposed W^T, both with padded leading dimensions — refreshed through per-element destination maps.
__global__ __launch_bounds__(256) void fit_adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, uint16_t* __restrict__ p16, const int* __restrict__ dst,
                                                        const int* __restrict__ dstT, long n, const float* __restrict__ scal, float b1,
                                                        float b2, float eps) {
    const float step_size = scal[0], bc2s = scal[1], decay = scal[2];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        g[i] = 0.f;
        float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        pi -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
        p[i] = pi; m[i] = mi; v[i] = vi;
        const uint16_t h = f2bf(pi);
        p16[dst[i]] = h;
        const int dt = dstT[i];
        if (dt >= 0) p16[dt] = h;
    }
}

}  // namespace fit
}  // namespace aadff

using namespace aadff;

extern "C" int aadff_fit_gemm_nt(const void* A, int lda, int na, const void* B, int ldb, int nb, int nc, int epilogue, void* out,
                                 int ld_out, void* outT, int ld_outT, const void* bias, const void* mask, int ld_mask,
                                 float* dbias, aadff_stream_t stream) {
    AADFF_CHECK_ARG(A && B && out, "fit_gemm_nt: NULL pointer");
    AADFF_CHECK_ARG(na > 0 && nb > 0 && nc > 0 && nc <= 256, "fit_gemm_nt: sizes na=%d nb=%d nc=%d (contraction <= 256)", na, nb, nc);
    AADFF_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ld_out % 4 == 0 && na % 4 == 0, "fit_gemm_nt: leading dimensions must be multiples of 8 (operands) / 4 (output), na a multiple of 4");
    AADFF_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "fit_gemm_nt: epilogue %d", epilogue);
    AADFF_CHECK_ARG(epilogue > 1 || bias, "fit_gemm_nt: forward epilogue needs a bias");
    AADFF_CHECK_ARG(epilogue != 2 || mask, "fit_gemm_nt: dX epilogue needs the forward output as mask");
    fit::GemmArgs g{static_cast<const uint16_t*>(A), lda, na, static_cast<const uint16_t*>(B), ldb, nb, nc, out, ld_out,
                    static_cast<uint16_t*>(outT), ld_outT, static_cast<const uint16_t*>(bias), static_cast<const uint16_t*>(mask), ld_mask, dbias};
    const dim3 grid((na + 31) / 32, (nb + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    const int ksteps = (nc + 31) / 32;
#define AADFF_FIT_LAUNCH(E, K) hipLaunchKernelGGL((fit::gemm_nt_kernel<E, K>), grid, dim3(256), 0, st, g)
#define AADFF_FIT_K(E)                                   \
    do {                                                 \
        if (ksteps <= 1) AADFF_FIT_LAUNCH(E, 1);         \
        else if (ksteps <= 2) AADFF_FIT_LAUNCH(E, 2);    \
        else if (ksteps <= 4) AADFF_FIT_LAUNCH(E, 4);    \
        else AADFF_FIT_LAUNCH(E, 8);                     \
    } while (0)
    switch (epilogue) {
        case 0: AADFF_FIT_K(fit::EPI_FWD); break;
        case 1: AADFF_FIT_K(fit::EPI_FWD_RELU); break;
        case 2: AADFF_FIT_K(fit::EPI_DX); break;
        default: AADFF_FIT_K(fit::EPI_DW);
    }
#undef AADFF_FIT_K
#undef AADFF_FIT_LAUNCH
    AADFF_CHECK_LAUNCH();
    return 0;
}

static int ksteps_class(int nc) { const int k = (nc + 31) / 32; return k <= 1 ? 1 : k <= 2 ? 2 : k <= 4 ? 4 : 8; }

extern "C" int aadff_fit_layer_bwd(const void* xT_prev, int ld_xT, int k, const void* dzT, int ld_dzT, int n, int batch, float* dW,
                                   const void* wT, int ld_wT, const void* dz, int ld_dz, const void* x_prev, int ld_x, void* dz_prev,
                                   int ld_dzp, void* dzT_prev, int ld_dzTp, float* dbias_prev, aadff_stream_t stream) {
    AADFF_CHECK_ARG(xT_prev && dzT && dW && k > 0 && n > 0 && batch > 0 && batch <= 256 && n <= 256, "fit_layer_bwd: bad arguments (batch, n <= 256)");
    AADFF_CHECK_ARG(ld_xT % 8 == 0 && ld_dzT % 8 == 0 && k % 4 == 0, "fit_layer_bwd: leading dimensions must be multiples of 8, k of 4");
    const bool with_dx = wT != nullptr;
    AADFF_CHECK_ARG(!with_dx || (dz && x_prev && dz_prev && ld_wT % 8 == 0 && ld_dz % 8 == 0 && ld_dzp % 4 == 0), "fit_layer_bwd: dX operands");
    fit::GemmArgs dw{static_cast<const uint16_t*>(xT_prev), ld_xT, k, static_cast<const uint16_t*>(dzT), ld_dzT, n, batch, dW, k,
                     nullptr, 0, nullptr, nullptr, 0, nullptr};
    fit::GemmArgs dx{static_cast<const uint16_t*>(wT), ld_wT, k, static_cast<const uint16_t*>(dz), ld_dz, batch, n, dz_prev, ld_dzp,
                     static_cast<uint16_t*>(dzT_prev), ld_dzTp, nullptr, static_cast<const uint16_t*>(x_prev), ld_x, dbias_prev};
    const int gy = with_dx ? std::max((n + 63) / 64, (batch + 63) / 64) : (n + 63) / 64;
    const dim3 grid((k + 31) / 32, gy, with_dx ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    const int kw = ksteps_class(batch), kx = ksteps_class(n);
#define AADFF_BWD_X(KW)                                                                                          \
    do {                                                                                                         \
        if (kx == 1) hipLaunchKernelGGL((fit::layer_bwd_kernel<KW, 1>), grid, dim3(256), 0, st, dw, dx);         \
        else if (kx == 2) hipLaunchKernelGGL((fit::layer_bwd_kernel<KW, 2>), grid, dim3(256), 0, st, dw, dx);    \
        else if (kx == 4) hipLaunchKernelGGL((fit::layer_bwd_kernel<KW, 4>), grid, dim3(256), 0, st, dw, dx);    \
        else hipLaunchKernelGGL((fit::layer_bwd_kernel<KW, 8>), grid, dim3(256), 0, st, dw, dx);                 \
    } while (0)
    if (kw == 1) AADFF_BWD_X(1);
    else if (kw == 2) AADFF_BWD_X(2);
    else if (kw == 4) AADFF_BWD_X(4);
    else AADFF_BWD_X(8);
#undef AADFF_BWD_X
    AADFF_CHECK_LAUNCH();
    return 0;
}


extern "C" int aadff_fit_chain(const aadff_fit_net* net, int* step_dev, float* scratch4, float lr0, int t_max, float beta1, float beta2,
                               float weight_decay, aadff_stream_t stream) {
    AADFF_CHECK_ARG(net && net->param_bf16 && net->scratch_bf16 && net->grad && net->inp && net->target && net->pred, "fit_chain: NULL pointer");
    const aadff_fit_net& a = *net;
    AADFF_CHECK_ARG(a.n_layers >= 2 && a.n_layers <= AADFF_FIT_MAX_LAYERS && a.batch > 0 && a.batch <= 256 && a.ld_batch % 8 == 0 && a.ld_batch >= a.batch,
                    "fit_chain: layers %d (2..%d), batch %d (<= 256), ld_batch %d", a.n_layers, AADFF_FIT_MAX_LAYERS, a.batch, a.ld_batch);
    AADFF_CHECK_ARG(!step_dev || (scratch4 && t_max > 0), "fit_chain: optimiser schedule needs scratch4 and t_max > 0");
    int xtotal = 0, maxw = 0, gx = 0, gy = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        AADFF_CHECK_ARG(a.k[l] > 0 && a.k[l] <= 256 && a.n[l] > 0 && a.n[l] <= 256 && a.k[l] % 4 == 0,
                        "fit_chain: layer %d: widths %d -> %d (<= 256, inputs multiples of 4)", l, a.k[l], a.n[l]);
        AADFF_CHECK_ARG(l == 0 || a.k[l] == a.n[l - 1], "fit_chain: layer %d input width %d != previous output %d", l, a.k[l], a.n[l - 1]);
        AADFF_CHECK_ARG((a.off_w[l] | a.off_wt[l] | a.off_b[l] | a.off_xt[l] | a.off_dzt[l + 1]) % 8 == 0 && a.off_gw[l] % 4 == 0,
                        "fit_chain: layer %d: buffer offsets must keep 16-byte alignment", l);
        gx = std::max(gx, (a.k[l] + 31) / 32);
        gy = std::max(gy, (a.n[l] + 63) / 64);
    }
    AADFF_CHECK_ARG(a.n[a.n_layers - 1] <= 128, "fit_chain: the head handles <= 128 outputs");
    for (int l = 0; l <= a.n_layers; ++l) {
        const int w = fit::chain_width(a, l);
        xtotal += fit::kChainRows * fit::chain_pitch(w);
        maxw = std::max(maxw, w);
    }
    const size_t lds_bytes = 2 * (size_t)(xtotal + 2 * fit::kChainRows * fit::chain_pitch(maxw)) + 64;
    AADFF_CHECK_ARG(lds_bytes <= 160 * 1024, "fit_chain: %zu bytes of LDS needed (160 KB per workgroup)", lds_bytes);
    static size_t lds_set = 0;
    if (lds_bytes > lds_set) {
        AADFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fit::fit_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        lds_set = lds_bytes;
    }
    hipStream_t st = (hipStream_t)stream;
    const AdamwSchedule sch{step_dev, scratch4, lr0, t_max, beta1, beta2, weight_decay};
    hipLaunchKernelGGL(fit::fit_chain_kernel, dim3((a.batch + fit::kChainRows - 1) / fit::kChainRows), dim3(fit::kChainThreads), lds_bytes, st, a, sch);
    AADFF_CHECK_LAUNCH();
    const dim3 grid(gx, gy, a.n_layers);
    switch (ksteps_class(a.batch)) {
        case 1: hipLaunchKernelGGL(fit::fit_dw_all_kernel<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(fit::fit_dw_all_kernel<2>, grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(fit::fit_dw_all_kernel<4>, grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(fit::fit_dw_all_kernel<8>, grid, dim3(256), 0, st, a);
    }
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_fit_input(const float* inp, void* x, int ld_x, void* xT, int ld_xT, int B, int K, aadff_stream_t stream) {
    AADFF_CHECK_ARG(inp && x && xT && B > 0 && K > 0 && ld_x >= K && ld_xT >= B, "fit_input: bad arguments");
    hipLaunchKernelGGL(fit::fit_input_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, inp, static_cast<uint16_t*>(x), ld_x,
                       static_cast<uint16_t*>(xT), ld_xT, B, K);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_fit_head(const void* z, int ld_z, const float* target, float* pred, void* dz, int ld_dz, void* dzT, int ld_dzT,
                              float* dbias, int B, int N, int* step_dev, float* scratch4, float lr0, int t_max, float beta1, float beta2,
                              float weight_decay, aadff_stream_t stream) {
    AADFF_CHECK_ARG(z && target && pred && dz && dzT && dbias && B > 0 && N > 0 && N <= 128, "fit_head: bad arguments (N <= 128)");
    AADFF_CHECK_ARG(!step_dev || (scratch4 && t_max > 0), "fit_head: optimiser schedule needs scratch4 and t_max > 0");
    const AdamwSchedule sch{step_dev, scratch4, lr0, t_max, beta1, beta2, weight_decay};
    hipLaunchKernelGGL(fit::fit_head_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, static_cast<const uint16_t*>(z), ld_z, target, pred,
                       static_cast<uint16_t*>(dz), ld_dz, static_cast<uint16_t*>(dzT), ld_dzT, dbias, B, N, sch);
    AADFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int aadff_fit_adamw(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, const int* dst,
                               const int* dst_t, long n, const float* scal4, float beta1, float beta2, float eps, aadff_stream_t stream) {
    AADFF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && param_bf16 && dst && dst_t && scal4 && n > 0, "fit_adamw: bad arguments");
    const int blocks = (int)std::min<long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(fit::fit_adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       static_cast<uint16_t*>(param_bf16), dst, dst_t, n, scal4, beta1, beta2, eps);
    AADFF_CHECK_LAUNCH();
    return 0;
}
