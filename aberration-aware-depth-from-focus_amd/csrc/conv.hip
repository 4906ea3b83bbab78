// Image-space PSF application for gfx950 (MI355X): patch-wise PSF-grid convolution
// (render_psf_map / render_psf, single slice and stack-fused) and per-pixel PSF gather
// (local_psf_render).  Semantics follow deeplens/render_psf.py of the reference; see
// include/aadff.h for the per-entry citations and DESIGN.md for the kernel design.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <type_traits>
#include <hip/hip_ext.h>
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
    // ceil(2^32 / d) for the uniform block-index decompositions: n / d == __umulhi(n, magic) for
    // n < 65536 (integer division has no scalar form on gfx950 and costs ~15 VALU slots each)
    unsigned m_ntx, m_nty, m_nchunk, m_c;
    // XCD-aware block order (xcd_remap): 1-D launches only.  gx, gy = logical grid extents; N = 8 xcd_q + xcd_r blocks (xcd_q = 0: plain order)
    unsigned gx, gy, xcd_q, xcd_r, m_gx, m_gy;      // m_gx / m_gy != 0: magic multipliers, valid while the launch has < 65536 blocks
};

// Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8; observed, not promised: speed only), each with an L2 of
// its own.  Neighbouring bands share halo rows / columns (34 x 108 staged for 24 x 96 rendered: 1.6x), so with the plain order
// every XCD's L2 fetches its halos from the fabric again (FETCH_SIZE 20 MB for a 12.6 MB image).  Remapped, XCD k works on the
// k-th CONTIGUOUS eighth of the logical block order, whose neighbours then hit in that XCD's L2.
// id = k + 8 j  ->  logical index start_k + j,  start_k = k q + min(k, r),  N = 8 q + r.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned q, unsigned r) {
    const unsigned k = id & 7u, j = id >> 3;
    return k * q + (k < r ? k : r) + j;
}

__host__ __device__ inline unsigned magic_of(unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ int udiv_magic(unsigned n, unsigned d, unsigned magic) { return d == 1 ? (int)n : (int)__umulhi(n, magic); }

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
//   lane = (k, q) owns a CX-wide x RR-tall block of outputs (2x8 or 4x4 = 16 accumulators);
//   input rows slide through registers read as whole ds_read_b64 / ds_read_b128 vectors;
//   PSF taps are wave-uniform (one patch per tile) -> SGPR operands of v_fmac_f32.
// Taps are consumed in groups of UG PSF rows so that UG*KS weights stay in SGPRs while
// each input row is re-read only ceil(KS/UG) times; the image tile is staged ONCE for
// all S slices of a stack.
// ------------------------------------------------------------------------------------
constexpr int TW = 32, TH = 32;
constexpr int CX = 4, RR = 4, QN = TW / CX;      // lane = (k = lane/8, q = lane%8): cols 4q..4q+3, rows 4k..4k+3

#ifndef AADFF_CONV_MINWAVES
#define AADFF_CONV_MINWAVES 4      // waves per SIMD the register allocator must leave room for
#endif

typedef float float2v __attribute__((ext_vector_type(2)));

// Packed fp32 FMA is the only way to the fp32 peak on gfx950 (measured, tools/fma_bench.hip:
// v_fmac_f32 with an SGPR tap 75 TFLOP/s, v_pk_fma_f32 with an SGPR-pair tap 141 TFLOP/s).
//   acc.lo += w * x.lo ; acc.hi += w * x.hi      with w = lo (HALF 0) or hi (HALF 1) half of an SGPR pair
template <int HALF>
__device__ __forceinline__ void pk_fma_bcast(float2v& acc, float2v wpair, float2v x) {
    if constexpr (HALF == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
}
// keeps a value "used" at this program point so the accumulation is not sunk into the masked stores
__device__ __forceinline__ void pin(float2v& v) { asm volatile("" : "+v"(v)); }

template <int KS>
struct ConvCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int TWP = TW + KS - 1;                  // staged columns
    static constexpr int THP = TH + KS - 1;                  // staged rows
    static constexpr int NIN = CX + KS - 1;                  // inputs a lane needs per row: in[0 .. KS+2]
    static constexpr int NE = (NIN + 1) / 2;                 // even-aligned pairs E[j] = (in[2j], in[2j+1])
    static constexpr int NO = (KS + 1) / 2;                  // shifted pairs     O[j] = (in[2j+1], in[2j+2])
    static constexpr int NVA = (2 * NE + 3) / 4;             // ds_read_b128 per row from tile A
    static constexpr int NVB = (2 * NO + 3) / 4;             // ds_read_b128 per row from tile B (= A shifted by one column)
    // pitches: 16-B slot step between lane-rows (RR=4 tile rows apart) must be 8 (mod 16) so that the four
    // lane-rows of a 16-lane ds_read_b128 group fall on disjoint bank quarters (64 banks x 4 B)
    static constexpr int need_a = (QN - 1) * CX + 4 * NVA, need_b = (QN - 1) * CX + 4 * NVB;
    static constexpr int PA = ((((need_a > TWP ? need_a : TWP)) - 8 + 15) / 16) * 16 + 8;
    static constexpr int PB = ((need_b - 8 + 15) / 16) * 16 + 8;
    static constexpr int NM = (KS + 1) / 2;                  // SGPR pairs per PSF row
    static constexpr int UG = KS <= 7 ? KS : (KS <= 13 ? 4 : 2);   // PSF rows per SGPR group
};

// One workgroup = NW waves = one 32x32 output tile of ONE patch and ONE channel plane for a CHUNK of
// NW slices: the reflect-padded input tile is staged once (tile A, plus tile B = A shifted left by one
// column so that odd taps also read ALIGNED register pairs) and wave w renders slice chunk*NW + w.
// Units are one slice long (short tail) yet NW waves share each 16 KB of LDS (>= 4 waves per SIMD).
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW, AADFF_CONV_MINWAVES) void conv_psf_map_kernel(
    const float* __restrict__ img, const float* __restrict__ psf, float* __restrict__ out, long sbc, long ss, int C, int S, int H, int W,
    int grid, int ntx, int nty, PatchBounds pb) {
    using Cfg = ConvCfg<KS>;
    constexpr int PA = Cfg::PA, PB = Cfg::PB;
    static_assert(Cfg::TWP <= kWave, "staging maps one tile column to one lane");
    __shared__ __attribute__((aligned(16))) float tileA[Cfg::THP * PA];
    __shared__ __attribute__((aligned(16))) float tileB[Cfg::THP * PB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform -> PSF taps via scalar loads
    const int pj = udiv_magic(blockIdx.x, ntx, pb.m_ntx), tx = blockIdx.x - pj * ntx;
    const int pi = udiv_magic(blockIdx.y, nty, pb.m_nty), ty = blockIdx.y - pi * nty;
    const int nchunk = (S + NW - 1) / NW;
    const int bc = udiv_magic(blockIdx.z, nchunk, pb.m_nchunk), chunk = blockIdx.z - bc * nchunk;
    const int c = bc - udiv_magic(bc, C, pb.m_c) * C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TW, y0 = pb.hb[pi] + ty * TH;
    if (x0 >= x_hi || y0 >= y_hi) return;

    // ---- stage the reflect-padded tile: column = lane (reflected once), rows strided over the waves
    //      (row index and its reflection are wave-uniform -> scalar ALU); every global load is issued
    //      before the first LDS write so the latencies overlap ----
    {
        const float* plane = img + (size_t)bc * H * W;
        constexpr int NR = (Cfg::THP + NW - 1) / NW;
        const int xx = reflect_idx(x0 - Cfg::PAD + lane, W);
        float v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = wave + j * NW;
            const int yy = reflect_idx(y0 - Cfg::PAD + r, H);
            v[j] = (lane < Cfg::TWP && r < Cfg::THP) ? plane[(size_t)yy * W + xx] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = wave + j * NW;
            if (lane < Cfg::TWP && r < Cfg::THP) {
                tileA[r * PA + lane] = v[j];
                if (lane >= 1 && lane - 1 < PB) tileB[r * PB + lane - 1] = v[j];
            }
        }
    }
    __syncthreads();

    const int q = lane % QN, k = lane / QN;
    const int G = grid * KS;
    const float* arow = &tileA[(RR * k) * PA + CX * q];
    const float* brow = &tileB[(RR * k) * PB + CX * q];

    const int s = chunk * NW + wave;
    if (s < S) {
        // PSF block of this patch; conv2d is a correlation with the FLIPPED PSF
        // (render_psf.py:62): w(u,v) = psf[KS-1-u][KS-1-v] = m_u[KS-1-v], m_u = PSF row KS-1-u.
        const float* wp = psf + ((size_t)(s * C + c) * G + pi * KS) * G + pj * KS;
        float2v acc[RR][2];
#pragma unroll
        for (int r = 0; r < RR; ++r) acc[r][0] = acc[r][1] = (float2v){0.f, 0.f};

#pragma unroll
        for (int u0 = 0; u0 < KS; u0 += Cfg::UG) {
            float2v M[Cfg::UG][Cfg::NM];                    // M[ug][j] = (m[2j], m[2j+1]) -> SGPR pairs
#pragma unroll
            for (int ug = 0; ug < Cfg::UG; ++ug) {
                const float* mrow = wp + (size_t)(KS - 1 - (u0 + ug < KS ? u0 + ug : KS - 1)) * G;
#pragma unroll
                for (int j = 0; j < Cfg::NM; ++j)
                    M[ug][j] = (float2v){mrow[2 * j], mrow[2 * j + 1 < KS ? 2 * j + 1 : KS - 1]};
            }
#pragma unroll
            for (int ir = u0; ir < u0 + Cfg::UG - 1 + RR; ++ir) {
                if (ir >= KS - 1 + RR) continue;
                float2v E[2 * Cfg::NVA], O[2 * Cfg::NVB];
                const float4* ap = reinterpret_cast<const float4*>(arow + ir * PA);
                const float4* bp = reinterpret_cast<const float4*>(brow + ir * PB);
#pragma unroll
                for (int h = 0; h < Cfg::NVA; ++h) {
                    const float4 t = ap[h];
                    E[2 * h] = (float2v){t.x, t.y};
                    E[2 * h + 1] = (float2v){t.z, t.w};
                }
#pragma unroll
                for (int h = 0; h < Cfg::NVB; ++h) {
                    const float4 t = bp[h];
                    O[2 * h] = (float2v){t.x, t.y};
                    O[2 * h + 1] = (float2v){t.z, t.w};
                }
                // tap loop outermost: the (up to) 4 output rows x 2 column pairs this input row feeds are
                // independent accumulators -> 8 interleaved dependency chains per wave
#pragma unroll
                for (int v = 0; v < KS; ++v) {
                    const int idx = KS - 1 - v;                // position of tap v in memory row m
#pragma unroll
                    for (int ug = 0; ug < Cfg::UG; ++ug) {
                        const int u = u0 + ug, r = ir - u;
                        if (u >= KS || r < 0 || r >= RR) continue;
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const float2v x = (v % 2 == 0) ? E[(v + 2 * p) / 2] : O[(v + 2 * p - 1) / 2];
                            if (idx % 2 == 0)
                                pk_fma_bcast<0>(acc[r][p], M[ug][idx / 2], x);
                            else
                                pk_fma_bcast<1>(acc[r][p], M[ug][idx / 2], x);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RR; ++r) { pin(acc[r][0]); pin(acc[r][1]); }
        }

        const int x = x0 + CX * q, yb = y0 + RR * k;
        float* o = out + (size_t)bc * sbc + (size_t)s * ss + (size_t)yb * W + x;
        const bool m0 = x < x_hi, m1 = x + 1 < x_hi, m2 = x + 2 < x_hi, m3 = x + 3 < x_hi;
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if (yb + r < y_hi) {
                if (m0) o[0] = acc[r][0].x;
                if (m1) o[1] = acc[r][0].y;
                if (m2) o[2] = acc[r][1].x;
                if (m3) o[3] = acc[r][1].y;
            }
            o += W;
        }
    }
}

// ------------------------------------------------------------------------------------
// MFMA path (KS <= 11): the patch convolution as a Toeplitz GEMM on v_mfma_f32_16x16x32_f16.
//   D[m][n] += sum_k A_u[m][k] B_u[k][n],  m = output row (16), n = output column (16), k = input column (32),
//   A_u[m][k] = in[y+m+u][x+k],  B_u[k][n] = w(u, k-n) for 0 <= k-n < KS else 0,   summed over the KS tap rows u.
// fp32 operands are split into two fp16 halves after an exact power-of-two pre-scale (x = hi + lo + O(2^-23 |x|)),
// and hi*hi + hi*lo + lo*hi are accumulated in fp32 by the MFMA: every fp16 product is exact in fp32, the dropped
// lo*lo term is 2^-22 relative, so a 121-tap sum carries ~1e-8 |sum| of split error - below the rounding of an
// fp32 FMA chain of the same length.  3*KS MFMAs (16 cycles each) per 16x16 block: 2112 MFMA cycles per 32x32
// tile and slice against 3872 VALU cycles for packed fp32 FMAs, with 88 instead of 131 ds_read_b128.
// LDS image: two fp16 planes [42][48]; the 48-half pitch (6 sixteen-byte slots per row) makes the A-fragment
// ds_read_b128 (lane = row l&15, k-group l>>4) bank-conflict free.  B fragments (Toeplitz taps, 8*KS VGPRs) are
// built once per wave and slice from the 121 taps.
// ------------------------------------------------------------------------------------
#ifndef AADFF_MFMA_SPW
#define AADFF_MFMA_SPW 1          // slices per wave from one staged tile (2 measured slower: 118 vs 74 us)
#endif
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}
// 2^(9 - floor(log2(amax))) and its inverse (exact powers of two; amax <= 0 or non-finite -> 1)
__device__ __forceinline__ void pow2_scale(float amax, float& s, float& inv) {
    const unsigned bits = __float_as_uint(amax);
    int e = (int)((bits >> 23) & 0xffu) - 127;
    const bool ok = amax > 0.f && e < 128;
    e = max(-100, min(e, 100));
    s = ok ? __uint_as_float((unsigned)(127 + 9 - e) << 23) : 1.f;
    inv = ok ? __uint_as_float((unsigned)(127 - 9 + e) << 23) : 1.f;
}

#ifdef AADFF_SB_TRACE
// Timeline instrumentation (tools/conv_timeline.py, build libaadff_sbtrace.so): wave 0 of every workgroup stamps the 100 MHz
// real-time counter at its start, after the band is staged, at the start of the matrix phase and at its end, plus HW_ID.
__device__ unsigned long long* g_sb_trace = nullptr;
#define AADFF_SB_STAMP(slot) do { if (g_sb_trace && threadIdx.x == 0) g_sb_trace[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AADFF_SB_STAMP(slot) do {} while (0)
#endif

template <int KS, int NW, int SPW>
__global__ __launch_bounds__(64 * NW) void conv_psf_map_mfma_kernel(
    const float* __restrict__ img, const float* __restrict__ psf, float* __restrict__ out, long sbc, long ss, int C, int S, int H, int W,
    int grid, int ntx, int nty, PatchBounds pb) {
    constexpr int PAD = KS / 2, TWP = TW + KS - 1, THP = TH + KS - 1, P = 48;
    static_assert(16 + KS - 1 <= 32 && TWP <= P, "Toeplitz band must fit K = 32");
    __shared__ __attribute__((aligned(16))) _Float16 Ahi[THP * P];
    __shared__ __attribute__((aligned(16))) _Float16 Alo[THP * P];
    // zero-padded flipped tap rows R_u[i] = w(u, i-15) (i = 15..15+KS-1, else 0) as fp16 hi/lo planes; a lane's 8
    // consecutive taps start at any half index: it reads the 5 covering dwords and funnel-shifts by 0 or 16 bits
    constexpr int RP = 48;
    __shared__ __attribute__((aligned(16))) _Float16 rows[NW][KS][2][RP + 2];
    __shared__ float red[NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pj = udiv_magic(blockIdx.x, ntx, pb.m_ntx), tx = blockIdx.x - pj * ntx;
    const int pi = udiv_magic(blockIdx.y, nty, pb.m_nty), ty = blockIdx.y - pi * nty;
    const int nchunk = (S + NW * SPW - 1) / (NW * SPW);
    const int bc = udiv_magic(blockIdx.z, nchunk, pb.m_nchunk), chunk = blockIdx.z - bc * nchunk;
    const int c = bc - udiv_magic(bc, C, pb.m_c) * C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TW, y0 = pb.hb[pi] + ty * TH;
    if (x0 >= x_hi || y0 >= y_hi) return;
    const int G = grid * KS;
    AADFF_SB_STAMP(0);

    // this wave's taps: lane i < RP owns padded index i, i.e. tap column v = i - 15, for every tap row u.  The loads of the
    // wave's FIRST slice are issued here, in front of the image loads, so that they are not a second exposed memory latency
    // behind the staging (a lone slice is one wave per workgroup: 1.5 of its 13.6 us, tools/conv_single_timeline.py)
    float wcol[KS];
    auto load_taps = [&](int s) {
        const int tv = lane - 15;
        const bool in = s < S && tv >= 0 && tv < KS;
        const float* wp = psf + ((size_t)((s < S ? s : 0) * C + c) * G + pi * KS) * G + pj * KS;
#pragma unroll
        for (int u = 0; u < KS; ++u) wcol[u] = in ? wp[(size_t)(KS - 1 - u) * G + (KS - 1 - tv)] : 0.f;       // w(u,v) = psf[KS-1-u][KS-1-v]
    };
    load_taps(chunk * SPW * NW + wave);

    // ---- stage: image window -> registers (column = lane, rows strided over waves), this wave's taps -> LDS ----
    constexpr int NR = (THP + NW - 1) / NW;
    float v[NR];
    float amax = 0.f;
    {
        const float* plane = img + (size_t)bc * H * W;
        const int xx = reflect_idx(x0 - PAD + lane, W);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = wave + j * NW;
            const int yy = reflect_idx(y0 - PAD + r, H);
            v[j] = (lane < TWP && r < THP) ? plane[(size_t)yy * W + xx] : 0.f;
            amax = fmaxf(amax, fabsf(v[j]));
        }
    }
    amax = wave_max(amax);
    AADFF_SB_STAMP(1);                                                   // image loads have arrived
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    float tmax = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tmax = fmaxf(tmax, red[w]);
    float sx, isx;
    pow2_scale(tmax, sx, isx);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = wave + j * NW;
        if (lane < P && r < THP) {
            const float xs = (lane < TWP) ? v[j] * sx : 0.f;
            const _Float16 h = (_Float16)xs;
            Ahi[r * P + lane] = h;
            Alo[r * P + lane] = (_Float16)(xs - (float)h);
        }
    }
    __syncthreads();
    AADFF_SB_STAMP(2);                                                   // tile in LDS

    // lane holds B[k = 8(l>>4)+j][n = l&15] = w(u, k-n) = R_u[st + j], st = 8(l>>4) - n + 15
    const int kq = lane >> 4, n = lane & 15;
    const int st = 8 * kq - n + 15;
    const unsigned sh = (st & 1) * 16;
    const unsigned* rbase = reinterpret_cast<const unsigned*>(&rows[wave][0][0][0]) + (st >> 1);
    constexpr int RSTRIDE = (RP + 2) / 2;                               // dwords per (u, plane) row
    auto load_frag = [&](int u, int plane) -> half8v {
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        const unsigned* q = rbase + (u * 2 + plane) * RSTRIDE;
        const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        uint4v r = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                    __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh)};
        return __builtin_bit_cast(half8v, r);
    };

#pragma unroll 1
    for (int sp = 0; sp < SPW; ++sp) {
    const int s = (chunk * SPW + sp) * NW + wave;
    if (s >= S) break;
    if (sp > 0) load_taps(s);
    float wmax = 0.f;
#pragma unroll
    for (int u = 0; u < KS; ++u) wmax = fmaxf(wmax, fabsf(wcol[u]));
    wmax = wave_max(wmax);
    float sw, isw;
    pow2_scale(wmax, sw, isw);
    if (lane < RP + 2) {
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const float w = lane < RP ? wcol[u] * sw : 0.f;
            const _Float16 h = (_Float16)w;
            rows[wave][u][0][lane] = h;
            rows[wave][u][1][lane] = (_Float16)(w - (float)h);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    AADFF_SB_STAMP(3);                                                   // tap rows in LDS

    const float inv = isx * isw;
    float* oplane = out + (size_t)bc * sbc + (size_t)s * ss;
    // tap row u outermost: one Toeplitz fragment pair live at a time, four 16x16 block accumulators
    constexpr int NBY = TH / 16, NBX = TW / 16;
    float4v acc[NBY][NBX];
#pragma unroll
    for (int by = 0; by < NBY; ++by)
#pragma unroll
        for (int bx = 0; bx < NBX; ++bx) acc[by][bx] = (float4v){0.f, 0.f, 0.f, 0.f};
    const int abase = n * P + 8 * kq;                                    // A fragment: row m = l&15, 8 halves at k-group l>>4
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        const half8v bh = load_frag(u, 0), bl = load_frag(u, 1);
#pragma unroll
        for (int by = 0; by < NBY; ++by)
#pragma unroll
            for (int bx = 0; bx < NBX; ++bx) {
                const int off = abase + (16 * by + u) * P + 16 * bx;
                const half8v ah = *reinterpret_cast<const half8v*>(&Ahi[off]);
                const half8v al = *reinterpret_cast<const half8v*>(&Alo[off]);
                acc[by][bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[by][bx], 0, 0, 0);
                acc[by][bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[by][bx], 0, 0, 0);
                acc[by][bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[by][bx], 0, 0, 0);
            }
    }
#ifdef AADFF_SB_TRACE
    asm volatile("s_nop 0" :: "v"(acc[NBY - 1][NBX - 1][0]) : "memory");   // the stamp waits for the last MFMA's result
    AADFF_SB_STAMP(4);
#endif
    // C/D layout: column n = l&15, rows 4(l>>4) + r
#pragma unroll
    for (int by = 0; by < NBY; ++by)
#pragma unroll
        for (int bx = 0; bx < NBX; ++bx) {
            const int x = x0 + 16 * bx + n, yb = y0 + 16 * by + 4 * kq;
            if (x < x_hi) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (yb + r < y_hi) oplane[(size_t)(yb + r) * W + x] = acc[by][bx][r] * inv;
            }
        }
#ifdef AADFF_SB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores retired
    AADFF_SB_STAMP(5);
#endif
    __builtin_amdgcn_wave_barrier();      // the next slice overwrites this wave's tap rows
    }
}

// ------------------------------------------------------------------------------------
// Toeplitz GEMM for the LARGE kernel sizes (round 5: odd ks 13 .. 51 - the reference's own render_single_img(method='psf') uses
// grid 7, ks 21, deeplens/optics.py:779-783; psf_map's default and analysis() ks 51, optics.py:1006).  The form of
// conv_psf_map_mfma_kernel with the Toeplitz band of a 16-column block spread over NK k-steps of 32 input columns
// (16 + ks - 1 <= 32 NK: NK = 1 up to ks 17, 2 up to ks 49, 3 for ks 51) and ks a run-time value:
//   D[y][x] += sum_u sum_kk sum_k in[y + u][x0 + 32 kk + k] * w(u, 32 kk + k - x)      per tap row u: 3 NK MFMAs (hi/lo split)
// Image tile as two fp16 planes [32 + ks - 1][P], P = the staged width rounded up to 8 (mod 16) halves: rows P / 2 dwords apart
// with P / 2 = 4 (mod 8) put the 16 rows of an A-fragment ds_read_b128 on 16 different bank quadruples.  Tap rows zero-padded,
// flipped, exact fp16 hi/lo, 32 NK + 16 (+2) halves each; a lane's 8 consecutive taps start at any half index (5 dwords +
// funnel shift).  One slice per wave, NW waves share the staged tile.  LDS is dynamic: tile planes + NW tap-row sets.
// Useful MACs are ks / (32 NK) of the issued ones (21 / 64 at ks 21), and it is still 3x the packed-FMA kernel: the matrix pipe
// has 16x the fp32 vector rate.
// ------------------------------------------------------------------------------------
// Workgroup = one 32 x 32 tile of one patch and plane, staged ONCE by its 4 waves for the `spw` slices of its chunk.  Every wave
// renders all four 16 x 16 blocks from one Toeplitz fragment per (tap row, k-step) - 12 MFMAs per fragment pair: a wave per block
// read the same fragments four times and ran LDS-bound at a quarter of the matrix rate - and the waves split
//   KSPLIT = false: the SLICES (wave w = slice 4 g + w of the chunk, its own tap rows in LDS; needs 4 tap-row sets: ks <= 31),
//   KSPLIT = true:  the TAP ROWS of one slice at a time (u = w, w + 4, ...; the four partial tiles meet in LDS): lone slices, ks > 31.
// MAXR / MAXT: tile rows / taps a thread stages in registers (the size class of ks: sized for ks 51 they cost ks 21 half its occupancy)
template <int NK, bool KSPLIT, int MAXR, int MAXT>
__global__ __launch_bounds__(256) void conv_psf_map_toeplitz_wide_kernel(
    const float* __restrict__ img, const float* __restrict__ psf, float* __restrict__ out, long sbc, long ss, int C, int S, int H, int W,
    int grid, int ks, int P, int spw, int ntx, int nty, PatchBounds pb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    constexpr int RP = 32 * NK + 16, RS = RP + 2;                        // padded tap-row length (halves), its LDS stride
    const int pad = ks / 2, TWP = TW + ks - 1, THP = TH + ks - 1;
    _Float16* Ahi = reinterpret_cast<_Float16*>(dyn);
    _Float16* Alo = Ahi + (size_t)THP * P;
    _Float16* rows0 = Alo + (size_t)THP * P;                             // KSPLIT: [ks][2][RS] (also the 16 KB reduction area); else [4][ks][2][RS]
    __shared__ float red[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pj = udiv_magic(blockIdx.x, ntx, pb.m_ntx), tx = blockIdx.x - pj * ntx;
    const int pi = udiv_magic(blockIdx.y, nty, pb.m_nty), ty = blockIdx.y - pi * nty;
    const int nchunk = (S + spw - 1) / spw;
    const int bc = udiv_magic(blockIdx.z, nchunk, pb.m_nchunk), chunk = blockIdx.z - bc * nchunk;
    const int c = bc - udiv_magic(bc, C, pb.m_c) * C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TW, y0 = pb.hb[pi] + ty * TH;
    if (x0 >= x_hi || y0 >= y_hi) return;
    const int G = grid * ks;
    const float* plane = img + (size_t)bc * H * W;

    // the taps of the first slice are requested FIRST: they are then not a second exposed memory latency behind the tile's
    const int nthr = KSPLIT ? 256 : 64, me = KSPLIT ? tid : lane;
    float tw[MAXT];
    auto load_taps = [&](int s, bool live) {
        const float* wp = psf + ((size_t)((live ? s : 0) * C + c) * G + pi * ks) * G + pj * ks;
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            const int e = me + nthr * j;
            const int u = e / ks, v = e - u * ks;
            tw[j] = (live && e < ks * ks) ? wp[(size_t)u * G + v] : 0.f;
        }
    };
    {
        const int s = KSPLIT ? chunk * spw : chunk * spw + wave;
        load_taps(s, s < S && (KSPLIT || wave < spw));
    }
    // ---- image tile: lane = columns (lane, lane + 64), rows strided over the 4 waves; every load in flight before the first use ----
    {
        float va[MAXR], vb[MAXR];
        float amax = 0.f;
        const int xa = reflect_idx(x0 - pad + lane, W), xb = reflect_idx(x0 - pad + 64 + lane, W);
        const bool ca = lane < TWP, cb = lane + 64 < TWP;
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const int r = wave + 4 * j;
            const float* row = plane + (size_t)reflect_idx(y0 - pad + (r < THP ? r : 0), H) * W;
            va[j] = (ca && r < THP) ? row[xa] : 0.f;
            vb[j] = (cb && r < THP) ? row[xb] : 0.f;
            amax = fmaxf(amax, fmaxf(fabsf(va[j]), fabsf(vb[j])));
        }
        amax = wave_max(amax);
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        const float tmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sx, isx0;
        pow2_scale(tmax, sx, isx0);
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const int r = wave + 4 * j;
            if (r < THP) {
                if (lane < P) {
                    const float xs = va[j] * sx;
                    const _Float16 h = (_Float16)xs;
                    Ahi[r * P + lane] = h;
                    Alo[r * P + lane] = (_Float16)(xs - (float)h);
                }
                if (lane + 64 < P) {
                    const float xs = vb[j] * sx;
                    const _Float16 h = (_Float16)xs;
                    Ahi[r * P + lane + 64] = h;
                    Alo[r * P + lane + 64] = (_Float16)(xs - (float)h);
                }
            }
        }
    }
    float sxd, isx;                                                      // (recomputed: keeps the staging registers out of the main loop)
    pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), sxd, isx);

    _Float16* rows = KSPLIT ? rows0 : rows0 + (size_t)wave * ks * 2 * RS;
    // zero padding of the tap rows (the taps themselves are rewritten per slice)
    for (int e = me; e < ks * RS; e += nthr) reinterpret_cast<unsigned*>(rows)[e] = 0u;

    // lane holds B[k = 8 kq + j][n] of k-step kk = w(u, 32 kk + k - n) = R_u[st + j], st = 32 kk + 8 kq - n + 15
    const int kq = lane >> 4, n = lane & 15;
    const int st0 = 8 * kq - n + 15;
    const unsigned sh = (st0 & 1) * 16;                                  // (32 kk keeps the parity)
    const unsigned* rbase = reinterpret_cast<const unsigned*>(rows) + (st0 >> 1);
    constexpr int RSD = RS / 2;                                          // dwords per (u, plane) row
    auto load_frag = [&](int u, int plane_i, int kk) -> half8v {
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        const unsigned* q = rbase + (u * 2 + plane_i) * RSD + 16 * kk;
        const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        uint4v r = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                    __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh)};
        return __builtin_bit_cast(half8v, r);
    };
    const int abase = n * P + 8 * kq;                                    // A fragment: row m = l & 15, 8 halves at k-group l >> 4
    constexpr int NBY = TH / 16, NBX = TW / 16;

    const int ngroups = KSPLIT ? spw : (spw + 3) / 4;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int s = KSPLIT ? chunk * spw + gi : chunk * spw + 4 * gi + wave;
        const bool live = s < S && (KSPLIT || 4 * gi + wave < spw);
        if (KSPLIT && !live) break;                                      // uniform
        // ---- the slice's taps -> registers -> scale -> flipped into the padded rows ----
        if (gi > 0) load_taps(s, live);
        float wmax = 0.f;
#pragma unroll
        for (int j = 0; j < MAXT; ++j) wmax = fmaxf(wmax, fabsf(tw[j]));
        wmax = wave_max(wmax);
        if constexpr (KSPLIT) {
            __syncthreads();                                             // the previous slice's reduction is done with the rows area; red is free
            if (lane == 0) red[wave] = wmax;
            __syncthreads();
            wmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        }
        float sw, isw;
        pow2_scale(wmax, sw, isw);
        if constexpr (KSPLIT) {                                          // (the reduction area overlaps the rows: restore their zero padding)
            if (gi > 0) {
                for (int e = tid; e < ks * RS; e += 256) reinterpret_cast<unsigned*>(rows)[e] = 0u;
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            const int e = me + nthr * j;
            if (e < ks * ks) {
                const int uu = e / ks, vv = e - uu * ks;                 // psf[uu][vv] = w(ks-1-uu, ks-1-vv)
                const float w = tw[j] * sw;
                const _Float16 h = (_Float16)w;
                const int at = ((ks - 1 - uu) * 2) * RS + (ks - 1 - vv) + 15;
                rows[at] = h;
                rows[at + RS] = (_Float16)(w - (float)h);
            }
        }
        if constexpr (KSPLIT) __syncthreads();
        else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (gi == 0) __syncthreads();                                // the tile (all waves staged it)
        }

        const float inv = isx * isw;
        float4v acc[NBY][NBX];
#pragma unroll
        for (int by = 0; by < NBY; ++by)
#pragma unroll
            for (int bx = 0; bx < NBX; ++bx) acc[by][bx] = (float4v){0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll 1
            for (int u = KSPLIT ? wave : 0; u < ks; u += KSPLIT ? 4 : 1) {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    const half8v bh = load_frag(u, 0, kk), bl = load_frag(u, 1, kk);
#pragma unroll
                    for (int by = 0; by < NBY; ++by)
#pragma unroll
                        for (int bx = 0; bx < NBX; ++bx) {
                            const int off = abase + (16 * by + u) * P + 16 * bx + 32 * kk;
                            const half8v ah = *reinterpret_cast<const half8v*>(&Ahi[off]);
                            const half8v al = *reinterpret_cast<const half8v*>(&Alo[off]);
                            acc[by][bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[by][bx], 0, 0, 0);
                            acc[by][bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[by][bx], 0, 0, 0);
                            acc[by][bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[by][bx], 0, 0, 0);
                        }
                }
            }
        }
        float* oplane = out + (size_t)bc * sbc + (size_t)(live ? s : 0) * ss;
        if constexpr (KSPLIT) {
            // the four partial tiles (tap rows u = w mod 4) meet in LDS: [wave][block][lane][4] floats = 16 KB over the rows area
            float4v* racc = reinterpret_cast<float4v*>(rows0);
            __syncthreads();                                             // every wave is done reading the tap rows
#pragma unroll
            for (int by = 0; by < NBY; ++by)
#pragma unroll
                for (int bx = 0; bx < NBX; ++bx) racc[(wave * 4 + by * 2 + bx) * 64 + lane] = acc[by][bx];
            __syncthreads();
            const float4v t = (racc[(0 * 4 + wave) * 64 + lane] + racc[(1 * 4 + wave) * 64 + lane]) +
                              (racc[(2 * 4 + wave) * 64 + lane] + racc[(3 * 4 + wave) * 64 + lane]);
            const int by = wave >> 1, bx = wave & 1;                     // wave b finishes block b
            const int x = x0 + 16 * bx + n, yb = y0 + 16 * by + 4 * kq;
            if (x < x_hi) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (yb + r < y_hi) oplane[(size_t)(yb + r) * W + x] = t[r] * inv;
            }
        } else if (live) {
            // C/D layout: column n = l & 15, rows 4 (l >> 4) + r
#pragma unroll
            for (int by = 0; by < NBY; ++by)
#pragma unroll
                for (int bx = 0; bx < NBX; ++bx) {
                    const int x = x0 + 16 * bx + n, yb = y0 + 16 * by + 4 * kq;
                    if (x < x_hi) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (yb + r < y_hi) oplane[(size_t)(yb + r) * W + x] = acc[by][bx][r] * inv;
                    }
                }
            __builtin_amdgcn_wave_barrier();                             // the next slice overwrites this wave's tap rows
        }
    }
}

// ------------------------------------------------------------------------------------
// Slice-batched MFMA path (ks = 11, stacks of >= 3 slices): the S slices of a stack share the image, so the image
// window is the shared GEMM operand and the slices ride on the M dimension:
//   D[m][n] += sum_k T[m][k] X[k][n]
//   n = cx (16): base pixel (y, X0 + 2 cx) of a 2-row x 32-column group
//   m = (sl, du, j) = 4 slices x 2 output rows x 2 output columns      (16)
//   k = (u, t), u = 0..11 input rows, t = 0..11 input columns           (144 of K = 160)
//   X[(u,t)][cx] = in[y + u][X0 + 2 cx + t],      T[(sl,du,j)][(u,t)] = w_sl(u - du, t - j)  (0 outside the 11x11 taps)
// -> 121 of 160 MACs per lane-slot useful for 4 slices at once (the Toeplitz form above: 11 of 32 for one slice):
// 15 MFMAs per 2x32 pixels x 4 slices instead of 66, and less than half the LDS operand bytes.
// K order: a lane's 8 consecutive k (k-group kg, step st) are the 2x4 block of input rows (2up, 2up+1) x columns
// (2dd0 .. 2dd0+3), slot q = 2 st + (kg >> 1): dd0 = 2 (q / 3), up = 2 (q % 3) + (kg & 1); slot 9 is padding (T = 0).
// The image tile is stored row-pair interleaved (tile[a][dd][row parity] dwords of fp16 pairs), so that block is 16
// contiguous bytes, 8-byte aligned: two ds_read_b64 per operand, and with a 224-dword row-pair pitch the 32 lanes of
// a read group (16 cx x {up, up+1}) cover all 64 banks once.  T fragments (40 VGPRs) are built once per wave from
// zero-padded tap rows in LDS; a workgroup = NC slice chunks x 2 row groups sharing one staged band of a patch, so
// the image is read from HBM once for all slices.  Same exact fp16 hi/lo operand split as above.
// ------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* lp_gptr_t;
typedef __attribute__((address_space(3))) void* lp_lptr_t;
namespace sb {
constexpr int KS = 11, PAD = 5;
constexpr int TCOLS = 96;                 // output columns per workgroup tile (3 column blocks of 32)
constexpr int WCOLS = TCOLS + 12;         // staged columns: t <= 11
constexpr int WDW = WCOLS / 2;            // 54 dword columns (fp16 pairs) per row
constexpr int RPP = 224;                  // row-PAIR pitch in dwords (== 32 mod 64): hi plane at 0, lo plane at LO
constexpr int LO = 112;
constexpr int PROWS = 14, PRD = 8;        // padded tap rows per slice (rows 0, 12, 13 zero), 8 dwords each
constexpr int PSL = PROWS * PRD + 2;      // slice stride 114 dwords: the 32 lanes of a T-build read hit 32 banks
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) f2u { float x, y; };       // 8-byte store at 4-byte alignment

// 16 operand bytes as two ds_read_b64 (full-rate LDS reads; a ds_read2_b64 or an 8-byte-aligned b128 is half rate
// or worse).  Not tracked by the compiler's waitcnt insertion: consume only after lds_wait().  The outputs are EARLY-CLOBBER:
// the statement holds two instructions, and with plain "=v" the allocator may give the first read's destination the
// address register - the second read then takes its address from a register the first one's returning data overwrites
// whenever the wave is held between the two for longer than the LDS latency (seen in round 3 as one wrong 2x2 output
// block per ~100 000, never the same one: `ds_read_b64 v[50:51], v50 ...; ds_read_b64 v[52:53], v50 ...`).
template <int OFF>
__device__ __forceinline__ void lds_read16(uint2v& a, uint2v& b, unsigned byte_addr) {
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(byte_addr), "n"(OFF), "n"(OFF + 8));
}
}  // namespace sb

// Workgroup = one band of RB output rows x 96 columns of one patch and channel plane; wave = one chunk of 4 slices
// (NC waves).  The band is staged once for all slices (HBM reads the image once), every wave builds the T fragments
// of its own chunk and walks the band's row pairs.
#ifndef AADFF_CONV_PAIR_DEFAULT
#define AADFF_CONV_PAIR_DEFAULT 0      // measured: every pairing is slower than one band per workgroup
#endif
// TIMED: the same code under a second name - the launches that carry aadff_time_next_launch's events (bench.py's solo leg) then
// have their own row in a rocprofv3 --stats summary of the very same command, separate from the launches of the timed region
// that share the device with the next stack's PSF-grid kernel.
// 5 waves per SIMD (96 VGPRs) = 6 workgroups per CU = 1536 slots: the 1452 workgroups of the bench launch are all resident in
// ONE round (round 2: 104 VGPRs, 5 per CU, 172 workgroups in a second round that ended 12 us after the first).  What made 96
// possible without spilling in the loop: two operand buffers instead of three.  Forms measured and dropped in round 3 (three
// accumulators per row pair, DPP operand sharing, 32/48-row bands, two row groups per chunk): DESIGN.md 4.1; git history has them.
// LAYERED (round 5, the M1-layered mode of SURVEY.md 8(d)): the S maps are (slice, layer) pairs p = slice * L + layer and a per-pixel
// layer index decides which candidate a pixel keeps: every lane still computes its 2 x 2 block for its map, but stores a pixel of
// slice p / L only where lidx[b][y][x] == p % L - the L candidates of a pixel sit in L different lanes (or waves), exactly one of
// them writes it.  Output bytes are those of ONE stack instead of L stacks plus a gather pass.
template <int RB, int NC, bool PAIR, bool TIMED = false, bool LAYERED = false>
__global__ __launch_bounds__(64 * NC, LAYERED ? 4 : 5) void conv_psf_map_sbatch_kernel(
    const float* __restrict__ img, const float* __restrict__ psf, float* __restrict__ out, long sbc, long ss, int C, int S, int H, int W,
    int grid, int ntx, int nty, int npass, PatchBounds pb, int stagger, int pair_mod, const unsigned char* __restrict__ lidx = nullptr, int L = 1) {
    using namespace sb;
    AADFF_SB_STAMP(0);
    if (stagger) {
        // Workgroups that share a CU are dispatched ~256 linear ids apart and would all run the same phase at the same
        // time (stage -> build T -> matrix phase -> stores).  A start delay by residency slot spreads the phases so that
        // one workgroup's HBM reads overlap another's MFMAs.
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int slot = (int)((lin >> 8) % 6u);
        for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    constexpr int NW = NC, THP = RB + KS - 1, NSL = 4 * NC;
    static_assert(RB % 2 == 0 && THP % 2 == 0, "bands are whole row pairs");
    static_assert(WDW <= 64, "one lane per dword column");
    __shared__ __attribute__((aligned(16))) unsigned tile[(THP / 2) * RPP];
    // [hi | lo] planes of the padded tap rows; PAIR: once the T fragments are built the same memory (and a little more) is
    // the landing zone of the second band's raw fp32 rows (LDS-DMA), THP rows x WCOLS floats
    constexpr int PROW_DW = 2 * NSL * PSL, STAGE_DW = PAIR ? THP * WCOLS : 0, POOL_DW = PROW_DW > STAGE_DW ? PROW_DW : STAGE_DW;
    __shared__ __attribute__((aligned(16))) unsigned pool[POOL_DW];
    unsigned (*prow)[NSL * PSL] = reinterpret_cast<unsigned (*)[NSL * PSL]>(pool);
    __shared__ float red[NW];
    __shared__ float s_isw[NSL];
    __shared__ __attribute__((aligned(4))) unsigned char lband[LAYERED ? RB * TCOLS : 4];     // layer index of the band's pixels (255 outside the patch)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pj = udiv_magic(blockIdx.x, ntx, pb.m_ntx), tx = blockIdx.x - pj * ntx;
    const int pi = udiv_magic(blockIdx.y, nty, pb.m_nty), ty = blockIdx.y - pi * nty;
    const int bc = udiv_magic(blockIdx.z, npass, pb.m_nchunk), pass = blockIdx.z - bc * npass;
    // PAIR: every pair_mod-th (patch, plane) renders its bands two per workgroup - the even band's workgroup also takes the
    // odd one, whose own workgroup leaves at once; pair_mod = 1 pairs everything (half the live workgroups), 4 pairs a
    // quarter: just enough fewer workgroups for the whole launch to be resident in one round (1280 slots at 5 per CU)
    bool paired = false;
    if constexpr (PAIR) {
        paired = (unsigned)((bc * grid + pi) * grid + pj) % (unsigned)pair_mod == 0u;
        if (paired && (ty & 1)) return;
    }
    const int c = bc - udiv_magic(bc, C, pb.m_c) * C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TCOLS, y0 = pb.hb[pi] + ty * RB;
    if (x0 >= x_hi || y0 >= y_hi) return;
    AADFF_SB_STAMP(1);
#ifdef AADFF_SB_TRACE
    if (g_sb_trace && threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_sb_trace[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + 6] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int G = grid * KS;
    const int s_base = pass * NSL;                       // first slice of this workgroup
    const int chunk = wave;
    const int kg = lane >> 4, lo4 = lane & 15;

    // ---- global loads up front: this wave's taps (4 slices, 2 per lane and slice), then its share of the image rows ----
    float tw0[4], tw1[4];
    const int t0 = lane, t1 = lane + 64;                                       // taps t0 (< 121 always) and t1 (< 121 for lane < 57)
    const int tu0 = t0 / KS, tc0 = t0 - tu0 * KS, tu1 = t1 / KS, tc1 = t1 - tu1 * KS;
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        const int s = s_base + 4 * chunk + k2;
        tw0[k2] = 0.f; tw1[k2] = 0.f;
        if (s < S) {
            // w(u,v) = psf[KS-1-u][KS-1-v]  (deeplens/render_psf.py:60 flips the kernel before conv2d)
            const float* wp = psf + ((size_t)(s * C + c) * G + pi * KS) * G + pj * KS;
            tw0[k2] = wp[(size_t)(KS - 1 - tu0) * G + (KS - 1 - tc0)];
            if (t1 < KS * KS) tw1[k2] = wp[(size_t)(KS - 1 - tu1) * G + (KS - 1 - tc1)];
        }
    }
    // image band: wave = row (strided), lane = dword column (two pixels)
    constexpr int NPT = (THP + NW - 1) / NW;
    float v0[NPT], v1[NPT];
    float amax = 0.f;
    {
        const float* plane = img + (size_t)bc * H * W;
        const int xa = reflect_idx(x0 - PAD + 2 * lane, W), xb = reflect_idx(x0 - PAD + 2 * lane + 1, W);
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            const int r = wave + e * NW;
            const bool in = lane < WDW && r < THP;
            const float* row = plane + (size_t)reflect_idx(y0 - PAD + r, H) * W;
            v0[e] = in ? row[xa] : 0.f;
            v1[e] = in ? row[xb] : 0.f;
            amax = fmaxf(amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
        }
    }
    // ---- padded fp16 hi/lo tap rows of this wave's chunk ----
    {
        _Float16* ph = reinterpret_cast<_Float16*>(&prow[0][0]);
        _Float16* pl = reinterpret_cast<_Float16*>(&prow[1][0]);
        for (int e = lane; e < 4 * PSL; e += 64) { prow[0][4 * chunk * PSL + e] = 0u; prow[1][4 * chunk * PSL + e] = 0u; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const int sl = 4 * chunk + k2;
            const float wmax = wave_max(fmaxf(fabsf(tw0[k2]), fabsf(tw1[k2])));
            float sw, isw;
            pow2_scale(wmax, sw, isw);
            if (lane == 0) s_isw[sl] = isw;
            {
                const float a = tw0[k2] * sw;
                const _Float16 h = (_Float16)a;
                const int base = (sl * PSL + (tu0 + 1) * PRD) * 2 + 2 + tc0;        // halves; taps start at half index 2
                ph[base] = h;
                pl[base] = (_Float16)(a - (float)h);
            }
            if (t1 < KS * KS) {
                const float a = tw1[k2] * sw;
                const _Float16 h = (_Float16)a;
                const int base = (sl * PSL + (tu1 + 1) * PRD) * 2 + 2 + tc1;
                ph[base] = h;
                pl[base] = (_Float16)(a - (float)h);
            }
        }
    }
    if constexpr (LAYERED) {
        const unsigned char* lp = lidx + (size_t)udiv_magic(bc, C, pb.m_c) * H * W;
        for (int e = tid; e < RB * TCOLS; e += 64 * NC) {
            const int r = e / TCOLS, cc = e - r * TCOLS, y = y0 + r, x = x0 + cc;
            lband[e] = (y < y_hi && x < x_hi) ? lp[(size_t)y * W + x] : (unsigned char)255;
        }
    }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    AADFF_SB_STAMP(2);                                                        // global loads have arrived (wave 0)
    __syncthreads();
    float tmax = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tmax = fmaxf(tmax, red[w]);
    float sx, isx;
    pow2_scale(tmax, sx, isx);
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const int r = wave + e * NW;
        if (lane < WDW && r < THP) {
            const float a = v0[e] * sx, b = v1[e] * sx;
            const _Float16 ah = (_Float16)a, bh = (_Float16)b;
            const _Float16 al = (_Float16)(a - (float)ah), bl = (_Float16)(b - (float)bh);
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            const int d = (r >> 1) * RPP + 2 * lane + (r & 1);
            tile[d] = __builtin_bit_cast(unsigned, (half2v){ah, bh});
            tile[d + LO] = __builtin_bit_cast(unsigned, (half2v){al, bl});
        }
    }

    // ---- T fragments of this wave's chunk: slot q = 2 st + (kg >> 1) -> column pair dd0, row pair up (see header) ----
    uint4v Th[5], Tl[5];
    {
        const int sl = lo4 >> 2, du = (lo4 >> 1) & 1, j = lo4 & 1;
        const unsigned* pbh = &prow[0][(chunk * 4 + sl) * PSL];
        const unsigned* pbl = &prow[1][(chunk * 4 + sl) * PSL];
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int q = 2 * st + (kg >> 1);
            const bool real = q < 9;
            const int dd0 = real ? 2 * (q / 3) : 0, up = 2 * (q % 3) + (kg & 1);
            const int row = real ? 2 * up - du + 1 : 12;                       // tap row u - du, stored at +1; rows 12, 13 are zero
            const int start = 2 + 2 * dd0 - j;                                 // half index of column t = 2 dd0
            const int e = start >> 1;
            const unsigned sh = (start & 1) * 16;
            const unsigned* a0 = pbh + row * PRD + e;
            const unsigned* a1 = pbl + row * PRD + e;
            Th[st] = (uint4v){__builtin_amdgcn_alignbit(a0[1], a0[0], sh), __builtin_amdgcn_alignbit(a0[PRD + 1], a0[PRD], sh),
                              __builtin_amdgcn_alignbit(a0[2], a0[1], sh), __builtin_amdgcn_alignbit(a0[PRD + 2], a0[PRD + 1], sh)};
            Tl[st] = (uint4v){__builtin_amdgcn_alignbit(a1[1], a1[0], sh), __builtin_amdgcn_alignbit(a1[PRD + 1], a1[PRD], sh),
                              __builtin_amdgcn_alignbit(a1[2], a1[1], sh), __builtin_amdgcn_alignbit(a1[PRD + 2], a1[PRD + 1], sh)};
        }
    }
    float inv = isx * s_isw[chunk * 4 + kg];                                  // D rows 4 kg + i belong to slice kg of the chunk
    __syncthreads();                                                          // the whole band is in LDS
    AADFF_SB_STAMP(3);

    // PAIR: the second band of this patch (same PSFs -> same T fragments, already in registers) is fetched NOW, by LDS-DMA
    // (no registers, nothing to wait for) into the memory the tap rows no longer need, while the first band is in its matrix
    // phase: its load latency (8 us of 11 us of prologue, tools/conv_timeline.py) is hidden and the launch has half as many
    // workgroups (all resident at once: no second residency round).
    const int y0b = y0 + RB;
    const bool second = PAIR && paired && y0b < y_hi;
    if constexpr (PAIR) {
        if (second) {
            const float* plane = img + (size_t)bc * H * W;
            const int xa = reflect_idx(x0 - PAD + lane, W), xb = reflect_idx(x0 - PAD + 64 + lane, W);
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                const int r = wave + e * NW;
                if (r < THP) {
                    const float* row = plane + (size_t)reflect_idx(y0b - PAD + r, H) * W;
                    // (the instruction offset would shift the global address too: the second piece gets its own LDS base)
                    __builtin_amdgcn_global_load_lds((lp_gptr_t)(row + xa), (lp_lptr_t)(pool + r * WCOLS), 4, 0, 0);
                    if (lane < WCOLS - 64) __builtin_amdgcn_global_load_lds((lp_gptr_t)(row + xb), (lp_lptr_t)(pool + r * WCOLS + 64), 4, 0, 0);
                }
            }
        }
    }

    const int s_out = s_base + chunk * 4 + kg;
    const bool s_ok = s_out < S;
    // stores: wave-uniform 64-bit base (first slice of the chunk) + 32-bit per-lane byte offset (host checks 16 x slice stride + 4 H W < 2^32)
    // LAYERED: map s_out is (slice s_out / L, layer s_out % L): base = the (b, c) plane stack, offset = the slice (host checks S x stride)
    const int so_l = LAYERED ? s_out / L : 0;
    const unsigned lyr = LAYERED ? (unsigned)(s_out - so_l * L) : 0u;
    char* wbase = reinterpret_cast<char*>(out + (size_t)bc * sbc + (LAYERED ? (size_t)0 : (size_t)(s_base + chunk * 4) * ss));
    const unsigned w4 = (unsigned)W * 4u, koff = (LAYERED ? (unsigned)so_l : (unsigned)kg) * (unsigned)(ss * 4);   // slice kg of the chunk
    bool pair_ok[3], one_ok[3];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const int x = x0 + 32 * cb + 2 * lo4;
        pair_ok[cb] = s_ok && x + 1 < x_hi;
        one_ok[cb] = s_ok && x + 1 == x_hi;
    }

    // X operand byte addresses: lane n = cx
    unsigned xaddr[5];
    const unsigned tile_base = (unsigned)(size_t)tile;                        // LDS byte address of the tile
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int q = 2 * st + (kg >> 1);
        const bool real = q < 9;
        const int dd0 = real ? 2 * (q / 3) : 0, up = real ? 2 * (q % 3) + (kg & 1) : (kg & 1);
        xaddr[st] = tile_base + 4u * (unsigned)(up * RPP + 2 * (lo4 + dd0));
    }

    // ---- matrix phase of one band (rows yb .. yb + RB - 1 of the image, staged in `tile`) ----
    auto run_band = [&](const int yb, const float inv) {
        int npairs = (y_hi - yb + 1) / 2;                   // row pairs of this band that hold valid rows
        npairs = npairs > RB / 2 ? RB / 2 : npairs;
        const unsigned rowb0 = 0u;
        {
            // Two operand buffers, reads ONE step ahead: 8 VGPRs fewer than the three-buffer form - what it takes to fit
            // 96 VGPRs = 5 waves per SIMD = 6 workgroups (18 waves) per CU, i.e. all 1452 workgroups of the bench launch
            // resident in ONE round (no 12 us tail of 172 late workgroups; tools/conv_timeline.py).  A row pair has 15 steps,
            // so the buffer roles swap from one row pair to the next (P).
            uint2v xq[2][4];
            auto issue = [&](auto bfc, auto stepc, unsigned rowb) {
                constexpr int bf = decltype(bfc)::value, step = decltype(stepc)::value, cb = step / 5, st = step % 5;
                const unsigned a = xaddr[st] + rowb;
                lds_read16<cb * 128>(xq[bf][0], xq[bf][1], a);
                lds_read16<LO * 4 + cb * 128>(xq[bf][2], xq[bf][3], a);
            };
            auto row_pair = [&](auto pc, const int rpi) {
                constexpr int P = decltype(pc)::value;
                const unsigned rowb = rowb0 + (unsigned)(rpi * RPP * 4);
                const unsigned rowb_next = rowb0 + (unsigned)((rpi + 1 < npairs ? rpi + 1 : rpi) * RPP * 4);
                const int yl = 2 * rpi;
                const bool row1 = yb + yl + 1 < y_hi;
                const unsigned loff = koff + (unsigned)(yb + yl) * w4 + (unsigned)(x0 + 2 * lo4) * 4u;
                float4v acc;
                auto step_fn = [&](auto stepc) {
                    constexpr int step = decltype(stepc)::value, cb = step / 5, st = step % 5, bf = (P + step) & 1, nx = bf ^ 1;
                    if constexpr (step + 1 < 15) issue(std::integral_constant<int, nx>{}, std::integral_constant<int, step + 1>{}, rowb);
                    else issue(std::integral_constant<int, nx>{}, std::integral_constant<int, 0>{}, rowb_next);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xq[bf][0]), "+v"(xq[bf][1]), "+v"(xq[bf][2]), "+v"(xq[bf][3]));
                    const uint4v h4 = {xq[bf][0].x, xq[bf][0].y, xq[bf][1].x, xq[bf][1].y};
                    const uint4v l4 = {xq[bf][2].x, xq[bf][2].y, xq[bf][3].x, xq[bf][3].y};
                    const half8v bh = __builtin_bit_cast(half8v, h4), bl = __builtin_bit_cast(half8v, l4);
                    const half8v th = __builtin_bit_cast(half8v, Th[st]), tl = __builtin_bit_cast(half8v, Tl[st]);
                    if constexpr (st == 0) acc = (float4v){0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl, bh, acc, 0, 0, 0);
                    if constexpr (st == 4) {
                        const float a0 = acc[0] * inv, b0 = acc[1] * inv, a1 = acc[2] * inv, b1 = acc[3] * inv;
                        char* o0 = wbase + loff + cb * 128;
                        char* o1 = wbase + (loff + w4) + cb * 128;
                        if constexpr (LAYERED) {
                            if (s_ok) {
                                const unsigned q0 = *reinterpret_cast<const unsigned short*>(&lband[yl * TCOLS + 32 * cb + 2 * lo4]);
                                const unsigned q1 = *reinterpret_cast<const unsigned short*>(&lband[(yl + 1) * TCOLS + 32 * cb + 2 * lo4]);
                                if ((q0 & 255u) == lyr) *reinterpret_cast<float*>(o0) = a0;
                                if ((q0 >> 8) == lyr) *reinterpret_cast<float*>(o0 + 4) = b0;
                                if ((q1 & 255u) == lyr) *reinterpret_cast<float*>(o1) = a1;
                                if ((q1 >> 8) == lyr) *reinterpret_cast<float*>(o1 + 4) = b1;
                            }
                        } else
                        if (pair_ok[cb]) {
                            *reinterpret_cast<f2u*>(o0) = (f2u){a0, b0};
                            if (row1) *reinterpret_cast<f2u*>(o1) = (f2u){a1, b1};
                        } else if (one_ok[cb]) {
                            *reinterpret_cast<float*>(o0) = a0;
                            if (row1) *reinterpret_cast<float*>(o1) = a1;
                        }
                    }
                };
                step_fn(std::integral_constant<int, 0>{}); step_fn(std::integral_constant<int, 1>{}); step_fn(std::integral_constant<int, 2>{});
                step_fn(std::integral_constant<int, 3>{}); step_fn(std::integral_constant<int, 4>{}); step_fn(std::integral_constant<int, 5>{});
                step_fn(std::integral_constant<int, 6>{}); step_fn(std::integral_constant<int, 7>{}); step_fn(std::integral_constant<int, 8>{});
                step_fn(std::integral_constant<int, 9>{}); step_fn(std::integral_constant<int, 10>{}); step_fn(std::integral_constant<int, 11>{});
                step_fn(std::integral_constant<int, 12>{}); step_fn(std::integral_constant<int, 13>{}); step_fn(std::integral_constant<int, 14>{});
            };
            issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rowb0);
            int rpi = 0;
#pragma unroll 1
            for (; rpi + 1 < npairs; rpi += 2) {
                row_pair(std::integral_constant<int, 0>{}, rpi);
                row_pair(std::integral_constant<int, 1>{}, rpi + 1);
            }
            if (rpi < npairs) row_pair(std::integral_constant<int, 0>{}, rpi);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(xq[0][0]), "+v"(xq[0][1]), "+v"(xq[0][2]), "+v"(xq[0][3]), "+v"(xq[1][0]), "+v"(xq[1][1]), "+v"(xq[1][2]), "+v"(xq[1][3])
                         :: "memory");
        }
    };
    run_band(y0, inv);
    AADFF_SB_STAMP(4);
    if constexpr (PAIR) {
        if (second) {
            // ---- second band: raw rows (DMA, landed during the first band's matrix phase) -> scaled fp16 hi/lo tile ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA pieces (and its stores of the first band)
            __syncthreads();                                      // every wave's pieces; nobody reads the old tile any more
            {
            const float* stg = reinterpret_cast<const float*>(pool);
            float bmax = 0.f;
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                const int r = wave + e * NW;
                const bool in = lane < WDW && r < THP;
                const float2 v = in ? *reinterpret_cast<const float2*>(stg + r * WCOLS + 2 * lane) : make_float2(0.f, 0.f);
                v0[e] = v.x; v1[e] = v.y;
                bmax = fmaxf(bmax, fmaxf(fabsf(v.x), fabsf(v.y)));
            }
            bmax = wave_max(bmax);
            if (lane == 0) red[wave] = bmax;
            __syncthreads();
            float t2 = red[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) t2 = fmaxf(t2, red[w]);
            float sx2, isx2;
            pow2_scale(t2, sx2, isx2);
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                const int r = wave + e * NW;
                if (lane < WDW && r < THP) {
                    const float a = v0[e] * sx2, b = v1[e] * sx2;
                    const _Float16 ah = (_Float16)a, bh = (_Float16)b;
                    const _Float16 al = (_Float16)(a - (float)ah), bl = (_Float16)(b - (float)bh);
                    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                    const int d = (r >> 1) * RPP + 2 * lane + (r & 1);
                    tile[d] = __builtin_bit_cast(unsigned, (half2v){ah, bh});
                    tile[d + LO] = __builtin_bit_cast(unsigned, (half2v){al, bl});
                }
            }
            inv = isx2 * s_isw[chunk * 4 + kg];
            }
            __syncthreads();
            run_band(y0b, inv);
        }
    }
#ifdef AADFF_SB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores retired
    AADFF_SB_STAMP(5);
#endif
}

// ------------------------------------------------------------------------------------
// Block-GEMM path for LONE slices (ks 9 / 11, S = 1: `render_psf_map` / `render_psf` themselves, deeplens/render_psf.py:12-73).
// With one slice there is nothing to batch on M, so a 4 x 4 block of OUTPUT PIXELS rides there instead:
//   D[m][n] += sum_k T[m][k] X[k][n]
//   n = (ry, cx): base pixel (y + 4 ry, X0 + 4 cx) of an 8-row x 32-column group          (2 x 8 = 16)
//   m = (du, j): output pixel (base row + du, base column + j), du, j = 0..3              (16)
//   k = (u, t): input rows u = 0..13, input columns t = 0..15 of the base's 14 x 14 window (224 = 7 k-steps; 196 real)
//   X[(u,t)][(ry,cx)] = in[y + 4 ry + u][X0 + 4 cx + t],   T[(du,j)][(u,t)] = w(u - du, t - j)  (0 outside the ks x ks taps)
// -> 21 MFMAs per 256 outputs where the Toeplitz form above needs 33, and per k-step ONE aligned ds_read_b128 per operand
// plane: k-step st = input row pair (2 st, 2 st + 1), lane group kg = columns 4 kg .. 4 kg + 3, and in the row-pair
// interleaved tile of the slice-batched kernel ([row pair][dword column][row parity]) that 2 x 4 block is 16 contiguous,
// 16-byte aligned bytes.  Row-pair pitch 240 dwords (== 16 mod 32): the two base rows of a 16-lane read group (8 lanes x 16 B
// each, 2 row pairs apart) fall on disjoint halves of the 64 banks.  A lane ends up with 4 consecutive output pixels of one
// row: 16-byte stores.  Staging (24-row x 96-column band of one patch and plane, image read as fp32, tile-wide power-of-two
// scale, exact fp16 hi/lo split) is the slice-batched kernel's; wave w renders column block w of the band (3 groups of 8 rows).
// ------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // 16-byte load / store at 4-byte alignment

namespace blk {
constexpr int TCOLS = 96, WCOLS = TCOLS + 12, WDW = WCOLS / 2, RB = 24, THP = RB + 10, NW = 3;
constexpr int RPP = 240, LO = 112;        // row-pair pitch, offset of the lo plane (dwords)
constexpr int TROWS = 17, TPD = 11;       // padded tap rows (a = -3 .. 13), dwords per row (b = -3 .. 18 as halves)
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte store at 4-byte alignment
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));       // 8-byte load at 4-byte alignment
}  // namespace blk

template <int KS, bool TIMED = false>
__global__ __launch_bounds__(64 * blk::NW, 5) void conv_psf_map_blk_kernel(
    const float* __restrict__ img, const float* __restrict__ psf, float* __restrict__ out, long sbc, long ss, int C, int S, int H, int W,
    int grid, int ntx, int nty, PatchBounds pb) {
    using namespace blk;
    constexpr int PAD = KS / 2;
    static_assert(KS == 9 || KS == 11, "the 14 x 14 window of a 4 x 4 block holds taps up to 11 x 11");
#ifdef AADFF_SB_TRACE
    // (one wave-uniform base pointer for all stamps of this kernel: with the global macro's per-stamp address arithmetic the
    // instrumented build spilled 68 registers at 96 VGPRs and its timeline meant nothing)
    unsigned long long* const trace_slot = g_sb_trace ? g_sb_trace + (size_t)blockIdx.x * 8 : nullptr;
#define AADFF_BLK_STAMP(slot) do { if (trace_slot && threadIdx.x == 0) trace_slot[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AADFF_BLK_STAMP(slot) do {} while (0)
#endif
    AADFF_BLK_STAMP(0);
    __shared__ __attribute__((aligned(16))) unsigned tile[(THP / 2) * RPP];
    __shared__ __attribute__((aligned(16))) unsigned ptap[2][TROWS * TPD];       // [hi | lo] planes of the zero-padded taps
    __shared__ float red[NW];
    __shared__ float s_isw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D launch in XCD-aware order: logical index -> (x fastest, then y, then plane / slice)
    const unsigned lin = pb.xcd_q ? xcd_remap(blockIdx.x, pb.xcd_q, pb.xcd_r) : blockIdx.x;
    // (an exact 32-bit division is ~25 instructions on the scalar path of every workgroup; launches below 65536 blocks - a
    // 1024^2 slice has 1452 - take the magic-multiplier form)
    const unsigned lrow = pb.m_gx ? (unsigned)udiv_magic(lin, pb.gx, pb.m_gx) : lin / pb.gx, bx = lin - lrow * pb.gx;
    const unsigned bz = pb.m_gx ? (unsigned)udiv_magic(lrow, pb.gy, pb.m_gy) : lrow / pb.gy, by = lrow - bz * pb.gy;
    const int pj = udiv_magic(bx, ntx, pb.m_ntx), tx = bx - pj * ntx;
    const int pi = udiv_magic(by, nty, pb.m_nty), ty = by - pi * nty;
    const int bc = udiv_magic(bz, S, pb.m_nchunk), s = bz - bc * S;
    const int c = bc - udiv_magic(bc, C, pb.m_c) * C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * TCOLS, y0 = pb.hb[pi] + ty * RB;
    if (x0 >= x_hi || y0 >= y_hi) return;
    const int G = grid * KS;

    // ---- global loads up front: the taps (wave 0, two per lane), then this wave's share of the image rows ----
    float tw0 = 0.f, tw1 = 0.f;
    const int t0 = lane, t1 = lane + 64;
    const int tu0 = t0 / KS, tc0 = t0 - tu0 * KS, tu1 = t1 / KS, tc1 = t1 - tu1 * KS;
    if (wave == 0) {
        // w(a,b) = psf[KS-1-a][KS-1-b]  (deeplens/render_psf.py:60 flips the kernel before conv2d)
        const float* wp = psf + ((size_t)(s * C + c) * G + pi * KS) * G + pj * KS;
        if (t0 < KS * KS) tw0 = wp[(size_t)(KS - 1 - tu0) * G + (KS - 1 - tc0)];
        if (t1 < KS * KS) tw1 = wp[(size_t)(KS - 1 - tu1) * G + (KS - 1 - tc1)];
    }
    constexpr int NPT = (THP + NW - 1) / NW;
    float v0[NPT], v1[NPT];
    float amax = 0.f;
    {
        const float* plane = img + (size_t)bc * H * W;
        // Interior bands (no reflection at an image border: all but the outermost ring of patches) take a branch-free form: one
        // base pointer, row r at + r W, a lane's pixel pair as ONE 8-byte load.  The staging's scalar address arithmetic is on
        // every workgroup's critical path: 515 scalar + 264 vector instructions per wave in front of the first barrier in the
        // general form (12 reflected rows, two loads each), which a wave issues in ~1.5 us.
        const bool interior = x0 - PAD >= 0 && x0 - PAD + WCOLS <= W && y0 - PAD >= 0 && y0 - PAD + THP <= H;     // workgroup-uniform
        if (interior) {
            const float* base = plane + (size_t)(y0 - PAD + wave) * W + (x0 - PAD) + 2 * lane;
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                v0[e] = 0.f; v1[e] = 0.f;
                if (lane < WDW && wave + e * NW < THP) {
                    const float2u t = *reinterpret_cast<const float2u*>(base + (size_t)(e * NW) * W);
                    v0[e] = t.x; v1[e] = t.y;
                }
            }
        } else {
            const int xa = reflect_idx(x0 - PAD + 2 * lane, W), xb = reflect_idx(x0 - PAD + 2 * lane + 1, W);
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                const int r = wave + e * NW;
                const bool in = lane < WDW && r < THP;
                const float* row = plane + (size_t)reflect_idx(y0 - PAD + r, H) * W;
                v0[e] = in ? row[xa] : 0.f;
                v1[e] = in ? row[xb] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < NPT; ++e) amax = fmaxf(amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
    }
    if (wave == 0) {
        // zero-padded fp16 hi/lo taps: tap (a, b) at half index (a + 3) * 2 TPD + (b + 3)
        for (int e = lane; e < TROWS * TPD; e += 64) { ptap[0][e] = 0u; ptap[1][e] = 0u; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float wmax = wave_max(fmaxf(fabsf(tw0), fabsf(tw1)));
        float sw, isw;
        pow2_scale(wmax, sw, isw);
        if (lane == 0) s_isw = isw;
        _Float16* ph = reinterpret_cast<_Float16*>(&ptap[0][0]);
        _Float16* pl = reinterpret_cast<_Float16*>(&ptap[1][0]);
        if (t0 < KS * KS) {
            const float a = tw0 * sw;
            const _Float16 h = (_Float16)a;
            ph[(tu0 + 3) * 2 * TPD + tc0 + 3] = h;
            pl[(tu0 + 3) * 2 * TPD + tc0 + 3] = (_Float16)(a - (float)h);
        }
        if (t1 < KS * KS) {
            const float a = tw1 * sw;
            const _Float16 h = (_Float16)a;
            ph[(tu1 + 3) * 2 * TPD + tc1 + 3] = h;
            pl[(tu1 + 3) * 2 * TPD + tc1 + 3] = (_Float16)(a - (float)h);
        }
    }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    AADFF_BLK_STAMP(1);                                                        // global loads have arrived (wave 0)
    __syncthreads();
    float tmax = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tmax = fmaxf(tmax, red[w]);
    float sx, isx;
    pow2_scale(tmax, sx, isx);
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const int r = wave + e * NW;
        if (lane < WDW && r < THP) {
            const float a = v0[e] * sx, b = v1[e] * sx;
            const _Float16 ah = (_Float16)a, bh = (_Float16)b;
            const _Float16 al = (_Float16)(a - (float)ah), bl = (_Float16)(b - (float)bh);
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            const int d = (r >> 1) * RPP + 2 * lane + (r & 1);
            tile[d] = __builtin_bit_cast(unsigned, (half2v){ah, bh});
            tile[d + LO] = __builtin_bit_cast(unsigned, (half2v){al, bl});
        }
    }

    // ---- T fragments: lane (m = (du, j), kg), k-step st: halves i = 0..7 = (row 2 st + (i >> 1 & 1), column 4 kg + (i & 1) + 2 (i >> 2)) ----
    const int kg = lane >> 4, n = lane & 15;
    uint4v Th[7], Tl[7];
    {
        const int du = n >> 2, j = n & 3;
        const int c0 = 4 * kg - j + 3;                                          // half index of tap column t - j, t = 4 kg
        const int e = c0 >> 1;
        const unsigned sh = (c0 & 1) * 16;
#pragma unroll
        for (int st = 0; st < 7; ++st) {
            const int ra = 2 * st - du + 3;                                     // padded row of tap row u - du, u = 2 st
            const unsigned* a0 = &ptap[0][ra * TPD + e];
            const unsigned* a1 = &ptap[1][ra * TPD + e];
            Th[st] = (uint4v){__builtin_amdgcn_alignbit(a0[1], a0[0], sh), __builtin_amdgcn_alignbit(a0[TPD + 1], a0[TPD], sh),
                              __builtin_amdgcn_alignbit(a0[2], a0[1], sh), __builtin_amdgcn_alignbit(a0[TPD + 2], a0[TPD + 1], sh)};
            Tl[st] = (uint4v){__builtin_amdgcn_alignbit(a1[1], a1[0], sh), __builtin_amdgcn_alignbit(a1[TPD + 1], a1[TPD], sh),
                              __builtin_amdgcn_alignbit(a1[2], a1[1], sh), __builtin_amdgcn_alignbit(a1[TPD + 2], a1[TPD + 1], sh)};
        }
    }
    const float inv = isx * s_isw;
    __syncthreads();                                                          // the whole band is in LDS
    AADFF_BLK_STAMP(2);

    // ---- matrix phase: column block `wave` of the band, groups of 8 rows x 7 k-steps.  (Measured, round 4: hand-pipelined
    //      operand reads - inline asm two steps ahead through three buffers, as in the slice-batched kernel - are SLOWER here,
    //      15.2 against 14.6 us: with 17 waves per CU the other waves already cover a step's LDS latency.) ----
    const int ry = n >> 3, cx = n & 7;
    const int xw = x0 + 32 * wave;
    if (xw < x_hi) {
        const unsigned* xptr = tile + (2 * ry) * RPP + 32 * wave + 4 * cx + 4 * kg;
        const int x = xw + 4 * cx;
        const bool full = x + 3 < x_hi;
        float* obase = out + (size_t)bc * sbc + (size_t)s * ss + x;
#pragma unroll
        for (int g = 0; g < RB / 8; ++g) {
            const int yg = y0 + 8 * g;
            if (yg < y_hi) {
                float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 7; ++st) {
                    const uint4v h4 = *reinterpret_cast<const uint4v*>(xptr + (4 * g + st) * RPP);
                    const uint4v l4 = *reinterpret_cast<const uint4v*>(xptr + (4 * g + st) * RPP + LO);
                    const half8v bh = __builtin_bit_cast(half8v, h4), bl = __builtin_bit_cast(half8v, l4);
                    const half8v th = __builtin_bit_cast(half8v, Th[st]), tl = __builtin_bit_cast(half8v, Tl[st]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl, bh, acc, 0, 0, 0);
                }
                // D[m = 4 kg + r][n]: du = kg, j = r -> out[yg + 4 ry + kg][x + r]
                const int y = yg + 4 * ry + kg;
                if (y < y_hi) {
                    float* o = obase + (size_t)y * W;
                    const float a0 = acc[0] * inv, a1 = acc[1] * inv, a2 = acc[2] * inv, a3 = acc[3] * inv;
                    // non-temporal: the 12.6 MB of output are not parked in L2 until the kernel's end-of-launch write-back (-5 %:
                    // 13.1-13.4 against 13.9-14.0 us, same box, tools/kbench.py)
                    if (full) __builtin_nontemporal_store((float4u){a0, a1, a2, a3}, reinterpret_cast<float4u*>(o));
                    else {
                        if (x < x_hi) o[0] = a0;
                        if (x + 1 < x_hi) o[1] = a1;
                        if (x + 2 < x_hi) o[2] = a2;
                    }
                }
            }
        }
    }
#ifdef AADFF_SB_TRACE
    AADFF_BLK_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores retired
    AADFF_BLK_STAMP(5);
#endif
}
#undef AADFF_BLK_STAMP

// ------------------------------------------------------------------------------------
// Block-GEMM path for ks 13 .. 21 (round 5; the reference's own render_single_img(method='psf') uses grid 7, ks 21:
// deeplens/optics.py:779-783).  conv_psf_map_blk_kernel's form with the input window of a 4 x 4 output block grown to
//   WR x WC = (KS + 3 rounded up to even) x (KS + 3 rounded up to a multiple of 4) pixels,
// cut into NB = (WR / 2)(WC / 4) blocks of 2 rows x 4 columns - the 16 contiguous, 16-byte aligned bytes one lane reads from the
// row-pair interleaved tile - and block b = 4 st + kg is lane group kg's share of k-step st (NST = ceil(NB / 4); the blocks past NB
// carry zero taps).  Useful MACs / issued: 169 / 256 (ks 13), 225 / 384 (15), 289 / 416 (17), 361 / 544 (19), 441 / 576 (21) -
// the Toeplitz form above issues 32 NK columns per tap row for ks of them: 13 / 32 ... 21 / 64.  At ks 21: 54 MFMAs per 256
// outputs against 126.  The T fragments (taps arranged for the 16 (du, j) outputs of a block) are loop-invariant per workgroup
// and live in registers: 8 NST VGPRs (144 at ks 21, two workgroups of 3 waves per CU).
// Staging, scaling and the exact fp16 hi / lo split are conv_psf_map_blk_kernel's; pixels of the window that no tap touches
// (beyond the ks - 1 halo) are staged as zeros, so a NaN there cannot reach an output it does not belong to.
// ------------------------------------------------------------------------------------
template <int KS, int RBV>
struct BlkW {
    static constexpr int PAD = KS / 2;
    // AFF: window rounded up to 4 rows x 8 columns, k-step = 2 row pairs x 2 quads (lane group kg = (row pair kg >> 1, quad kg & 1)):
    // the tile offset of a k-step is a compile-time constant plus one per-lane term, no per-step offset registers (ks 21 holds
    // 144 VGPRs of T fragments); costs nothing at ks 13 / 21 (16 x 16, 24 x 24 are the minimal windows), one step at ks 19.
    // Otherwise (ks 15 / 17): minimal window, blocks dealt to the lane groups in order, offsets in registers.
    static constexpr bool AFF = KS == 13 || KS == 19 || KS == 21;
    static constexpr int WR = AFF ? (KS + 3 + 3) / 4 * 4 : (KS + 3 + 1) / 2 * 2, WC = AFF ? (KS + 3 + 7) / 8 * 8 : (KS + 3 + 3) / 4 * 4;
    static constexpr int NQ = WC / 4, NB = (WR / 2) * NQ, NST = (NB + 3) / 4;
    static constexpr int TCOLS = 96, RB = RBV, NW = 3;                           // RB: rows of the band (24, 32 or 48: groups of 8)
    static constexpr int WCOLS = TCOLS + WC - 4, WDW = WCOLS / 2, THP = RB + WR - 4;
    static constexpr int NEEDC = TCOLS + KS - 1, NEEDR = RB + KS - 1;          // columns / rows some tap touches
    static constexpr int LO = (WCOLS + 3) / 4 * 4, RPP = 240;                   // lo-plane offset, row-pair pitch (dwords; 240 = 16 mod 32)
    static constexpr int TROWS = WR + 3, TPD = (WC + 6) / 2;                    // padded tap rows (a = -3 ..), dwords per row
    static constexpr int NTP = (KS * KS + 63) / 64;                             // taps a lane loads (every wave loads all of them)
    static_assert(WDW <= 64 && LO + WCOLS <= RPP && THP % 2 == 0, "tile layout");
};

template <int KS, int RBV>
__global__ __launch_bounds__(192, KS <= 11 ? 5 : 2) void conv_psf_map_blkw_kernel(
    const float* __restrict__ img, const float* __restrict__ psf, float* __restrict__ out, long sbc, long ss, int C, int S, int H, int W,
    int grid, int ntx, int nty, PatchBounds pb) {
    using Z = BlkW<KS, RBV>;
    using blk::uint4v; using blk::float4u; using blk::float2u;
    constexpr int PAD = Z::PAD, NW = Z::NW, RPP = Z::RPP, LO = Z::LO, TPD = Z::TPD, NST = Z::NST, NQ = Z::NQ;
    __shared__ __attribute__((aligned(16))) unsigned tile[(Z::THP / 2) * RPP];
    __shared__ __attribute__((aligned(16))) unsigned ptap[2][Z::TROWS * TPD];
    __shared__ float red[NW];
#ifdef AADFF_SB_TRACE
    unsigned long long* const trace_slot = g_sb_trace ? g_sb_trace + (size_t)blockIdx.x * 8 : nullptr;
#define AADFF_BLK_STAMP(slot) do { if (trace_slot && threadIdx.x == 0) trace_slot[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AADFF_BLK_STAMP(slot) do {} while (0)
#endif
    AADFF_BLK_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lin = pb.xcd_q ? xcd_remap(blockIdx.x, pb.xcd_q, pb.xcd_r) : blockIdx.x;
    const unsigned lrow = pb.m_gx ? (unsigned)udiv_magic(lin, pb.gx, pb.m_gx) : lin / pb.gx, bx = lin - lrow * pb.gx;
    const unsigned bz = pb.m_gx ? (unsigned)udiv_magic(lrow, pb.gy, pb.m_gy) : lrow / pb.gy, by = lrow - bz * pb.gy;
    const int pj = udiv_magic(bx, ntx, pb.m_ntx), tx = bx - pj * ntx;
    const int pi = udiv_magic(by, nty, pb.m_nty), ty = by - pi * nty;
    const int bc = udiv_magic(bz, S, pb.m_nchunk), s = bz - bc * S;
    const int c = bc - udiv_magic(bc, C, pb.m_c) * C;
    const int x_hi = pb.wb[pj + 1], y_hi = pb.hb[pi + 1];
    const int x0 = pb.wb[pj] + tx * Z::TCOLS, y0 = pb.hb[pi] + ty * Z::RB;
    if (x0 >= x_hi || y0 >= y_hi) return;
    const int G = grid * KS;

    // ---- global loads up front: the taps (every wave loads all of them: NTP per lane, its own maximum and scale - no cross-wave
    //      reduction), then this wave's share of the image rows ----
    float tw[Z::NTP];
    {
        // w(a,b) = psf[KS-1-a][KS-1-b]  (deeplens/render_psf.py:60 flips the kernel before conv2d)
        const float* wp = psf + ((size_t)(s * C + c) * G + pi * KS) * G + pj * KS;
#pragma unroll
        for (int e = 0; e < Z::NTP; ++e) {
            const int t = lane + e * 64;
            const int tu = t / KS, tc = t - tu * KS;
            tw[e] = t < KS * KS ? wp[(KS - 1 - tu) * G + (KS - 1 - tc)] : 0.f;
        }
    }
    constexpr int NPT = (Z::THP + NW - 1) / NW;
    float v0[NPT], v1[NPT];
    float amax = 0.f;
    {
        const float* plane = img + (size_t)bc * H * W;
        const bool interior = x0 - PAD >= 0 && x0 - PAD + Z::WCOLS <= W && y0 - PAD >= 0 && y0 - PAD + Z::THP <= H;     // workgroup-uniform
        const bool ca = 2 * lane < Z::NEEDC, cb = 2 * lane + 1 < Z::NEEDC;
        if (interior) {
            const float* base = plane + (size_t)(y0 - PAD + wave) * W + (x0 - PAD) + 2 * lane;
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                v0[e] = 0.f; v1[e] = 0.f;
                if (lane < Z::WDW && wave + e * NW < Z::NEEDR) {
                    const float2u t = *reinterpret_cast<const float2u*>(base + (size_t)(e * NW) * W);
                    v0[e] = ca ? t.x : 0.f; v1[e] = cb ? t.y : 0.f;
                }
            }
        } else {
            const int xa = reflect_idx(x0 - PAD + 2 * lane, W), xb = reflect_idx(x0 - PAD + 2 * lane + 1, W);
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                const int r = wave + e * NW;
                const bool in = lane < Z::WDW && r < Z::NEEDR;
                const float* row = plane + (size_t)reflect_idx(y0 - PAD + r, H) * W;
                v0[e] = in && ca ? row[xa] : 0.f;
                v1[e] = in && cb ? row[xb] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < NPT; ++e) amax = fmaxf(amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
    }
    for (int e = tid; e < Z::TROWS * TPD; e += 64 * NW) { ptap[0][e] = 0u; ptap[1][e] = 0u; }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    AADFF_BLK_STAMP(1);                                                        // global loads have arrived (wave 0)
    __syncthreads();                                                          // band maximum known, tap array zeroed
    float tmax = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tmax = fmaxf(tmax, red[w]);
    float sx, isx;
    pow2_scale(tmax, sx, isx);
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const int r = wave + e * NW;
        if (lane < Z::WDW && r < Z::THP) {
            const float a = v0[e] * sx, b = v1[e] * sx;
            const _Float16 ah = (_Float16)a, bh = (_Float16)b;
            const _Float16 al = (_Float16)(a - (float)ah), bl = (_Float16)(b - (float)bh);
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            const int d = (r >> 1) * RPP + 2 * lane + (r & 1);
            tile[d] = __builtin_bit_cast(unsigned, (half2v){ah, bh});
            tile[d + LO] = __builtin_bit_cast(unsigned, (half2v){al, bl});
        }
    }
    const int kg = lane >> 4, n = lane & 15;
    const int ry = n >> 3, cx = n & 7;
    const int xw = x0 + 32 * wave;
    // block b = 4 st + kg of the window -> (row pair, column quad) -> dword offset in the tile
    constexpr int NXO = Z::AFF ? 1 : NST;
    int xo[NXO];
    if constexpr (Z::AFF) xo[0] = (kg >> 1) * RPP + 4 * (kg & 1);
    else {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int b = 4 * st + kg;
            const int bb = b < Z::NB ? b : 0;
            const int rp = bb / NQ, q = bb - rp * NQ;
            xo[st] = rp * RPP + 4 * q;
        }
    }
    // tile offset (dwords) of lane group kg's block in k-step st
    auto xoff = [&](int st) { if constexpr (Z::AFF) return xo[0] + 2 * (st / (NQ / 2)) * RPP + 8 * (st % (NQ / 2)); else return xo[st]; };
    float sw, isw;
    {
        // zero-padded fp16 hi / lo taps: tap (a, b) at half index (a + 3) * 2 TPD + (b + 3); wave w writes every NW-th group
        float wmax = 0.f;
#pragma unroll
        for (int e = 0; e < Z::NTP; ++e) wmax = fmaxf(wmax, fabsf(tw[e]));
        pow2_scale(wave_max(wmax), sw, isw);
        _Float16* ph = reinterpret_cast<_Float16*>(&ptap[0][0]);
        _Float16* pl = reinterpret_cast<_Float16*>(&ptap[1][0]);
#pragma unroll
        for (int e = 0; e < Z::NTP; ++e) {
            const int t = lane + e * 64;
            const int tu = t / KS, tc = t - tu * KS;
            if (t < KS * KS && e % NW == wave) {
                const float a = tw[e] * sw;
                const _Float16 h = (_Float16)a;
                ph[(tu + 3) * 2 * TPD + tc + 3] = h;
                pl[(tu + 3) * 2 * TPD + tc + 3] = (_Float16)(a - (float)h);
            }
        }
    }
    __syncthreads();                                                          // band and taps are in LDS
    AADFF_BLK_STAMP(2);

    // ---- T fragments: lane (m = (du, j), kg), k-step st: block b = 4 st + kg = (row pair rp, column quad q);
    //      halves i = 0..7 = (row 2 rp + (i >> 1 & 1), column 4 q + (i & 1) + 2 (i >> 2)) ----
    uint4v Th[NST], Tl[NST];
    {
        const int du = n >> 2, j = n & 3;
        // AFF: padded tap row / dword column of the block = per-lane part + compile-time part of the k-step (one LDS base register)
        const int cl = Z::AFF ? 4 * (kg & 1) - j + 3 : 0;
        const unsigned* tl0 = &ptap[0][(2 * (kg >> 1) - du + 3) * TPD + (cl >> 1)];
        const unsigned shl = (cl & 1) * 16;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const unsigned *a0, *a1;
            unsigned sh;
            bool valid = true;
            if constexpr (Z::AFF) {
                a0 = tl0 + 4 * (st / (NQ / 2)) * TPD + 4 * (st % (NQ / 2));
                a1 = a0 + Z::TROWS * TPD;
                sh = shl;
            } else {
                const int b = 4 * st + kg;
                valid = b < Z::NB;
                const int bb = valid ? b : 0;
                const int rp = bb / NQ, q = bb - rp * NQ;
                const int c0 = 4 * q - j + 3;                                   // half index of tap column t - j, t = 4 q
                const int ra = 2 * rp - du + 3;                                 // padded row of tap row u - du, u = 2 rp
                a0 = &ptap[0][ra * TPD + (c0 >> 1)];
                a1 = &ptap[1][ra * TPD + (c0 >> 1)];
                sh = (c0 & 1) * 16;
            }
            const uint4v th = (uint4v){__builtin_amdgcn_alignbit(a0[1], a0[0], sh), __builtin_amdgcn_alignbit(a0[TPD + 1], a0[TPD], sh),
                                       __builtin_amdgcn_alignbit(a0[2], a0[1], sh), __builtin_amdgcn_alignbit(a0[TPD + 2], a0[TPD + 1], sh)};
            const uint4v tl = (uint4v){__builtin_amdgcn_alignbit(a1[1], a1[0], sh), __builtin_amdgcn_alignbit(a1[TPD + 1], a1[TPD], sh),
                                       __builtin_amdgcn_alignbit(a1[2], a1[1], sh), __builtin_amdgcn_alignbit(a1[TPD + 2], a1[TPD + 1], sh)};
            const uint4v zero = (uint4v){0u, 0u, 0u, 0u};
            Th[st] = valid ? th : zero;
            Tl[st] = valid ? tl : zero;
        }
    }
    const float inv = isx * isw;
#ifdef AADFF_SB_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragments built
    AADFF_BLK_STAMP(3);
#endif

    // ---- matrix phase: column block `wave` of the band, groups of 8 rows x NST k-steps ----
    if (xw < x_hi) {
        const unsigned* xptr = tile + (2 * ry) * RPP + 32 * wave + 4 * cx;
        const int x = xw + 4 * cx;
        const bool full = x + 3 < x_hi;
        float* obase = out + (size_t)bc * sbc + (size_t)s * ss + x;
#pragma unroll
        for (int g = 0; g < Z::RB / 8; ++g) {
            const int yg = y0 + 8 * g;
            if (yg < y_hi) {
                float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const uint4v h4 = *reinterpret_cast<const uint4v*>(xptr + 4 * g * RPP + xoff(st));
                    const uint4v l4 = *reinterpret_cast<const uint4v*>(xptr + 4 * g * RPP + xoff(st) + LO);
                    const half8v bh = __builtin_bit_cast(half8v, h4), bl = __builtin_bit_cast(half8v, l4);
                    const half8v th = __builtin_bit_cast(half8v, Th[st]), tl = __builtin_bit_cast(half8v, Tl[st]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl, bh, acc, 0, 0, 0);
                }
                // D[m = 4 kg + r][n]: du = kg, j = r -> out[yg + 4 ry + kg][x + r]
                const int y = yg + 4 * ry + kg;
                if (y < y_hi) {
                    float* o = obase + (size_t)y * W;
                    const float a0 = acc[0] * inv, a1 = acc[1] * inv, a2 = acc[2] * inv, a3 = acc[3] * inv;
                    if (full) __builtin_nontemporal_store((float4u){a0, a1, a2, a3}, reinterpret_cast<float4u*>(o));
                    else {
                        if (x < x_hi) o[0] = a0;
                        if (x + 1 < x_hi) o[1] = a1;
                        if (x + 2 < x_hi) o[2] = a2;
                    }
                }
            }
        }
    }
#ifdef AADFF_SB_TRACE
    AADFF_BLK_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores retired
    AADFF_BLK_STAMP(5);
#endif
}
#undef AADFF_BLK_STAMP

// ------------------------------------------------------------------------------------
// Generic path (any odd ks <= AADFF_MAX_KS): same tiling, runtime loops, PSF taps staged
// flipped in LDS and read as broadcasts.  Correctness path for unusual kernel sizes.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_psf_map_generic_kernel(const float* __restrict__ img,
                                                                    const float* __restrict__ psf,
                                                                    float* __restrict__ out, long sbc, long ss, int C, int S,
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
        float* oplane = out + (size_t)bc * sbc + (size_t)s * ss;
        const int x = x0 + lx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + ly * 4 + r;
            if (y < y_hi && x < x_hi) oplane[(size_t)y * W + x] = acc[r];
        }
    }
}

// events armed by aadff_time_next_launch for the next stack-convolution / PSF-grid launch of this host thread (common.h)
thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;

template <int KS>
static int launch_fast(const float* img, const float* psf, float* out, long sbc, long ss, int B, int C, int S, int H, int W,
                       int grid, int ntx, int nty, const PatchBounds& pb, hipStream_t st) {
    // waves per workgroup = slices of one chunk (one slice per wave): least padding, at most 5
    int nw = 1, best = 1 << 30;
    const int maxnw = KS == 11 ? 5 : 4;
    for (int n = maxnw; n >= (S > 1 ? 2 : 1); --n) {      // larger n wins ties; a lone wave per tile only for S == 1
        if (KS != 11 && n == 3) continue;
        const int waste = (S + n - 1) / n * n - S;
        if (waste < best) { best = waste; nw = n; }
    }
    const char* env = getenv("AADFF_CONV_NW");                 // tuning override, validated: garbage / out-of-range values are ignored
    if (env && KS == 11) {
        const int v = atoi(env);
        if (v >= 1 && v <= maxnw) nw = v;
    }
    const int nchunk = (S + nw - 1) / nw;
    AADFF_CHECK_ARG((size_t)B * C * nchunk <= 65535, "render_psf_map: B*C*chunks too large");
    PatchBounds pbm = pb;
    pbm.m_ntx = magic_of(ntx); pbm.m_nty = magic_of(nty); pbm.m_nchunk = magic_of(nchunk); pbm.m_c = magic_of(C);
    dim3 g(ntx * grid, nty * grid, B * C * nchunk);
    // path: "mfma" (Toeplitz GEMM on fp16-split MFMA, KS <= 11) or "valu" (packed fp32 FMA); AADFF_CONV_PATH overrides
    const char* penv = getenv("AADFF_CONV_PATH");
    const bool use_mfma = KS <= 11 && !(penv && penv[0] == 'v');
    if constexpr (KS == 11) {
        // stacks: slice-batched GEMM (the image is the shared operand); "toeplitz" / "valu" force the older paths
        if (S >= 3 && !penv) {
            constexpr int RB = 24;          // band rows (8/12/16/32/48 measured slower)
            const int nc = S <= 4 ? 1 : (S <= 8 ? 2 : (S <= 12 ? 3 : 4));
            const int npass = (S + 4 * nc - 1) / (4 * nc);
            int mh = 0, mw = 0;
            for (int i = 0; i < grid; ++i) {
                mh = std::max(mh, pb.hb[i + 1] - pb.hb[i]);
                mw = std::max(mw, pb.wb[i + 1] - pb.wb[i]);
            }
            const int sntx = (mw + sb::TCOLS - 1) / sb::TCOLS, snty = (mh + RB - 1) / RB;
            AADFF_CHECK_ARG((size_t)B * C * npass <= 65535 && (size_t)snty * grid <= 65535, "render_psf_map: grid too large");
            AADFF_CHECK_ARG((size_t)H * W <= ((size_t)1 << 27) && 16 * (size_t)ss + 4 * (size_t)H * W < ((size_t)1 << 32),
                            "render_psf_map: image planes above 2^27 pixels / slice strides above 2^28 elements are not supported on the stack path");
            // paired bands (default): a workgroup renders two consecutive bands of its patch, the second one prefetched by LDS-DMA
            // AADFF_CONV_PAIR: 0 = one band per workgroup (round-2 form), N >= 1 = bands in pairs for every N-th (patch, plane)
            const int pair_mod = [] { const char* e = getenv("AADFF_CONV_PAIR"); const int v = e ? atoi(e) : AADFF_CONV_PAIR_DEFAULT; return v < 0 ? 0 : v; }();   // read per launch: tests switch it
            const bool pair = pair_mod > 0;
            const int gny = snty;
            PatchBounds pbs = pb;
            pbs.m_ntx = magic_of(sntx); pbs.m_nty = magic_of(gny); pbs.m_nchunk = magic_of(npass); pbs.m_c = magic_of(C);
            dim3 gs(sntx * grid, gny * grid, B * C * npass);
            static const int stagger = [] { const char* e = getenv("AADFF_CONV_STAGGER"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : (v > 64 ? 64 : v); }();   // default 1: -1 % in bench, -7 % back to back
            // aadff_time_next_launch: the two events ride ON this dispatch (kernel begin / end timestamps)
            hipEvent_t ev0 = g_time_start, ev1 = g_time_stop;
            g_time_start = g_time_stop = nullptr;
#define AADFF_LAUNCH_S3(NCV, PR) do { \
                if (ev0) hipExtLaunchKernelGGL((conv_psf_map_sbatch_kernel<RB, NCV, PR, true>), gs, dim3(64 * NCV), 0, st, ev0, ev1, 0, img, psf, out, sbc, ss, C, S, H, W, grid, sntx, gny, npass, pbs, stagger, pair_mod, (const unsigned char*)nullptr, 1); \
                else hipLaunchKernelGGL((conv_psf_map_sbatch_kernel<RB, NCV, PR, false>), gs, dim3(64 * NCV), 0, st, img, psf, out, sbc, ss, C, S, H, W, grid, sntx, gny, npass, pbs, stagger, pair_mod, (const unsigned char*)nullptr, 1); } while (0)
#define AADFF_LAUNCH_S(NCV) do { if (pair) AADFF_LAUNCH_S3(NCV, true); else AADFF_LAUNCH_S3(NCV, false); } while (0)
            switch (nc) {
                case 1: AADFF_LAUNCH_S(1); break;
                case 2: AADFF_LAUNCH_S(2); break;
                case 3: AADFF_LAUNCH_S(3); break;
                default: AADFF_LAUNCH_S(4);
            }
#undef AADFF_LAUNCH_S
#undef AADFF_LAUNCH_S3
            return 0;
        }
    }
    if constexpr (KS == 9 || KS == 11) {
        // lone slices (render_psf_map / render_psf as the reference calls them) and pairs: block-GEMM form, 21 instead of 33
        // MFMAs per 256 outputs and the slice-batched kernel's staging; AADFF_CONV_PATH = toeplitz / valu force the older paths
        // S == 1 only - measured (tools/conv_paths_bench.py, 1024^2, us): ks 11: S = 1 14.5-16.6 against 19.3-19.8 Toeplitz, S = 2
        // 23.2 / 23.0 (re-staging per slice cancels the gain; a form that renders three slices from one staged band was built
        // and was no faster: 23.0 / 28.9 for S = 2 / 3); ks 9: S = 1 14.5 / 18.5, S = 2 23.8 / 21.0, S = 10 86 / 72-78
        if (S == 1 && B * C <= 65535 && !penv && (size_t)H * W <= ((size_t)1 << 30)) {
            int mh = 0, mw = 0;
            for (int i = 0; i < grid; ++i) {
                mh = std::max(mh, pb.hb[i + 1] - pb.hb[i]);
                mw = std::max(mw, pb.wb[i + 1] - pb.wb[i]);
            }
            const int bntx = (mw + blk::TCOLS - 1) / blk::TCOLS, bnty = (mh + blk::RB - 1) / blk::RB;
            const size_t gx = (size_t)bntx * grid, gy = (size_t)bnty * grid, total = gx * gy * B * C * S;
            if (total < ((size_t)1 << 31)) {
                PatchBounds pbb = pb;
                pbb.m_ntx = magic_of(bntx); pbb.m_nty = magic_of(bnty); pbb.m_nchunk = magic_of(S); pbb.m_c = magic_of(C);
                pbb.gx = (unsigned)gx; pbb.gy = (unsigned)gy;
                const bool magic = total < 65536;
                pbb.m_gx = magic ? (gx == 1 ? 1u : magic_of((unsigned)gx)) : 0u;           // (udiv_magic returns n for d == 1; the flag is m_gx != 0)
                pbb.m_gy = magic ? (gy == 1 ? 1u : magic_of((unsigned)gy)) : 0u;
                pbb.xcd_q = total >= 64 ? (unsigned)(total / 8) : 0u;
                pbb.xcd_r = (unsigned)(total % 8);
                hipLaunchKernelGGL((conv_psf_map_blk_kernel<KS>), dim3((unsigned)total), dim3(64 * blk::NW), 0, st, img, psf, out, sbc, ss, C, S, H, W, grid, bntx, bnty, pbb);
                return 0;
            }
        }
    }
    if constexpr (KS <= 11) {
        if (use_mfma) {
            const int nchunk_m = (S + nw * AADFF_MFMA_SPW - 1) / (nw * AADFF_MFMA_SPW);
            dim3 gm(ntx * grid, nty * grid, B * C * nchunk_m);
            PatchBounds pbmm = pbm;
            pbmm.m_nchunk = magic_of(nchunk_m);
#define AADFF_LAUNCH_M(NWV) hipLaunchKernelGGL((conv_psf_map_mfma_kernel<KS, NWV, AADFF_MFMA_SPW>), gm, dim3(64 * NWV), 0, st, img, psf, out, sbc, ss, C, S, H, W, grid, ntx, nty, pbmm)
            switch (nw) {
                case 5: if constexpr (KS == 11) { AADFF_LAUNCH_M(5); break; }
                case 4: AADFF_LAUNCH_M(4); break;
                case 3: if constexpr (KS == 11) { AADFF_LAUNCH_M(3); break; }
                case 2: AADFF_LAUNCH_M(2); break;
                default: AADFF_LAUNCH_M(1);
            }
#undef AADFF_LAUNCH_M
            return 0;
        }
    }
#define AADFF_LAUNCH(NWV) hipLaunchKernelGGL((conv_psf_map_kernel<KS, NWV>), g, dim3(64 * NWV), 0, st, img, psf, out, sbc, ss, C, S, H, W, grid, ntx, nty, pbm)
    if constexpr (KS == 11) {
        switch (nw) {
            case 5: AADFF_LAUNCH(5); break;
            case 4: AADFF_LAUNCH(4); break;
            case 3: AADFF_LAUNCH(3); break;
            case 2: AADFF_LAUNCH(2); break;
            default: AADFF_LAUNCH(1);
        }
    } else {
        switch (nw) {
            case 4: AADFF_LAUNCH(4); break;
            case 2: AADFF_LAUNCH(2); break;
            default: AADFF_LAUNCH(1);
        }
    }
#undef AADFF_LAUNCH
    return 0;
}

// out plane of (b, c, s) starts at out + (b*C + c)*sbc + s*ss (elements); the contiguous [B,C,S,H,W] stack is sbc = S*H*W, ss = H*W
static int conv_dispatch(const float* img, const float* psf, float* out, long sbc, long ss, int B, int C, int S, int H, int W,
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
    if (ks >= 3 && ks <= 21 && !getenv("AADFF_CONV_PATH")) {
        // block-GEMM form (conv_psf_map_blkw_kernel): AADFF_CONV_BLKW = 0 never, 1 always, unset: where it measured faster
        // (tools/conv_blkw_probe.py -> profiles/r05_y_conv_blkw_probe.txt, 1024^2, us per launch blkw / wide Toeplitz / packed-FMA -
        // lone slices at grid 7: ks 13 15.0 / 31.6 / 30.8, ks 15 22.3 / 32.0 / 41.3, ks 17 21.4 / 33.5 / 94.8, ks 19 30.7 / 41.4 / 110.5,
        // ks 21 27.3 / 43.6 / 79.8 (grid 11: 24.3 / 38.7 / 80.4); 10-slice stacks (one workgroup per slice and band here, the Toeplitz
        // form shares its staged tile between four slices) at grid 7 / 11: ks 13 105 / 113 and 78 / 105, ks 15 156 / 124 and 113 / 109,
        // ks 17 134 / 131 and 100 / 124, ks 19 291 / 247 and 198 / 217, ks 21 224 / 270 and 158 / 237)
        const char* benv = getenv("AADFF_CONV_BLKW");
        const int blkw = benv ? atoi(benv) : -1;
        // ks 9 / 11 lone slices (us, conv_psf_map_blk_kernel / this kernel with 24- / 32- / 48-row bands; ks 9's window here is 12 x 12 =
        // 5 k-steps, the older kernel's fixed 14 x 14 = 7): ks 9 grid 7 16.3 / 13.0 / 12.1 / 13.2, grid 11 12.9 / 11.1 / 11.4 / 12.0, grid 5
        // 15.1 / 11.7 / 12.1 / 12.8; ks 11 grid 7 15.4 / 15.9 / 13.2 / 14.9, grid 11 12.3 / 12.9 / 13.4 / 12.9, grid 5 14.8 / 15.7 / 14.0 / 14.7
        // ks 3 / 5 / 7 (window ks + 3: 2 / 2 / 4 k-steps; the Toeplitz kernel (b) issues 3 ks MFMAs per 16 x 16 block, this form 3 NST per
        // 16 x 16): lone slices 9 - 12 us against 13 - 17 at every grid, 10-slice stacks faster at ks 5 only (grid 7 / 11: 53 / 47 against 62 / 54)
        const bool use = blkw == 1 || (blkw == -1 && (ks >= 13 ? (S == 1 || ks == 13 || ks == 17 || ks == 21)
                                                                : (S == 1 ? (ks <= 9 || mh > 96) : ks == 5)));
        // rows per band: every workgroup pays ~5.5 us of loads / staging / fragment building in front of 0.76 us of matrix work per
        // 8-row group (tools/conv_single_timeline.py --ks 21: profiles/r05_y_conv_blkw_timeline_ks21.json), so taller bands win until the
        // launch is too few workgroups for two rounds on 512 slots.  Measured at 1024^2 (ks 21, us, RB 24 / 32 / 48): grid 7 (147-row
        // patches) 32.4 / 27.8 / 30.6, grid 11 (94) 25.2 / 25.3 / 24.2, grid 5 (205) 29.1 / 26.8 / 25.4; the other ks alike.
        const char* renv = getenv("AADFF_CONV_BLKW_RB");
        int rb = renv ? atoi(renv) : 0;
        if (rb != 24 && rb != 32 && rb != 48) {
            if (ks <= 11) {
                // 6 workgroups per CU: 32-row bands when that saves a round of workgroups on the chip's 1536 slots (1024^2: grid 7 yes,
                // grid 5 / 11 no - measured, ks 5: 9.4 / 9.1, 9.2 / 9.7, 8.7 / 8.9 us with 24 / 32 rows)
                const size_t per = (size_t)((mw + 95) / 96) * grid * grid * B * C * S;
                const size_t w24 = per * ((mh + 23) / 24), w32 = per * ((mh + 31) / 32);
                rb = (w24 + 1535) / 1536 > (w32 + 1535) / 1536 ? 32 : 24;
            } else rb = mh <= 24 ? 24 : (mh <= 32 ? 32 : (mh <= 48 ? 48 : (mh <= 64 ? 32 : (mh <= 96 ? 48 : 32))));
        }
        const int bntx = (mw + 96 - 1) / 96, bnty = (mh + rb - 1) / rb;
        const size_t gx = (size_t)bntx * grid, gy = (size_t)bnty * grid, total = gx * gy * B * C * S;
        if (use && total < ((size_t)1 << 31) && (size_t)H * W <= ((size_t)1 << 30)) {
            PatchBounds pbb = pb;
            pbb.m_ntx = magic_of(bntx); pbb.m_nty = magic_of(bnty); pbb.m_nchunk = magic_of(S); pbb.m_c = magic_of(C);
            pbb.gx = (unsigned)gx; pbb.gy = (unsigned)gy;
            const bool magic = total < 65536;
            pbb.m_gx = magic ? (gx == 1 ? 1u : magic_of((unsigned)gx)) : 0u;
            pbb.m_gy = magic ? (gy == 1 ? 1u : magic_of((unsigned)gy)) : 0u;
            pbb.xcd_q = total >= 64 ? (unsigned)(total / 8) : 0u;
            pbb.xcd_r = (unsigned)(total % 8);
#define AADFF_BLKW2(K, R) hipLaunchKernelGGL((conv_psf_map_blkw_kernel<K, R>), dim3((unsigned)total), dim3(192), 0, st, img, psf, out, sbc, ss, C, S, H, W, grid, bntx, bnty, pbb)
#define AADFF_BLKW(K) case K: if (rb == 24) AADFF_BLKW2(K, 24); else if (rb == 32) AADFF_BLKW2(K, 32); else AADFF_BLKW2(K, 48); break;
            switch (ks) { AADFF_BLKW(3) AADFF_BLKW(5) AADFF_BLKW(7) AADFF_BLKW(9) AADFF_BLKW(11) AADFF_BLKW(13) AADFF_BLKW(15) AADFF_BLKW(17) AADFF_BLKW(19) AADFF_BLKW(21) }
#undef AADFF_BLKW
#undef AADFF_BLKW2
            AADFF_CHECK_LAUNCH();
            return 0;
        }
    }
    if (ks >= 13 && !getenv("AADFF_CONV_PATH")) {
        // large kernels on the matrix cores (Toeplitz GEMM over NK k-steps); AADFF_CONV_PATH=valu keeps the packed-FMA / generic kernels
        const int nk = (16 + ks - 1 + 31) / 32;                         // 1: ks <= 17, 2: ks <= 49, 3: ks 51
        const int twp = TW + ks - 1, thp = TH + ks - 1;
        int P = (twp + 7) / 8 * 8;
        while (P % 16 != 8) P += 8;                                      // P / 2 dwords = 4 (mod 8): conflict-free A-fragment reads
        if (P < 32 * nk + 16 + 8) P = 32 * nk + 24;                      // the last k-step's 8 halves stay inside the row (zero padding)
        while (P % 16 != 8) P += 8;
        const size_t tile_b = (size_t)2 * thp * P * sizeof(_Float16), rows_b = (size_t)ks * 2 * (32 * nk + 18) * sizeof(_Float16);
        // waves = slices when four tap-row sets fit beside the tile and there are slices to share it; else the waves split the tap rows
        const bool ksplit = S < 2 || tile_b + 4 * rows_b > 64 * 1024 - 64 || ks > 31;
        const size_t lds = tile_b + (ksplit ? std::max(rows_b, (size_t)16384) : 4 * rows_b);
        AADFF_CHECK_ARG(lds <= 64 * 1024 - 64, "render_psf_map: ks %d needs %zu bytes of LDS", ks, lds);
        // slices per workgroup: the whole stack from one staged tile unless that leaves the chip short of workgroups
        const long tiles = (long)ntx * grid * nty * grid * B * C;
        int spw = S;
        while (spw > (ksplit ? 1 : 4) && tiles * ((S + spw - 1) / spw) < 2048) spw = (spw + 1) / 2;
        if (!ksplit) spw = (spw + 3) / 4 * 4;
        const int nchunk = (S + spw - 1) / spw;
        AADFF_CHECK_ARG((size_t)B * C * nchunk <= 65535, "render_psf_map: B*C*chunks too large");
        PatchBounds pbm = pb;
        pbm.m_ntx = magic_of(ntx); pbm.m_nty = magic_of(nty); pbm.m_nchunk = magic_of(nchunk); pbm.m_c = magic_of(C);
        dim3 g(ntx * grid, nty * grid, B * C * nchunk);
#define AADFF_WIDE(NKV, KSP, MR, MT) hipLaunchKernelGGL((conv_psf_map_toeplitz_wide_kernel<NKV, KSP, MR, MT>), g, dim3(256), lds, st, img, psf, out, sbc, ss, C, S, H, \
                                                        W, grid, ks, P, spw, ntx, nty, pbm)
        // size classes: rows per thread = ceil((32 + ks - 1) / 4), taps per thread = ceil(ks^2 / 256) (tap rows split) or / 64 (slices split)
        if (ks <= 17) { if (ksplit) AADFF_WIDE(1, true, 12, 2); else AADFF_WIDE(1, false, 12, 5); }
        else if (ks <= 21) { if (ksplit) AADFF_WIDE(2, true, 13, 2); else AADFF_WIDE(2, false, 13, 7); }
        else if (ks <= 31) { if (ksplit) AADFF_WIDE(2, true, 16, 4); else AADFF_WIDE(2, false, 16, 16); }
        else if (ks <= 49) AADFF_WIDE(2, true, 20, 10);
        else AADFF_WIDE(3, true, 21, 11);
#undef AADFF_WIDE
        AADFF_CHECK_LAUNCH();
        return 0;
    }
    switch (ks) {
#define AADFF_CASE(K) case K: { int rc = launch_fast<K>(img, psf, out, sbc, ss, B, C, S, H, W, grid, ntx, nty, pb, st); if (rc) return rc; } break;
        AADFF_CASE(3) AADFF_CASE(5) AADFF_CASE(7) AADFF_CASE(9) AADFF_CASE(11) AADFF_CASE(13)
        AADFF_CASE(15) AADFF_CASE(21)
#undef AADFF_CASE
        default: {
            dim3 g(ntx * grid, nty * grid, B * C);
            const size_t lds = ((size_t)(TH + ks - 1) * (TW + ks - 1) + (size_t)ks * ks) * sizeof(float);
            hipLaunchKernelGGL(conv_psf_map_generic_kernel, g, dim3(256), lds, st, img, psf, out, sbc, ss, C, S, H, W, grid,
                               ks, ntx, nty, pb);
        }
    }
    AADFF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------
// Per-pixel PSF gather (local_psf_render): HBM-bound on the PSF tensor (ks*ks*4 B/pixel + 24 B/pixel).
// One wave per run of 64 consecutive pixels of one image row:
//   * lane l streams ITS pixel's ks*ks taps straight from global memory with 16-byte loads (the taps of
//     a pixel are contiguous; every byte of every cache line is used, the lines live in L1 for the few
//     loads that share them) - no LDS round trip for the 484 B/pixel stream, so 8+ waves per CU keep
//     tens of KB of loads in flight;
//   * only the C x ks x (64+ks-1) replicate-clamped image window is staged in LDS (read at lane+v:
//     conflict-free);
//   * no flip (render_psf.py:99-105 multiplies unfold() patches with the kernel as is).
// ------------------------------------------------------------------------------------

constexpr int LP_NPX = 64, LP_MAXC = 4;

// Replicate-clamped image window [C][KS][64+KS-1] of one 64-pixel run into LDS.  Every (channel, row) is one coalesced
// load of 64 floats plus a KS-1 float tail, and ALL of them are issued before the first is written to LDS: as a loop of
// dependent load -> store iterations this staging was a chain of ~38 memory latencies per run and, not the PSF stream,
// set the speed of the gather kernels (measured: 196 -> see DESIGN.md).
template <int KS, int CN, int NWV = 1>      // CN > 0: compile-time channel count (no per-channel branches); CN == 0: runtime C <= LP_MAXC
__device__ __forceinline__ void lp_stage_window(const float* __restrict__ img, float* tl, int b, int C, int H, int W, int y, int x0, int lane,
                                                int wave = 0) {                         // NWV waves share the rows: row index % NWV == wave
    constexpr int PAD = KS / 2, TWD = LP_NPX + KS - 1;
    constexpr int MC = CN > 0 ? CN : LP_MAXC;
    float v0[MC * KS], v1[MC * KS];
    const int xa = min(max(x0 - PAD + lane, 0), W - 1);
    const int xb = min(max(x0 - PAD + LP_NPX + lane, 0), W - 1);
#pragma unroll
    for (int cc = 0; cc < MC; ++cc) {
        if (CN > 0 || cc < C) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                if (NWV > 1 && (cc * KS + u) % NWV != wave) continue;
                const int yy = min(max(y - PAD + u, 0), H - 1);
                const float* row = img + ((size_t)(b * C + cc) * H + yy) * W;
                v0[cc * KS + u] = row[xa];
                v1[cc * KS + u] = lane < KS - 1 ? row[xb] : 0.f;
            }
        }
    }
#pragma unroll
    for (int cc = 0; cc < MC; ++cc) {
        if (CN > 0 || cc < C) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                if (NWV > 1 && (cc * KS + u) % NWV != wave) continue;
                tl[(cc * KS + u) * TWD + lane] = v0[cc * KS + u];
                if (lane < KS - 1) tl[(cc * KS + u) * TWD + LP_NPX + lane] = v1[cc * KS + u];
            }
        }
    }
}

template <int KS, int CN>
__global__ __launch_bounds__(64) void local_psf_kernel(const float* __restrict__ img, const float* __restrict__ psf,
                                                        float* __restrict__ out, int C, int H, int W) {
    constexpr int MC = CN > 0 ? CN : LP_MAXC;
    constexpr int KK = KS * KS, TWD = LP_NPX + KS - 1;
    __shared__ float tl[MC * KS * TWD];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * LP_NPX, y = blockIdx.y, b = blockIdx.z;
    const int npx = min(LP_NPX, W - x0);
    const bool act = lane < npx;
    const float* pp = psf + ((size_t)(b * H + y) * W + x0 + (act ? lane : 0)) * KK;

    // first batch of tap loads is in flight while the image window is staged
    constexpr int NB = 8;                                    // 16-byte loads per batch
    constexpr int NV = KK / 4, REM = KK - 4 * NV;            // KK = 4*NV + REM
    f4u wq[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (i < NV) wq[i] = *reinterpret_cast<const f4u*>(pp + 4 * i);

    lp_stage_window<KS, CN>(img, tl, b, C, H, W, y, x0, lane);
    __syncthreads();

    float acc[MC] = {};
    auto tap = [&](int t, float wv) {
        const int u = t / KS, v = t - u * KS;
#pragma unroll
        for (int cc = 0; cc < MC; ++cc)
            if (CN > 0 || cc < C) acc[cc] = fmaf(tl[(cc * KS + u) * TWD + lane + v], wv, acc[cc]);
    };
#pragma unroll
    for (int base = 0; base < NV; base += NB) {
        f4u cur[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) cur[i] = wq[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)                           // next batch goes out before this one is consumed
            if (base + NB + i < NV) wq[i] = *reinterpret_cast<const f4u*>(pp + 4 * (base + NB + i));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (base + i < NV) {
                const int t = 4 * (base + i);
                tap(t, cur[i].x); tap(t + 1, cur[i].y); tap(t + 2, cur[i].z); tap(t + 3, cur[i].w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) tap(4 * NV + i, pp[4 * NV + i]);
    if (act) {
#pragma unroll
        for (int cc = 0; cc < MC; ++cc)
            if (CN > 0 || cc < C) out[((size_t)(b * C + cc) * H + y) * W + x0 + lane] = acc[cc];
    }
}

// LDS-DMA form (default when the rows are 16-byte aligned: W % 4 == 0).  The taps of a 64-pixel run are ONE contiguous
// block of 64*ks*ks*4 bytes (30 976 B at ks 11), so the run is fetched by `global_load_lds_dwordx4` wave-instructions of
// 1 KiB each (64 lanes x 16 consecutive bytes: every request a full line, nothing passes through VGPRs) that are all in
// flight at once, and each lane then reads ITS pixel's taps from LDS at a pitch of ks*ks dwords (odd: the 32 lanes of a
// ds_read_b32 group hit 32 different banks).  The old form let every lane walk its own 484-byte row with 16-byte loads:
// one wave-instruction touched 64 different rows.  One wave per workgroup, 40 KB of LDS -> 4 runs (120 KB) in flight
// per CU while other workgroups of the CU compute.  A workgroup is 2 or 4 waves that split the pieces, the window rows and the
// tap rows of the run (partial sums meet through the freed tap buffer): the shorter a workgroup computes, the larger the
// share of its life it spends with loads in flight.
#ifndef AADFF_LP_DMA_AUX
#define AADFF_LP_DMA_AUX 0          // 2 = nt (streamed-once hint)
#endif
template <int KS, int CN, int NWV>
__global__ __launch_bounds__(64 * NWV) void local_psf_dma_kernel(const float* __restrict__ img, const float* __restrict__ psf,
                                                                  float* __restrict__ out, int C, int H, int W) {
    constexpr int KK = KS * KS, TWD = LP_NPX + KS - 1;
    constexpr int RUN_BYTES = LP_NPX * KK * 4, PIECES = (RUN_BYTES + 1023) / 1024;
    extern __shared__ __attribute__((aligned(16))) float lp_smem[];      // (64*KK + C*KS*TWD) floats: 40 744 B at ks 11, C 3 -> 4 per CU
    float* wl = lp_smem;
    float* tl = lp_smem + LP_NPX * KK;
    const int lane = threadIdx.x & 63;
    const int wave = NWV > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int x0 = blockIdx.x * LP_NPX, y = blockIdx.y, b = blockIdx.z;
    const int npx = min(LP_NPX, W - x0);
    const int bytes = npx * KK * 4;                          // multiple of 16 (W % 4 == 0)
    const char* src = reinterpret_cast<const char*>(psf + ((size_t)(b * H + y) * W + x0) * KK);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        if (NWV > 1 && p % NWV != wave) continue;           // the waves of a workgroup share the run's pieces
        const int off = p * 1024 + lane * 16;
        if (off < bytes)
            __builtin_amdgcn_global_load_lds((lp_gptr_t)(src + off), (lp_lptr_t)(reinterpret_cast<char*>(wl) + p * 1024), 16, 0, AADFF_LP_DMA_AUX);
    }
    lp_stage_window<KS, CN, NWV>(img, tl, b, C, H, W, y, x0, lane, wave);
    __syncthreads();                                         // also waits for the DMA pieces (vmcnt(0))
    constexpr int MC = CN > 0 ? CN : LP_MAXC;
    const float* wr = wl + lane * KK;
    float acc[MC] = {};
    // One wave per SIMD cannot hide LDS latency by occupancy: read a whole tap row (KS taps + MC*KS window values)
    // into registers in one burst, one row ahead of the FMAs that consume it.  With NWV = 2 the tap rows alternate
    // between the two waves (half the compute latency per run: the workgroup spends more of its life with loads in flight).
    auto load_row = [&](int u, float (&w)[KS], float (&x)[MC][KS]) {
#pragma unroll
        for (int v = 0; v < KS; ++v) w[v] = wr[u * KS + v];
#pragma unroll
        for (int cc = 0; cc < MC; ++cc)
            if (CN > 0 || cc < C) {
#pragma unroll
                for (int v = 0; v < KS; ++v) x[cc][v] = tl[(cc * KS + u) * TWD + lane + v];
            }
    };
    float w0[KS], x0r[MC][KS], w1[KS], x1r[MC][KS];
    const int ustart = wave;                                 // rows ustart, ustart + NWV, ...
    if (ustart < KS) load_row(ustart, w0, x0r);
#pragma unroll
    for (int k = 0; k < (KS + NWV - 1) / NWV; ++k) {
        const int u = ustart + k * NWV;
        if (u >= KS) break;
        if (u + NWV < KS) {
            if (k & 1) load_row(u + NWV, w0, x0r); else load_row(u + NWV, w1, x1r);
        }
        asm volatile("" ::: "memory");                       // keep the next row's reads ahead of this row's FMAs
#pragma unroll
        for (int v = 0; v < KS; ++v) {
#pragma unroll
            for (int cc = 0; cc < MC; ++cc)
                if (CN > 0 || cc < C) acc[cc] = fmaf((k & 1) ? x1r[cc][v] : x0r[cc][v], (k & 1) ? w1[v] : w0[v], acc[cc]);
        }
    }
    if (NWV > 1) {                                           // wave 1 hands its partial sums over through the (now free) tap buffer
        __syncthreads();
        if (wave != 0) {
#pragma unroll
            for (int cc = 0; cc < MC; ++cc) wl[((wave - 1) * MC + cc) * 64 + lane] = acc[cc];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < NWV; ++w)
#pragma unroll
            for (int cc = 0; cc < MC; ++cc) acc[cc] += wl[((w - 1) * MC + cc) * 64 + lane];
    }
    if (lane >= npx) return;
#pragma unroll
    for (int cc = 0; cc < MC; ++cc)
        if (CN > 0 || cc < C) out[((size_t)(b * C + cc) * H + y) * W + x0 + lane] = acc[cc];
}

// ------------------------------------------------------------------------------------
// Thin-lens baseline (ThinLens.coc + ThinLens.render, deeplens/psfnet.py:503-570) with the PSF evaluated IN the gather
// kernel: per pixel  coc -> Gaussian exp(-(x^2+y^2)/(2 rad^2)) cut at x^2+y^2 < rad^2 -> L1 normalise -> 11x11 gather over
// the replicate-padded image (render_psf.py:76-107, no flip).  24 B/pixel of HBM traffic + 4 B of depth instead of the
// 508 B/pixel of materialising the [N,H,W,ks,ks] PSF tensor and streaming it through local_psf_render.
//   coc chain in the reference's fp32 op order with IEEE divisions (the hard cut r^2 < rad^2 is discontinuous: a 1-ulp
//   difference in rad^2 next to an integer would flip a whole ring of taps), Gaussian as a product of two 1-D factors
//   (differs from exp of the sum by rounding only), normalisation folded into one division of the gathered sums.
// ------------------------------------------------------------------------------------
template <int KS, int CN>
__global__ __launch_bounds__(64) void thinlens_kernel(const float* __restrict__ img, const float* __restrict__ depth,
                                                       const float* __restrict__ foc_dist, const int* __restrict__ negate,
                                                       float* __restrict__ out, int C, int H, int W, float a_coc, float foc_len,
                                                       float inv_ps, float d_min, float d_max) {
    constexpr int MC = CN > 0 ? CN : LP_MAXC;
    constexpr int PAD = KS / 2, TWD = LP_NPX + KS - 1;
    __shared__ float tl[MC * KS * TWD];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * LP_NPX, y = blockIdx.y, b = blockIdx.z;
    const int npx = min(LP_NPX, W - x0);
    const bool act = lane < npx;
    float d = depth[((size_t)b * H + y) * W + x0 + (act ? lane : 0)];
    float fd = foc_dist[b];
    lp_stage_window<KS, CN>(img, tl, b, C, H, W, y, x0, lane);
    __syncthreads();
    if (negate && *negate) { d = -d; fd = -fd; }            // `if (depth < 0).any()` is a whole-tensor test (psfnet.py:505)
    d = fminf(fmaxf(d, d_min), d_max);
    float rad2;
    {
#pragma clang fp contract(off)
        float coc = a_coc * fabsf(d - fd);                   // foc_len / fnum * |depth - foc_dist| / depth * foc_len / (foc_dist - foc_len)
        coc = coc / d;
        coc = coc * foc_len;
        coc = coc / (fd - foc_len);
        coc = fmaxf(coc * inv_ps, 0.1f);                     // tensor / python-scalar = tensor * (1 / scalar) in ATen
        const float rad = coc * 0.5f;
        rad2 = rad * rad;
    }
    float e[PAD + 1];                                        // exp(-k^2 / 2 / rad^2), k = 0..PAD
#pragma unroll
    for (int k = 0; k <= PAD; ++k) e[k] = __expf((float)(-(k * k)) * 0.5f / rad2);
    float acc[MC] = {};
    float wsum = 0.f;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
#pragma unroll
        for (int v = 0; v < KS; ++v) {
            const int du = u < PAD ? PAD - u : u - PAD, dv = v < PAD ? PAD - v : v - PAD;
            const float wv = (float)(du * du + dv * dv) < rad2 ? e[du] * e[dv] : 0.f;
            wsum += wv;
#pragma unroll
            for (int cc = 0; cc < MC; ++cc)
                if (CN > 0 || cc < C) acc[cc] = fmaf(tl[(cc * KS + u) * TWD + lane + v], wv, acc[cc]);
        }
    }
    if (act) {
        const float inv = 1.f / wsum;
#pragma unroll
        for (int cc = 0; cc < MC; ++cc)
            if (CN > 0 || cc < C) out[((size_t)(b * C + cc) * H + y) * W + x0 + lane] = acc[cc] * inv;
    }
}

// any odd ks / any channel count: PSFs through LDS (the previous design), correctness path
template <int NPX>
__global__ __launch_bounds__(64) void local_psf_generic_kernel(const float* __restrict__ img, const float* __restrict__ psf,
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
    return conv_dispatch(img, psf_map, out, (long)H * W, (long)H * W, B, C, 1, H, W, grid, ks, (hipStream_t)stream);
}

static int stack_launch(const float* img, const float* psf_maps, float* out, long sbc, long ss, int B, int C, int S, int H,
                        int W, int grid, int ks, hipStream_t st) {
    hipEvent_t e0 = g_time_start, e1 = g_time_stop;
    // armed by aadff_time_next_launch: the slice-batched launch attaches the events to its dispatch; every other path
    // records them around the call
    const bool bracket = e0 && !(ks == 11 && S >= 3 && !getenv("AADFF_CONV_PATH"));
    if (bracket) {
        g_time_start = g_time_stop = nullptr;
        AADFF_CHECK_HIP(hipEventRecord(e0, st));
    }
    const int rc = conv_dispatch(img, psf_maps, out, sbc, ss, B, C, S, H, W, grid, ks, st);
    g_time_start = g_time_stop = nullptr;
    if (bracket && rc == 0) AADFF_CHECK_HIP(hipEventRecord(e1, st));
    return rc;
}

int aadff_render_psf_map_stack(const float* img, const float* psf_maps, float* out, int B, int C, int S, int H,
                               int W, int grid, int ks, aadff_stream_t stream) {
    return stack_launch(img, psf_maps, out, (long)S * H * W, (long)H * W, B, C, S, H, W, grid, ks, (hipStream_t)stream);
}

int aadff_render_psf_map_stack_layered(const float* img, const float* psf_maps, const unsigned char* layer_idx, float* out, int B, int C, int S,
                                       int L, int H, int W, int grid, int ks, aadff_stream_t stream) {
    AADFF_CHECK_ARG(img && psf_maps && layer_idx && out, "render_psf_map_stack_layered: NULL pointer");
    AADFF_CHECK_ARG(B > 0 && C > 0 && S > 0 && L >= 1 && L <= 254 && H > 0 && W > 0, "render_psf_map_stack_layered: B=%d C=%d S=%d L=%d H=%d W=%d", B, C, S, L, H, W);
    AADFF_CHECK_ARG(grid >= 1 && grid <= AADFF_MAX_GRID && grid <= H && grid <= W, "render_psf_map_stack_layered: grid %d", grid);
    if (ks != 11) {
        set_error("render_psf_map_stack_layered: the fused form exists for ks 11 only (ks %d: compose aadff_render_psf_map_stack and a gather)", ks);
        return AADFF_EUNSUPPORTED;
    }
    AADFF_CHECK_ARG(5 < H && 5 < W, "render_psf_map_stack_layered: reflect padding 5 needs H,W > pad");
    const long ss = (long)H * W, sbc = (long)S * ss;
    const int maps = S * L;
    PatchBounds pb;
    std::memset(&pb, 0, sizeof(pb));
    fill_bounds(pb.hb, grid, H);
    fill_bounds(pb.wb, grid, W);
    int mh = 0, mw = 0;
    for (int i = 0; i < grid; ++i) {
        mh = std::max(mh, pb.hb[i + 1] - pb.hb[i]);
        mw = std::max(mw, pb.wb[i + 1] - pb.wb[i]);
    }
    constexpr int RB = 24;
    const int nc = maps <= 4 ? 1 : (maps <= 8 ? 2 : (maps <= 12 ? 3 : 4));
    const int npass = (maps + 4 * nc - 1) / (4 * nc);
    const int sntx = (mw + sb::TCOLS - 1) / sb::TCOLS, snty = (mh + RB - 1) / RB;
    AADFF_CHECK_ARG((size_t)B * C * npass <= 65535 && (size_t)snty * grid <= 65535, "render_psf_map_stack_layered: grid too large");
    AADFF_CHECK_ARG((size_t)H * W <= ((size_t)1 << 27) && 4 * ((size_t)S * ss + (size_t)H * W) < ((size_t)1 << 32),
                    "render_psf_map_stack_layered: a (b, c) plane stack above 2^30 elements is not supported");
    pb.m_ntx = magic_of(sntx); pb.m_nty = magic_of(snty); pb.m_nchunk = magic_of(npass); pb.m_c = magic_of(C);
    dim3 gs(sntx * grid, snty * grid, B * C * npass);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = g_time_start, e1 = g_time_stop;                             // aadff_time_next_launch: bracket this launch
    g_time_start = g_time_stop = nullptr;
    if (e0) AADFF_CHECK_HIP(hipEventRecord(e0, st));
#define AADFF_LAUNCH_L(NCV) hipLaunchKernelGGL((conv_psf_map_sbatch_kernel<RB, NCV, false, false, true>), gs, dim3(64 * NCV), 0, st, img, psf_maps, out, sbc, ss, C, \
                                               maps, H, W, grid, sntx, snty, npass, pb, 0, 0, layer_idx, L)
    switch (nc) {
        case 1: AADFF_LAUNCH_L(1); break;
        case 2: AADFF_LAUNCH_L(2); break;
        case 3: AADFF_LAUNCH_L(3); break;
        default: AADFF_LAUNCH_L(4);
    }
#undef AADFF_LAUNCH_L
    if (e0) AADFF_CHECK_HIP(hipEventRecord(e1, st));
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_render_psf_map_stack_strided(const float* img, const float* psf_maps, float* out, long stride_bc, long stride_s,
                                       int B, int C, int S, int H, int W, int grid, int ks, aadff_stream_t stream) {
    AADFF_CHECK_ARG(stride_bc >= (long)H * W && stride_s >= (long)H * W, "render_psf_map_stack_strided: strides %ld / %ld below one plane (%d x %d)",
                    stride_bc, stride_s, H, W);
    // the S*B*C planes must not overlap: either slices are innermost (stride_bc >= S*stride_s) or (b, c) planes are (stride_s >= B*C*stride_bc)
    AADFF_CHECK_ARG(stride_bc >= (long)S * stride_s || stride_s >= (long)B * C * stride_bc,
                    "render_psf_map_stack_strided: planes overlap (stride_bc %ld, stride_s %ld, B*C %d, S %d)", stride_bc, stride_s, B * C, S);
    return stack_launch(img, psf_maps, out, stride_bc, stride_s, B, C, S, H, W, grid, ks, (hipStream_t)stream);
}

#ifdef AADFF_SB_TRACE
extern "C" int aadff_sb_trace_buffer(unsigned long long* dev_buf) {      // not part of the ABI: instrumentation builds only
    AADFF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(aadff::g_sb_trace), &dev_buf, sizeof(dev_buf)));
    return 0;
}
#endif

int aadff_time_next_launch(void* start_event, void* stop_event) {
    AADFF_CHECK_ARG((start_event == nullptr) == (stop_event == nullptr), "time_next_launch: give both events or neither");
    g_time_start = (hipEvent_t)start_event;
    g_time_stop = (hipEvent_t)stop_event;
    return 0;
}

int aadff_render_psf(const float* img, const float* psf, float* out, int B, int C, int H, int W, int ks,
                     aadff_stream_t stream) {
    // a 1x1 PSF grid: the [C,ks,ks] PSF is its own map (render_psf.py:12-28 vs :31-73)
    return conv_dispatch(img, psf, out, (long)H * W, (long)H * W, B, C, 1, H, W, 1, ks, (hipStream_t)stream);
}

int aadff_local_psf_render(const float* img, const float* psf, float* out, int B, int C, int H, int W, int ks,
                           aadff_stream_t stream) {
    AADFF_CHECK_ARG(img && psf && out, "local_psf_render: NULL pointer");
    AADFF_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "local_psf_render: empty tensor");
    AADFF_CHECK_ARG(ks % 2 == 1 && ks >= 1 && ks <= AADFF_MAX_KS, "local_psf_render: ks %d must be odd and <= %d", ks, AADFF_MAX_KS);
    AADFF_CHECK_ARG(H <= 65535 && B <= 65535, "local_psf_render: H or B too large for the launch grid");
    hipStream_t st = (hipStream_t)stream;
    if (C <= LP_MAXC && (ks == 3 || ks == 5 || ks == 7 || ks == 9 || ks == 11 || ks == 13)) {
        dim3 g((W + LP_NPX - 1) / LP_NPX, H, B);
        // LDS-DMA form needs 16-byte aligned runs (rows of W*ks*ks floats, W % 4 == 0); AADFF_LOCAL_PSF=direct forces the
        // per-lane streaming form for A/B runs
        static const bool force_direct = [] { const char* e = std::getenv("AADFF_LOCAL_PSF"); return e && e[0] == 'd'; }();
        const bool dma = !force_direct && W % 4 == 0 && (reinterpret_cast<uintptr_t>(psf) & 15) == 0;
        const size_t dlds = (size_t)(64 * ks * ks + C * ks * (64 + ks - 1)) * sizeof(float);
        // waves per workgroup (they split the run's DMA pieces, window rows and tap rows; ks 11 at 1024^2: 1 -> 145 us,
        // 2 -> 122, 4 -> 101, 8 -> 111; ks 5: 2 -> 25, 4 -> 27): 4 for ks >= 9, 2 below
#define AADFF_LPC(K, CN) do { constexpr int NWV_ = K >= 9 ? 4 : 2; \
                              if (dma) hipLaunchKernelGGL((local_psf_dma_kernel<K, CN, NWV_>), g, dim3(64 * NWV_), dlds, st, img, psf, out, C, H, W); \
                              else hipLaunchKernelGGL((local_psf_kernel<K, CN>), g, dim3(64), 0, st, img, psf, out, C, H, W); } while (0)
#define AADFF_LP(K) case K: if (C == 3) AADFF_LPC(K, 3); else if (C == 1) AADFF_LPC(K, 1); else AADFF_LPC(K, 0); break;
        switch (ks) {
            AADFF_LP(3) AADFF_LP(5) AADFF_LP(7) AADFF_LP(9) AADFF_LP(11) AADFF_LP(13)
        }
#undef AADFF_LP
#undef AADFF_LPC
    } else {
        const int npx = ks <= 15 ? 64 : 16;
        const size_t lds = ((size_t)npx * ks * ks + (size_t)C * ks * (npx + ks - 1)) * sizeof(float);
        AADFF_CHECK_ARG(lds <= 160 * 1024, "local_psf_render: C=%d ks=%d needs %zu B of LDS", C, ks, lds);
        dim3 g((W + npx - 1) / npx, H, B);
        if (npx == 64) {
            if (lds > 64 * 1024)
                AADFF_CHECK_HIP(hipFuncSetAttribute((const void*)local_psf_generic_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(local_psf_generic_kernel<64>, g, dim3(64), lds, st, img, psf, out, C, H, W, ks);
        } else {
            if (lds > 64 * 1024)
                AADFF_CHECK_HIP(hipFuncSetAttribute((const void*)local_psf_generic_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(local_psf_generic_kernel<16>, g, dim3(64), lds, st, img, psf, out, C, H, W, ks);
        }
    }
    AADFF_CHECK_LAUNCH();
    return 0;
}

int aadff_thinlens_render(const float* img, const float* depth, const float* foc_dist, const int* negate_or_null, float* out,
                          int B, int C, int H, int W, int ks, float foc_len_over_fnum, float foc_len, float inv_pixel_size,
                          float d_min, float d_max, aadff_stream_t stream) {
    AADFF_CHECK_ARG(img && depth && foc_dist && out, "thinlens_render: NULL pointer");
    AADFF_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "thinlens_render: empty tensor");
    AADFF_CHECK_ARG(C <= LP_MAXC, "thinlens_render: at most %d channels", LP_MAXC);
    AADFF_CHECK_ARG(ks == 3 || ks == 5 || ks == 7 || ks == 9 || ks == 11 || ks == 13, "thinlens_render: ks %d not in {3,5,...,13}", ks);
    AADFF_CHECK_ARG(H <= 65535 && B <= 65535, "thinlens_render: H or B too large for the launch grid");
    hipStream_t st = (hipStream_t)stream;
    dim3 g((W + LP_NPX - 1) / LP_NPX, H, B);
#define AADFF_TLC(K, CN) hipLaunchKernelGGL((thinlens_kernel<K, CN>), g, dim3(64), 0, st, img, depth, foc_dist, negate_or_null, out, C, H, W, \
                                            foc_len_over_fnum, foc_len, inv_pixel_size, d_min, d_max)
#define AADFF_TL(K) case K: if (C == 3) AADFF_TLC(K, 3); else AADFF_TLC(K, 0); break;
    switch (ks) { AADFF_TL(3) AADFF_TL(5) AADFF_TL(7) AADFF_TL(9) AADFF_TL(11) AADFF_TL(13) }
#undef AADFF_TL
#undef AADFF_TLC
    AADFF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
