// Host-side continuation of torch's CPU generator stream (no GPU code).
//
// The reference draws its pupil samples with torch.rand on the HOST generator
// (deeplens/optics.py:480-481, deeplens/surfaces.py:192-193); sample-for-sample parity needs
// exactly that stream.  torch.rand costs ~10 ns per float; this routine produces the same
// floats from the same MT19937 state in ~1.5 ns each (scalar) / ~0.3 ns each (AVX2 / AVX-512 builds of the same loops, picked at run
// time) and hands the advanced state back, so torch's global generator stays where the reference's call sequence would
// have left it.
//
// State layout = at::CPUGeneratorImpl::get_state(): CPUGeneratorImplStateLegacy
//   u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] | f64 normal_x,y,rho | i32 normal_valid (+pad)
//   | f32 next_float_normal | bool valid (+pad)            = 5056 bytes
// float32 uniform = (y & (2^24 - 1)) * 2^-24 with y the tempered 32-bit output
// (ATen/core/DistributionsHelper.h uniform_real_distribution<float>), one draw per element, serial.
#include <cstdint>
#include <cstring>
#include "aadff.h"
#include "common.h"

namespace {
constexpr int N = 624, M = 397;
constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;

// The three loops below are written once and compiled twice (baseline x86-64 and AVX2): every iteration reads st[j],
// st[j+1] (not yet updated) and an element >= 227 places away, so blocks of 8 iterations are independent and the
// compiler vectorises them; the tempering + int->float conversion is element-wise.
#define AADFF_MT_BODY(VW)                                                                                              \
    static inline uint32_t twist(uint32_t u, uint32_t v) {                                                         \
        return (((u & UPPER) | (v & LOWER)) >> 1) ^ ((0u - (v & 1u)) & MATRIX_A);                                   \
    }                                                                                                              \
    static void next_state(uint32_t* __restrict__ st) {                                                            \
        int j = 0;                                                                                                 \
        _Pragma(AADFF_STR(clang loop vectorize_width(VW)))                                                         \
        for (; j < N - M; ++j) st[j] = st[j + M] ^ twist(st[j], st[j + 1]);                                         \
        _Pragma(AADFF_STR(clang loop vectorize_width(VW)))                                                         \
        for (; j < N - 1; ++j) st[j] = st[j + M - N] ^ twist(st[j], st[j + 1]);                                     \
        st[N - 1] = st[M - 1] ^ twist(st[N - 1], st[0]);                                                           \
    }                                                                                                              \
    static void temper(const uint32_t* __restrict__ st, float* __restrict__ out, long n) {                         \
        _Pragma(AADFF_STR(clang loop vectorize_width(VW)))                                                         \
        for (long k = 0; k < n; ++k) {                                                                             \
            uint32_t y = st[k];                                                                                    \
            y ^= (y >> 11);                                                                                        \
            y ^= (y << 7) & 0x9d2c5680u;                                                                           \
            y ^= (y << 15) & 0xefc60000u;                                                                          \
            y ^= (y >> 18);                                                                                        \
            out[k] = (float)(int32_t)(y & 0xffffffu) * (1.0f / 16777216.0f);                                       \
        }                                                                                                          \
    }                                                                                                              \
    static void fill(uint32_t* st, int& left, int& nxt, long n, float* out) {                                      \
        long i = 0;                                                                                                \
        while (i < n) {                                                                                            \
            /* at::mt19937::operator(): if (--left == 0) next_state();  y = state[next++] */                       \
            if (left == 1) {                                                                                       \
                next_state(st);                                                                                    \
                left = N + 1;                                                                                      \
                nxt = 0;                                                                                           \
            }                                                                                                      \
            long run = left - 1; /* outputs available before the next regeneration */                              \
            if (run > n - i) run = n - i;                                                                          \
            if (out) temper(st + nxt, out + i, run); /* out == NULL: discard (aadff_host_mt19937_discard) */        \
            i += run;                                                                                              \
            nxt += (int)run;                                                                                       \
            left -= (int)run;                                                                                      \
        }                                                                                                          \
    }

#define AADFF_STR(x) #x
struct Base { AADFF_MT_BODY(4) };
#pragma clang attribute push(__attribute__((target("avx2"))), apply_to = function)
struct Avx2 { AADFF_MT_BODY(8) };
#pragma clang attribute pop
#pragma clang attribute push(__attribute__((target("avx512f,avx512bw,avx512vl"))), apply_to = function)
struct Avx512 { AADFF_MT_BODY(16) };
#pragma clang attribute pop
}  // namespace

struct Gen { uint32_t st[N]; int left, nxt; };

static int load_gen(const unsigned char* torch_state, long state_bytes, Gen& g) {
    AADFF_CHECK_ARG(state_bytes == 5056, "host_mt19937: unexpected torch CPU generator state size %ld (want 5056)", state_bytes);
    int32_t left;
    uint64_t next64;
    std::memcpy(&left, torch_state + 8, 4);
    std::memcpy(&next64, torch_state + 16, 8);
    AADFF_CHECK_ARG(left >= 1 && left <= N && next64 <= (uint64_t)N, "host_mt19937: corrupt generator state (left=%d next=%llu)", left, (unsigned long long)next64);
    for (int i = 0; i < N; ++i) {
        uint64_t v;
        std::memcpy(&v, torch_state + 24 + 8 * i, 8);
        g.st[i] = (uint32_t)v;
    }
    g.left = left;
    g.nxt = (int)next64;
    return 0;
}

static void store_gen(unsigned char* torch_state, const Gen& g) {
    const int32_t left = g.left;
    const uint64_t next64 = (uint64_t)g.nxt;
    std::memcpy(torch_state + 8, &left, 4);
    std::memcpy(torch_state + 16, &next64, 8);
    for (int k = 0; k < N; ++k) {
        const uint64_t v = g.st[k];
        std::memcpy(torch_state + 24 + 8 * k, &v, 8);
    }
}

static void fill_any(Gen& g, long n, float* out) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    static const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    if (avx512) Avx512::fill(g.st, g.left, g.nxt, n, out);
    else if (avx2) Avx2::fill(g.st, g.left, g.nxt, n, out);
    else Base::fill(g.st, g.left, g.nxt, n, out);
}

static int advance(unsigned char* torch_state, long state_bytes, long n, float* out) {
    Gen g;
    if (int rc = load_gen(torch_state, state_bytes, g)) return rc;
    fill_any(g, n, out);
    store_gen(torch_state, g);
    return 0;
}

extern "C" int aadff_host_mt19937_uniform_f32(unsigned char* torch_state, long state_bytes, long n, float* out) {
    AADFF_CHECK_ARG(torch_state && out && n >= 0, "host_mt19937: NULL pointer or negative count");
    return advance(torch_state, state_bytes, n, out);
}

// The state after `n` further float32 draws, without producing them (regenerations only, no tempering / conversion / stores):
// a rank of a sharded job that owns some slices of a scene's stack skips the draws of the others.
extern "C" int aadff_host_mt19937_discard(unsigned char* torch_state, long state_bytes, long n) {
    AADFF_CHECK_ARG(torch_state && n >= 0, "host_mt19937_discard: NULL pointer or negative count");
    return advance(torch_state, state_bytes, n, nullptr);
}

// Rows of draws in TWO passes (single-stack latency: what the first launch of a focal stack needs are the 2 x 2048 focus draws at the
// head of every slice's block, the reference's order puts slice s + 1's behind slice s's PSF draws - SURVEY.md Appendix B):
//   phase 0: from the generator state, for every row r: produce the first `head` draws of the row into out[r * row_len ..], keep a
//            snapshot of the generator there (snapshots[r]: 2504 bytes = 624 words + left + next), SKIP the other row_len - head
//            (regenerations only); the state is left behind the whole block - exactly where torch.rand(n_rows * row_len) leaves it;
//   phase 1: for every row r: from snapshots[r] produce the remaining row_len - head draws into out[r * row_len + head ..]
//            (torch_state is not touched).  The block then equals the one a single pass produces, bit for bit.
extern "C" int aadff_host_mt19937_rows(unsigned char* torch_state, long state_bytes, int n_rows, long row_len, long head, float* out,
                                       unsigned char* snapshots, int phase) {
    AADFF_CHECK_ARG(out && snapshots && n_rows >= 0 && row_len >= 0 && head >= 0 && head <= row_len && (phase == 0 || phase == 1),
                    "host_mt19937_rows: bad arguments");
    static_assert(sizeof(Gen) == 2504, "snapshot size");
    if (phase == 0) {
        AADFF_CHECK_ARG(torch_state, "host_mt19937_rows: NULL state");
        Gen g;
        if (int rc = load_gen(torch_state, state_bytes, g)) return rc;
        for (int r = 0; r < n_rows; ++r) {
            fill_any(g, head, out + (long)r * row_len);
            std::memcpy(snapshots + (size_t)r * sizeof(Gen), &g, sizeof(Gen));
            fill_any(g, row_len - head, nullptr);
        }
        store_gen(torch_state, g);
    } else {
        for (int r = 0; r < n_rows; ++r) {
            Gen g;
            std::memcpy(&g, snapshots + (size_t)r * sizeof(Gen), sizeof(Gen));
            fill_any(g, row_len - head, out + (long)r * row_len + head);
        }
    }
    return 0;
}

// Pupil / aperture points of the strict-parity mode from host uniforms, in the reference's float32 operations
// (deeplens/optics.py:480-486, deeplens/surfaces.py:188-199):  theta = (u_t * 2) * pi,  r = sqrt(u_r * R^2),
// point = (r * cos(theta), r * sin(theta), z) - every step a separately rounded float32 operation, the cosine and sine the SAME
// vector routines torch's CPU kernels call, exported by the libtorch_cpu.so of the caller's process: MKL's vmsCos / vmsSin (and
// vmsSqrt for the root) when torch is built with MKL (ATen/cpu/vml.h; kind 1), else Sleef's u10 functions `Sleef_cosf16_u10` / `Sleef_sinf16_u10` (torch CPU
// capability AVX512; kind 16) or `Sleef_cosf8_u10` / `Sleef_sinf8_u10` (AVX2; kind 8).  The caller resolves them and passes the
// addresses; this library does not link torch.  The host mirror compares a block against torch itself before it trusts this path.
// Row i reads n theta uniforms at u + theta_off[i], n radius uniforms at u + r_off[i] and writes out[i][n][3].
#include <immintrin.h>

namespace {
#pragma clang attribute push(__attribute__((target("avx512f,avx512bw,avx512vl"))), apply_to = function)
void pupil_rows_16(const float* u, long n_rows, const long* t_off, const long* r_off, long n, float pi_f, float R2, float z, float* out,
                   const void* cos_fn, const void* sin_fn) {
    typedef __m512 (*fn_t)(__m512);
    const fn_t cosv = (fn_t)cos_fn, sinv = (fn_t)sin_fn;
    alignas(64) float bx[16], by[16];
    for (long i = 0; i < n_rows; ++i) {
        const float *ut = u + t_off[i], *ur = u + r_off[i];
        float* o = out + i * n * 3;
        for (long k = 0; k < n; k += 16) {
            const int m = (int)(n - k < 16 ? n - k : 16);
            const __mmask16 mask = (__mmask16)((1u << m) - 1u);
            const __m512 tu = _mm512_maskz_loadu_ps(mask, ut + k), ru = _mm512_maskz_loadu_ps(mask, ur + k);
            const __m512 theta = _mm512_mul_ps(_mm512_mul_ps(tu, _mm512_set1_ps(2.f)), _mm512_set1_ps(pi_f));
            const __m512 r = _mm512_sqrt_ps(_mm512_mul_ps(ru, _mm512_set1_ps(R2)));
            _mm512_store_ps(bx, _mm512_mul_ps(r, cosv(theta)));
            _mm512_store_ps(by, _mm512_mul_ps(r, sinv(theta)));
            for (int j = 0; j < m; ++j) {
                o[(k + j) * 3 + 0] = bx[j];
                o[(k + j) * 3 + 1] = by[j];
                o[(k + j) * 3 + 2] = z;
            }
        }
    }
}
#pragma clang attribute pop
#pragma clang attribute push(__attribute__((target("avx2,fma"))), apply_to = function)
void pupil_rows_8(const float* u, long n_rows, const long* t_off, const long* r_off, long n, float pi_f, float R2, float z, float* out,
                  const void* cos_fn, const void* sin_fn) {
    typedef __m256 (*fn_t)(__m256);
    const fn_t cosv = (fn_t)cos_fn, sinv = (fn_t)sin_fn;
    alignas(32) float bt[8], br[8], bx[8], by[8];
    for (long i = 0; i < n_rows; ++i) {
        const float *ut = u + t_off[i], *ur = u + r_off[i];
        float* o = out + i * n * 3;
        for (long k = 0; k < n; k += 8) {
            const int m = (int)(n - k < 8 ? n - k : 8);
            for (int j = 0; j < 8; ++j) {
                bt[j] = j < m ? ut[k + j] : 0.f;
                br[j] = j < m ? ur[k + j] : 0.f;
            }
            const __m256 theta = _mm256_mul_ps(_mm256_mul_ps(_mm256_load_ps(bt), _mm256_set1_ps(2.f)), _mm256_set1_ps(pi_f));
            const __m256 r = _mm256_sqrt_ps(_mm256_mul_ps(_mm256_load_ps(br), _mm256_set1_ps(R2)));
            _mm256_store_ps(bx, _mm256_mul_ps(r, cosv(theta)));
            _mm256_store_ps(by, _mm256_mul_ps(r, sinv(theta)));
            for (int j = 0; j < m; ++j) {
                o[(k + j) * 3 + 0] = bx[j];
                o[(k + j) * 3 + 1] = by[j];
                o[(k + j) * 3 + 2] = z;
            }
        }
    }
}
#pragma clang attribute pop

// torch built with MKL evaluates cos / sin / sqrt of a contiguous float32 CPU tensor with MKL's vector-math calls
// vmsCos / vmsSin / vmsSqrt(n, in, out, VML_HA | VML_FTZDAZ_OFF | VML_ERRMODE_IGNORE) (ATen/cpu/vml.h), not with Sleef / the IEEE root.
void pupil_rows_vml(const float* u, long n_rows, const long* t_off, const long* r_off, long n, float pi_f, float R2, float z, float* out,
                    const void* cos_fn, const void* sin_fn, const void* sqrt_fn) {
    typedef void (*fn_t)(int, const float*, float*, long long);
    const fn_t cosv = (fn_t)cos_fn, sinv = (fn_t)sin_fn, sqrtv = (fn_t)sqrt_fn;
    constexpr long long kMode = 0x2 | 0x00140000 | 0x100;
    constexpr long CH = 512;                             // small calls: MKL never fans them out over its threads
    float th[CH], c[CH], sn[CH], q[CH], r[CH];
    for (long i = 0; i < n_rows; ++i) {
        const float *ut = u + t_off[i], *ur = u + r_off[i];
        float* o = out + i * n * 3;
        for (long k = 0; k < n; k += CH) {
            const long m = n - k < CH ? n - k : CH;
            for (long j = 0; j < m; ++j) {
                th[j] = (ut[k + j] * 2.f) * pi_f;
                q[j] = ur[k + j] * R2;
            }
            cosv((int)m, th, c, kMode);
            sinv((int)m, th, sn, kMode);
            sqrtv((int)m, q, r, kMode);                  // torch.sqrt is MKL's too (not the correctly rounded root everywhere)
            for (long j = 0; j < m; ++j) {
                o[(k + j) * 3 + 0] = r[j] * c[j];
                o[(k + j) * 3 + 1] = r[j] * sn[j];
                o[(k + j) * 3 + 2] = z;
            }
        }
    }
}
}  // namespace

extern "C" int aadff_host_pupil_points(const float* u, long n_rows, const long* theta_off, const long* r_off, long n, float pi_f, float R2,
                                       float z, float* out, const void* cos_fn, const void* sin_fn, const void* sqrt_fn, int width) {
    AADFF_CHECK_ARG(u && theta_off && r_off && out && cos_fn && sin_fn && (sqrt_fn || width != 1) && n_rows >= 0 && n >= 0, "host_pupil_points: NULL pointer or negative size");
    AADFF_CHECK_ARG(width == 1 || width == 8 || width == 16, "host_pupil_points: kind %d (1 = MKL vmsCos / vmsSin, 8 = the AVX2 Sleef routines, 16 = the AVX-512 ones)", width);
    if (width == 1) {
        pupil_rows_vml(u, n_rows, theta_off, r_off, n, pi_f, R2, z, out, cos_fn, sin_fn, sqrt_fn);
    } else if (width == 16) {
        AADFF_CHECK_ARG(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl"),
                        "host_pupil_points: width 16 on a CPU without AVX-512");
        pupil_rows_16(u, n_rows, theta_off, r_off, n, pi_f, R2, z, out, cos_fn, sin_fn);
    } else {
        AADFF_CHECK_ARG(__builtin_cpu_supports("avx2"), "host_pupil_points: width 8 on a CPU without AVX2");
        pupil_rows_8(u, n_rows, theta_off, r_off, n, pi_f, R2, z, out, cos_fn, sin_fn);
    }
    return 0;
}

// ---- np.mean of the countable values of each row (refocus, deeplens/optics.py:1175-1178) in numpy's own arithmetic ------------------
// numpy sums a contiguous float32 vector PAIRWISE (numpy/_core/src/umath/loops_utils.h.src: fewer than 8 elements in order; up to 128
// in eight interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the remainder added in order; longer vectors
// split at n/2 rounded down to a multiple of 8) and np.mean divides that float32 sum by the count IN FLOAT64, then rounds to float32
// (numpy/_core/_methods.py: `ret.dtype.type(ret / rcount)` with rcount an intp scalar).  Checked against numpy itself by the caller
// (aadff/strict_stack.py: _HostFast) and in tests/test_host_logic.py.
namespace {
float np_pairwise_sum(const float* a, long n) {
    if (n < 8) {
        float res = 0.f;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        long i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}
}  // namespace

extern "C" int aadff_host_masked_mean_f32(const float* values, const float* weights, long rows, long n, float* scratch, float* out) {
    AADFF_CHECK_ARG(values && weights && scratch && out && rows >= 0 && n >= 0, "host_masked_mean_f32: NULL pointer or negative size");
    for (long k = 0; k < rows; ++k) {
        const float *v = values + k * n, *w = weights + k * n;
        long m = 0;
        for (long i = 0; i < n; ++i)
            if (w[i] > 0.f && v[i] > 0.f) scratch[m++] = v[i];            // ra > 0, not NaN, > 0: the reference's three filters
        out[k] = m ? (float)((double)np_pairwise_sum(scratch, m) / (double)m) : __builtin_nanf("");
    }
    return 0;
}
