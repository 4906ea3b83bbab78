// Victim kernels for tools/concurrency_isa_probe.py: each runs ONE class of instruction the strict trace is made of in a long
// dependent chain on seeded inputs and writes the chain's last value, so that two launches on the same inputs must agree bit for
// bit.  The probe launches them beside a busy second stream and counts launches whose output differs from a quiet launch.
// Built by the probe: hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -Iinclude -I<pkg>/csrc tools/conc_victims.hip -o tools/conc_victims.so
#include <hip/hip_runtime.h>
#include "strict_math2.h"
using aadff::strict::div2; using aadff::strict::sqrt2; using aadff::strict::normalize32; using aadff::strict::recip2;

#pragma clang fp contract(off)

using aadff::strict::f2;
using namespace aadff::strict;
typedef const __attribute__((address_space(4))) aadff_surface_t* csurf_t;
typedef const __attribute__((address_space(4))) float* cfloat_t;

template <int OP>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int iters, const float* table) {
    __shared__ float sh[256];
    __shared__ unsigned bits[8];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 8) bits[threadIdx.x] = 0u;
    sh[threadIdx.x] = in[i % n] * 0.5f;
    __syncthreads();
    float x = in[i % n], y = in[(i + 977) % n];
    f2 v = (f2){x, y};
    const f2 a = (f2){1.0000001f, 0.9999999f}, b = (f2){1e-3f, -1e-3f};
    Surf sf{};
    if (OP >= 17) sf = make_surf_from((csurf_t)table + 8, 1);                                   // rf50mm surface 8: the first aspheric one
    for (int k = 0; k < iters; ++k) {
        if (OP == 0) v = __builtin_elementwise_fma(v, a, b);                                       // v_pk_fma_f32
        else if (OP == 1) v = v * a + b;                                                           // v_pk_mul_f32, v_pk_add_f32
        else if (OP == 2) { x = __builtin_fmaf(x, 1.0000001f, 1e-3f); y = __builtin_fmaf(y, 0.9999999f, -1e-3f); }   // v_fma_f32
        else if (OP == 3) { x = __builtin_amdgcn_rcpf(x) + 1.5f; y = __builtin_amdgcn_rcpf(y) + 2.5f; }              // v_rcp_f32
        else if (OP == 4) { x = __builtin_amdgcn_sqrtf(x) + 1.5f; y = __builtin_amdgcn_sqrtf(y) + 2.5f; }            // v_sqrt_f32
        else if (OP == 5) { x = (x + 3.f) / (y + 2.f) + 1.f; y = (y + 1.f) / (x + 5.f) + 1.f; }                      // the compiler's IEEE division
        else if (OP == 6) { x = __builtin_amdgcn_div_fixupf(x * 0.99f, y + 2.f, x) + 1e-3f; y = y * 0.999f + 1e-3f; }  // v_div_fixup_f32
        else if (OP == 7) {                                                                        // LDS traffic: neighbour's value, wave-local
            const float t = sh[(threadIdx.x & 192) | ((threadIdx.x + k) & 63)];
            x = x * 0.5f + t * 0.25f;
            if ((k & 15) == 0) { const unsigned m = 1u << (__float_as_uint(x) & 31); if (m & ~bits[k & 7]) atomicOr(&bits[k & 7], m); }
        } else if (OP == 8) {                                                                      // scalar loads of a constant table
            const cfloat_t t = (cfloat_t)table;
            x = x * 0.5f + t[(k * 7 + (blockIdx.x & 3)) & 1023] * 0.25f;
        } else if (OP == 9) {                                                                      // coherent vector loads + readfirstlane
            const float t = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int,
                                __hip_atomic_load(table + ((k * 7 + (blockIdx.x & 3)) & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
            x = x * 0.5f + t * 0.25f;
        } else if (OP == 10) {                                                                     // float64 powers (v_mul_f64, conversions)
            double p = (double)x, r = p;
            for (int q = 1; q < 5; ++q) r *= p;
            x = (float)r * 0.03f + 0.9f;
            double p2 = (double)y, r2 = p2 * p2;
            r2 *= p2; r2 *= p2;
            y = (float)r2 * 0.05f + 0.8f;
        } else if (OP == 11) {                                                                     // the aspheric terms: scalar coefficients x float64 powers
            const cfloat_t t = (cfloat_t)table;
            float z = x * 0.25f;
            for (int j = 1; j < 6; ++j) {
                double p = (double)x, r = p;
                for (int q = 1; q < j + 1; ++q) r *= p;
                z = z + (t[j] * 1e-3f) * (float)r;
            }
            x = z * 0.5f + 0.6f;
        } else if (OP == 12) {                                                                     // div2: v_rcp_f32 pair -> packed refinement
            v = div2(v + a, v * a + 1.5f) + 0.75f;
        } else if (OP == 13) {                                                                     // sqrt2: v_sqrt_f32 pair -> packed residuals
            v = sqrt2(v + a) + 0.75f;
        } else if (OP == 14) {                                                                     // normalize32
            f2 p = v, q = v * a + b, r = v + 1.f;
            normalize32(p, q, r);
            v = p + q * 0.5f + r * 0.25f + 0.3f;
        } else if (OP == 15) {                                                                     // v_rcp_f32 pair then ONE packed op on the pair
            const f2 r0 = (f2){__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)};
            v = __builtin_elementwise_fma(r0, a, b) + 1.25f;
        } else if (OP == 16) {                                                                     // the same with an independent instruction slot after the pair
            f2 r0 = (f2){__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)};
            asm volatile("s_nop 1" : "+v"(r0));
            v = __builtin_elementwise_fma(r0, a, b) + 1.25f;
        } else if (OP == 17) {                                                                     // the even-polynomial terms, half by half
            const f2 r2 = v * 20.f;
            v = (f2){sag_poly(sf, r2.x, v.x), sag_poly(sf, r2.y, v.y)} * 0.9f + 0.1f;
        } else if (OP == 18) {
            const f2 r2 = v * 20.f;
            v = (f2){dsag_poly(sf, r2.x, v.x), dsag_poly(sf, r2.y, v.y)} * 0.9f + 0.1f;
        } else if (OP == 19) {                                                                     // dsag: IEEE divisions, sqrtf, polynomial
            const f2 r2 = v * 20.f;
            v = (f2){dsag(sf, r2.x), dsag(sf, r2.y)} * 0.5f + 1.f;
        } else if (OP == 20) {                                                                     // sag_dsag2: packed conic part + polynomial
            const f2 r2 = v * 20.f;
            f2 z, g;
            sag_dsag2(sf, r2, z, g);
            v = z * 0.1f + g + 1.f;
        } else if (OP == 22) {                                                                     // control: sag_dsag2 with the polynomial branch off
            Surf s0 = sf; s0.n_ai = 0;
            const f2 r2 = v * 20.f;
            f2 z, g;
            sag_dsag2(s0, r2, z, g);
            v = z * 0.1f + g + 1.f;
        } else if (OP >= 23 && OP <= 27) {                                                         // the first instructions of the polynomial block, exactly
            // y half as the compiler emits it: cvt_f64 | pk_mov (t = {v.hi, S}) | mul_f64 | pk_mul (t * v.hi); variants take pieces away
            double dd, qq;
            f2 t;
            const float S = 1.0000002f;
            if (OP == 23) asm volatile("v_cvt_f64_f32 %0, %3\n\tv_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n\tv_mul_f64 %1, %0, %0\n\tv_pk_mul_f32 %2, %2, %4 op_sel:[0,1]"
                                       : "=&v"(dd), "=&v"(qq), "=&v"(t) : "v"(v.y), "v"(v), "s"((f2){S, S}));
            else if (OP == 24) asm volatile("v_cvt_f64_f32 %0, %3\n\tv_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 %2, %2, %4 op_sel:[0,1]\n\tv_mul_f64 %1, %0, %0"
                                       : "=&v"(dd), "=&v"(qq), "=&v"(t) : "v"(v.y), "v"(v), "s"((f2){S, S}));
            else if (OP == 25) asm volatile("v_cvt_f64_f32 %0, %3\n\tv_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n\tv_mul_f64 %1, %0, %0\n\ts_nop 0\n\tv_pk_mul_f32 %2, %2, %4 op_sel:[0,1]"
                                       : "=&v"(dd), "=&v"(qq), "=&v"(t) : "v"(v.y), "v"(v), "s"((f2){S, S}));
            else if (OP == 26) asm volatile("v_cvt_f64_f32 %0, %3\n\tv_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n\tv_mul_f64 %1, %0, %0\n\ts_nop 4\n\tv_pk_mul_f32 %2, %2, %4 op_sel:[0,1]"
                                       : "=&v"(dd), "=&v"(qq), "=&v"(t) : "v"(v.y), "v"(v), "s"((f2){S, S}));
            else asm volatile("v_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 %2, %2, %4 op_sel:[0,1]\n\tv_cvt_f64_f32 %0, %3\n\tv_mul_f64 %1, %0, %0"
                                       : "=&v"(dd), "=&v"(qq), "=&v"(t) : "v"(v.y), "v"(v), "s"((f2){S, S}));
            v = (f2){t.y * 0.25f + 0.9f, t.x * 0.2f + (float)qq * 0.1f + 0.6f};
        } else if (OP >= 28 && OP <= 32) {                                                         // ... taken apart further: no float64 at all
            f2 t;
            const float S = 1.0000002f;
            if (OP == 28) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "=&v"(t) : "v"(v), "s"((f2){S, S}));
            else if (OP == 29) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(t) : "v"(v), "s"((f2){S, S}));          // t = {v.hi, S}
            else if (OP == 30) { t = v * 1.5f; asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(t) : "v"(v)); }          // {t.lo v.hi, t.hi v.hi}
            else if (OP == 31) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(t) : "v"(v), "v"((f2){S, S}));          // VGPR second source
            else { t = v * 1.5f; asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(t) : "v"(v)); }                      // {t.lo v.lo, t.hi v.lo}
            v = (f2){t.y * 0.25f + 0.9f, t.x * 0.2f + 0.7f};
        } else if (OP == 43) {                                                                     // SGPR pair as the SECOND source, its high half to both lanes
            f2 t = v * 1.5f;                                                                       // (how the compiler broadcasts an odd scalar register)
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(t) : "s"((f2){1.0000002f, 0.9999998f}));
            v = (f2){t.y * 0.25f + 0.9f, t.x * 0.2f + 0.7f};
        } else if (OP >= 33 && OP <= 42) {                                                         // which packed forms: op_sel on the other source, SGPR sources, fma, add
            f2 t = v * 1.5f;
            const f2 S2 = (f2){1.0000002f, 0.9999998f};
            if (OP == 33) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0]" : "+v"(t) : "v"(v));                                    // {t.hi v.lo, t.hi v.hi}
            else if (OP == 34) asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0]" : "=&v"(t) : "v"(v), "s"(S2));                     // SGPR pair, high half to the low lane
            else if (OP == 35) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[1,0,0]" : "+v"(t) : "v"(v));
            else if (OP == 36) asm volatile("v_pk_fma_f32 %0, %2, %1, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=&v"(t) : "v"(v), "s"(S2));   // conv.hip:90 (VALU path)
            else if (OP == 37) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,0,1]" : "+v"(t) : "v"(v));
            else if (OP == 38) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(t) : "v"(v));
            else if (OP == 39) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(t) : "v"(v));               // swapped halves (csrc/strict.hip)
            else if (OP == 40) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(t) : "v"(v));
            else if (OP == 41) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1]" : "+v"(t) : "v"(v));                               // both sources from the high halves
            else asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,0]" : "+v"(t) : "v"(v));                                          // both lanes from the low halves
            v = (f2){t.y * 0.25f + 0.9f, t.x * 0.2f + 0.7f};
        } else if (OP == 21) {                                                                     // one loose Newton residual
            const R32 o = {v, v * 0.5f, f2s(40.f)}, d = {f2s(0.05f), f2s(-0.03f), f2s(0.998f)};
            f2 ft, dfdt;
            residual2<false, true>(sf, o, d, f2s(1.f), v + 9.f, ft, dfdt);
            v = ft * 0.01f + dfdt * 0.1f + 1.2f;
        }
    }
    if (OP <= 1 || OP >= 12) { x = v.x; y = v.y; }
    if (OP == 7) { __syncthreads(); y = (float)bits[threadIdx.x & 7]; }
    out[2 * (size_t)i] = x;
    out[2 * (size_t)i + 1] = y;
}

extern "C" int victim_launch(int op, const float* in, float* out, int n, int blocks, int iters, const float* table, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (op) {
#define V(K) case K: victim_kernel<K><<<blocks, 256, 0, s>>>(in, out, n, iters, table); break;
        V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(8) V(9) V(10) V(11) V(12) V(13) V(14) V(15) V(16) V(17) V(18) V(19) V(20) V(21) V(22) V(23) V(24) V(25) V(26) V(27) V(28) V(29) V(30) V(31) V(32) V(33) V(34) V(35) V(36) V(37) V(38) V(39) V(40) V(41) V(42) V(43)
#undef V
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- aggressors: one feature of the convolution kernels each ------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int OP>
__global__ __launch_bounds__(256) void aggressor_kernel(float* __restrict__ out, int iters) {
    __shared__ float big[OP == 1 || OP == 3 ? 6500 : 64];
    const int tid = threadIdx.x;
    float r = 0.f;
    if (OP == 0 || OP == 3) {                                                                      // MFMA chain (3: with 26 KB of LDS held)
        half8 ah, bh;
        for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(0.001f * (tid + j)); bh[j] = (_Float16)(0.002f * (tid - j)); }
        float4v acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < iters; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        r = acc[0] + acc[1] + acc[2] + acc[3];
        if (OP == 3) { big[tid] = r; __syncthreads(); r += big[(tid * 7) & 255]; }
    } else if (OP == 1) {                                                                          // 26 KB of LDS, written and read all the time
        for (int k = 0; k < iters; ++k) {
            big[(tid + k * 256) % 6500] = (float)k;
            r += big[(tid * 3 + k * 64) % 6500];
        }
    } else if (OP == 2) {                                                                          // plain VALU
        float x = (float)tid;
        for (int k = 0; k < iters; ++k) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        r = x;
    }
    out[(size_t)blockIdx.x * 256 + tid] = r;
}

extern "C" int aggressor_launch(int op, float* out, int blocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (op) {
        case 0: aggressor_kernel<0><<<blocks, 256, 0, s>>>(out, iters); break;
        case 1: aggressor_kernel<1><<<blocks, 256, 0, s>>>(out, iters); break;
        case 2: aggressor_kernel<2><<<blocks, 256, 0, s>>>(out, iters); break;
        case 3: aggressor_kernel<3><<<blocks, 256, 0, s>>>(out, iters); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
