// Microbenchmark: fp32 FMA issue rate on gfx950, plain v_fmac (SGPR or VGPR weight) vs v_pk_fma_f32,
// at 1/2/4/8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/fma_bench.hip -o /tmp/fma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, const float* wsrc, int iters) {
    float w = wsrc[blockIdx.x & 1];                      // uniform -> SGPR
    float wv = wsrc[threadIdx.x & 1];                    // VGPR weight
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    if (MODE == 0 || MODE == 1) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = 0.f;
        const float ww = MODE == 0 ? w : wv;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(ww, x[i], a[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(a[i]));
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += a[i];
        out[blockIdx.x * 64 + threadIdx.x] = s;
    } else {
        float2v a[8], xx[8];
        for (int i = 0; i < 8; ++i) { a[i] = (float2v){0.f, 0.f}; xx[i] = (float2v){x[2 * i], x[2 * i + 1]}; }
        const float2v ww = MODE == 2 ? (float2v){w, w} : (float2v){wv, wv};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(ww, xx[i], a[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
        out[blockIdx.x * 64 + threadIdx.x] = s;
    }
}

template <int MODE>
void bench(const char* name, float* out, float* w) {
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, w, iters);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, w, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double flop = 2.0 * 128 * iters * 64.0 * blocks;     // 128 scalar FMAs per iter per lane
        printf("%-28s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, wps, ms, flop / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float *out, *w;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    hipMalloc(&w, 16);
    float hw[4] = {1.0001f, 0.9999f, 1.f, 1.f};
    hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
    bench<0>("v_fmac  sgpr weight", out, w);
    bench<1>("v_fmac  vgpr weight", out, w);
    bench<2>("v_pk_fma sgpr weight", out, w);
    bench<3>("v_pk_fma vgpr weight", out, w);
    return 0;
}
