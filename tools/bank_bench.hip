// Does v_pk_fma_f32 slow down when the accumulator pair and the x pair sit in the same VGPR bank pair?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    // acc pairs: v[8:9] v[12:13] ... (all = 0 mod 4);  x pairs: MODE0 -> v[42:43].. (2 mod 4), MODE1 -> v[40:41].. (0 mod 4)
    asm volatile(
        "s_mov_b32 s20, 0x3f800054\n s_mov_b32 s21, 0x3f7fff58\n"
        "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n"
        "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
        "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 0.5\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n"
        ::: "v8","v9","v12","v13","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","v40","v41","v42","v43","v44","v45","v46","v47","s20","s21");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile(
                ".rept 8\n"
                "v_pk_fma_f32 v[8:9], s[20:21], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[12:13], s[20:21], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 v[16:17], s[20:21], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[20:21], s[20:21], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 v[24:25], s[20:21], v[42:43], v[24:25] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[28:29], s[20:21], v[46:47], v[28:29] op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 v[32:33], s[20:21], v[42:43], v[32:33] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[36:37], s[20:21], v[46:47], v[36:37] op_sel_hi:[0,1,1]\n"
                ".endr\n" ::: "v8","v9","v12","v13","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37");
        } else if (MODE == 1) {
            asm volatile(
                ".rept 8\n"
                "v_pk_fma_f32 v[8:9], s[20:21], v[40:41], v[8:9] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[12:13], s[20:21], v[44:45], v[12:13] op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 v[16:17], s[20:21], v[40:41], v[16:17] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[20:21], s[20:21], v[44:45], v[20:21] op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 v[24:25], s[20:21], v[40:41], v[24:25] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[28:29], s[20:21], v[44:45], v[28:29] op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 v[32:33], s[20:21], v[40:41], v[32:33] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[36:37], s[20:21], v[44:45], v[36:37] op_sel_hi:[0,1,1]\n"
                ".endr\n" ::: "v8","v9","v12","v13","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37");
        } else {   // MODE 2: only 2 accumulators alternating (dependency distance 2), no bank conflict
            asm volatile(
                ".rept 32\n"
                "v_pk_fma_f32 v[8:9], s[20:21], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[12:13], s[20:21], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n"
                ".endr\n" ::: "v8","v9","v12","v13");
        }
    }
    float r;
    asm volatile("v_add_f32 %0, v8, v12" : "=v"(r));
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> void bench(const char* name, float* out) {
    const int iters = 4096;
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-34s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, wps, ms, 2.0 * 2 * 64 * iters * 64.0 * blocks / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    bench<0>("pk_fma acc/x different banks", out);
    bench<1>("pk_fma acc/x SAME bank pair", out);
    bench<2>("pk_fma 2 chains (dep distance 2)", out);
    return 0;
}
