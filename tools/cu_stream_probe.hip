// How fast can ONE workgroup (8 waves on one CU) stream weights it reads once?  Two access shapes for the same bytes:
//   rows:      lane (row = l & 15, kg = l >> 4) reads 16 B at row * 512 + s * 64 + kg * 16   (MFMA A fragments of a row-major
//              [256][256] bf16 matrix: one wave-instruction touches 16 rows x 64 B)
//   fragments: lane l reads 16 B at (instr * 64 + l) * 16                                     (matrix stored in fragment order:
//              one wave-instruction = 1 KiB contiguous)
// 8 workgroups (one per XCD, like the fit chain kernel at batch 128), each streaming `passes` x 128 KB; 16 loads in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O2 tools/cu_stream_probe.hip -o build/cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void stream(const uint4v* __restrict__ w, unsigned* out, int passes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4v acc = {0u, 0u, 0u, 0u};
    for (int p = 0; p < passes; ++p) {
        const char* base = reinterpret_cast<const char*>(w) + (size_t)p * 131072;         // one 256 x 256 bf16 matrix per pass
        uint4v v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            size_t off;
            if (SHAPE == 0) {                                                              // rows: tile (wave + 8 (i & 1)), k-step i >> 1
                const int tile = wave + 8 * (i & 1), s = i >> 1;
                off = (size_t)(tile * 16 + (lane & 15)) * 512 + s * 64 + (lane >> 4) * 16;
            } else {
                off = ((size_t)(wave * 16 + i) * 64 + lane) * 16;
            }
            v[i] = *reinterpret_cast<const uint4v*>(base + off);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1;
}
int main() {
    const int passes = 20, wgs = 8;
    uint4v* w; unsigned* out;
    hipMalloc(&w, (size_t)passes * 131072); hipMalloc(&out, 4);
    hipMemset(w, 1, (size_t)passes * 131072);
    char* junk; hipMalloc(&junk, 512u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(junk, rep, 512u << 20, 0);                                      // cold caches
            hipEventRecord(e0, 0);
            if (shape == 0) hipLaunchKernelGGL(stream<0>, dim3(wgs), dim3(512), 0, 0, w, out, passes);
            else hipLaunchKernelGGL(stream<1>, dim3(wgs), dim3(512), 0, 0, w, out, passes);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %d passes x 128 KB per workgroup in %.1f us = %.1f GB/s per CU\n", shape ? "fragments" : "rows     ", passes, ms * 1e3,
                   passes * 131072.0 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
