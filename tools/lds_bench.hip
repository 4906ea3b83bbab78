// LDS read micro-benchmark: cycles per wave-instruction of ds_read2_b32 / ds_read_b64 for given per-lane
// dword addresses (bank-conflict rules on gfx950).  Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/lds_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
#include <string>

template <int OFF1, int MODE>   // MODE 0: ds_read2_b32 offset0:0 offset1:OFF1 ; 1: ds_read_b64 ; 2: ds_read_b32
__global__ void k(const int* addr, long long* cycles, unsigned* sink, int iters) {
    __shared__ unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)addr[threadIdx.x & 63] * 4u;
    unsigned long long acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        unsigned long long r[16];
#define RD2(i) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:%2" : "=v"(r[i]) : "v"(a), "n"(OFF1))
#define RD64(i) asm volatile("ds_read_b64 %0, %1" : "=v"(r[i]) : "v"(a))
#define RD32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(*(unsigned*)&r[i]) : "v"(a))
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (MODE == 0) RD2(i);
            else if constexpr (MODE == 1) RD64(i);
            else { r[i] = 0; RD32(i); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(r[i])); acc ^= r[i]; }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)acc ^ (unsigned)(acc >> 32);
}

int main() {
    int* d_addr; long long* d_cyc; unsigned* d_sink;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_cyc, 8 * 8); hipMalloc(&d_sink, 1024 * 4 * 4);
    const int iters = 1000, waves = 16;      // 4 waves = one per SIMD on one CU: LDS-throughput bound
    auto run = [&](const char* name, std::function<int(int)> f, int mode, int off1) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) a[l] = f(l);
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d_addr, d_cyc, d_sink, iters); };
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 1) launch(k<0, 1>);
            else if (mode == 2) launch(k<0, 2>);
            else if (off1 == 1) launch(k<1, 0>);
            else if (off1 == 136) launch(k<136, 0>);
            else if (off1 == 72) launch(k<72, 0>);
            else if (off1 == 32) launch(k<32, 0>);
            else if (off1 == 160) launch(k<160, 0>);
            else if (off1 == 144) launch(k<144, 0>);
            else if (off1 == 16) launch(k<16, 0>);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        // clock64 ticks at 100 MHz on gfx9: report ticks per instruction-wave as relative numbers
        printf("%-58s  ticks per wave-instr (16 waves) %.4f\n", name, (double)c / iters / 16);
    };
    auto cxkg = [](int l, int rpd, int kgmul) { int cx = l & 15, kg = l >> 4; return cx + (kg & 1) * kgmul * rpd + (kg >> 1) * 3; };
    run("b32 linear (conflict-free)", [](int l) { return l; }, 2, 0);
    run("b32 all lanes stride 32 (32-way)", [](int l) { return (l & 31) * 32; }, 2, 0);
    run("read2 linear, off1=32", [](int l) { return l; }, 0, 32);
    run("read2 linear, off1=16", [](int l) { return l; }, 0, 16);
    run("read2 linear, off1=1", [](int l) { return l; }, 0, 1);
    run("read2 mine: cx + kg*2*136, off1=136", [&](int l) { return cxkg(l, 136, 2); }, 0, 136);
    run("read2 cx + kg*2*72, off1=72", [&](int l) { return cxkg(l, 72, 2); }, 0, 72);
    run("read2 cx + (kg&1)*16, off1=32", [](int l) { return (l & 15) + ((l >> 4) & 1) * 16 + (l >> 5) * 64; }, 0, 32);
    run("read2 cx + (kg&1)*16, off1=160", [](int l) { return (l & 15) + ((l >> 4) & 1) * 16 + (l >> 5) * 64; }, 0, 160);
    run("read2 cx + (kg&1)*16, off1=144", [](int l) { return (l & 15) + ((l >> 4) & 1) * 16 + (l >> 5) * 64; }, 0, 144);
    run("read2 cx + (kg&1)*16, off1=136", [](int l) { return (l & 15) + ((l >> 4) & 1) * 16 + (l >> 5) * 64; }, 0, 136);
    run("read2 cx + (kg&1)*16, off1=1", [](int l) { return (l & 15) + ((l >> 4) & 1) * 16 + (l >> 5) * 64; }, 0, 1);
    run("b64 linear 2 dwords per lane", [](int l) { return 2 * l; }, 1, 0);
    run("b64 misaligned (odd dword) linear", [](int l) { return 2 * l + 1; }, 1, 0);
    run("b64 lane stride 1 dword (overlapping windows)", [](int l) { return l; }, 1, 0);
    run("b64 even lanes A, odd lanes B(+80), kg*32", [](int l) { int cx = l & 15, kg = l >> 4; return (cx & ~1) + (cx & 1) * 80 + (kg & 1) * 32 + (kg >> 1) * 160; }, 1, 0);
    run("b64 even lanes A, odd lanes B(+64), kg*32", [](int l) { int cx = l & 15, kg = l >> 4; return (cx & ~1) + (cx & 1) * 64 + (kg & 1) * 32 + (kg >> 1) * 160; }, 1, 0);
    return 0;
}
