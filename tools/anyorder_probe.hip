// Does hipExtAnyOrderLaunch drop the barrier between two kernels of one stream on gfx950?  Kernel A: many workgroups that
// each spin ~T us and record their end time; kernel B records its start time.  Reports B.start - A.end (negative = overlap).
// Build: hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <algorithm>
__global__ void spin(unsigned long long* t_end, long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(t_end, wall_clock64());
}
__global__ void stamp(unsigned long long* t_start) {
    if (threadIdx.x == 0) atomicMin(t_start, wall_clock64());
}
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipStream_t s, s2;
    hipStreamCreate(&s); hipStreamCreate(&s2);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);    // kHz
    const long long ticks = (long long)rate * 100 / 1000;                                // 100 us
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            h[0] = 0; h[1] = ~0ull;
            hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, ticks);
            if (mode == 0) hipLaunchKernelGGL(stamp, dim3(64), dim3(64), 0, s, d + 1);
            if (mode == 1) hipExtLaunchKernelGGL(stamp, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 1);
            if (mode == 2) hipLaunchKernelGGL(stamp, dim3(64), dim3(64), 0, s2, d + 1);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("mode %d (%s): B.start - A.end = %.2f us\n", mode, mode == 0 ? "plain" : mode == 1 ? "any-order" : "other stream",
                   ((double)h[1] - (double)h[0]) / rate * 1e3);
        }
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
