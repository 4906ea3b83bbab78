#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB> __global__ __launch_bounds__(320) void k(float* o) { __shared__ float s[KB * 256]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = s[(threadIdx.x * 7) % (KB * 256)]; }
template <int KB> void probe() { int n = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<KB>, 320, 0); printf("static LDS %3d KB, 320 threads: max active blocks/CU = %d\n", KB, n); }
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    probe<8>(); probe<11>(); probe<16>(); probe<29>(); probe<32>(); probe<40>(); probe<64>();
    return 0;
}
