#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(int* o) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  sharedMemPerMultiprocessor %zu  maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d\n", p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    for (int kb : {8, 16, 20, 24, 28, 30, 32, 34, 36, 37, 38, 40, 48, 52, 56, 64, 80}) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, (size_t)kb * 1024);
        printf("dyn LDS %3d KiB -> %d blocks/CU (%s)\n", kb, n, hipGetErrorString(e));
    }
    return 0;
}
