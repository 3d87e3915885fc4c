#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ float sm[];
__global__ __launch_bounds__(256, 2) void k(float* o) { sm[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = sm[255 - threadIdx.x]; }
int main() {
    for (int bytes : {65536, 80000, 81920, 82000, 83000, 86016, 98304}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, bytes);
        printf("lds %d -> %d blocks/CU\n", bytes, nb);
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu sharedMemPerBlock %zu\n", p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
}
