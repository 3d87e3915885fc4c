// probe: semantics of ds_read_b64_tr_b16 on gfx950 (which source lane/element lands where)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned short* out) {
    __shared__ unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + threadIdx.x * 8;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipMalloc(&d, 256 * 2);
    k<<<1, 64>>>(d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d   (src lane,elem: %d.%d %d.%d %d.%d %d.%d)\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3],
        h[4*l]/4, h[4*l]%4, h[4*l+1]/4, h[4*l+1]%4, h[4*l+2]/4, h[4*l+2]%4, h[4*l+3]/4, h[4*l+3]%4);
    return 0;
}
