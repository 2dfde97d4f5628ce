#include <hip/hip_runtime.h>
#include <cstdio>
// Does ds_read ignore address bits above the LDS size?  (float-bits-as-address trick)
__global__ void k(float* out, unsigned hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* w = (float2*)smem;
    for (int i = threadIdx.x; i < 4096; i += 64) w[i] = make_float2((float)i, -(float)i);
    __syncthreads();
    unsigned addr = (threadIdx.x * 3 * 8) | hi;
    float2 v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x] = v.x;
}
int main() {
    float* o; hipMalloc(&o, 256);
    for (unsigned hi : {0u, 0x00040000u, 0x4B000000u, 0x4B400000u << 3, 0x80000000u}) {
        k<<<1, 64, 32768>>>(o, hi); float h[64]; hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
        printf("hi=%08x: %g %g %g %g ... %g\n", hi, h[0], h[1], h[2], h[3], h[63]);
    }
}
