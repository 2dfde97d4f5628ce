// ds_read_b128 at 8-byte-aligned (not 16-byte-aligned) LDS addresses on gfx950: does it work, and at what rate, compared with
// 2 x ds_read_b64?  Gather pattern of the DAS kernel: lane l reads samples starting at base + l*stride (8-byte samples).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void probe(float *out, int iters, int stride8, int odd) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // 8x8 wave footprint: depth = lane & 7 (stride samples), column = lane >> 3 (jitter 1 sample)
    uint32_t addr = (uint32_t)(((lane & 7) * stride8 / 8 * 2 + (lane >> 3)) * 2 + odd) * 8u % 16384u;
    v4f acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            v2f a, b, c, d;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr));
            acc += (v4f){a.x + c.x, a.y + c.y, b.x + d.x, b.y + d.y};
        } else {
            v4f a, b;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr));
            acc += (v4f){a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        }
        addr = (addr + 64u) % 16384u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int CU = p.multiProcessorCount;
    float *out; CHK(hipMalloc(&out, sizeof(float) * CU * 1024));
    float *h = (float *)malloc(sizeof(float) * CU * 1024);
    for (int odd = 0; odd < 2; ++odd) for (int stride8 : {8, 16}) {
        double ms[2]; float v[2];
        for (int mode = 0; mode < 2; ++mode) {
            auto launch = [&]() { if (mode == 0) probe<0><<<CU, 1024, 65536>>>(out, 4000, stride8, odd); else probe<1><<<CU, 1024, 65536>>>(out, 4000, stride8, odd); };
            launch(); CHK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float t; CHK(hipEventElapsedTime(&t, e0, e1)); ms[mode] = t;
            CHK(hipMemcpy(h, out, sizeof(float) * 1024, hipMemcpyDeviceToHost)); v[mode] = h[5];
        }
        printf("odd=%d stride=%d samples/lane: 4x b64 %.3f ms, 2x b128 %.3f ms, values %s (%g vs %g)\n", odd, stride8 / 8 * 2, ms[0], ms[1], v[0] == v[1] ? "EQUAL" : "DIFFER", v[0], v[1]);
    }
    return 0;
}
