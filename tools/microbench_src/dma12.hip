#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
template <int SZ> __global__ void k(const unsigned* x, unsigned* y, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeef;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    if constexpr (SZ == 12) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)smem, 12, lane * 12, 8, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)smem, 16, lane * 16, 8, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) y[i] = ((unsigned*)smem)[i];
}
int main() {
    unsigned *x, *y; hipMalloc(&x, 8192); hipMalloc(&y, 4096);
    unsigned h[2048]; for (int i = 0; i < 2048; ++i) h[i] = i; hipMemcpy(x, h, 8192, hipMemcpyHostToDevice);
    for (int sz : {12, 16}) {
        if (sz == 12) k<12><<<1, 64, 8192>>>(x, y, 8192); else k<16><<<1, 64, 8192>>>(x, y, 8192);
        unsigned o[1024]; hipMemcpy(o, y, 4096, hipMemcpyDeviceToHost);
        printf("size %d:", sz); for (int i = 0; i < 40; ++i) printf(" %x", o[i]); printf(" ... [190..200]:"); for (int i = 188; i < 200; ++i) printf(" %x", o[i]); printf(" [252..260]:"); for (int i = 252; i < 260; ++i) printf(" %x", o[i]); printf("\n");
    }
    return 0;
}
