#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_dpp(float v) {
    v = fminf(v, dppf<0xB1>(v));        // quad_perm [1,0,3,2]
    v = fminf(v, dppf<0x4E>(v));        // quad_perm [2,3,0,1]
    v = fminf(v, dppf<0x141>(v));       // row_half_mirror
    v = fminf(v, dppf<0x140>(v));       // row_mirror
    v = fminf(v, dppf<0x142, 0xa>(v));  // row_bcast:15 -> rows 1,3
    v = fminf(v, dppf<0x143, 0xc>(v));  // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(__float_as_int(v), 63) == 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)) : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__global__ void k(const float *in, float *out) {
    const float v = in[threadIdx.x];
    out[threadIdx.x] = wave_min_dpp(v);
}
int main() {
    float h[64], *d, *o, r[64];
    for (int t = 0; t < 20; ++t) {
        float ref = 1e30f;
        for (int i = 0; i < 64; ++i) { h[i] = (float)((i * 7919 + t * 104729) % 1000) - 300.f; if (h[i] < ref) ref = h[i]; }
        hipMalloc(&d, 256); hipMalloc(&o, 256);
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, o);
        hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
        int ok = 1; for (int i = 0; i < 64; ++i) if (r[i] != ref) ok = 0;
        if (!ok) { printf("MISMATCH t=%d ref=%g got=%g\n", t, ref, r[0]); return 1; }
    }
    printf("dpp wave_min OK\n");
    return 0;
}
