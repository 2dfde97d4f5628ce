#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R8(S) S S S S S S S S
template <int MODE> __global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2}, q = {a0 * 0.5f, a1 * 0.25f}, r = {a2 * 0.1f, a3 * 0.2f};
    float sc = seed * 1.0001f; unsigned long long sc2 = ((unsigned long long)__float_as_uint(sc) << 32) | __float_as_uint(sc);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(R8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(r));
        if (MODE == 1) asm volatile(R8("v_pk_fma_f32 %0, %0, %4, %5 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %2, %2, %4, %5 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %3, %3, %4, %5 op_sel_hi:[1,1,0]\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "s"(sc2));
        if (MODE == 2) asm volatile(R8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        if (MODE == 3) asm volatile(R8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        if (MODE == 4) asm volatile(R8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q.x), "s"(sc));
        if (MODE == 5) asm volatile(R8("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q.x), "v"(r.x));
        if (MODE == 6) asm volatile(R8("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q.x));
        if (MODE == 7) asm volatile(R8("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %5, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(r));
        if (MODE == 8) asm volatile(R8("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); const int CU = p.multiProcessorCount;
    float* out; hipMalloc(&out, sizeof(float) * CU * 4 * 256);
    const char* nm[] = {"pk_fma 3vgpr", "pk_fma 2vgpr+sgpr", "pk_mul", "pk_add", "fma 2vgpr+sgpr", "fmac 3vgpr", "mul", "pk_fma opsel-bcast (MAC form)", "lshl_add"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 9; ++mode) for (int wps : {2, 4}) {
        const int iters = 4000, blocks = CU * wps;
        auto L = [&]() { switch (mode) { case 0: k<0><<<blocks,256>>>(out,iters,1.f); break; case 1: k<1><<<blocks,256>>>(out,iters,1.f); break; case 2: k<2><<<blocks,256>>>(out,iters,1.f); break; case 3: k<3><<<blocks,256>>>(out,iters,1.f); break; case 4: k<4><<<blocks,256>>>(out,iters,1.f); break; case 5: k<5><<<blocks,256>>>(out,iters,1.f); break; case 6: k<6><<<blocks,256>>>(out,iters,1.f); break; case 7: k<7><<<blocks,256>>>(out,iters,1.f); break; case 8: k<8><<<blocks,256>>>(out,iters,1.f); break; } };
        L(); hipDeviceSynchronize(); float best = 1e9;
        for (int r = 0; r < 3; ++r) { hipEventRecord(e0); L(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-32s waves/SIMD=%d: %.3f ns per wave-instr per SIMD\n", nm[mode], wps, best * 1e6 / ((double)iters * 32 * wps));
    }
}
