// microbench.hip -- gfx950 instruction-throughput probes that decide the DAS kernel's inner-loop shape.
//   build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench
// Reports cycles per wave-instruction per SIMD (VALU probes, 1..8 waves/SIMD) and LDS cycles per
// wave-instruction per CU for the gather patterns of the tiled kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void valu_probe(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0001f, c = 0.5f;
    const v2f m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) {        // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE == 1) { // 8 independent v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
            } else if (MODE == 2) { // pk_fma with op_sel broadcast of src1 low half
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %4, %4, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                             "v_pk_fma_f32 %6, %6, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
            } else if (MODE == 3) { // transcendental: v_sin_f32
                asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 4) { // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
            } else if (MODE == 5) { // v_cvt_u32_f32 + v_fract mix (cheap VALU)
                asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 6) { // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 7) { // v_cndmask
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

// LDS gather: each lane reads 4 consecutive 8-byte samples starting at base + lane*stride (in samples)
template <int MODE>
__global__ void lds_probe(float *out, int iters, int stride8, int jitter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *w = (float2 *)smem;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) w[i] = make_float2(i, -i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // stride8 = stride in 1/8 samples (16 -> 2.0 samples per lane)
    int idx = (lane * stride8) >> 3;
    idx += (jitter ? ((lane * 7) & 1) : 0);
    float ax = 0.f, ay = 0.f;
    for (int i = 0; i < iters; ++i) {
        const int a = (idx + (i & 63)) * 8;
        if (MODE == 0) {          // 4 x ds_read_b64
            float2 s0, s1, s2, s3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a));
            ax += s0.x + s1.x + s2.x + s3.x; ay += s0.y + s1.y + s2.y + s3.y;
        } else if (MODE == 1) {   // 2 x ds_read2_b64
            float4 s0, s1;
            asm volatile("ds_read2_b64 %0, %2 offset1:1\n ds_read2_b64 %1, %2 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(s0), "=&v"(s1) : "v"(a));
            ax += s0.x + s0.z + s1.x + s1.z; ay += s0.y + s0.w + s1.y + s1.w;
        } else if (MODE == 2) {   // de-interleaved: 2 bases (even / odd halves), 4 x ds_read_b64
            const int h = (idx + (i & 63)) >> 1, par = (idx + (i & 63)) & 1;
            const int ae = (h + par) * 8, ao = (2048 + h) * 8;
            float2 s0, s1, s2, s3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %5\n ds_read_b64 %3, %5 offset:8\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(ae), "v"(ao));
            ax += s0.x + s1.x + s2.x + s3.x; ay += s0.y + s1.y + s2.y + s3.y;
        } else if (MODE == 3) {   // 4 x ds_read_b32 (fp16 complex samples, 4-byte)
            float s0, s1, s2, s3;
            const int a4 = a >> 1;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a4));
            ax += s0 + s1; ay += s2 + s3;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay;
}

template <typename F> static float time_ms(F f, int reps = 3) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(e0)); f(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int CU = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz nominal\n", p.name, CU, ghz);
    float *out; CHK(hipMalloc(&out, sizeof(float) * CU * 8 * 1024));
    const char *vn[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel", "v_sin_f32", "v_fma_f64", "v_fract_f32", "v_rcp_f32", "v_cndmask_b32"};
    const int iters = 2000;
    for (int mode = 0; mode < 8; ++mode) {
        for (int wps : {1, 2, 4}) {                       // waves per SIMD
            const int threads = 256, blocks = CU * wps;   // 4 waves/block -> 1 wave per SIMD per block
            auto launch = [&]() {
                switch (mode) {
                    case 0: valu_probe<0><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 1: valu_probe<1><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 2: valu_probe<2><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 3: valu_probe<3><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 4: valu_probe<4><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 5: valu_probe<5><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 6: valu_probe<6><<<blocks, threads>>>(out, iters, 1.f); break;
                    case 7: valu_probe<7><<<blocks, threads>>>(out, iters, 1.f); break;
                }
            };
            const float ms = time_ms(launch);
            const double inst_per_simd = (double)iters * 64 * wps;      // wave-instructions issued on one SIMD
            printf("VALU %-22s waves/SIMD=%d : %.3f ms -> %.2f cycles/wave-instr/SIMD @nominal clock\n", vn[mode], wps, ms,
                   ms * 1e-3 * ghz * 1e9 / inst_per_simd);
        }
    }
    const char *ln[] = {"4x ds_read_b64", "2x ds_read2_b64", "deinterleaved 4x b64", "4x ds_read_b32"};
    for (int mode = 0; mode < 4; ++mode)
        for (int stride8 : {8, 12, 16, 20}) for (int jit : {0, 1}) {
            const int threads = 256, wpc = 12, blocks = CU * (wpc / 4);
            const int it = 4000;
            auto launch = [&]() {
                switch (mode) {
                    case 0: lds_probe<0><<<blocks, threads, 48 * 1024>>>(out, it, stride8, jit); break;
                    case 1: lds_probe<1><<<blocks, threads, 48 * 1024>>>(out, it, stride8, jit); break;
                    case 2: lds_probe<2><<<blocks, threads, 48 * 1024>>>(out, it, stride8, jit); break;
                    case 3: lds_probe<3><<<blocks, threads, 48 * 1024>>>(out, it, stride8, jit); break;
                }
            };
            const float ms = time_ms(launch);
            const double sets_per_cu = (double)it * wpc;                 // 4-tap gathers (one per lane) per CU
            printf("LDS  %-22s stride=%.2f jitter=%d : %.3f ms -> %.1f cycles per 64-lane 4-tap gather per CU\n", ln[mode], stride8 / 8.0, jit, ms,
                   ms * 1e-3 * ghz * 1e9 / sets_per_cu);
        }
    return 0;
}
