// issue.hip -- gfx950 issue-cost probes in SHADER CYCLES (s_memtime), not wall time: the clock an instruction mix sustains differs
// (DVFS), so wall-time probes cannot tell a 2-cycle from a 4-cycle instruction.  One workgroup per CU (LDS-limited), WAVES waves;
// every wave runs REP x an unrolled body of 64 instructions between two s_memtime reads; reported: cycles per wave-instruction per
// SIMD = elapsed / (REP * 64 * waves per SIMD), mean over all waves, plus the wall clock the mix ran at.
//   build: hipcc -O3 --offload-arch=gfx950 tools/microbench_src/issue.hip -o tools/scratch/issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define R8(S) S S S S S S S S
enum { FMA32, PKFMA, PKFMA_BC, PKMUL, PKADD, FMA64, ADD32, MIXED_PK_DS, MIXED_FMA_DS, DS64, DS128, PKFMA_SGPR, CVT, NMODES };
static const char *NAMES[NMODES] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel bcast", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f64", "v_add_f32",
                                    "pair-loop mix: 32 pk_fma + 21 pk + 32 ds_read_b64", "same with 64 v_fma_f32 for the MACs", "ds_read_b64 only (4-tap gathers)",
                                    "ds_read_b128 only (2 per gather, 8-byte aligned)", "v_pk_fma_f32 with an SGPR-pair operand", "v_cvt_f64_f32 / v_cvt_f32_f64"};

template <int MODE>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, float *sink, int rep, float seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *w = (float2 *)smem;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) w[i] = make_float2(i * 1e-3f, -i * 1e-3f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0001f, c = 0.5f;
    const v2f m2 = {m, m}, c2 = {c, c};
    // the kernel's gather pattern: 8 x 8 pixels per wave, 2 samples per pixel of depth, a column every ~3 samples; 8-byte samples
    const uint32_t addr = ((uint32_t)(lane & 7) * 2u + (uint32_t)(lane >> 3) * 3u + (uint32_t)wave * 64u) * 8u;
    v2f t0, t1, t2, t3, t4, t5, t6, t7;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f q0, q1, q2, q3;
    asm volatile("s_barrier");
    const unsigned long long s0 = __builtin_readcyclecounter();
    for (int i = 0; i < rep; ++i) {
        if constexpr (MODE == FMA32) {
            asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (MODE == ADD32) {
            asm volatile(R8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                            "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if constexpr (MODE == PKFMA) {
            asm volatile(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                            "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
        } else if constexpr (MODE == PKFMA_BC) {    // acc = tap * w.x(broadcast) + acc, as the MACs of the pair loop
            asm volatile(R8("v_pk_fma_f32 %0, %9, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %9, %8, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                            "v_pk_fma_f32 %2, %9, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %9, %8, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                            "v_pk_fma_f32 %4, %9, %8, %4 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %9, %8, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                            "v_pk_fma_f32 %6, %9, %8, %6 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %9, %8, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
        } else if constexpr (MODE == PKFMA_SGPR) {
            asm volatile(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                            "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(m2), "v"(c2));
        } else if constexpr (MODE == PKMUL) {
            asm volatile(R8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                            "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2));
        } else if constexpr (MODE == PKADD) {
            asm volatile(R8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                            "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
        } else if constexpr (MODE == FMA64) {
            asm volatile(R8("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                            "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
        } else if constexpr (MODE == CVT) {
            asm volatile(R8("v_cvt_f64_f32 %0, %8\n v_cvt_f32_f64 %9, %0\n v_cvt_f64_f32 %1, %8\n v_cvt_f32_f64 %9, %1\n"
                            "v_cvt_f64_f32 %2, %8\n v_cvt_f32_f64 %9, %2\n v_cvt_f64_f32 %3, %8\n v_cvt_f32_f64 %9, %3\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(a0), "+v"(a1));
        } else if constexpr (MODE == DS64) {          // 64 ds_read_b64 = 16 four-tap gathers, a wait per 16 reads
            #pragma unroll
            for (int g = 0; g < 4; ++g) {
                asm volatile("ds_read_b64 %0, %8 offset:0\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                             "ds_read_b64 %4, %8 offset:1024\n ds_read_b64 %5, %8 offset:1032\n ds_read_b64 %6, %8 offset:1040\n ds_read_b64 %7, %8 offset:1048\n"
                             "ds_read_b64 %0, %8 offset:2048\n ds_read_b64 %1, %8 offset:2056\n ds_read_b64 %2, %8 offset:2064\n ds_read_b64 %3, %8 offset:2072\n"
                             "ds_read_b64 %4, %8 offset:3072\n ds_read_b64 %5, %8 offset:3080\n ds_read_b64 %6, %8 offset:3088\n ds_read_b64 %7, %8 offset:3096\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr + g * 4096u));
                p0 += t0 + t4; p1 += t1 + t5; p2 += t2 + t6; p3 += t3 + t7;
            }
        } else if constexpr (MODE == DS128) {         // the same bytes as 32 ds_read_b128 (8-byte aligned addresses: unaligned access mode)
            #pragma unroll
            for (int g = 0; g < 4; ++g) {
                asm volatile("ds_read_b128 %0, %4 offset:0\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:1024\n ds_read_b128 %3, %4 offset:1040\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr + g * 4096u));
                p0 += (v2f){q0.x + q1.z, q0.y + q1.w}; p1 += (v2f){q2.x + q3.z, q2.y + q3.w};
                asm volatile("ds_read_b128 %0, %4 offset:2048\n ds_read_b128 %1, %4 offset:2064\n ds_read_b128 %2, %4 offset:3072\n ds_read_b128 %3, %4 offset:3088\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr + g * 4096u));
                p2 += (v2f){q0.x + q1.z, q0.y + q1.w}; p3 += (v2f){q2.x + q3.z, q2.y + q3.w};
            }
        } else if constexpr (MODE == MIXED_PK_DS || MODE == MIXED_FMA_DS) {
            // one transmit pair of the reciprocal + mirror loop: 32 ds_read_b64 (2 transmits x 4 sets x 4 taps), 21 packed index / weight
            // instructions between issue and wait, 32 packed MACs (or 64 scalar FMAs)
            #pragma unroll
            for (int half = 0; half < 2; ++half) {
                asm volatile("ds_read_b64 %0, %8 offset:0\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                             "ds_read_b64 %4, %8 offset:1024\n ds_read_b64 %5, %8 offset:1032\n ds_read_b64 %6, %8 offset:1040\n ds_read_b64 %7, %8 offset:1048\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr + half * 16384u));
                v2f u0, u1, u2, u3, u4, u5, u6, u7;
                asm volatile("ds_read_b64 %0, %8 offset:2048\n ds_read_b64 %1, %8 offset:2056\n ds_read_b64 %2, %8 offset:2064\n ds_read_b64 %3, %8 offset:2072\n"
                             "ds_read_b64 %4, %8 offset:3072\n ds_read_b64 %5, %8 offset:3080\n ds_read_b64 %6, %8 offset:3088\n ds_read_b64 %7, %8 offset:3096\n"
                             : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(addr + half * 16384u));
                if (half == 0) {   // "weights": 21 packed ops on p4..p7
                    asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %0, %4\n v_pk_add_f32 %2, %1, %4\n v_pk_mul_f32 %3, %2, %2\n v_pk_mul_f32 %0, %3, %5\n"
                                 "v_pk_fma_f32 %0, %3, %5, %4\n v_pk_fma_f32 %1, %3, %5, %4\n v_pk_fma_f32 %2, %3, %5, %4\n v_pk_fma_f32 %0, %0, %3, %4\n"
                                 "v_pk_fma_f32 %1, %1, %3, %4\n v_pk_fma_f32 %2, %2, %3, %4\n v_pk_fma_f32 %0, %0, %3, %5\n v_pk_fma_f32 %1, %1, %3, %5\n"
                                 "v_pk_fma_f32 %2, %2, %3, %5\n v_pk_mul_f32 %0, %0, %3\n v_pk_mul_f32 %1, %1, %3\n v_pk_fma_f32 %2, %2, %3, %0\n"
                                 "v_pk_fma_f32 %0, %0, %3, %1\n v_pk_fma_f32 %1, %1, %3, %2\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5\n"
                                 : "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(m2));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7),
                             "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
                if constexpr (MODE == MIXED_PK_DS) {
                    p0 = t0 * p4.x + p0; p1 = t4 * p4.y + p1; p2 = u0 * p4.x + p2; p3 = u4 * p4.y + p3;
                    p0 = t1 * p5.x + p0; p1 = t5 * p5.y + p1; p2 = u1 * p5.x + p2; p3 = u5 * p5.y + p3;
                    p0 = t2 * p6.x + p0; p1 = t6 * p6.y + p1; p2 = u2 * p6.x + p2; p3 = u6 * p6.y + p3;
                    p0 = t3 * p7.x + p0; p1 = t7 * p7.y + p1; p2 = u3 * p7.x + p2; p3 = u7 * p7.y + p3;
                } else {
#define SF(ACC, TAP, W) asm volatile("v_fma_f32 %0, %2, %4, %0\n v_fma_f32 %1, %3, %4, %1" : "+v"(ACC.x), "+v"(ACC.y) : "v"(TAP.x), "v"(TAP.y), "v"(W))
                    float ax = p0.x, ay = p0.y, bx = p1.x, by = p1.y, cx = p2.x, cy = p2.y, dx = p3.x, dy = p3.y;
#define SF2(X, Y, TAP, W) asm volatile("v_fma_f32 %0, %2, %4, %0\n v_fma_f32 %1, %3, %4, %1" : "+v"(X), "+v"(Y) : "v"(TAP.x), "v"(TAP.y), "v"(W))
                    SF2(ax, ay, t0, p4.x); SF2(bx, by, t4, p4.y); SF2(cx, cy, u0, p4.x); SF2(dx, dy, u4, p4.y);
                    SF2(ax, ay, t1, p5.x); SF2(bx, by, t5, p5.y); SF2(cx, cy, u1, p5.x); SF2(dx, dy, u5, p5.y);
                    SF2(ax, ay, t2, p6.x); SF2(bx, by, t6, p6.y); SF2(cx, cy, u2, p6.x); SF2(dx, dy, u6, p6.y);
                    SF2(ax, ay, t3, p7.x); SF2(bx, by, t7, p7.y); SF2(cx, cy, u3, p7.x); SF2(dx, dy, u7, p7.y);
                    p0 = (v2f){ax, ay}; p1 = (v2f){bx, by}; p2 = (v2f){cx, cy}; p3 = (v2f){dx, dy};
                }
            }
        }
    }
    const unsigned long long s1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = s1 - s0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE> static void run(int CU, int waves, unsigned long long *dout, float *sink) {
    const int rep = (MODE == MIXED_PK_DS || MODE == MIXED_FMA_DS) ? 2000 : 4000;
    const int threads = waves * 64, blocks = CU;
    const size_t lds = 100 * 1024;             // one workgroup per CU
    CHK(hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    probe<MODE><<<blocks, threads, lds>>>(dout, sink, rep, 1.f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    probe<MODE><<<blocks, threads, lds>>>(dout, sink, rep, 1.f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * waves);
    CHK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : h) sum += (double)v;
    const double ticks = sum / h.size();
    const double wps = waves / 4.0;
    // instructions per body: 64 for the single-instruction modes; mixes: 2 halves x (16 ds) ... reported per body instead
    const bool mix = (MODE == MIXED_PK_DS || MODE == MIXED_FMA_DS);
    const double per = mix ? ticks / rep : ticks / (rep * 64.0 * (wps < 1 ? 1 : wps));
    printf("%-58s waves/SIMD=%d : %9.0f ticks, %8.3f ms -> %s %.2f  (ticks per ms: %.3e = GHz if a tick is a shader cycle)\n", NAMES[MODE], (int)wps, ticks, ms,
           mix ? "ticks per transmit pair (32 reads + 21 + 32|64 VALU) per WAVE:" : "ticks per wave-instruction per SIMD:", per, ticks / ms * 1e-6);
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int CU = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, CU);
    unsigned long long *dout; float *sink;
    CHK(hipMalloc(&dout, sizeof(unsigned long long) * CU * 16));
    CHK(hipMalloc(&sink, sizeof(float) * CU * 1024));
    for (int waves : {4, 16}) {
        run<FMA32>(CU, waves, dout, sink); run<ADD32>(CU, waves, dout, sink); run<PKFMA>(CU, waves, dout, sink); run<PKFMA_BC>(CU, waves, dout, sink);
        run<PKFMA_SGPR>(CU, waves, dout, sink); run<PKMUL>(CU, waves, dout, sink); run<PKADD>(CU, waves, dout, sink); run<FMA64>(CU, waves, dout, sink);
        run<CVT>(CU, waves, dout, sink); run<DS64>(CU, waves, dout, sink); run<DS128>(CU, waves, dout, sink);
        run<MIXED_PK_DS>(CU, waves, dout, sink); run<MIXED_FMA_DS>(CU, waves, dout, sink);
    }
    return 0;
}
