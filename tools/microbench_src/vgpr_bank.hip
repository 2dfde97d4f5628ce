// VGPR operand-bank experiment for v_pk_fma_f32 / v_fma_f32 on gfx950: same instruction count, different source-register placement.
// hipcc -O3 --offload-arch=gfx950 tools/microbench_src/vgpr_bank.hip -o /tmp/vgpr_bank && /tmp/vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

// 8 accumulator pairs A0..A7 (explicit registers), operands M (pair) and C (pair or SGPR pair)
#define PK8(A0,A1,A2,A3,A4,A5,A6,A7,M,C) \
    "v_pk_fma_f32 " A0 ", " A0 ", " M ", " C "\n" "v_pk_fma_f32 " A1 ", " A1 ", " M ", " C "\n" \
    "v_pk_fma_f32 " A2 ", " A2 ", " M ", " C "\n" "v_pk_fma_f32 " A3 ", " A3 ", " M ", " C "\n" \
    "v_pk_fma_f32 " A4 ", " A4 ", " M ", " C "\n" "v_pk_fma_f32 " A5 ", " A5 ", " M ", " C "\n" \
    "v_pk_fma_f32 " A6 ", " A6 ", " M ", " C "\n" "v_pk_fma_f32 " A7 ", " A7 ", " M ", " C "\n"
// MAC form: acc = w * g + acc with three distinct pairs
#define MAC8(A0,A1,A2,A3,A4,A5,A6,A7,W,G) \
    "v_pk_fma_f32 " A0 ", " W ", " G ", " A0 "\n" "v_pk_fma_f32 " A1 ", " W ", " G ", " A1 "\n" \
    "v_pk_fma_f32 " A2 ", " W ", " G ", " A2 "\n" "v_pk_fma_f32 " A3 ", " W ", " G ", " A3 "\n" \
    "v_pk_fma_f32 " A4 ", " W ", " G ", " A4 "\n" "v_pk_fma_f32 " A5 ", " W ", " G ", " A5 "\n" \
    "v_pk_fma_f32 " A6 ", " W ", " G ", " A6 "\n" "v_pk_fma_f32 " A7 ", " W ", " G ", " A7 "\n"
#define F8(A0,A1,A2,A3,A4,A5,A6,A7,M,C) \
    "v_fma_f32 " A0 ", " A0 ", " M ", " C "\n" "v_fma_f32 " A1 ", " A1 ", " M ", " C "\n" \
    "v_fma_f32 " A2 ", " A2 ", " M ", " C "\n" "v_fma_f32 " A3 ", " A3 ", " M ", " C "\n" \
    "v_fma_f32 " A4 ", " A4 ", " M ", " C "\n" "v_fma_f32 " A5 ", " A5 ", " M ", " C "\n" \
    "v_fma_f32 " A6 ", " A6 ", " M ", " C "\n" "v_fma_f32 " A7 ", " A7 ", " M ", " C "\n"

#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","s40","s41"

template <int MODE>
__global__ void probe(float *out, int iters) {
    // init: all working registers = small values
    asm volatile("v_mov_b32 v64, 1.0\n v_mov_b32 v65, 1.0\n v_mov_b32 v66, 1.0\n v_mov_b32 v67, 1.0\n v_mov_b32 v68, 1.0\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, 1.0\n v_mov_b32 v71, 1.0\n"
                 "v_mov_b32 v72, 1.0\n v_mov_b32 v73, 1.0\n v_mov_b32 v74, 1.0\n v_mov_b32 v75, 1.0\n v_mov_b32 v76, 1.0\n v_mov_b32 v77, 1.0\n v_mov_b32 v78, 1.0\n v_mov_b32 v79, 1.0\n"
                 "v_mov_b32 v80, 1.0\n v_mov_b32 v81, 1.0\n v_mov_b32 v82, 1.0\n v_mov_b32 v83, 1.0\n v_mov_b32 v84, 1.0\n v_mov_b32 v85, 1.0\n v_mov_b32 v86, 1.0\n v_mov_b32 v87, 1.0\n"
                 "v_mov_b32 v88, 1.0\n v_mov_b32 v89, 1.0\n v_mov_b32 v90, 1.0\n v_mov_b32 v91, 1.0\n v_mov_b32 v92, 1.0\n v_mov_b32 v93, 1.0\n v_mov_b32 v94, 1.0\n v_mov_b32 v95, 1.0\n"
                 "v_mov_b32 v96, 0.5\n v_mov_b32 v97, 0.5\n v_mov_b32 v98, 0.5\n v_mov_b32 v99, 0.5\n v_mov_b32 v100, 0.5\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 0.5\n"
                 "s_mov_b32 s40, 0.5\n s_mov_b32 s41, 0.5\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            // pair classes: a pair v[2j:2j+1] has class j & 1  (banks {0,1} or {2,3} if the file has 4 banks)
            if (MODE == 0)      asm volatile(PK8("v[64:65]","v[66:67]","v[68:69]","v[70:71]","v[72:73]","v[74:75]","v[76:77]","v[78:79]","v[96:97]","v[98:99]") ::: CLOB);   // consecutive accs, M class0, C class1
            else if (MODE == 1) asm volatile(PK8("v[64:65]","v[68:69]","v[72:73]","v[76:77]","v[80:81]","v[84:85]","v[88:89]","v[92:93]","v[98:99]","s[40:41]") ::: CLOB);   // accs class0, M class1, C sgpr
            else if (MODE == 2) asm volatile(PK8("v[64:65]","v[68:69]","v[72:73]","v[76:77]","v[80:81]","v[84:85]","v[88:89]","v[92:93]","v[96:97]","s[40:41]") ::: CLOB);   // accs class0, M class0, C sgpr
            else if (MODE == 3) asm volatile(PK8("v[64:65]","v[68:69]","v[72:73]","v[76:77]","v[80:81]","v[84:85]","v[88:89]","v[92:93]","v[98:99]","v[102:103]") ::: CLOB); // accs class0, M class1, C class1
            else if (MODE == 4) asm volatile(PK8("v[64:65]","v[68:69]","v[72:73]","v[76:77]","v[80:81]","v[84:85]","v[88:89]","v[92:93]","v[98:99]","v[100:101]") ::: CLOB); // accs class0, M class1, C class0
            else if (MODE == 5) asm volatile(MAC8("v[64:65]","v[68:69]","v[72:73]","v[76:77]","v[80:81]","v[84:85]","v[88:89]","v[92:93]","v[98:99]","v[102:103]") ::: CLOB); // MAC form: acc class0, w class1, g class1
            else if (MODE == 6) asm volatile(PK8("v[64:65]","v[68:69]","v[72:73]","v[76:77]","v[80:81]","v[84:85]","v[88:89]","v[92:93]","1.0","0.5") ::: CLOB);              // inline constants only
            else if (MODE == 7) asm volatile(F8("v64","v68","v72","v76","v80","v84","v88","v92","v97","v98") ::: CLOB);        // fma: acc bank0, m bank1, c bank2
            else if (MODE == 8) asm volatile(F8("v64","v68","v72","v76","v80","v84","v88","v92","v96","v100") ::: CLOB);       // fma: all bank0
            else if (MODE == 9) asm volatile(F8("v64","v68","v72","v76","v80","v84","v88","v92","1.0","0.5") ::: CLOB);        // fma: constants
            else if (MODE == 10) asm volatile(F8("v64","v68","v72","v76","v80","v84","v88","v92","v97","s40") ::: CLOB);       // fma: acc bank0, m bank1, sgpr
        }
    }
    float r;
    asm volatile("v_add_f32 %0, v64, v68\n v_add_f32 %0, %0, v72\n v_add_f32 %0, %0, v65" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int CU = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    float *out; CHK(hipMalloc(&out, sizeof(float) * CU * 4 * 256));
    const char *nm[] = {"pk: accs consecutive, M c0, C c1 (3 VGPR pairs)", "pk: acc c0, M c1, C sgpr", "pk: acc c0, M c0, C sgpr", "pk: acc c0, M c1, C c1", "pk: acc c0, M c1, C c0",
                        "pk MAC form: acc c0, w c1, g c1", "pk: inline constants", "fma: banks 0,1,2", "fma: banks 0,0,0", "fma: constants", "fma: banks 0,1 + sgpr"};
    const int iters = 2000, wps = 4;
    for (int mode = 0; mode < 11; ++mode) {
        auto launch = [&]() {
            switch (mode) {
#define L(M) case M: probe<M><<<CU * wps, 256>>>(out, iters); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10)
            }
        };
        launch(); CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) { CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-52s %.3f ms -> %.2f cycles/wave-instr/SIMD @%.2f GHz nominal\n", nm[mode], best, best * 1e-3 * ghz * 1e9 / ((double)iters * 64 * wps), ghz);
    }
    return 0;
}
