// Second reproducer attempt: temporaries from the default stream-ordered pool (release threshold 0: returned to the driver at every
// synchronisation), written by one kernel and read by the next through SCALAR loads, with hipMalloc / hipFree of other buffers in between -- the pattern of
// qdas_shift_sum behind a call-per-launch gateway.   hipcc --offload-arch=gfx950 stale_pool.hip -o stale_pool && ./stale_pool [rounds] [keep_pool]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)
typedef const __attribute__((address_space(4))) int *cint;

__global__ void write_kernel(const float *src, int *tab, int n, int salt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = (int)(src[i] * 16.f) + salt;
}
__global__ void read_kernel(const int *tab, int *out, int n) {           // uniform index: scalar loads
    int acc = 0;
    for (int i = 0; i < n; ++i) acc += ((cint)tab)[i] * (i + 1);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, keep = argc > 2 ? atoi(argv[2]) : 0, mode = argc > 3 ? atoi(argv[3]) : 0, n = 15;
    // mode bit 0: no hipMalloc / hipFree churn (the staged buffers live across the calls); bit 1: no pool temporaries (hipMalloc'ed once)
    float *p_big = nullptr, *p_src = nullptr;
    int *p_out = nullptr, *p_tab = nullptr, *p_blk = nullptr;
    if (mode & 1) { CK(hipMalloc(&p_big, 300 * 1024)); CK(hipMalloc(&p_src, n * 4)); CK(hipMalloc(&p_out, 64)); }
    if (mode & 2) { CK(hipMalloc(&p_tab, n * 4)); CK(hipMalloc(&p_blk, 40)); }
    if (keep) {
        hipMemPool_t pool; uint64_t thr = 64ull << 20;
        CK(hipDeviceGetDefaultMemPool(&pool, 0)); CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    }
    int bad = 0;
    std::vector<float> h(n);
    for (int r = 0; r < rounds; ++r) {
        for (int call = 0; call < 2; ++call) {               // two calls per round: the gateway's and the driver's
            for (int i = 0; i < n; ++i) h[i] = 1.f + i + 0.25f * ((r + call) % 7);
            float *big = nullptr, *src = nullptr;
            int *out = nullptr, *tab = nullptr, *blk = nullptr;
            if (mode & 1) { src = p_src; out = p_out; }
            else { if (call == 0) CK(hipMalloc(&big, 300 * 1024)); CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&out, 64)); }
            CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
            if (mode & 2) tab = p_tab; else CK(hipMallocAsync((void **)&tab, n * 4, nullptr));
            write_kernel<<<1, 64>>>(src, tab, n, r * 2 + call);
            if (mode & 2) blk = p_blk; else CK(hipMallocAsync((void **)&blk, 40, nullptr));
            read_kernel<<<1, 64>>>(tab, out, n);
            if (!(mode & 2)) { CK(hipFreeAsync(tab, nullptr)); CK(hipFreeAsync(blk, nullptr)); }
            int got = 0, want = 0;
            CK(hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) want += ((int)(h[i] * 16.f) + r * 2 + call) * (i + 1);
            if (got != want) { ++bad; if (bad <= 3) printf("round %d call %d: read %d, expected %d\n", r, call, got, want); }
            if (!(mode & 1)) { if (big) CK(hipFree(big)); CK(hipFree(src)); CK(hipFree(out)); }
        }
    }
    printf("%d of %d calls read stale data (pool release threshold %s, mode %d)\n", bad, 2 * rounds, keep ? "64 MiB" : "0", mode);
    return bad ? 1 : 0;
}
