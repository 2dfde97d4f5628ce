// Minimal reproducer attempt for the stale read seen in tests/fake_mex: a free -> malloc -> upload -> launch cycle whose
// allocations come back at the old addresses in another order.  hipcc --offload-arch=gfx950 stale_read.hip -o stale_read && ./stale_read
// Prints how many of the rounds read something other than what was just uploaded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)

__global__ void read_kernel(const float *a, const float *b, float *out, int n) {       // out = {a, b}: what the kernel saw
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = a[i]; out[n + i] = b[i]; }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, n = 15;
    int bad = 0, swapped_seen = 0;
    std::vector<float> ha(n), hb(n), ho(2 * n);
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < n; ++i) { ha[i] = 1.f + i + 0.001f * r; hb[i] = 0.5f - 0.01f * i; }
        float *big = nullptr, *a = nullptr, *b = nullptr, *out = nullptr;
        // phase 1: the gateway's call: allocate [big, a, b, out], upload, launch, read back, free all
        CK(hipMalloc(&big, 300 * 1024)); CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&out, 2 * n * 4));
        CK(hipMemset(big, 0, 300 * 1024));
        CK(hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), n * 4, hipMemcpyHostToDevice));
        read_kernel<<<1, 64>>>(a, b, out, n);
        CK(hipMemcpy(ho.data(), out, 2 * n * 4, hipMemcpyDeviceToHost));
        CK(hipFree(big)); CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(out));
        // phase 2: the driver's call: allocate in ANOTHER order (the old addresses come back swapped), upload the same arrays, launch at once
        float *a2 = nullptr, *b2 = nullptr, *out2 = nullptr;
        CK(hipMalloc(&b2, n * 4)); CK(hipMalloc(&a2, n * 4)); CK(hipMalloc(&out2, 2 * n * 4));
        CK(hipMemcpy(a2, ha.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, hb.data(), n * 4, hipMemcpyHostToDevice));
        read_kernel<<<1, 64>>>(a2, b2, out2, n);
        CK(hipMemcpy(ho.data(), out2, 2 * n * 4, hipMemcpyDeviceToHost));
        bool ok = true, sw = true;
        for (int i = 0; i < n; ++i) {
            if (ho[i] != ha[i] || ho[n + i] != hb[i]) ok = false;
            if (ho[i] != hb[i] || ho[n + i] != ha[i]) sw = false;
        }
        if (!ok) { ++bad; if (sw) ++swapped_seen; if (bad <= 3) printf("round %d: kernel read a[0] = %.6g b[0] = %.6g, uploaded %.6g %.6g (a2 %s old b, b2 %s old a)\n", r, ho[0], ho[n], ha[0], hb[0], (void *)a2 == (void *)b ? "==" : "!=", (void *)b2 == (void *)a ? "==" : "!="); }
        CK(hipFree(a2)); CK(hipFree(b2)); CK(hipFree(out2));
    }
    printf("%d of %d rounds read stale data (%d of them exactly the arrays swapped)\n", bad, rounds, swapped_seen);
    return bad ? 1 : 0;
}
