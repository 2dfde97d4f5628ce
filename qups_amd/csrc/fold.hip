// fold.hip -- the reciprocity fold of a full-synthetic-aperture frame (host + device): one streaming pass over HBM in front of the fused kernel.
//
// A reciprocal acquisition (transmit elements == receive elements, one t0: detected by the plan, qdas_api.hip) has tau(n,m) == tau(m,n) for every
// pixel, and the interpolators are linear in the data: w x[:,n,m] sampled at tau plus w' x[:,m,n] sampled at the same tau is the sample of
// (w x[:,n,m] + w' x[:,m,n]).  The reference forms both terms per pixel (src/bf.cu:96-141: I * N * M gathers); here the two traces of an
// unordered pair are added ONCE per frame,
//     xs[:, n, m] = w[n,m] x[:, n, m] + w[m,n] x[:, m, n]   (n < m),      xs[:, n, n] = w[n,n] x[:, n, n],
// and the fused kernel (das_tile_impl.h TileCfg::FOLD) walks the upper triangle: half the staging, half the gathers, half the multiply-accumulates.
// Pixel-independent apodization (the plan's folded N x M table) is applied here, so the folded kernels carry no weight table; a zero weight never lets
// its trace in (src/bf.cu:122,126: dead channels may hold non-finite samples).  The sum of two fp32 samples is rounded once (2^-24 relative): far
// below the path's tolerance.  Bound: HBM -- reads the frame once, writes half of it.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "qdas_kernels.h"

namespace qdas {

// fp16 frames (4-byte samples {re, im}) fold into a complex64 copy as well: the folded fp32 kernels then serve fp16 data -- a folded sample rounded
// back to fp16 would cost 2^-11 relative, so the sum is kept in fp32 (twice the bytes of the fp16 frame, as many as an fp32 upper triangle)
static __device__ __forceinline__ float2 fold_ld(const float2 *p, uint64_t k) { return p[k]; }
static __device__ __forceinline__ float2 fold_ld(const uint32_t *p, uint64_t k) {
    const uint32_t v = p[k];
    return make_float2(__half2float(__ushort_as_half((unsigned short)(v & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(v >> 16))));
}
// complex32 image <- complex64 image (the fp16 plans' output)
__global__ void __launch_bounds__(256) fold_y16_kernel(const float2 *__restrict__ y32, uint32_t *__restrict__ y16, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 v = y32[i];
    y16[i] = (uint32_t)__half_as_ushort(__float2half(v.x)) | ((uint32_t)__half_as_ushort(__float2half(v.y)) << 16);
}
hipError_t launch_y32_to_y16(const void *y32, void *y16, uint64_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    fold_y16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const float2 *)y32, (uint32_t *)y16, n);
    return hipGetLastError();
}

// grid (m, n): workgroups with n > m leave at once; 256 lanes x 16 bytes (two complex samples) per step when the traces are 16-byte aligned
template <bool V4, bool WT, typename IN>
__global__ void __launch_bounds__(256) fold_kernel(const IN *__restrict__ x, float2 *__restrict__ xs, const float2 *__restrict__ wtab,
                                                   uint64_t T, uint64_t N, uint64_t strN, uint64_t strM) {
    const uint64_t m = blockIdx.x, n = blockIdx.y;
    if (n > m) return;
    const IN *a = x + n * strN + m * strM;              // (rx n, tx m)
    const IN *b = x + m * strN + n * strM;              // (rx m, tx n): the reciprocal trace
    float2 *o = xs + n * strN + m * strM;
    float2 wa = make_float2(1.f, 0.f), wb = make_float2(1.f, 0.f);
    if constexpr (WT) { wa = wtab[n + N * m]; wb = wtab[m + N * n]; }
    const bool diag = n == m;
    const bool za = WT && wa.x == 0.f && wa.y == 0.f, zb = diag || (WT && wb.x == 0.f && wb.y == 0.f);
    auto term = [](float2 v, float2 w) { return WT ? make_float2(w.x * v.x - w.y * v.y, w.x * v.y + w.y * v.x) : v; };
    if constexpr (V4 && sizeof(IN) == 8) {
        const float4 *a4 = (const float4 *)a, *b4 = (const float4 *)b;
        float4 *o4 = (float4 *)o;
        for (uint64_t k = threadIdx.x; k < T / 2; k += 256) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!za) { const float4 v = a4[k]; const float2 p = term(make_float2(v.x, v.y), wa), q = term(make_float2(v.z, v.w), wa); r = make_float4(p.x, p.y, q.x, q.y); }
            if (!zb) { const float4 v = b4[k]; const float2 p = term(make_float2(v.x, v.y), wb), q = term(make_float2(v.z, v.w), wb); r.x += p.x; r.y += p.y; r.z += q.x; r.w += q.y; }
            o4[k] = r;
        }
    } else {
        for (uint64_t k = threadIdx.x; k < T; k += 256) {
            float2 r = make_float2(0.f, 0.f);
            if (!za) r = term(fold_ld(a, k), wa);
            if (!zb) { const float2 p = term(fold_ld(b, k), wb); r.x += p.x; r.y += p.y; }
            o[k] = r;
        }
    }
}

hipError_t launch_fold(const void *x, void *xs, const void *wtab, uint64_t T, uint64_t N, uint64_t strN, uint64_t strM, hipStream_t s, int in_f16) {
    if (!T || !N) return hipSuccess;
    if (N > 65535) return hipErrorInvalidValue;
    const dim3 g((unsigned)N, (unsigned)N), b(256);
    const float2 *wt = (const float2 *)wtab;
    float2 *xo = (float2 *)xs;
    if (in_f16) {
        const uint32_t *xi = (const uint32_t *)x;
        if (wt) fold_kernel<false, true, uint32_t><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); else fold_kernel<false, false, uint32_t><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM);
        return hipGetLastError();
    }
    const bool v4 = T % 2 == 0 && strN % 2 == 0 && strM % 2 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)xs & 15) == 0;
    const float2 *xi = (const float2 *)x;
    if (v4) { if (wt) fold_kernel<true, true, float2><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); else fold_kernel<true, false, float2><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); }
    else    { if (wt) fold_kernel<false, true, float2><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); else fold_kernel<false, false, float2><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); }
    return hipGetLastError();
}

}  // namespace qdas
