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
#include <stdint.h>
#include "qdas_kernels.h"

namespace qdas {

// grid (m, n): workgroups with n > m leave at once; 256 lanes x 16 bytes (two complex samples) per step when the traces are 16-byte aligned
template <bool V4, bool WT>
__global__ void __launch_bounds__(256) fold_kernel(const float2 *__restrict__ x, float2 *__restrict__ xs, const float2 *__restrict__ wtab,
                                                   uint64_t T, uint64_t N, uint64_t strN, uint64_t strM) {
    const uint64_t m = blockIdx.x, n = blockIdx.y;
    if (n > m) return;
    const float2 *a = x + n * strN + m * strM;          // (rx n, tx m)
    const float2 *b = x + m * strN + n * strM;          // (rx m, tx n): the reciprocal trace
    float2 *o = xs + n * strN + m * strM;
    float2 wa = make_float2(1.f, 0.f), wb = make_float2(1.f, 0.f);
    if constexpr (WT) { wa = wtab[n + N * m]; wb = wtab[m + N * n]; }
    const bool diag = n == m;
    const bool za = WT && wa.x == 0.f && wa.y == 0.f, zb = diag || (WT && wb.x == 0.f && wb.y == 0.f);
    auto term = [](float2 v, float2 w) { return WT ? make_float2(w.x * v.x - w.y * v.y, w.x * v.y + w.y * v.x) : v; };
    if constexpr (V4) {
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
            if (!za) r = term(a[k], wa);
            if (!zb) { const float2 p = term(b[k], wb); r.x += p.x; r.y += p.y; }
            o[k] = r;
        }
    }
}

hipError_t launch_fold(const void *x, void *xs, const void *wtab, uint64_t T, uint64_t N, uint64_t strN, uint64_t strM, hipStream_t s) {
    if (!T || !N) return hipSuccess;
    if (N > 65535) return hipErrorInvalidValue;
    const bool v4 = T % 2 == 0 && strN % 2 == 0 && strM % 2 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)xs & 15) == 0;
    const dim3 g((unsigned)N, (unsigned)N), b(256);
    const float2 *xi = (const float2 *)x, *wt = (const float2 *)wtab;
    float2 *xo = (float2 *)xs;
    if (v4) { if (wt) fold_kernel<true, true><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); else fold_kernel<true, false><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); }
    else    { if (wt) fold_kernel<false, true><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); else fold_kernel<false, false><<<g, b, 0, s>>>(xi, xo, wt, T, N, strN, strM); }
    return hipGetLastError();
}

}  // namespace qdas
