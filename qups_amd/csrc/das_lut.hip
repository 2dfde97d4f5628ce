// das_lut.hip -- split-delay ("look-up table") delay-and-sum: the bfDAS / bfDASLUT flavour.
//
// Computes what reference bfDASLUT -> ChannelData.sample2sep -> wsinterpd2 computes
// (reference src/UltrasoundSystem.m:4641-4660, src/ChannelData.m:1431-1445,
//  kern/wsinterpd2.m:236, kernel body src/interpd.cu:344-396):
//
//     y[i,(n),(m)] = sum  w[i,n,m] * exp(j*omega*s) * sample(x[:,n,m], s),   s = tau_rx[i,n] + tau_tx[i,m]
//
// with host-supplied delay tables (already in samples: (tau - t0)*fs).  Differences by design:
// every output element is OWNED by one lane and accumulated in a register in a fixed (m, n)
// order -- deterministic -- where the reference adds with float atomics (src/interpd.cu:339,393);
// the summed dimension is the inner loop; non-finite delays are skipped (src/interpd.cu:390).
#include "qdas_device.h"
#include "qdas_kernels.h"

namespace qdas {

template <int INTERP, typename TY>
__global__ void __launch_bounds__(256) das_lut_kernel(const LutParams P) {
    using R  = typename TY::real;
    using ST = typename TY::store;
    using AR = typename TY::apod_real_t;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.I) return;
    const size_t I = P.I, N = P.N, M = P.M, T = P.T;
    const R *__restrict__ trx = (const R *)P.tau_rx, *__restrict__ ttx = (const R *)P.tau_tx;
    const ST *__restrict__ x = (const ST *)P.x;
    ST *__restrict__ y = (ST *)P.y;
    const bool keep_rx = P.flag & 8, keep_tx = P.flag & 16, tpose = P.flag & 32;
    const R omega = (R)P.omega;

    auto pair = [&](size_t n, size_t m, R t1, R t2) -> cplx<R> {
        const R s = t1 + t2;                                           // src/interpd.cu:389
        if (!(fabs((double)s) <= 1.0e300) ) return {(R)0, (R)0};          // inf / nan: skip (src/interpd.cu:390)
        const size_t nm = tpose ? (m + n * M) : (n + m * N);
        cplx<R> v = sample_global<INTERP, R, ST>(x + nm * T, (long)T, s);
        if (omega != (R)0) {                                           // src/interpd.cu:391
            R sn, cs;
            if constexpr (sizeof(R) == 8) sincos((double)(omega * s), (double *)&sn, (double *)&cs);
            else sincosf((float)(omega * s), (float *)&sn, (float *)&cs);
            v = cmul(v, cplx<R>{cs, sn});
        }
        if (P.w) {
            const size_t k = i * P.wst[0] + n * P.wst[1] + m * P.wst[2];
            if (P.w_real) { const R w = (R)ldr((const AR *)P.w, k); v.x *= w; v.y *= w; }
            else v = cmul(v, ld((const ST *)P.w, k));
        }
        return v;
    };

    if (keep_rx && keep_tx) {
        for (size_t m = 0; m < M; ++m) { const R t2 = ttx[i + I * m];
            for (size_t n = 0; n < N; ++n) st(y, i + I * (n + N * m), pair(n, m, trx[i + I * n], t2)); }
    } else if (keep_rx) {
        for (size_t n = 0; n < N; ++n) { const R t1 = trx[i + I * n]; cplx<R> acc = {(R)0, (R)0};
            for (size_t m = 0; m < M; ++m) { const cplx<R> v = pair(n, m, t1, ttx[i + I * m]); acc.x += v.x; acc.y += v.y; }
            st(y, i + I * n, acc); }
    } else if (keep_tx) {
        for (size_t m = 0; m < M; ++m) { const R t2 = ttx[i + I * m]; cplx<R> acc = {(R)0, (R)0};
            for (size_t n = 0; n < N; ++n) { const cplx<R> v = pair(n, m, trx[i + I * n], t2); acc.x += v.x; acc.y += v.y; }
            st(y, i + I * m, acc); }
    } else {
        cplx<R> acc = {(R)0, (R)0};
        for (size_t m = 0; m < M; ++m) { const R t2 = ttx[i + I * m];
            for (size_t n = 0; n < N; ++n) { const cplx<R> v = pair(n, m, trx[i + I * n], t2); acc.x += v.x; acc.y += v.y; } }
        st(y, i, acc);
    }
}

template <typename TY> static hipError_t launch_lut_t(const LutParams &P, hipStream_t s) {
    const dim3 g((unsigned)((P.I + 255) / 256)), b(256);
    switch (P.flag & 7) {
        case 0: das_lut_kernel<0, TY><<<g, b, 0, s>>>(P); break;
        case 1: case 4: das_lut_kernel<1, TY><<<g, b, 0, s>>>(P); break;
        case 2: das_lut_kernel<2, TY><<<g, b, 0, s>>>(P); break;
        case 3: das_lut_kernel<3, TY><<<g, b, 0, s>>>(P); break;
        case 5: das_lut_kernel<5, TY><<<g, b, 0, s>>>(P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lut(const LutParams &P, int dtype, hipStream_t s) {
    if (P.I == 0) return hipSuccess;
    switch (dtype) {
        case 0: return launch_lut_t<st_f64>(P, s);
        case 1: return launch_lut_t<st_f32>(P, s);
        case 2: return launch_lut_t<st_f16>(P, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
