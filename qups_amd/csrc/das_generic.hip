// das_generic.hip -- the "any shape" DAS kernel: one pixel per lane.
//
// Serves every case the tiled kernel (das_tile_impl.h) does not: fp64, the keep_rx/keep_tx
// modes, N-D sound speed, arbitrary broadcast apodization stacks, and pixel tiles whose
// delay window does not fit in LDS.  It computes exactly what reference src/bf.cu:49-142
// (`DAS_temp`) defines, but is organised differently:
//   * the summed dimension is the INNER loop and lives in a register accumulator, so the
//     SYN/MUL modes never read-modify-write global memory (reference src/bf.cu:131-133 does);
//   * the transmit distance is hoisted out of the receive loop (reference recomputes it per
//     pair, src/bf.cu:104-108);
//   * interpolation is a compile-time variant, not a runtime flag switch (src/interpd.cu:152-167).
// Gathers go through L1/L2 (consecutive lanes = consecutive depth pixels -> neighbouring
// fast-time samples of the same trace).
#include "qdas_device.h"
#include "qdas_kernels.h"

namespace qdas {

template <typename R> __device__ __forceinline__ R qcospi(R v);
template <> __device__ __forceinline__ float  qcospi(float v)  { return cospif(v); }
template <> __device__ __forceinline__ double qcospi(double v) { return cospi(v); }
template <typename R> __device__ __forceinline__ R qsqrt(R v);
template <> __device__ __forceinline__ float  qsqrt(float v)  { return sqrtf(v); }
template <> __device__ __forceinline__ double qsqrt(double v) { return sqrt(v); }
template <typename R> __device__ __forceinline__ R qcopysign(R a, R b);
template <> __device__ __forceinline__ float  qcopysign(float a, float b)   { return copysignf(a, b); }
template <> __device__ __forceinline__ double qcopysign(double a, double b) { return copysign(a, b); }

// out of line: four inlined copies of the fp64 acos / cos code (one per accumulation mode) cost ~200 spilled scalars
static __device__ __noinline__ double rx_apod_weight_call(int kind, double p0, double p1, double rx, double ry, double rz, double nx, double ny,
                                                         double nz, double pix_x, double pix_z, double el_x) {
    return rx_apod_weight(kind, p0, p1, rx, ry, rz, nx, ny, nz, pix_x, pix_z, el_x);
}

template <int INTERP, typename TY>
__global__ void __launch_bounds__(256)
das_generic_kernel(const GenericParams P) {
    using R  = typename TY::real;
    using ST = typename TY::store;
    using AR = typename TY::apod_real_t;
    const R  *__restrict__ Pi = (const R *)P.Pi, *__restrict__ Pr = (const R *)P.Pr;
    const R  *__restrict__ Pv = (const R *)P.Pv, *__restrict__ Nv = (const R *)P.Nv;
    const R  *__restrict__ cinv = (const R *)P.cinv;
    const ST *__restrict__ x = (const ST *)P.x;
    ST *__restrict__ y = (ST *)P.y;

    size_t il;
    if (P.tile_list) {   // fallback launch: blockIdx.x enumerates 64 x TX pixel tiles that overflowed LDS
        if (blockIdx.x / P.blocks_per_tile >= P.tile_list[0]) return;      // uniform: nothing (more) to redo
        const uint32_t t = P.tile_list[1 + blockIdx.x / P.blocks_per_tile];
        const uint32_t within = (blockIdx.x % P.blocks_per_tile) * blockDim.x + threadIdx.x; // 0..64*TX
        const uint64_t tz = t % P.tiles_z, tx = t / P.tiles_z;
        const uint32_t wz = within & ((1u << P.tile_zl) - 1u), wc = within >> P.tile_zl;
        const uint64_t i1 = (tz << P.tile_zl) + wz, col = tx * P.tile_cols + wc;
        if (i1 >= P.I1 || wc >= P.tile_cols || col >= P.I2 * P.I3) return;
        const uint64_t i = i1 + P.I1 * col;
        if (i < P.i_begin || i >= P.i_begin + P.i_count) return;
        il = i - P.i_begin;
    } else {
        il = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (il >= P.i_count) return;
    }
    const size_t i = P.i_begin + il;
    const size_t i1 = i % P.I1, i2 = (i / P.I1) % P.I2, i3 = i / (P.I1 * P.I2);
    const size_t N = P.N, M = P.M, T = P.T;
    const int S = P.S;
    const bool keep_rx = P.flag & 8, keep_tx = P.flag & 16, tpose = P.flag & 32;
    const R fs = (R)P.fs, fc = (R)P.fmod;

    const R px = Pi[3 * i], py = Pi[3 * i + 1], pz = Pi[3 * i + 2];
    const size_t cbase = i1 * P.cst[0] + i2 * P.cst[1] + i3 * P.cst[2];
    size_t abase[QDAS_MAX_APOD];
#pragma unroll
    for (int s = 0; s < QDAS_MAX_APOD; ++s)
        abase[s] = (s < S) ? P.ast[6 * s + 5] + i1 * P.ast[6 * s] + i2 * P.ast[6 * s + 1] + i3 * P.ast[6 * s + 2] : 0;

    auto tx_dist = [&](size_t m) -> R {               // reference src/bf.cu:104-108
        const R rx = px - Pv[4 * m], ry = py - Pv[4 * m + 1], rz = pz - Pv[4 * m + 2];
        const R dot = rx * Nv[3 * m] + ry * Nv[3 * m + 1] + rz * Nv[3 * m + 2];
        if (!P.VS) return dot;
        const R len = qsqrt(rx * rx + ry * ry + rz * rz);
        return P.DV ? len : qcopysign(len, dot);
    };
    auto rx_dist = [&](size_t n) -> R {               // reference src/bf.cu:110
        const R rx = px - Pr[3 * n], ry = py - Pr[3 * n + 1], rz = pz - Pr[3 * n + 2];
        return qsqrt(rx * rx + ry * ry + rz * rz);
    };
    auto pair = [&](size_t n, size_t m, R dv, R dr) -> cplx<R> {
        const size_t nm = tpose ? (m + n * M) : (n + m * N);                 // src/bf.cu:100
        const R ci = cinv[cbase + n * P.cst[3] + m * P.cst[4]];
        const R tau = ci * (dv + dr) - Pv[4 * m + 3];                        // src/bf.cu:113-114
        cplx<R> a = {(R)1, (R)0};
#pragma unroll
        for (int s = 0; s < QDAS_MAX_APOD; ++s) {
            if (s < S && (a.x != (R)0 || a.y != (R)0)) {                     // src/bf.cu:121-123
                const size_t k = abase[s] + n * P.ast[6 * s + 3] + m * P.ast[6 * s + 4];
                if (P.apod_real) { const R w = (R)ldr((const AR *)P.apod, k); a.x *= w; a.y *= w; }
                else a = cmul(a, ld((const ST *)P.apod, k));
            }
        }
        if (P.gen_kind && (a.x != (R)0 || a.y != (R)0)) {                      // generated receive apodization (qdas.h QDAS_RXAPOD_*)
            const R *rn = (const R *)P.rxn;
            const double ex = (double)Pr[3 * n], ey = (double)Pr[3 * n + 1], ez = (double)Pr[3 * n + 2];
            const double nx = rn ? (double)rn[3 * n] : 0.0, ny = rn ? (double)rn[3 * n + 1] : 0.0, nz = rn ? (double)rn[3 * n + 2] : 1.0;
            const R w = (R)rx_apod_weight_call(P.gen_kind, P.gen_p0, P.gen_p1, (double)px - ex, (double)py - ey, (double)pz - ez,
                                          nx, ny, nz, (double)px, (double)pz, ex);
            a.x *= w; a.y *= w;
        }
        if (a.x == (R)0 && a.y == (R)0) return {(R)0, (R)0};                  // zero weight: skip the gather
        cplx<R> v = sample_global<INTERP, R, ST>(x + nm * T, (long)T, tau * fs);
        if (fc != (R)0) {                                                    // src/bf.cu:117
            const cplx<R> w = {qcospi<R>((R)2 * fc * tau), qsinpi((R)2 * fc * tau)};
            v = cmul(v, w);
        }
        return cmul(a, v);
    };

    if (keep_rx && keep_tx) {                         // 'BF': store every (n, m) plane (src/bf.cu:134-135)
        for (size_t m = 0; m < M; ++m) {
            const R dv = tx_dist(m);
            for (size_t n = 0; n < N; ++n) {
                const size_t nm = tpose ? (m + n * M) : (n + m * N);
                st(y, il + nm * P.y_ld, pair(n, m, dv, rx_dist(n)));
            }
        }
    } else if (keep_rx) {                             // 'SYN': sum over transmits, one plane per receiver
        for (size_t n = 0; n < N; ++n) {
            const R dr = rx_dist(n);
            cplx<R> acc = {(R)0, (R)0};
            for (size_t m = 0; m < M; ++m) { const cplx<R> v = pair(n, m, tx_dist(m), dr); acc.x += v.x; acc.y += v.y; }
            st(y, il + n * P.y_ld, acc);
        }
    } else if (keep_tx) {                             // 'MUL': sum over receivers, one plane per transmit
        for (size_t m = 0; m < M; ++m) {
            const R dv = tx_dist(m);
            cplx<R> acc = {(R)0, (R)0};
            for (size_t n = 0; n < N; ++n) { const cplx<R> v = pair(n, m, dv, rx_dist(n)); acc.x += v.x; acc.y += v.y; }
            st(y, il + m * P.y_ld, acc);
        }
    } else {                                          // 'DAS': sum over both (src/bf.cu:137,140)
        cplx<R> acc = {(R)0, (R)0};
        for (size_t m = 0; m < M; ++m) {
            const R dv = tx_dist(m);
            for (size_t n = 0; n < N; ++n) { const cplx<R> v = pair(n, m, dv, rx_dist(n)); acc.x += v.x; acc.y += v.y; }
        }
        st(y, il, acc);
    }
}

// 'delays' (reference src/bf.cu:209-298): tau[i + n*I + m*I*N] = cinv * (dv + dr)
template <typename R>
__global__ void __launch_bounds__(256) delays_kernel(const GenericParams P, R *__restrict__ tau, R cinv) {
    const size_t il = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (il >= P.i_count) return;
    const size_t i = P.i_begin + il;
    const R *Pi = (const R *)P.Pi, *Pr = (const R *)P.Pr, *Pv = (const R *)P.Pv, *Nv = (const R *)P.Nv;
    const R px = Pi[3 * i], py = Pi[3 * i + 1], pz = Pi[3 * i + 2];
    for (size_t m = 0; m < P.M; ++m) {
        const R rx = px - Pv[4 * m], ry = py - Pv[4 * m + 1], rz = pz - Pv[4 * m + 2];
        const R dot = rx * Nv[3 * m] + ry * Nv[3 * m + 1] + rz * Nv[3 * m + 2];
        const R len = qsqrt(rx * rx + ry * ry + rz * rz);
        const R dv = P.VS ? (P.DV ? len : qcopysign(len, dot)) : dot;
        for (size_t n = 0; n < P.N; ++n) {
            const R ax = px - Pr[3 * n], ay = py - Pr[3 * n + 1], az = pz - Pr[3 * n + 2];
            tau[il + n * P.y_ld + m * P.y_ld * P.N] = cinv * (dv + qsqrt(ax * ax + ay * ay + az * az));
        }
    }
}

template <typename TY>
static hipError_t launch_generic_t(const GenericParams &P, unsigned grid, hipStream_t s) {
    const dim3 g(grid), b(256);
    switch (P.flag & 7) {
        case 0: das_generic_kernel<0, TY><<<g, b, 0, s>>>(P); break;
        case 1: case 4: das_generic_kernel<1, TY><<<g, b, 0, s>>>(P); break;
        case 2: das_generic_kernel<2, TY><<<g, b, 0, s>>>(P); break;
        case 3: das_generic_kernel<3, TY><<<g, b, 0, s>>>(P); break;
        case 5: das_generic_kernel<5, TY><<<g, b, 0, s>>>(P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_generic(const GenericParams &P, int dtype, unsigned grid, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    switch (dtype) {
        case 0: return launch_generic_t<st_f64>(P, grid, s);
        case 1: return launch_generic_t<st_f32>(P, grid, s);
        case 2: return launch_generic_t<st_f16>(P, grid, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_delays(const GenericParams &P, int dtype, void *tau, double cinv, hipStream_t s) {
    const unsigned grid = (unsigned)((P.i_count + 255) / 256);
    if (grid == 0) return hipSuccess;
    if (dtype == 0) delays_kernel<double><<<dim3(grid), dim3(256), 0, s>>>(P, (double *)tau, cinv);
    else            delays_kernel<float><<<dim3(grid), dim3(256), 0, s>>>(P, (float *)tau, (float)cinv);
    return hipGetLastError();
}

}  // namespace qdas
