// qdas_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Written for CDNA4 (wave64) only: no CUDA shims, no dual paths.  Semantics of the
// interpolators follow the reference's definitions (reference src/interpd.cu:68-150)
// with the edge rule of SURVEY.md section 8 a5: a sample is in support iff all taps lie in
// [0, T) and tau >= 0; everything else is exactly 0.
#pragma once
#include "tile_rtc.h"

namespace qdas {

// ---------------------------------------------------------------- complex helpers
template <typename R> struct cplx { R x, y; };
template <typename R> __device__ __forceinline__ cplx<R> cmul(cplx<R> a, cplx<R> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// storage <-> compute conversion. Storage types: double2 / float2 / __half2 (as uint32).
struct st_f64 { using store = double2; using real = double; using apod_real_t = double; };
struct st_f32 { using store = float2;  using real = float;  using apod_real_t = float; };
struct st_f16 { using store = uint32_t; using real = float; using apod_real_t = uint16_t; };

__device__ __forceinline__ cplx<double> ld(const double2 *p, size_t i) { double2 v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ cplx<float>  ld(const float2 *p, size_t i)  { float2 v = p[i];  return {v.x, v.y}; }
__device__ __forceinline__ cplx<float>  ld(const uint32_t *p, size_t i) {
    const uint32_t v = p[i];
    return {__half2float(__ushort_as_half((unsigned short)(v & 0xffffu))),
            __half2float(__ushort_as_half((unsigned short)(v >> 16)))};
}
__device__ __forceinline__ double ldr(const double *p, size_t i) { return p[i]; }
__device__ __forceinline__ float  ldr(const float *p, size_t i)  { return p[i]; }
__device__ __forceinline__ float  ldr(const uint16_t *p, size_t i) { return __half2float(__ushort_as_half(p[i])); }

__device__ __forceinline__ void st(double2 *p, size_t i, cplx<double> v) { p[i] = make_double2(v.x, v.y); }
__device__ __forceinline__ void st(float2 *p, size_t i, cplx<float> v)   { p[i] = make_float2(v.x, v.y); }
__device__ __forceinline__ void st(uint32_t *p, size_t i, cplx<float> v) {
    p[i] = (uint32_t)__half_as_ushort(__float2half_rn(v.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(v.y)) << 16);
}

// ---------------------------------------------------------------- generated receive apodization (qdas.h QDAS_RXAPOD_*)
// Evaluated in fp64 from the fp32/fp64 inputs: the masks are step functions of the geometry, and a float64 host evaluation
// (reference src/UltrasoundSystem.m:5165-5430 runs in double) must give the same side of the step.
// r = pixel - element, n = element normal, (pix_x, pix_z) the pixel and el_x the element's x coordinate.
__device__ __forceinline__ double rx_apod_weight(int kind, double p0, double p1, double rx, double ry, double rz,
                                                 double nx, double ny, double nz, double pix_x, double pix_z, double el_x) {
    if (kind == 1 || kind == 2) {
        const double c = (rx * nx + ry * ny + rz * nz) / sqrt(rx * rx + ry * ry + rz * rz);   // NaN at the element itself
        if (kind == 1) return (c >= p0) ? 1.0 : 0.0;                                            // NaN -> 0, like MATLAB's >=
        const double cc = (c == c) ? fmin(1.0, fmax(-1.0, c)) : 1.0;    // MATLAB's max(-1, min(1, NaN)) == 1
        return cos(fmin(1.5707963267948966, p0 * acos(cc)));
    }
    if (kind == 3) {
        const double d2 = fabs(2.0 * (el_x - pix_x));
        return (pix_z > p0 * d2 && d2 < p1) ? 1.0 : 0.0;
    }
    if (kind == 4) {
        const double d2 = fabs(2.0 * (rx * nz - rz * nx)), z = fabs(rx * nx + rz * nz);
        return (z > p0 * d2 && d2 < p1) ? 1.0 : 0.0;
    }
    return 1.0;
}

// ---------------------------------------------------------------- interpolation weights
__device__ __forceinline__ float  qfloor(float v)  { return floorf(v); }
__device__ __forceinline__ double qfloor(double v) { return floor(v); }
__device__ __forceinline__ float  qsinpi(float v)  { return sinpif(v); }
__device__ __forceinline__ double qsinpi(double v) { return sinpi(v); }

// L(v) = 2 sin(pi v) sin(pi v/2) / (pi^2 v^2), L(0) = 1   (reference src/interpd.cu:116-127, a = 2)
template <typename R> __device__ __forceinline__ R lanczos2(R v) {
    const R pi2 = (R)9.86960440108935861883;
    return (v == (R)0) ? (R)1 : (R)2 * qsinpi(v) * qsinpi(v * (R)0.5) / (pi2 * v * v);
}

// taps/offset per interp code (bits 0-2 of QUPS_BF_FLAG)
__host__ __device__ constexpr int interp_taps(int interp) { return interp == 0 ? 1 : ((interp == 1 || interp == 4) ? 2 : 4); }

// weights for fractional offset u in [0,1)
template <int INTERP, typename R> __device__ __forceinline__ void interp_weights(R u, R w[4]) {
    if constexpr (INTERP == 1 || INTERP == 4) {          // lerp (src/interpd.cu:84)
        w[0] = (R)1 - u; w[1] = u;
    } else if constexpr (INTERP == 2) {                  // Catmull-Rom (comment src/interpd.cu:108-111)
        w[0] = (R)0.5 * (u * ((R)-1 + u * ((R)2 - u)));
        w[1] = (R)0.5 * ((R)2 + u * u * ((R)3 * u - (R)5));
        w[2] = (R)0.5 * (u * ((R)1 + u * ((R)4 - (R)3 * u)));
        w[3] = (R)0.5 * (u * u * (u - (R)1));
    } else if constexpr (INTERP == 5) {                  // device Horner lines as executed (src/interpd.cu:103-106)
        w[0] = (R)0.5 * (u * ((R)-1 + u * ((R)2 * u - (R)1)));
        w[1] = (R)0.5 * ((R)2 + u * (u * ((R)-5 * u + (R)3)));
        w[2] = (R)0.5 * (u * ((R)1 + u * ((R)4 * u - (R)3)));
        w[3] = (R)0.5 * (u * (u * ((R)1 - u)));
    } else if constexpr (INTERP == 3) {                  // lanczos, window 2 (src/interpd.cu:145-148)
        w[0] = lanczos2(u + (R)1); w[1] = lanczos2(u); w[2] = lanczos2(u - (R)1); w[3] = lanczos2(u - (R)2);
    }
}

// One sample of a trace in global memory at fractional index s (0-based).
template <int INTERP, typename R, typename ST>
__device__ __forceinline__ cplx<R> sample_global(const ST *__restrict__ tr, long T, R s) {
    cplx<R> out = {(R)0, (R)0};
    if (!(s >= (R)0)) return out;                        // tau >= 0; rejects NaN
    if constexpr (INTERP == 0) {                         // nearest (src/interpd.cu:70-72)
        const R r = qfloor(s + (R)0.5);
        if (r < (R)T) out = ld(tr, (size_t)r);
        return out;
    } else {
        const R fl = qfloor(s);
        constexpr int K = interp_taps(INTERP);
        constexpr int OFF = (K == 2) ? 0 : -1;
        if (!(fl + (R)(K - 1 + OFF) < (R)T) || fl + (R)OFF < (R)0) return out;   // all taps in [0,T); rejects +inf
        const size_t first = (size_t)((long)fl + OFF);
        R w[4];
        interp_weights<INTERP>(s - fl, w);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const cplx<R> v = ld(tr, first + k);
            out.x += w[k] * v.x; out.y += w[k] * v.y;
        }
        return out;
    }
}

}  // namespace qdas
