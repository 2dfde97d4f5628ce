/* das_ref_tuned.c -- the CPU baseline a host-tuned build of the reference's CPU algorithm reaches (TEST INFRASTRUCTURE ONLY).
 *
 * Same algorithm and loop nest as das_ref.c / das_ref_body.inc -- the reference's CPU branch kern/das_spec.m:451-482: distances
 * for all pixels first (:427-440), then "parfor m / for n" with the sample VECTORISED OVER PIXELS (:462-481, interp1 over all I) --
 * restricted to what bench.py's `cpu_baseline` leg times: 'DAS' mode, fp32, scalar sound speed, no apodization, fmod = 0.
 * What is tuned (and nothing else): pixels are the innermost, unit-stride loop with every (n, m)-dependent quantity hoisted,
 * weights are evaluated in float (Catmull-Rom exactly; Lanczos by the even/odd minimax polynomials in s = u - 1/2, |err| <= 3e-6,
 * coefficients from tools/gen_lanczos_poly.py 3 3 3 3) instead of eight double-precision sin() per pair, out-of-record samples are
 * masked instead of branched on, and the file is compiled for THIS host (-O3 -march=native -fopenmp, no -ffast-math) by
 * oracle/das_ref.py at first use -- never shipped pre-built, because the build host is not the GPU box's host.
 * tests/test_oracle_pins.py checks it against the straight port (das_ref.c) to 2e-5 of the image maximum.
 * The product (qups_amd/) never links or calls this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct das_ref_sizes {
    uint64_t T, N, M, I, I1, I2, I3, S;
    int32_t flag;
    int32_t VS, DV;
} das_ref_sizes;

static const float EI[4] = { 5.731605671e-01f, -2.590694898e-01f, -1.665395708e-01f, 1.290811049e-01f };
static const float OI[4] = { -1.392318724e+00f, 1.791504781e+00f, -9.550008342e-01f, 2.644278493e-01f };
static const float EO[4] = { -6.368713894e-02f, 1.749036498e-01f, 3.857659640e-01f, -2.657193793e-01f };
static const float OO[4] = { 1.849471404e-01f, -8.496553157e-01f, 4.664762821e-01f, -1.080671528e-01f };

#define BLK 512   /* pixels per inner block: delays, indices and weights of a block stay in L1 */

/* returns 0 on success, 3 when the call is outside the tuned subset (the caller then uses das_ref_f32) */
int das_ref_tuned_f32(const das_ref_sizes *sz, float *y, const float *Pi, const float *Pr, const float *Pv4, const float *Nv,
                      const float *apod, const float *cinv, const uint64_t *acstride, const float *x, const float *fsfc,
                      int nthreads) {
    const long I = (long)sz->I, N = (long)sz->N, M = (long)sz->M, T = (long)sz->T;
    const int flag = sz->flag, interp = (flag & 7) == 4 ? 1 : (flag & 7);
    const int tpose = (flag >> 5) & 1;
    (void)apod;
    if ((flag & 24) || sz->S || fsfc[1] != 0.0f || interp > 3) return 3;
    for (int k = 0; k < 5; ++k) if (acstride[k]) return 3;
    const float fs = fsfc[0], ci = cinv[0];
    const int K = interp == 0 ? 1 : (interp == 1 ? 2 : 4);
    if (T < K) { memset(y, 0, sizeof(float) * 2 * (size_t)I); return 0; }

    float *dv = (float *)malloc(sizeof(float) * (size_t)I * M);
    float *dr = (float *)malloc(sizeof(float) * (size_t)I * N);
    if (!dv || !dr) { free(dv); free(dr); return 1; }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    /* kern/das_spec.m:427-440; the sound speed and sampling rate are folded in: sample index = ci*fs*(dv + dr) - t0*fs */
    #pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m) {
        const float vx = Pv4[4 * m], vy = Pv4[4 * m + 1], vz = Pv4[4 * m + 2], nx = Nv[3 * m], ny = Nv[3 * m + 1], nz = Nv[3 * m + 2];
        for (long i = 0; i < I; ++i) {
            const float rx = Pi[3 * i] - vx, ry = Pi[3 * i + 1] - vy, rz = Pi[3 * i + 2] - vz;
            const float dot = rx * nx + ry * ny + rz * nz;
            const float len = sqrtf(rx * rx + ry * ry + rz * rz);
            dv[i + m * I] = sz->VS ? (sz->DV ? len : copysignf(len, dot)) : dot;
        }
    }
    #pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
        const float ex = Pr[3 * n], ey = Pr[3 * n + 1], ez = Pr[3 * n + 2];
        for (long i = 0; i < I; ++i) {
            const float rx = Pi[3 * i] - ex, ry = Pi[3 * i + 1] - ey, rz = Pi[3 * i + 2] - ez;
            dr[i + n * I] = sqrtf(rx * rx + ry * ry + rz * rz);
        }
    }
    memset(y, 0, sizeof(float) * 2 * (size_t)I);

    const long nblk = (I + BLK - 1) / BLK;
    #pragma omp parallel for schedule(dynamic, 1)
    for (long b = 0; b < nblk; ++b) {
        const long i0 = b * BLK, cnt = (i0 + BLK <= I) ? BLK : I - i0;
        float accr[BLK], acci[BLK], w0[BLK], w1[BLK], w2[BLK], w3[BLK], ok[BLK];
        int32_t first[BLK];
        for (long j = 0; j < cnt; ++j) { accr[j] = 0.f; acci[j] = 0.f; }
        for (long m = 0; m < M; ++m) {                                   /* kern/das_spec.m:462 */
            const float t0 = Pv4[4 * m + 3];
            const float *dvm = dv + i0 + m * I;
            for (long n = 0; n < N; ++n) {                               /* :466 */
                const long nm = tpose ? (m + n * M) : (n + m * N);
                const float *tr = x + 2 * (size_t)nm * T;
                const float *drn = dr + i0 + n * I;
                /* pass 1 (vectorises): index, in-support mask, weights -- kern/das_spec.m:469, src/interpd.cu:68-150 */
                #pragma omp simd
                for (long j = 0; j < cnt; ++j) {
                    const float tau = ci * (dvm[j] + drn[j]) - t0;
                    const float s = tau * fs;
                    const float base = interp == 0 ? floorf(s + 0.5f) : floorf(s);
                    const float u = s - base;
                    const float f0 = interp >= 2 ? base - 1.0f : base;                /* first tap */
                    const int good = (s >= 0.0f) & (f0 >= 0.0f) & (f0 + (float)(K - 1) < (float)T);
                    const float fc = good ? f0 : 0.0f;
                    first[j] = (int32_t)fc;
                    ok[j] = good ? 1.0f : 0.0f;
                    if (interp == 0) { w0[j] = 1.0f; }
                    else if (interp == 1) { w0[j] = 1.0f - u; w1[j] = u; }
                    else if (interp == 2) {                                            /* Catmull-Rom, src/interpd.cu:108-111 */
                        w0[j] = 0.5f * (u * (-1.0f + u * (2.0f - u)));
                        w1[j] = 0.5f * (2.0f + u * u * (3.0f * u - 5.0f));
                        w2[j] = 0.5f * (u * (1.0f + u * (4.0f - 3.0f * u)));
                        w3[j] = 0.5f * (u * u * (u - 1.0f));
                    } else {                                                           /* Lanczos a = 2, src/interpd.cu:116-149 */
                        const float sh = u - 0.5f, q = sh * sh;
                        const float ei = EI[0] + q * (EI[1] + q * (EI[2] + q * EI[3])), oi = OI[0] + q * (OI[1] + q * (OI[2] + q * OI[3]));
                        const float eo = EO[0] + q * (EO[1] + q * (EO[2] + q * EO[3])), oo = OO[0] + q * (OO[1] + q * (OO[2] + q * OO[3]));
                        w1[j] = ei + sh * oi; w2[j] = ei - sh * oi; w0[j] = eo + sh * oo; w3[j] = eo - sh * oo;
                    }
                }
                /* pass 2: gather + accumulate (:476-478) */
                if (K == 4) {
                    #pragma omp simd
                    for (long j = 0; j < cnt; ++j) {
                        const float *p = tr + 2 * (size_t)first[j];
                        const float re = w0[j] * p[0] + w1[j] * p[2] + w2[j] * p[4] + w3[j] * p[6];
                        const float im = w0[j] * p[1] + w1[j] * p[3] + w2[j] * p[5] + w3[j] * p[7];
                        accr[j] += ok[j] * re; acci[j] += ok[j] * im;
                    }
                } else if (K == 2) {
                    #pragma omp simd
                    for (long j = 0; j < cnt; ++j) {
                        const float *p = tr + 2 * (size_t)first[j];
                        accr[j] += ok[j] * (w0[j] * p[0] + w1[j] * p[2]); acci[j] += ok[j] * (w0[j] * p[1] + w1[j] * p[3]);
                    }
                } else {
                    #pragma omp simd
                    for (long j = 0; j < cnt; ++j) {
                        const float *p = tr + 2 * (size_t)first[j];
                        accr[j] += ok[j] * p[0]; acci[j] += ok[j] * p[1];
                    }
                }
            }
        }
        for (long j = 0; j < cnt; ++j) { y[2 * (i0 + j)] = accr[j]; y[2 * (i0 + j) + 1] = acci[j]; }
    }
    free(dv); free(dr);
    return 0;
}

int das_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
