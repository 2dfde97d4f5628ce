// pre.hip -- the step in front of the DAS path: real RF traces -> analytic (complex) channel data, optionally downmixed.
//
// Replaces ChannelData.hilbert (reference src/ChannelData.m:935-966: fft along time to N points, weights
// w = [1; 2 x (floor(N/2)-1); 1 + mod(N,2); 0 ...], ifft) fused with ChannelData.downmix (src/ChannelData.m:757-766:
// data .* exp(-2i*pi*fc*time)), so that real (fp32 or int16) traces can be uploaded -- half / a quarter of the bytes of
// complex64 channel data over PCIe -- and turned into the beamformer's input on the device.  The FFTs are hipFFT plans
// (a library FFT, like the reference's MATLAB fft); the packing, weighting and downmix kernels are HIP.
//
// For real input, ifft(w .* fft(x)) = x + i*H(x): the real part IS the (padded / truncated) input, only the imaginary part
// needs an inverse transform, and it is real: H(x) = C2R of (-i sgn(f) X(f)) over the half spectrum.  So the pass is
// pad -> R2C -> half-spectrum weights -> C2R -> interleave (x, H(x)) [* phasor]: two half-size real transforms instead of a real
// forward + a full complex inverse one (whose length-2816 decomposition costs hipFFT two extra transposes).
//
// One-pass path (lengths N = 2^a 3^b 5^c 7^d 11^e 13^f up to 8192 whose stages fit a workgroup -- every record length of the BASELINE configurations, e.g. 2816 = 2^8 * 11):
// hilbert_lds_kernel keeps a whole trace pair in LDS.  Two real traces ride one complex transform (z = x1 + i x2; the analytic
// filter A = ifft(w .* fft(.)) is linear over C, so A z = (x1 - H x2) + i (x2 + H x1)), forward and inverse mixed-radix Stockham
// stages run in LDS, and the last stage writes (x, H x) [* phasor] -- HBM sees the real input once and the complex output once.
// Everything else (other lengths, QDAS_PRE_HIPFFT=1) takes the hipFFT passes below.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "qdas_kernels.h"
extern "C" int qdas_internal_upload(void *dst, const void *src, size_t bytes);      // qdas_api.hip: host -> device through pinned staging

namespace qdas {

// ---------------------------------------------------------------------------------------------------------------- one-pass path
// Stage s has radix r[s]; thread j < N / r[s] owns butterfly j of every stage and keeps its r[s] points in registers, so the trace pair
// lives in ONE LDS buffer (read -> barrier -> write -> barrier) -- 8 N bytes per workgroup, seven workgroups per CU at N = 2816.
// The first forward stage reads the real traces from HBM straight into the butterfly (its points j + t N/r are a coalesced pattern);
// the inverse transform runs the radices in the order r[1], ..., r[n-1], r[0], so its last stage produces points j + t N/r[0] --
// again coalesced -- and writes the result to HBM from registers.
struct FftStages { int n; int r[14]; };

// complex arithmetic on packed fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one instruction per complex add, two per complex multiply)
typedef float pre_v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ pre_v2f pv(float2 a) { return (pre_v2f){a.x, a.y}; }
static __device__ __forceinline__ float2 pf(pre_v2f a) { return make_float2(a.x, a.y); }
static __device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    const pre_v2f bb = pv(b);
    return pf((pre_v2f){-a.y, a.y} * bb.yx + (pre_v2f){a.x, a.x} * bb);
}
static __device__ __forceinline__ float2 caddf(float2 a, float2 b) { return pf(pv(a) + pv(b)); }
static __device__ __forceinline__ float2 csubf(float2 a, float2 b) { return pf(pv(a) - pv(b)); }
static __device__ __forceinline__ float2 cmulmi(float2 a) { return make_float2(a.y, -a.x); }              // a * (-i)

// length-4 DFT of (a, b, c, d) in place
static __device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 s02 = caddf(a, c), d02 = csubf(a, c), s13 = caddf(b, d), d13 = cmulmi(csubf(b, d));
    a = caddf(s02, s13); c = csubf(s02, s13); b = caddf(d02, d13); d = csubf(d02, d13);
}

// exp(-2 pi i m / 16), m = 0..9
static __device__ __forceinline__ float2 w16(int m) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    switch (m) {
        case 0: return make_float2(1.f, 0.f);
        case 1: return make_float2(C1, -S1);
        case 2: return make_float2(H, -H);
        case 3: return make_float2(S1, -C1);
        case 4: return make_float2(0.f, -1.f);
        case 5: return make_float2(-S1, -C1);
        case 6: return make_float2(-H, -H);
        case 7: return make_float2(-C1, -S1);
        case 8: return make_float2(-1.f, 0.f);
        default: return make_float2(-C1, S1);
    }
}

// length-R DFT of v in place.  2, 4, 8, 16: split-radix style networks with literal constants; odd R: the R x R product with
// wr[m] = exp(-2 pi i m / R) (uniform loads from the plan's table)
template <int R> static __device__ __forceinline__ void dft_small(float2 (&v)[R], const float2 *__restrict__ tw, uint32_t NR) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = caddf(a, b); v[1] = csubf(a, b);
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 8) {                                       // t = 2a + b, u = u1 + 4 u2
        dft4(v[0], v[2], v[4], v[6]);                                    // y0[u1]
        dft4(v[1], v[3], v[5], v[7]);                                    // y1[u1]
        v[3] = cmulf(v[3], w16(2)); v[5] = cmulmi(v[5]); v[7] = cmulf(v[7], w16(6));
        float2 o[8];
#pragma unroll
        for (int u1 = 0; u1 < 4; ++u1) { o[u1] = caddf(v[2 * u1], v[2 * u1 + 1]); o[u1 + 4] = csubf(v[2 * u1], v[2 * u1 + 1]); }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = o[u];
    } else if constexpr (R == 16) {                                      // t = 4a + b, u = u1 + 4 u2
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);      // y_b[u1] sits in v[4 u1 + b]
#pragma unroll
        for (int u1 = 1; u1 < 4; ++u1)
#pragma unroll
            for (int b = 1; b < 4; ++b) v[4 * u1 + b] = cmulf(v[4 * u1 + b], w16(b * u1));
        float2 o[16];
#pragma unroll
        for (int u1 = 0; u1 < 4; ++u1) {
            dft4(v[4 * u1], v[4 * u1 + 1], v[4 * u1 + 2], v[4 * u1 + 3]);           // over b -> u2
#pragma unroll
            for (int u2 = 0; u2 < 4; ++u2) o[u1 + 4 * u2] = v[4 * u1 + u2];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = o[u];
    } else {                                                             // odd R: pair the points t and R - t
        static_assert(R % 2 == 1, "even radices have their own networks");
        constexpr int H = (R - 1) / 2;
        float c[H + 1], sn[H + 1];                                       // cos, sin of 2 pi m / R, m = 0..H (uniform loads; m = 0 occurs for R = 9)
        c[0] = 1.f; sn[0] = 0.f;
#pragma unroll
        for (int m = 1; m <= H; ++m) { const float2 w = tw[m * NR]; c[m] = w.x; sn[m] = -w.y; }
        float2 a[H + 1], b[H + 1];
#pragma unroll
        for (int t = 1; t <= H; ++t) { a[t] = caddf(v[t], v[R - t]); b[t] = csubf(v[t], v[R - t]); }
        const float2 v0 = v[0];
        float2 sum = v0;
#pragma unroll
        for (int t = 1; t <= H; ++t) sum = caddf(sum, a[t]);
        v[0] = sum;
#pragma unroll
        for (int u = 1; u <= H; ++u) {                                   // X[u] = P - i Q, X[R-u] = P + i Q
            float2 P = v0, Q = make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 1; t <= H; ++t) {
                const int m = (u * t) % R;                               // cos(2 pi m/R) = cos(2 pi (R-m)/R), sin flips
                const float cc = m <= H ? c[m] : c[R - m], ss = m <= H ? sn[m] : -sn[R - m];
                P = pf(pv(a[t]) * cc + pv(P)); Q = pf(pv(b[t]) * ss + pv(Q));
            }
            v[u] = make_float2(P.x + Q.y, P.y - Q.x);
            v[R - u] = make_float2(P.x - Q.y, P.y + Q.x);
        }
    }
}

// LDS index of point i: one spare slot per 16 points, so that the stride-r writes of the early stages (points r j + t: 128-byte strides for
// r = 16 -- every second lane on the same pair of banks) spread over all banks
static __device__ __forceinline__ uint32_t lds_pad(uint32_t i) { return i + (i >> 4); }

struct HilbertArgs {
    const void *x; float2 *y; const float2 *tw;
    uint32_t T, N; uint64_t K;
    FftStages st;
    double fd, t0, fs;
};

// One Stockham stage of radix R (sub-transform length so far: Ns), in place in `buf` through registers.
//   ING: the stage opens the forward transform -- its points come from the two real traces in HBM
//   WGT: the stage opens the inverse transform -- its loads apply the analytic weights and the conjugate (ifft(X) = conj(fft(conj(X))) / N)
//   OUTG: the stage closes the inverse transform -- its results are finished (x, H x) [* phasor] and go to HBM
template <int R, bool ING, bool WGT, bool OUTG, typename TI, bool BIG>
static __device__ __forceinline__ void fft_stage(float2 *buf, const HilbertArgs &A, const uint32_t N, const uint32_t Ns, uint64_t k1, bool two) {
    constexpr int ITER = (BIG && R <= 8) ? 2 : 1;                        // butterflies per thread (the long-record variant doubles up on the light radices)
    const uint32_t NR = N / R;
    float2 v[ITER][R];
    const TI *x1 = (const TI *)A.x + (uint64_t)A.T * k1, *x2 = x1 + A.T;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const uint32_t j = threadIdx.x + it * blockDim.x;
        if (j < NR) {
            const uint32_t k = j % Ns;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const uint32_t idx = j + t * NR;
                if constexpr (ING) v[it][t] = make_float2(idx < A.T ? (float)x1[idx] : 0.f, (two && idx < A.T) ? (float)x2[idx] : 0.f);
                else v[it][t] = buf[lds_pad(idx)];
                if constexpr (WGT) {                                     // w = [1; 2 ...; 1 + mod(N,2); 0 ...]  (src/ChannelData.m:961-963)
                    const uint32_t h = N / 2;
                    const float g = idx == 0 ? 1.f : idx < h ? 2.f : idx == h ? (float)(1 + (N & 1)) : 0.f;
                    v[it][t] = make_float2(v[it][t].x * g, -v[it][t].y * g);
                }
            }
            if (Ns > 1) {                                                // twiddles exp(-2 pi i k t / (Ns R)): one table entry, the rest by products
                float2 w[R];
                w[1] = A.tw[k * (NR / Ns)];
#pragma unroll
                for (int t = 2; t < R; ++t) w[t] = cmulf(w[t / 2], w[t - t / 2]);
#pragma unroll
                for (int t = 1; t < R; ++t) v[it][t] = cmulf(v[it][t], w[t]);
            }
            dft_small<R>(v[it], A.tw, NR);
        }
    }
    if constexpr (!ING) __syncthreads();                                 // every butterfly has read its points
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const uint32_t j = threadIdx.x + it * blockDim.x;
        if (j < NR) {
            const uint32_t k = j % Ns, j0 = (j - k) * R + k;
            if constexpr (!OUTG) {
#pragma unroll
                for (int t = 0; t < R; ++t) buf[lds_pad(j0 + t * Ns)] = v[it][t];
            } else {                                                     // Ns == N / R here: j0 == j, points j + t NR
                const float sc = 1.0f / (float)N;
                float2 *y1 = A.y + (uint64_t)N * k1, *y2 = y1 + N;
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const uint32_t i = j0 + t * Ns;
                    const float2 z = v[it][t];                           // conj(z) / N = (x1 - H x2) + i (x2 + H x1)
                    const float r1 = i < A.T ? (float)x1[i] : 0.f, r2 = (two && i < A.T) ? (float)x2[i] : 0.f;
                    float2 v1 = make_float2(r1, -z.y * sc - r2), v2 = make_float2(r2, r1 - z.x * sc);
                    if (A.fd != 0.0) {
                        const double cyc = A.fd * (A.t0 + (double)i / A.fs);   // cycles; reduced in fp64 before the fp32 sincos
                        const float ph = (float)(cyc - floor(cyc));
                        const float c = __builtin_amdgcn_cosf(ph), sn = -__builtin_amdgcn_sinf(ph);
                        v1 = make_float2(v1.x * c - v1.y * sn, v1.x * sn + v1.y * c);
                        v2 = make_float2(v2.x * c - v2.y * sn, v2.x * sn + v2.y * c);
                    }
                    y1[i] = v1;
                    if (two) y2[i] = v2;
                }
            }
        }
    }
    if constexpr (!OUTG) __syncthreads();
}

template <bool ING, bool WGT, bool OUTG, typename TI, bool BIG>
static __device__ __forceinline__ void fft_stage_r(int R, float2 *buf, const HilbertArgs &A, uint32_t Ns, uint64_t k1, bool two) {
    const uint32_t N = A.N;
    switch (R) {
        case 2: fft_stage<2, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 3: fft_stage<3, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 4: fft_stage<4, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 5: fft_stage<5, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 7: fft_stage<7, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 8: fft_stage<8, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 9: fft_stage<9, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 11: fft_stage<11, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        case 13: fft_stage<13, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
        default: fft_stage<16, ING, WGT, OUTG, TI, BIG>(buf, A, N, Ns, k1, two); break;
    }
}

// BIG: up to 512 threads, two butterflies per thread on radices <= 8 (records up to 8192 samples); otherwise up to 256 threads
template <typename TI, bool BIG>
__global__ void __launch_bounds__(BIG ? 512 : 256) hilbert_lds_kernel(const HilbertArgs A) {
    extern __shared__ float2 pre_lds[];
    const uint64_t k1 = 2ull * blockIdx.x;
    const bool two = k1 + 1 < A.K;
    const int n = A.st.n;
    uint32_t Ns = 1;
    fft_stage_r<true, false, false, TI, BIG>(A.st.r[0], pre_lds, A, Ns, k1, two);       // forward
    Ns = A.st.r[0];
    for (int s = 1; s < n; ++s) { fft_stage_r<false, false, false, TI, BIG>(A.st.r[s], pre_lds, A, Ns, k1, two); Ns *= A.st.r[s]; }
    if (n == 1) { fft_stage_r<false, true, true, TI, BIG>(A.st.r[0], pre_lds, A, 1, k1, two); return; }
    fft_stage_r<false, true, false, TI, BIG>(A.st.r[1], pre_lds, A, 1, k1, two);        // weights, inverse: radices 1 .. n-1, then 0
    Ns = A.st.r[1];
    for (int s = 2; s < n; ++s) { fft_stage_r<false, false, false, TI, BIG>(A.st.r[s], pre_lds, A, Ns, k1, two); Ns *= A.st.r[s]; }
    fft_stage_r<false, false, true, TI, BIG>(A.st.r[0], pre_lds, A, Ns, k1, two);
}

// The same transform with the stage list (hence N, every stride and every twiddle step) known at compile time: the lists of the usual record
// lengths are prebuilt (FIXED_LISTS); other lengths take the run-time kernel above, whose register budget is the worst radix's.
template <typename TI, bool BIG, int R0, int R1, int R2, int R3>
__global__ void __launch_bounds__(BIG ? 512 : 256) hilbert_fixed_kernel(const HilbertArgs A) {
    extern __shared__ float2 pre_lds[];
    constexpr uint32_t N = (uint32_t)R0 * R1 * R2 * R3;
    const uint64_t k1 = 2ull * blockIdx.x;
    const bool two = k1 + 1 < A.K;
    fft_stage<R0, true, false, false, TI, BIG>(pre_lds, A, N, 1, k1, two);
    fft_stage<R1, false, false, false, TI, BIG>(pre_lds, A, N, R0, k1, two);
    if constexpr (R2 > 1) fft_stage<R2, false, false, false, TI, BIG>(pre_lds, A, N, R0 * R1, k1, two);
    if constexpr (R3 > 1) fft_stage<R3, false, false, false, TI, BIG>(pre_lds, A, N, R0 * R1 * R2, k1, two);
    fft_stage<R1, false, true, false, TI, BIG>(pre_lds, A, N, 1, k1, two);
    if constexpr (R2 > 1) fft_stage<R2, false, false, false, TI, BIG>(pre_lds, A, N, R1, k1, two);
    if constexpr (R3 > 1) fft_stage<R3, false, false, false, TI, BIG>(pre_lds, A, N, R1 * R2, k1, two);
    fft_stage<R0, false, false, true, TI, BIG>(pre_lds, A, N, R1 * R2 * R3, k1, two);
}

typedef void (*HilbertFn)(const HilbertArgs);
struct FixedList { int r[4]; bool big; HilbertFn f32, i16; };
#define QFIX(BIG, A, B, C, D) {{A, B, C, D}, BIG, hilbert_fixed_kernel<float, BIG, A, B, C, D>, hilbert_fixed_kernel<int16_t, BIG, A, B, C, D>}
static const FixedList FIXED_LISTS[] = {
    QFIX(false, 16, 16, 1, 1),    //  256
    QFIX(false, 8, 8, 8, 1),      //  512
    QFIX(false, 16, 8, 8, 1),     // 1024
    QFIX(false, 3, 8, 8, 8),      // 1536
    QFIX(false, 16, 16, 8, 1),    // 2048  (C1, C2, C5)
    QFIX(false, 5, 8, 8, 8),      // 2560
    QFIX(false, 11, 16, 16, 1),   // 2816  (C3)
    QFIX(true, 3, 16, 8, 8),      // 3072
    QFIX(false, 13, 16, 16, 1),   // 3328
    QFIX(true, 7, 8, 8, 8),       // 3584
    QFIX(false, 16, 16, 16, 1),   // 4096
    QFIX(true, 16, 8, 8, 8),      // 8192
};
#undef QFIX

static const FixedList *fixed_list(const FftStages &st, bool big) {
    for (const FixedList &f : FIXED_LISTS) {
        if (f.big != big) continue;
        bool same = true;
        for (int q = 0; q < 4; ++q) same = same && (q < st.n ? st.r[q] : 1) == f.r[q];
        if (same && st.n <= 4 && st.n >= 2) return &f;
    }
    return nullptr;
}

// N <= 8192 = product of radices {16, 8, 4, 2, 9, 3, 5, 7, 11, 13} whose stages fit the workgroup (256 threads, or 512 with two butterflies per thread on radices <= 8)?  The largest odd radix goes first (the
// first stage has no twiddles).
static bool fft_factor(uint64_t N, FftStages &st, unsigned &threads) {
    st.n = 0;
    if (N < 2 || N > 8192) return false;
    uint64_t n = N;
    auto push = [&](int r) { if (st.n >= 14) return false; st.r[st.n++] = r; n /= r; return true; };
    const int odd[6] = {13, 11, 7, 5, 9, 3};
    for (int r : odd) while (n % r == 0) if (!push(r)) return false;
    int e = 0;
    while (((n >> e) & 1) == 0) ++e;                                     // 2^e: ceil(e/4) stages of (nearly) equal radix, e.g. 2^13 = 16 8 8 8
    if (e) {
        const int m = (e + 3) / 4, base = e / m, extra = e % m;
        for (int q = 0; q < m; ++q) if (!push(1 << (base + (q < extra ? 1 : 0)))) return false;
    }
    if (n != 1) return false;
    uint64_t small = 0, big = 0;                                         // threads the 256-thread / the 512-thread variant needs
    for (int s = 0; s < st.n; ++s) {
        const uint64_t nr = N / st.r[s], nb = st.r[s] <= 8 ? (nr + 1) / 2 : nr;
        if (nr > small) small = nr;
        if (nb > big) big = nb;
    }
    if (small <= 256) threads = (unsigned)((small + 63) / 64 * 64);
    else if (big <= 512) threads = 0x10000u | (unsigned)((big + 63) / 64 * 64);     // flag: the long-record variant
    else return false;
    return true;
}

// real traces (T x K, fp32 or int16) -> zero-padded / truncated fp32 (N x K)
template <typename TI>
__global__ void __launch_bounds__(256) pre_pad_kernel(const TI *x, float *xr, uint64_t T, uint64_t N, uint64_t K) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    const uint64_t t = i % N, k = i / N;
    xr[i] = t < T ? (float)x[t + T * k] : 0.f;
}

// half spectrum X (N/2+1 bins per trace) -> spectrum of the Hilbert transform, scaled by the 1/N of the inverse transform:
// -i*sgn(f)*X(f)/N for 0 < f < N/2, 0 at DC and (even N) at Nyquist -- the imaginary part of what the reference's weights
// [1; 2...; 1 + mod(N,2); 0...] produce (src/ChannelData.m:961-963)
__global__ void __launch_bounds__(256) pre_spectrum_kernel(float2 *half, uint64_t N, uint64_t K, float scale) {
    const uint64_t H = N / 2 + 1;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= H * K) return;
    const uint64_t f = i % H;
    const bool zero = (f == 0) || ((N & 1) == 0 && f == N / 2);
    const float2 v = half[i];
    half[i] = zero ? make_float2(0.f, 0.f) : make_float2(v.y * scale, -v.x * scale);
}

// y = (x, H(x)) times the downmix phasor exp(-2j pi fd (t0 + t/fs)) (fd == 0: none)
__global__ void __launch_bounds__(256) pre_finish_kernel(const float *xr, const float *hr, float2 *y, uint64_t N, uint64_t K, double fd, double t0, double fs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    float2 v = make_float2(xr[i], hr[i]);
    if (fd != 0.0) {
        const double cyc = fd * (t0 + (double)(i % N) / fs);           // cycles; reduced in fp64 before the fp32 sincos
        const float ph = (float)(cyc - floor(cyc));
        const float c = __builtin_amdgcn_cosf(ph), s = -__builtin_amdgcn_sinf(ph);
        v = make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
    }
    y[i] = v;
}

struct PrePlan {
    uint64_t T, K, N;
    int in_type;          // 0: fp32, 1: int16
    double fs, t0, fd;
    hipfftHandle r2c = 0, c2r = 0;
    float *xr = nullptr;  // N x K real (padded input)
    float *hr = nullptr;  // N x K real (its Hilbert transform)
    float2 *half = nullptr;
    bool have = false;
    bool lds = false;     // one-pass path
    FftStages st{};
    unsigned threads = 0;
    float2 *tw = nullptr; // exp(-2 pi i k / N), k < N
};

int pre_create(PrePlan **out, uint64_t T, uint64_t K, uint64_t N, int in_type, double fs, double t0, double fd) {
    PrePlan *p = new PrePlan();
    p->T = T; p->K = K; p->N = N ? N : T; p->in_type = in_type; p->fs = fs; p->t0 = t0; p->fd = fd;
    if (p->N == 0 || K == 0) { *out = p; return 0; }
    const char *force = getenv("QDAS_PRE_HIPFFT");
    if (!(force && force[0] == '1') && p->N <= 0xffffffffull && T <= 0xffffffffull && fft_factor(p->N, p->st, p->threads)) {
        std::vector<float2> h(p->N);
        for (uint64_t k = 0; k < p->N; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)p->N;
            h[k] = make_float2((float)cos(a), (float)sin(a));
        }
        const size_t lds_bytes = sizeof(float2) * (p->N + p->N / 16 + 1);
        bool ok = hipMalloc(&p->tw, sizeof(float2) * p->N) == hipSuccess &&
                  qdas_internal_upload(p->tw, h.data(), sizeof(float2) * p->N) == (int)hipSuccess;
        if (ok && lds_bytes > 65536) {
            const void *fns[4] = {(const void *)hilbert_lds_kernel<float, false>, (const void *)hilbert_lds_kernel<float, true>,
                                  (const void *)hilbert_lds_kernel<int16_t, false>, (const void *)hilbert_lds_kernel<int16_t, true>};
            for (const void *fn : fns) ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess;
            for (const FixedList &fl : FIXED_LISTS) ok = ok && hipFuncSetAttribute((const void *)fl.f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess
                                                         && hipFuncSetAttribute((const void *)fl.i16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess;
        }
        if (ok) { p->lds = p->have = true; *out = p; return 0; }
        if (p->tw) { (void)hipFree(p->tw); p->tw = nullptr; }
        (void)hipGetLastError();
    }
    int n[1] = {(int)p->N};
    if (hipfftPlanMany(&p->r2c, 1, n, nullptr, 1, (int)p->N, nullptr, 1, (int)(p->N / 2 + 1), HIPFFT_R2C, (int)K) != HIPFFT_SUCCESS) { delete p; return 1; }
    if (hipfftPlanMany(&p->c2r, 1, n, nullptr, 1, (int)(p->N / 2 + 1), nullptr, 1, (int)p->N, HIPFFT_C2R, (int)K) != HIPFFT_SUCCESS) { hipfftDestroy(p->r2c); delete p; return 1; }
    if (hipMalloc(&p->xr, sizeof(float) * p->N * K) != hipSuccess || hipMalloc(&p->hr, sizeof(float) * p->N * K) != hipSuccess ||
        hipMalloc(&p->half, sizeof(float2) * (p->N / 2 + 1) * K) != hipSuccess) {
        hipfftDestroy(p->r2c); hipfftDestroy(p->c2r); if (p->xr) (void)hipFree(p->xr); if (p->hr) (void)hipFree(p->hr); delete p; return 2;
    }
    p->have = true;
    *out = p;
    return 0;
}

void pre_destroy(PrePlan *p) {
    if (!p) return;
    if (p->lds) { (void)hipFree(p->tw); delete p; return; }
    if (p->have) { hipfftDestroy(p->r2c); hipfftDestroy(p->c2r); (void)hipFree(p->xr); (void)hipFree(p->hr); (void)hipFree(p->half); }
    delete p;
}

bool pre_one_pass(const PrePlan *p) { return p && p->lds; }

int pre_execute(PrePlan *p, const void *x, void *y, hipStream_t s) {
    if (!p->have) return 0;
    if (p->lds) {
        HilbertArgs A{x, (float2 *)y, p->tw, (uint32_t)p->T, (uint32_t)p->N, p->K, p->st, p->fd, p->t0, p->fs};
        const unsigned nb = (unsigned)((p->K + 1) / 2);
        const size_t lds_bytes = sizeof(float2) * (p->N + p->N / 16 + 1);
        const unsigned th = p->threads & 0xffffu;
        const bool big = (p->threads >> 16) != 0;
        if (const FixedList *f = fixed_list(p->st, big)) {
            (p->in_type == 1 ? f->i16 : f->f32)<<<nb, th, lds_bytes, s>>>(A);
            return hipGetLastError() == hipSuccess ? 0 : 3;
        }
        if (p->in_type == 1) { if (big) hilbert_lds_kernel<int16_t, true><<<nb, th, lds_bytes, s>>>(A); else hilbert_lds_kernel<int16_t, false><<<nb, th, lds_bytes, s>>>(A); }
        else { if (big) hilbert_lds_kernel<float, true><<<nb, th, lds_bytes, s>>>(A); else hilbert_lds_kernel<float, false><<<nb, th, lds_bytes, s>>>(A); }
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    const uint64_t NK = p->N * p->K;
    const unsigned g = (unsigned)((NK + 255) / 256);
    if (p->in_type == 1) pre_pad_kernel<int16_t><<<g, 256, 0, s>>>((const int16_t *)x, p->xr, p->T, p->N, p->K);
    else pre_pad_kernel<float><<<g, 256, 0, s>>>((const float *)x, p->xr, p->T, p->N, p->K);
    if (hipfftSetStream(p->r2c, s) != HIPFFT_SUCCESS || hipfftSetStream(p->c2r, s) != HIPFFT_SUCCESS) return 1;
    if (hipfftExecR2C(p->r2c, p->xr, (hipfftComplex *)p->half) != HIPFFT_SUCCESS) return 1;
    const uint64_t HK = (p->N / 2 + 1) * p->K;
    pre_spectrum_kernel<<<(unsigned)((HK + 255) / 256), 256, 0, s>>>(p->half, p->N, p->K, 1.0f / (float)p->N);
    if (hipfftExecC2R(p->c2r, (hipfftComplex *)p->half, p->hr) != HIPFFT_SUCCESS) return 1;
    pre_finish_kernel<<<g, 256, 0, s>>>(p->xr, p->hr, (float2 *)y, p->N, p->K, p->fd, p->t0, p->fs);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}


// ---------------------------------------------------------------------------------------------------------------- FFT convolution (convd, long filters)
// convd of complex64 traces with ONE filter (ChannelData.filter's band-pass: reference kern/convd.m + src/convd.cu:95-146 form every product, M x N per
// trace) through the same LDS-resident transform: a trace is read once, zero-padded to a length Nf >= M + N - 1 that the stage list above takes,
// transformed, multiplied by the filter's spectrum, transformed back (ifft(Z) = conj(fft(conj(Z))) / Nf) and the outputs [off, off + L) of the linear
// convolution are written -- O(Nf log Nf) per trace instead of M x N products: from about a hundred taps on the direct kernel (conv.hip, bound by
// packed-FMA issue) is the slower one.  One trace per workgroup.  The spectrum H (times 1 / Nf) and the twiddle table are made on the stream by two
// tiny kernels in double precision; nothing is synchronised.  Rounding: a few 1e-7 of the largest output, like the direct sum of fp32 products.
struct FftConvArgs {
    const float2 *x; float2 *z; const float2 *H; const float2 *tw;
    uint32_t M, N, off, L; uint64_t K;
    FftStages st;
};

__global__ void __launch_bounds__(256) fftconv_tables_kernel(const void *taps, int taps_real, uint32_t ntaps, uint32_t N, float2 *H, float2 *tw) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)N, &sn, &cs);
    tw[k] = make_float2((float)cs, (float)sn);
    double re = 0.0, im = 0.0;
    for (uint32_t n = 0; n < ntaps; ++n) {
        const double yr = taps_real ? (double)((const float *)taps)[n] : (double)((const float2 *)taps)[n].x;
        const double yi = taps_real ? 0.0 : (double)((const float2 *)taps)[n].y;
        sincospi(-2.0 * (double)(((uint64_t)k * n) % N) / (double)N, &sn, &cs);
        re += yr * cs - yi * sn; im += yr * sn + yi * cs;
    }
    H[k] = make_float2((float)(re / (double)N), (float)(im / (double)N));
}

//   ING: opens the forward transform (points from the trace in HBM, zero beyond M);  WGT: opens the inverse one (times H, conjugated);
//   OUTG: closes it (conjugate back, outputs [off, off + L) to HBM)
template <int R, bool ING, bool WGT, bool OUTG, bool BIG>
static __device__ __forceinline__ void fftconv_stage(float2 *buf, const FftConvArgs &A, const uint32_t N, const uint32_t Ns, uint64_t k1) {
    constexpr int ITER = (BIG && R <= 8) ? 2 : 1;
    const uint32_t NR = N / R;
    float2 v[ITER][R];
    const float2 *x = A.x + (uint64_t)A.M * k1;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const uint32_t j = threadIdx.x + it * blockDim.x;
        if (j < NR) {
            const uint32_t k = j % Ns;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const uint32_t idx = j + t * NR;
                if constexpr (ING) v[it][t] = idx < A.M ? x[idx] : make_float2(0.f, 0.f);
                else v[it][t] = buf[lds_pad(idx)];
                if constexpr (WGT) { const float2 p = cmulf(v[it][t], A.H[idx]); v[it][t] = make_float2(p.x, -p.y); }
            }
            if (Ns > 1) {
                float2 w[R];
                w[1] = A.tw[k * (NR / Ns)];
#pragma unroll
                for (int t = 2; t < R; ++t) w[t] = cmulf(w[t / 2], w[t - t / 2]);
#pragma unroll
                for (int t = 1; t < R; ++t) v[it][t] = cmulf(v[it][t], w[t]);
            }
            dft_small<R>(v[it], A.tw, NR);
        }
    }
    if constexpr (!ING) __syncthreads();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const uint32_t j = threadIdx.x + it * blockDim.x;
        if (j < NR) {
            const uint32_t k = j % Ns, j0 = (j - k) * R + k;
            if constexpr (!OUTG) {
#pragma unroll
                for (int t = 0; t < R; ++t) buf[lds_pad(j0 + t * Ns)] = v[it][t];
            } else {
                float2 *z = A.z + (uint64_t)A.L * k1;
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const uint32_t i = j0 + t * Ns;
                    if (i >= A.off && i - A.off < A.L) z[i - A.off] = make_float2(v[it][t].x, -v[it][t].y);
                }
            }
        }
    }
    if constexpr (!OUTG) __syncthreads();
}

template <bool ING, bool WGT, bool OUTG, bool BIG>
static __device__ __forceinline__ void fftconv_stage_r(int R, float2 *buf, const FftConvArgs &A, uint32_t Ns, uint64_t k1) {
    const uint32_t N = A.N;
    switch (R) {
        case 2: fftconv_stage<2, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 3: fftconv_stage<3, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 4: fftconv_stage<4, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 5: fftconv_stage<5, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 7: fftconv_stage<7, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 8: fftconv_stage<8, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 9: fftconv_stage<9, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 11: fftconv_stage<11, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        case 13: fftconv_stage<13, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
        default: fftconv_stage<16, ING, WGT, OUTG, BIG>(buf, A, N, Ns, k1); break;
    }
}

template <bool BIG>
__global__ void __launch_bounds__(BIG ? 512 : 256) fftconv_lds_kernel(const FftConvArgs A) {
    extern __shared__ float2 pre_lds[];
    const uint64_t k1 = blockIdx.x;
    const int n = A.st.n;
    uint32_t Ns = 1;
    fftconv_stage_r<true, false, false, BIG>(A.st.r[0], pre_lds, A, Ns, k1);
    Ns = A.st.r[0];
    for (int s = 1; s < n; ++s) { fftconv_stage_r<false, false, false, BIG>(A.st.r[s], pre_lds, A, Ns, k1); Ns *= A.st.r[s]; }
    if (n == 1) { fftconv_stage_r<false, true, true, BIG>(A.st.r[0], pre_lds, A, 1, k1); return; }
    fftconv_stage_r<false, true, false, BIG>(A.st.r[1], pre_lds, A, 1, k1);
    Ns = A.st.r[1];
    for (int s = 2; s < n; ++s) { fftconv_stage_r<false, false, false, BIG>(A.st.r[s], pre_lds, A, Ns, k1); Ns *= A.st.r[s]; }
    fftconv_stage_r<false, false, true, BIG>(A.st.r[0], pre_lds, A, Ns, k1);
}

// 0: done (asynchronously on `s`), 1: not this path (no transform length up to 8192 takes M + ntaps - 1 points), 2: HIP error
int fftconv_launch(const void *x, const void *taps, int taps_real, void *z, uint64_t M, uint64_t ntaps, uint64_t K, uint64_t off, uint64_t L, hipStream_t s) {
    const uint64_t need = M + ntaps - 1;
    if (need > 8192 || K == 0 || K >= (1ull << 31)) return 1;
    FftStages st{};
    unsigned threads = 0;
    uint64_t N = 0;
    // the transform length: among the lengths up to a quarter beyond M + ntaps - 1 that the stage list takes, the one with the least points x stages
    // (every stage is a pass over the trace in LDS between two barriers: 2944 points needed -> 3328 = 13 x 16 x 16 beats 2970 = 11 x 5 x 9 x 3 x 2)
    double best = 0.0;
    for (uint64_t n = need; n <= 8192 && n <= need + need / 4 + 64; ++n) {
        FftStages c{};
        unsigned th = 0;
        if (!fft_factor(n, c, th)) continue;
        const double cost = (double)n * ((double)c.n + ((th >> 16) ? 0.5 : 0.0));
        if (!N || cost < best) { N = n; st = c; threads = th; best = cost; }
    }
    if (!N) return 1;
    const bool big = (threads >> 16) != 0;
    const unsigned th = threads & 0xffffu;
    const size_t lds_bytes = sizeof(float2) * (N + N / 16 + 1);
    if (lds_bytes > 65536) {
        const void *fn = big ? (const void *)fftconv_lds_kernel<true> : (const void *)fftconv_lds_kernel<false>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 1; }
    }
    Scratch scratch(s);
    float2 *tab = (float2 *)scratch.get(sizeof(float2) * 2 * N);         // [H | tw]
    if (!tab) return 1;
    fftconv_tables_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(taps, taps_real, (uint32_t)ntaps, (uint32_t)N, tab, tab + N);
    FftConvArgs A{(const float2 *)x, (float2 *)z, tab, tab + N, (uint32_t)M, (uint32_t)N, (uint32_t)off, (uint32_t)L, K, st};
    if (big) fftconv_lds_kernel<true><<<(unsigned)K, th, lds_bytes, s>>>(A); else fftconv_lds_kernel<false><<<(unsigned)K, th, lds_bytes, s>>>(A);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : 2;
}

}  // namespace qdas
