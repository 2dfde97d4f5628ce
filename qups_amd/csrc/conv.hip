// conv.hip -- batched 1-D convolution along one dimension (the band-pass / matched-filter step in front of DAS).
//
// Computes what the reference kernels conv / convf / convc / convcf compute (reference src/convd.cu:95-127, launched from
// kern/convd.m:150-199) for x: C x M x S and y: C x N x S (C = product of the dimensions in front of the convolved one):
//
//   z[c, l, s] = sum_i x[c, i, s] * y[c, (l + off) - i, s],   l = 0 .. L-1,
//
// `off` = index of output 0 in the full convolution: 0 ('full', L = M+N-1), N-1 - floor((N-1)/2) ('same', L = M),
// N-1 ('valid', L = M-N+1) (kern/convd.m:103-114: l0 = -lags(1); the kernel reads y[N-1 - (l0 - l + i)]).
//
// MI355X mapping.  The reference runs one thread per output that walks ALL of x (M iterations, most of them masked off) with two
// dependent global loads per product.  Here
//   * time-contiguous data (C == 1): a workgroup owns 1024 consecutive outputs of one slice; taps are processed in chunks of 256:
//     the x span a chunk needs (1024 + 256 samples, zero-filled outside the record) is staged in LDS, de-interleaved by
//     (index mod 8) so that a wave's reads are consecutive words.  Each lane produces 8 consecutive outputs and slides a
//     16-sample register window over the span; the taps are uniform and arrive through the scalar cache: 8 LDS reads feed 64
//     multiply-accumulates (256 FMAs for complex data).
//   * strided time (C > 1): lanes run along the contiguous column index c, every lane slides the same register window over its
//     own column straight from global memory (the loads of a wave are coalesced; the taps of a filter shared by all columns are
//     uniform and come through the scalar cache).
// A singleton column / slice dimension of either operand is broadcast by a zero stride instead of being replicated
// (the reference repmat's, kern/convd.m:75-84).
#include <type_traits>
#include "qdas_device.h"
#include "qdas_kernels.h"
#include "../../include/qdas.h"

namespace qdas {

constexpr int CV_TL = 1024;   // outputs per workgroup (time-contiguous kernel)
constexpr int CV_KC = 256;    // taps per LDS chunk

template <typename T> struct cv_zero;
template <> struct cv_zero<float>   { static __device__ __forceinline__ float   v() { return 0.f; } };
template <> struct cv_zero<double>  { static __device__ __forceinline__ double  v() { return 0.0; } };
template <> struct cv_zero<float2>  { static __device__ __forceinline__ float2  v() { return make_float2(0.f, 0.f); } };
template <> struct cv_zero<double2> { static __device__ __forceinline__ double2 v() { return make_double2(0.0, 0.0); } };

// storage type S -> compute type T and back: half-precision data (the reference's convh / convch, src/convd.cu:141,153) is multiplied and
// accumulated in fp32 and rounded once at the store (the reference accumulates in half)
template <typename T, typename S> struct cv_io {
    static __device__ __forceinline__ T ld(const S *p, uint64_t i) { return p[i]; }
    static __device__ __forceinline__ void st(S *p, uint64_t i, T v) { p[i] = v; }
};
template <> struct cv_io<float, _Float16> {
    static __device__ __forceinline__ float ld(const _Float16 *p, uint64_t i) { return (float)p[i]; }
    static __device__ __forceinline__ void st(_Float16 *p, uint64_t i, float v) { p[i] = (_Float16)v; }
};
struct cv_half2 { _Float16 x, y; };
template <> struct cv_io<float2, cv_half2> {
    static __device__ __forceinline__ float2 ld(const cv_half2 *p, uint64_t i) { const cv_half2 h = p[i]; return make_float2((float)h.x, (float)h.y); }
    static __device__ __forceinline__ void st(cv_half2 *p, uint64_t i, float2 v) { p[i] = cv_half2{(_Float16)v.x, (_Float16)v.y}; }
};

__device__ __forceinline__ void cv_mac(float &a, float x, float y)    { a = fmaf(x, y, a); }
__device__ __forceinline__ void cv_mac(double &a, double x, double y) { a = fma(x, y, a); }
__device__ __forceinline__ void cv_mac(float2 &a, float2 x, float2 y) {
    a.x = fmaf(x.x, y.x, a.x); a.x = fmaf(-x.y, y.y, a.x);
    a.y = fmaf(x.x, y.y, a.y); a.y = fmaf(x.y, y.x, a.y);
}
// complex data, real taps (a real band-pass / matched filter on analytic traces): half the multiplies
__device__ __forceinline__ void cv_mac(float2 &a, float2 x, float y)    { a.x = fmaf(x.x, y, a.x); a.y = fmaf(x.y, y, a.y); }
__device__ __forceinline__ void cv_mac(double2 &a, double2 x, double y) { a.x = fma(x.x, y, a.x); a.y = fma(x.y, y, a.y); }
__device__ __forceinline__ void cv_mac(double2 &a, double2 x, double2 y) {
    a.x = fma(x.x, y.x, a.x); a.x = fma(-x.y, y.y, a.x);
    a.y = fma(x.x, y.y, a.y); a.y = fma(x.y, y.x, a.y);
}

// ---- time-contiguous: x (M x S), y (N x S), z (L x S).  128 lanes x 8 consecutive outputs; the span is de-interleaved by
// (index mod 8) with rows padded so that both the staging writes and the window reads are bank-conflict free; taps are uniform
// per workgroup and come through the scalar cache (no LDS traffic): per 8 taps a lane issues 8 LDS reads for 64 MACs.
template <typename T> struct cv_row {      // padded row length: row stride = 8 words x (words per element)  (mod 64 banks)
    static constexpr int WPE = sizeof(T) / 4 > 4 ? 4 : sizeof(T) / 4;
    static constexpr int Q = (CV_TL + CV_KC) / 8;
    static constexpr int MOD = 64 / WPE;
    static constexpr int LEN = Q + ((8 - Q % MOD) % MOD + MOD) % MOD;
};

template <typename T, typename S, typename TT, typename ST>
__global__ void __launch_bounds__(128) conv_time_kernel(const ConvParams P) {
    constexpr int QP = cv_row<T>::LEN;
    using IO = cv_io<T, S>;
    using IOT = cv_io<TT, ST>;                   // taps: the data's type, or its real type
    __shared__ T X[8][QP];                       // X[w][k] = span element 8k + w
    const S *__restrict__ x = (const S *)P.x + (uint64_t)blockIdx.x * P.xss;     // slices along grid.x (may exceed 65535)
    const ST *__restrict__ y = (const ST *)P.y + (uint64_t)blockIdx.x * P.yss;
    S *__restrict__ z = (S *)P.z + (uint64_t)blockIdx.x * P.L;
    const int t = threadIdx.x;
    const int64_t M = (int64_t)P.M, N = (int64_t)P.N, L = (int64_t)P.L;
    const int64_t lfb = (int64_t)blockIdx.y * CV_TL + P.off;          // full-convolution index of this tile's first output
    // taps that meet the record for some output of the tile: j in [lfb - (M-1), lfb + TL - 1]
    int64_t jlo = lfb - (M - 1); if (jlo < 0) jlo = 0;
    int64_t jhi = lfb + CV_TL; if (jhi > N) jhi = N;                  // exclusive
    T acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = cv_zero<T>::v();
    for (int64_t j0 = jlo & ~(int64_t)7; j0 < jhi; j0 += CV_KC) {
        const int64_t o = lfb - j0 - CV_KC;                           // record index of span element 0
        const int nt = (int)((jhi - j0 < CV_KC) ? (jhi - j0) : CV_KC);
        const int no = (nt + 7) >> 3;                                 // groups of 8 taps
        {   // stage span elements [KC - 8 no, TL + KC): all loads in flight before the first LDS write
            constexpr int NLD = (CV_TL + CV_KC) / 128;
            const int e0 = CV_KC - 8 * no + t;
            T v[NLD];
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int e = e0 + 128 * q;
                const int64_t i = o + e;
                v[q] = (e < CV_TL + CV_KC && i >= 0 && i < M) ? IO::ld(x, (uint64_t)i) : cv_zero<T>::v();
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int e = e0 + 128 * q;
                if (e < CV_TL + CV_KC) X[e & 7][e >> 3] = v[q];
            }
            __syncthreads();
        }
        // window: win[w + 8] = span element (KC + 8t - 8g) + w, w = -8 .. 7  (output r, tap u of the group uses w = r - u)
        T win[16];
        int k = CV_KC / 8 + t;
#pragma unroll
        for (int w = 0; w < 8; ++w) win[8 + w] = X[w][k];
        // two groups per pass: the second one loads its eight new entries into the UPPER half and reads the window rotated by eight -- no register copies
        // (round 4 moved the lower half up after every group: 16 moves per 64 multiply-accumulates)
        for (int g = 0; g < no; g += 2) {
            --k;
#pragma unroll
            for (int w = 0; w < 8; ++w) win[w] = X[w][k];
            const int64_t jb = j0 + 8 * g;                            // uniform: scalar loads
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const TT tap = (jb + u < N) ? IOT::ld(y, (uint64_t)(jb + u)) : cv_zero<TT>::v();
#pragma unroll
                for (int r = 0; r < 8; ++r) cv_mac(acc[r], win[8 + r - u], tap);
            }
            if (g + 1 < no) {                                         // (uniform)
                --k;
#pragma unroll
                for (int w = 0; w < 8; ++w) win[8 + w] = X[w][k];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const TT tap = (jb + 8 + u < N) ? IOT::ld(y, (uint64_t)(jb + 8 + u)) : cv_zero<TT>::v();
#pragma unroll
                    for (int r = 0; r < 8; ++r) cv_mac(acc[r], win[(16 + r - u) & 15], tap);
                }
            }
        }
    }
    // outputs 8t .. 8t+7 of a lane go through LDS (same de-interleaved mapping as the span) so that the stores are coalesced
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) X[r][t] = acc[r];
    __syncthreads();
    const int64_t l0 = (int64_t)blockIdx.y * CV_TL;
#pragma unroll
    for (int q = 0; q < CV_TL / 128; ++q) {
        const int e = t + 128 * q;
        if (l0 + e < L) IO::st(z, (uint64_t)(l0 + e), X[e & 7][e >> 3]);
    }
}

// ---- strided time: x (C x M x S), y (C x N x S), z (C x L x S); lanes along c, 16 (double complex: 8) consecutive outputs per lane, the same
// kind of register window fed straight from global memory (a wave's loads are coalesced along c).  YB: y has one column
// (a filter shared by all traces): its taps are uniform and come through the scalar cache.
// outputs per lane (register budget: R accumulators + R + 8 window entries): 32 for 4-byte data and for complex fp32 data with REAL taps (the band-pass case),
// 16 for 8-byte data whose taps are as wide (complex x complex: the product's temporaries), 8 for double2
template <typename T, typename TT = T> struct cv_opl { static constexpr int V = sizeof(T) > 8 ? 8 : (sizeof(T) == 8 && sizeof(TT) == 8) ? 16 : 32; };

template <int N, class F> __device__ __forceinline__ void cv_unroll(F &&f) {
    if constexpr (N > 0) { cv_unroll<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

template <typename T, typename S, typename TT, typename ST, bool YB>
__global__ void __launch_bounds__(256) conv_col_kernel(const ConvParams P) {
    using IO = cv_io<T, S>;
    using IOT = cv_io<TT, ST>;
    constexpr int R = cv_opl<T, TT>::V;                                   // outputs per lane; taps go in groups of 8
    constexpr int WN = R + 8;                                             // registers of the sliding window
    constexpr int NG = WN / 8;                                            // groups after which the window's rotation is back where it started
    const uint32_t ncb = (uint32_t)((P.C + 63) / 64);
    const uint64_t sl = blockIdx.x / ncb;                                // slice; column block = blockIdx.x % ncb
    const uint64_t c = (uint64_t)(blockIdx.x % ncb) * 64 + threadIdx.x;
    // (the four waves of a block take four consecutive output ranges: the range -- and with it every record index below -- is UNIFORM per wave: scalar
    //  registers, scalar bounds tests, scalar address products; round 4 formed them per lane: 64-bit multiplies, compares and exec masks around every load)
    const int ty = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const int64_t l = ((int64_t)blockIdx.y * 4 + ty) * R;
    const int64_t M = (int64_t)P.M, N = (int64_t)P.N, L = (int64_t)P.L;
    if (l >= L) return;
    const bool mine = c < P.C;
    const uint64_t cc = mine ? c : P.C - 1;                              // (lanes past the last column compute on a real one and store nothing: no divergence inside the loop)
    const S *__restrict__ x = (const S *)P.x + sl * P.xss + cc * P.xcs;
    const ST *__restrict__ y = (const ST *)P.y + sl * P.yss + (YB ? 0 : cc * P.ycs);
    S *__restrict__ z = (S *)P.z + (sl * P.L) * P.C + cc;
    // (32-bit, scalar: M, N < 2^31 by the host's check; readfirstlane tells the compiler what it cannot see -- that these are uniform)
    const int Mi = __builtin_amdgcn_readfirstlane((int)M), Ni = __builtin_amdgcn_readfirstlane((int)N);
    const int lf = __builtin_amdgcn_readfirstlane((int)(l + P.off));
    int jlo = lf - (Mi - 1); if (jlo < 0) jlo = 0;
    int jhi = lf + R; if (jhi > Ni) jhi = Ni;
    const uint64_t xts = P.xts;
    auto ldx = [&](int i) -> T {                                          // i is uniform: one scalar test, one scalar product
        T v = cv_zero<T>::v();
        if (i >= 0 && i < Mi) v = IO::ld(x, (uint64_t)(uint32_t)i * xts);
        return v;
    };
    // The window: logical entry w = -8 .. R-1 holds x[lf - j + w] (output r, tap u of the group uses w = r - u).  Advancing j by 8 moves every entry up by 8:
    // instead of copying R registers per group (a quarter of the loop's instructions in round 4) the PHYSICAL register of logical w in group g is
    // (w + 8 - 8 g) mod (R + 8) -- the loop is unrolled over the NG groups of one rotation, all indices are compile-time constants.
    T acc[R], win[WN];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = cv_zero<T>::v();
    int j = jlo;
#pragma unroll
    for (int w = 0; w < R; ++w) win[8 + w] = ldx(lf - j + w);
    while (j < jhi) {
        cv_unroll<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (j < jhi) {                                                // (uniform)
                auto ph = [](int w) constexpr { return ((w + 8 - 8 * g) % WN + WN) % WN; };
                const int i0 = lf - j - 8;                                // record index of the group's first new row
                // (ONE path: an interior-group fast path without the tests doubled the loop's code to ~45 KB and ran 40 % slower -- the instruction cache)
#pragma unroll
                for (int w = 0; w < 8; ++w) win[ph(w - 8)] = ldx(i0 + w);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const TT tap = (j + u < Ni) ? IOT::ld(y, (uint64_t)(uint32_t)(j + u) * (YB ? 1 : P.yts)) : cv_zero<TT>::v();
#pragma unroll
                    for (int r = 0; r < R; ++r) cv_mac(acc[r], win[ph(r - u)], tap);
                }
                j += 8;
            }
        });
    }
    if (mine) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (l + r < L) IO::st(z, (uint64_t)(l + r) * P.C, acc[r]);
    }
}

template <typename T, typename S = T, typename TT = T, typename ST = S>
static hipError_t launch_conv_t(const ConvParams &P, hipStream_t s) {
    if (P.C == 1) {
        dim3 grid((unsigned)P.S, (unsigned)((P.L + CV_TL - 1) / CV_TL));
        hipLaunchKernelGGL((conv_time_kernel<T, S, TT, ST>), grid, dim3(128), 0, s, P);
    } else {
        dim3 grid((unsigned)(((P.C + 63) / 64) * P.S), (unsigned)((P.L + 4 * cv_opl<T, TT>::V - 1) / (4 * cv_opl<T, TT>::V)));
        if (P.ycs == 0 && P.yts == 1) hipLaunchKernelGGL((conv_col_kernel<T, S, TT, ST, true>), grid, dim3(64, 4), 0, s, P);
        else hipLaunchKernelGGL((conv_col_kernel<T, S, TT, ST, false>), grid, dim3(64, 4), 0, s, P);
    }
    return hipGetLastError();
}

// y_real: complex x and z, real y (cplx must be set)
hipError_t launch_conv(const ConvParams &P, int dtype, int cplx, int y_real, hipStream_t s) {
    if (cplx && y_real) {
        if (dtype == QDAS_F16) return launch_conv_t<float2, cv_half2, float, _Float16>(P, s);
        if (dtype == QDAS_F32) return launch_conv_t<float2, float2, float, float>(P, s);
        return launch_conv_t<double2, double2, double, double>(P, s);
    }
    if (dtype == QDAS_F16) return cplx ? launch_conv_t<float2, cv_half2>(P, s) : launch_conv_t<float, _Float16>(P, s);
    if (dtype == QDAS_F32) return cplx ? launch_conv_t<float2>(P, s) : launch_conv_t<float>(P, s);
    return cplx ? launch_conv_t<double2>(P, s) : launch_conv_t<double>(P, s);
}

}  // namespace qdas
