// greens.hip -- point-scatterer channel-data simulator: the scatter-side dual of delay-and-sum.
//
// Computes what the reference kernel greens[f] computes (reference src/greens.cu:8-86, launched from
// UltrasoundSystem.greens, src/UltrasoundSystem.m:681-718):
//
//   y[s, n, m] = 1/fsr * sum_i sum_ne sum_me  a_i * sample(x, fsr * (s - (cinv*(r1 + r2) + t0 - s0)*fs)) / (max(r1, R0) * max(r2, R0))
//
// r1 = |P_i - Pr[n, ne]|, r2 = |P_i - Pv[m, me]|, x = the transmit-receive waveform sampled at fsr*fs, "sample" the DAS
// interpolators (src/interpd.cu:68-167), zero outside the waveform.  R0 = 0 means no propagation loss (the reference's CPU
// branch, src/UltrasoundSystem.m:797-803; its device kernel divides by R0^2 and returns inf there, src/greens.cu:84).
//
// MI355X mapping.  The reference runs one thread per (s, n, m) that recomputes both distances for every scatterer.  Here a
// workgroup owns 1024 samples of one (n, m) trace: the delay and amplitude of every (scatterer, sub-aperture pair) is computed ONCE per
// block by the workgroup (256 entries per pass), the entries are binned in LDS by the wave whose 256 samples they reach, and every wave
// walks its own lists only; the waveform itself sits in LDS when it fits.
#include "qdas_device.h"
#include "qdas_kernels.h"

namespace qdas {

constexpr int GR_CHUNK = 256;
constexpr int GR_SPT = 4;                              // output samples per lane: a wave owns 64 * GR_SPT CONSECUTIVE samples of the block
__device__ __forceinline__ float  gsqrt(float v)  { return sqrtf(v); }
__device__ __forceinline__ double gsqrt(double v) { return sqrt(v); }

// Round 3: the entries of a chunk are BINNED by the wave whose samples they touch.  Pass 1 (all lanes, one entry each): delay and amplitude,
// then per consumer wave a ballot + prefix count appends the entry to that wave's list (one segment per producer wave: no cross-wave prefix,
// and the order within a list is the entry order -- sums are formed in the same order as before, bit for bit).  Pass 2: every wave walks ITS
// lists only -- entries that actually arrive in its 256 samples -- instead of testing every entry of the chunk.
template <int INTERP, typename TY>
__global__ void __launch_bounds__(256) greens_kernel(const GreensParams P) {
    using R  = typename TY::real;
    using ST = typename TY::store;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    cplx<R> *amp = (cplx<R> *)gsm;                       // [4 consumer][4 producer][64] a_i / (r1 r2 fsr)
    R *dly = (R *)(amp + 16 * 64);                       // [4][4][64] first-sample delay of the entry, in output samples
    int *cnt = (int *)(dly + 16 * 64);                   // [4][4]
    ST *xl = (ST *)(cnt + 16);                           // [T] waveform copy (only if P.x_in_lds)
    const uint32_t n = blockIdx.y, m = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const R *Ps = (const R *)P.Ps, *Pr = (const R *)P.Pr, *Pv = (const R *)P.Pv;
    const ST *a = (const ST *)P.a;
    const ST *x = (const ST *)P.x;
    const uint64_t T = P.T, S = P.S, I = P.I;
    if (P.x_in_lds) {
        for (uint64_t k = tid; k < T; k += 256) xl[k] = x[k];
        x = xl;
    }
    const R fs = (R)P.fs, fsr = (R)P.fsr, cinv = (R)P.cinv, R0 = (R)P.R0, toff = (R)(P.t0 - P.s0);
    const R span = (R)((double)T / P.fsr) + (R)2;        // output samples one waveform can touch (+ interpolator margin)
    const int EE = P.En * P.Em;
    const uint64_t entries = I * (uint64_t)EE;
    constexpr int SPT = GR_SPT, WS = 64 * SPT;           // samples per wave
    // this lane's output samples: s = blockIdx.x * 256 * SPT + wave * 64 * SPT + lane + 64 * q
    const uint64_t s0i = (uint64_t)blockIdx.x * (256 * SPT) + (uint64_t)wave * WS + lane;
    cplx<R> acc[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) acc[q] = {(R)0, (R)0};
    const R blk_lo = (R)((uint64_t)blockIdx.x * (256 * SPT));

    for (uint64_t e0 = 0; e0 < entries; e0 += GR_CHUNK) {
        __syncthreads();                                 // the lists of the previous chunk are consumed (and the waveform copy is complete)
        {   // one entry per lane: (scatterer i, receive sub-aperture ne, transmit sub-aperture me)
            const uint64_t e = e0 + tid;
            R d = (R)0;
            cplx<R> w = {(R)0, (R)0};
            int b0 = 1, b1 = 0;                          // consumer waves the entry touches: none
            if (e < entries) {
                const uint64_t i = e / EE;
                const int sub = (int)(e % EE), ne = sub % P.En, me = sub / P.En;
                const R px = Ps[3 * i], py = Ps[3 * i + 1], pz = Ps[3 * i + 2];
                const size_t kr = 3 * ((size_t)n + (size_t)ne * P.N), kv = 3 * ((size_t)m + (size_t)me * P.M);
                const R ax = px - Pr[kr], ay = py - Pr[kr + 1], az = pz - Pr[kr + 2];
                const R bx = px - Pv[kv], by = py - Pv[kv + 1], bz = pz - Pv[kv + 2];
                R r1 = gsqrt(ax * ax + ay * ay + az * az), r2 = gsqrt(bx * bx + by * by + bz * bz);   // src/greens.cu:61-62
                d = (cinv * (r1 + r2) + toff) * fs;                                                   // src/greens.cu:65
                if (R0 != (R)0) { r1 = r1 < R0 ? R0 : r1; r2 = r2 < R0 ? R0 : r2; } else { r1 = (R)1; r2 = (R)1; }
                const cplx<R> ai = ld(a, i);
                const R g = (R)1 / (r1 * r2 * fsr);
                w = {ai.x * g, ai.y * g};
                // the entry touches output samples (d - 2, d + span): which waves' ranges [blk_lo + 256 b, blk_lo + 256 (b + 1)) meet that?
                const R lo = (d - (R)2) - blk_lo, hi = (d + span) - blk_lo;
                if (hi >= (R)0 && lo < (R)(4 * WS)) {    // (a non-finite delay fails both tests: never listed, as before)
                    b0 = lo <= (R)0 ? 0 : (int)(lo * (R)(1.0 / WS));
                    b1 = hi >= (R)(4 * WS) ? 3 : (int)(hi * (R)(1.0 / WS));
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool in = b0 <= b && b <= b1;
                const uint64_t mask = __ballot(in);
                if (in) {
                    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    const uint32_t k = (uint32_t)(b * 4 + (int)wave) * 64u + pos;
                    dly[k] = d; amp[k] = w;
                }
                if (lane == 0) cnt[b * 4 + (int)wave] = __builtin_popcountll(mask);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int pw = 0; pw < 4; ++pw) {                 // this wave's lists, producer by producer: entry order
            const int nk = __builtin_amdgcn_readfirstlane(cnt[(int)wave * 4 + pw]);
            const uint32_t base = (uint32_t)((int)wave * 4 + pw) * 64u;
            for (int k = 0; k < nk; ++k) {
                const R d = dly[base + k];               // broadcast LDS reads
                const cplx<R> w = amp[base + k];
#pragma unroll
                for (int q = 0; q < SPT; ++q) {
                    const R tau = (R)(s0i + 64 * q) - d; // src/greens.cu:65: kernel time of this sample, in output samples
                    if (tau > (R)-2 && tau < span) {
                        const cplx<R> v = sample_global<INTERP, R, ST>(x, (long)T, fsr * tau);   // src/greens.cu:79
                        acc[q].x += w.x * v.x - w.y * v.y; acc[q].y += w.x * v.y + w.y * v.x;
                    }
                }
            }
        }
    }
    ST *y = (ST *)P.y + ((size_t)n + (size_t)m * P.N) * S;                                      // src/greens.cu:84
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const uint64_t s = s0i + 64 * q;
        if (s < S) st(y, (size_t)s, acc[q]);
    }
}

template <typename TY>
static hipError_t launch_greens_t(const GreensParams &P, hipStream_t s) {
    GreensParams p = P;
    const size_t esz = sizeof(typename TY::store), rsz = sizeof(typename TY::real);
    size_t lds = 16 * 64 * (rsz + esz) + 16 * sizeof(int);
    p.x_in_lds = (P.T * esz <= 96 * 1024) ? 1 : 0;
    if (p.x_in_lds) lds += P.T * esz;
    const dim3 g((unsigned)((P.S + 1023) / 1024), (unsigned)P.N, (unsigned)P.M), b(256);
#define QG(I)                                                                                       \
    do {                                                                                            \
        auto kfn = greens_kernel<I, TY>;                                                            \
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                              \
        kfn<<<g, b, lds, s>>>(p);                                                                   \
    } while (0)
    switch (P.interp) {
        case 0: QG(0); break;
        case 1: case 4: QG(1); break;
        case 2: QG(2); break;
        case 3: QG(3); break;
        case 5: QG(5); break;
        default: return hipErrorInvalidValue;
    }
#undef QG
    return hipGetLastError();
}

hipError_t launch_greens(const GreensParams &P, int dtype, hipStream_t s) {
    if (P.S == 0 || P.N == 0 || P.M == 0) return hipSuccess;
    switch (dtype) {
        case 0: return launch_greens_t<st_f64>(P, s);
        case 1: return launch_greens_t<st_f32>(P, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
