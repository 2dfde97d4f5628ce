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
#include <math.h>
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------
// Many scatterers: IMPULSE TRAINS (round 4).  With an integer waveform-to-data sampling ratio Q = fsr the waveform position of output sample s for
// an entry with delay d (output samples) is p = Q s - Q d = (Q s - c) + u with c = ceil(Q d) and u = c - Q d in [0, 1): the tap index ti = Q s - c
// runs over the samples, the FRACTION u -- hence the interpolation weights -- is ONE per entry.  So
//     y[s] = sum_k sum_t H_k[Q s - t] x[t + o_k],     H_k[c] = sum over the entries with ceil(Q d_i) = c of  a_i g_i w_k(u_i),
// (o_k: tap offsets of the interpolator; t over the tap indices the edge rule of qdas_device.h admits): every entry costs its delay, its K weights
// and K complex adds into K impulse trains -- whatever the length of the waveform --, and ONE dense convolution of the trains with the waveform per
// block of samples follows (K x T multiply-adds per output sample, whatever the number of scatterers).  The kernel above evaluates the waveform per
// (entry, sample) -- ~1.3 divergent sample evaluations of 256 lanes per entry and wave --: 100 000 scatterers on the C1 geometry 95 ms per call; this one 7.5 ms.
// Exactly the reference's sum (src/greens.cu:8-86) re-associated, with the edge rule per train.
// The trains are accumulated in LDS with 64-bit INTEGER atomics (ds_add_u64) on fixed-point values: integer addition is associative, so the result
// does not depend on the order in which the lanes arrive -- bit-reproducible, unlike float atomics (the reference's own wsinterpd2 path, src/interpd.cu:393).
// Scale: 2^46 / (largest single contribution, greens_bound_kernel): 17 bits of headroom for coincident entries; a contribution 2^-22 of the largest
// still carries fp32's 24 bits.  fp32 data only (fixed point would cost fp64 data its last bits); non-integer fsr and few entries keep the kernel above.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) greens_bound_kernel(const GreensParams P, unsigned int *bound_bits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.I) return;
    const float *Ps = (const float *)P.Ps, *Pr = (const float *)P.Pr, *Pv = (const float *)P.Pv;
    const float2 ai = ((const float2 *)P.a)[i];
    float b = sqrtf(ai.x * ai.x + ai.y * ai.y) / (float)P.fsr;
    const float R0 = (float)P.R0;
    if (R0 != 0.f) {                                     // 1 / (max(r1, R0) max(r2, R0)) at the nearest receive / transmit (sub-)element
        const float px = Ps[3 * i], py = Ps[3 * i + 1], pz = Ps[3 * i + 2];
        float m1 = INFINITY, m2 = INFINITY;
        for (uint64_t k = 0; k < P.N * (uint64_t)P.En; ++k) { const float ax = px - Pr[3 * k], ay = py - Pr[3 * k + 1], az = pz - Pr[3 * k + 2]; m1 = fminf(m1, ax * ax + ay * ay + az * az); }
        for (uint64_t k = 0; k < P.M * (uint64_t)P.Em; ++k) { const float ax = px - Pv[3 * k], ay = py - Pv[3 * k + 1], az = pz - Pv[3 * k + 2]; m2 = fminf(m2, ax * ax + ay * ay + az * az); }
        b /= fmaxf(sqrtf(m1), R0) * fmaxf(sqrtf(m2), R0);
    }
    b *= 1.5f;                                           // |w_k| <= 1 for every interpolator here (Lanczos peaks at 1, the cubics at 1): margin for rounding
    if (!(b >= 0.f) || b > 3.0e38f) b = INFINITY;        // a non-finite amplitude: the trains cannot carry it (the kernel writes NaN)
    atomicMax(bound_bits, __float_as_uint(b));           // non-negative floats order like their bit patterns
    if (ai.y != 0.f) bound_bits[1] = 1u;                 // (any complex amplitude: real clouds -- the usual ones -- skip the imaginary trains' arithmetic; same value from every writer)
}

// element-to-scatterer distances, ONCE per launch: the delay is separable, r1 depends on (scatterer, receive element) and r2 on (scatterer, transmit
// element) only -- (N En + M Em) I square roots instead of the N En M Em I x 2 (x blocks per trace) of a scan that recomputes them.
// One workgroup = one CHUNK of 256 scatterers of one element: it also leaves the chunk's {smallest, largest} distance (fminf / fmaxf: a NaN distance
// never lands anywhere and is ignored; a chunk of NaNs gets {+inf, -inf}: never visited)
#ifndef QDAS_GT_CHUNK
#define QDAS_GT_CHUNK 256
#endif
constexpr int GT_CHUNK = QDAS_GT_CHUNK;          // scatterers per chunk of the lists (a build-time experiment knob: 128 / 256 / 512)
__global__ void __launch_bounds__(GT_CHUNK) greens_dist_kernel(const float *__restrict__ Ps, const float *__restrict__ Pe, float *__restrict__ R, float2 *__restrict__ cb, uint64_t I) {
    __shared__ float2 part[GT_CHUNK / 64];
    const uint64_t i = (uint64_t)blockIdx.x * GT_CHUNK + threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    if (i < I) {
        const size_t k = 3 * (size_t)blockIdx.y;
        const float ax = Ps[3 * i] - Pe[k], ay = Ps[3 * i + 1] - Pe[k + 1], az = Ps[3 * i + 2] - Pe[k + 2];
        const float r = sqrtf(ax * ax + ay * ay + az * az);                      // src/greens.cu:61-62
        R[(size_t)blockIdx.y * I + i] = r;
        lo = fminf(lo, r); hi = fmaxf(hi, r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = make_float2(lo, hi);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < GT_CHUNK / 64; ++w) { lo = fminf(lo, part[w].x); hi = fmaxf(hi, part[w].y); }
        cb[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = make_float2(lo, hi);
    }
}

// Scatterers grouped by the MORTON cell of their positions (4 to 6 bits per axis of the cloud's bounding box: about four scatterers per cell of a volume;
// a counting sort: histogram, prefix, scatter --
// the order inside a cell is whatever the atomics make it): a chunk of 256 consecutive scatterers is a compact run of cells, so its distances to any
// one element span a few cell diameters and a workgroup -- 1 / (blocks per trace) of the delay range -- can tell from the chunk bounds alone that
// most chunks do not reach it.  The trains are INTEGER sums: the order of the scatterers does not change a bit of the result.
constexpr uint32_t GT_CELLS = 1u << 18;                // (6 bits per axis: the most)
static int gt_cell_bits(uint64_t I) { int b = 4; while (b < 6 && (4ull << (3 * b)) < I) ++b; return b; }
__device__ __forceinline__ uint32_t gt_ordered(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float gt_unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__global__ void __launch_bounds__(256) greens_bbox_kernel(const float *__restrict__ Ps, uint64_t I, uint32_t *__restrict__ bb) {      // bb: {min x, y, z, max x, y, z}, ordered bits
    __shared__ float part[4][6];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    bool ok = false;
    if (i < I) { v[0] = Ps[3 * i]; v[1] = Ps[3 * i + 1]; v[2] = Ps[3 * i + 2]; ok = fabsf(v[0]) <= 3.0e38f && fabsf(v[1]) <= 3.0e38f && fabsf(v[2]) <= 3.0e38f; }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float lo = ok ? v[d] : INFINITY, hi = ok ? v[d] : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        if ((threadIdx.x & 63u) == 0) { part[threadIdx.x >> 6][d] = lo; part[threadIdx.x >> 6][3 + d] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 3u) {                              // (six atomics per workgroup, not per wave: they all land on the same six words)
        const int d = (int)threadIdx.x;
        float lo = part[0][d], hi = part[0][3 + d];
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, part[w][d]); hi = fmaxf(hi, part[w][3 + d]); }
        if (lo <= hi) { atomicMin(bb + d, gt_ordered(lo)); atomicMax(bb + 3 + d, gt_ordered(hi)); }
    }
}
__device__ __forceinline__ uint32_t gt_spread3(uint32_t v) {       // (up to) 6 bits -> every third bit
    v &= 0x3fu; v = (v | (v << 8)) & 0x300fu; v = (v | (v << 4)) & 0x30c3u; v = (v | (v << 2)) & 0x9249u;
    return v;
}
__global__ void __launch_bounds__(256) greens_key_kernel(const float *__restrict__ Ps, uint64_t I, const uint32_t *__restrict__ bb, int bits, uint32_t *__restrict__ key, uint32_t *__restrict__ hist) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    uint32_t k = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float lo = gt_unordered(bb[d]), hi = gt_unordered(bb[3 + d]), v = Ps[3 * i + d];
        const float top = (float)((1 << bits) - 1), sc = hi > lo ? top / (hi - lo) : 0.f;
        float c = (v - lo) * sc;
        c = c >= 0.f ? (c <= top ? c : top) : 0.f;                                // (NaN -> 0: any order is a valid order)
        k |= gt_spread3((uint32_t)c) << d;
    }
    key[i] = k;
    atomicAdd(hist + k, 1u);
}
// counts -> first positions, in place (one workgroup: 1024 threads x 256 cells)
__global__ void __launch_bounds__(1024) greens_cellscan_kernel(uint32_t *__restrict__ hist, uint32_t cells) {
    __shared__ uint32_t part[1024];
    const uint32_t PER = cells / 1024, tid = threadIdx.x;          // (cells: 4096 and up)
    uint32_t sum = 0;
    for (uint32_t c = 0; c < PER; ++c) sum += hist[tid * PER + c];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;                                               // exclusive
    for (uint32_t c = 0; c < PER; ++c) { const uint32_t n = hist[tid * PER + c]; hist[tid * PER + c] = run; run += n; }
}
__global__ void __launch_bounds__(256) greens_gather_kernel(const float *__restrict__ Ps, const float2 *__restrict__ a, const uint32_t *__restrict__ key, uint32_t *__restrict__ cursor, uint64_t I,
                                                            float *__restrict__ Pso, float2 *__restrict__ ao) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    const size_t j = atomicAdd(cursor + key[i], 1u);
    Pso[3 * j] = Ps[3 * i]; Pso[3 * j + 1] = Ps[3 * i + 1]; Pso[3 * j + 2] = Ps[3 * i + 2];
    ao[j] = a[i];
}

// |t| < 2^51 float -> the nearest 64-bit integer (ties to even), without the generic conversion sequence: the float is exact as a double, and
// adding 1.5 x 2^52 leaves the integer in the low mantissa bits -- a conversion, a double add and a 64-bit subtract (the two-rounding fp32 split that
// stood here took eleven instructions and was a fifth of the work-off loop)
__device__ __forceinline__ long long gt_fixed(float t) {
    const double d = (double)t + 6755399441055744.0;
    return (long long)__double_as_longlong(d) - 0x4338000000000000ll;
}

// ... and back: (float)v without the generic 64-bit conversion sequence (two 32-bit conversions and an FMA; the sum is rounded once more, 2^-24 relative)
__device__ __forceinline__ float gt_float(long long v) {
    return fmaf((float)(int)(v >> 32), 4294967296.0f, (float)(unsigned int)(v & 0xffffffffll));
}

// The convolution's tap list, ONCE per launch: with t = Q j + p the sum over the taps of train k is a sum over the phases p of plain FIR filters on
// unit-stride sequences (see the kernel below).  Every (train, phase) segment is cut into GROUPS of 8 taps (the last one padded with zero taps):
// per group the offset of its first element in the de-interleaved trains, and the 8 waveform taps {x, i x} as float4 -- read by the kernel through
// SCALAR loads (group and taps are the same for a whole wave): one wait per 8 taps.
constexpr int GT_SEGS = 64;                              // K Q at most
typedef float gt_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) gt_f4 *gt_cf4;      // (constant address space: uniform indices become s_load)
typedef const __attribute__((address_space(4))) int *gt_ci;
__global__ void __launch_bounds__(256) greens_xtab_kernel(const float2 *__restrict__ x, int K, int Q, int O0, int tlo, int thi0, int thi1, int ALP, int jmax, int *__restrict__ grp,
                                                          float4 *__restrict__ xt) {
    __shared__ int4 sg[GT_SEGS + 1];                     // {first j, taps, first group, groups}
    const int tid = (int)threadIdx.x;
    if (tid < K * Q) {
        const int k = tid / Q, ph = tid % Q, thk = k == 0 ? thi0 : thi1;
        const int jlo = tlo > ph ? 1 : 0, jhi = thk >= ph ? (thk - ph) / Q : -1, cnt = jhi >= jlo ? jhi - jlo + 1 : 0;
        sg[tid] = make_int4(jlo, cnt, 0, (cnt + 7) / 8);
    }
    __syncthreads();
    if (tid == 0) { int base = 0; for (int c = 0; c < K * Q; ++c) { sg[c].z = base; base += sg[c].w; } sg[GT_SEGS] = make_int4(base, 0, 0, 0); }
    __syncthreads();
    const long G = sg[GT_SEGS].x;
    if (blockIdx.x == 0 && tid == 0) grp[0] = (int)G;    // grp[0]: the number of groups; grp[1 + g]: group g
    const long st = (long)blockIdx.x * 256 + tid;
    if (st >= 8 * G) return;
    const long g = st >> 3;
    int c = 0;
    while (c + 1 < K * Q && g >= (long)sg[c + 1].z) ++c;
    const int k = c / Q, ph = c % Q, u = (int)(st - 8 * (long)sg[c].z), j = sg[c].x + u;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);          // (a padded tap)
    if (u < sg[c].y) { const float2 xv = x[(long)Q * j + ph + O0 + k]; v = make_float4(xv.x, xv.y, -xv.y, xv.x); }
    xt[st] = v;
    if ((st & 7) == 0) grp[1 + g] = c * ALP + jmax - j;  // element of A_kp[s - j] for output 0 (output so: + so; the next taps: - 1 each)
}

constexpr int GT_WQ = 128;                               // entries a wave queues before it works 64 of them off

// One block of SB output samples of one (receiver, transmit) trace, in three stages over the same LDS (TrainBlock::deposit, ::convert, ::convolve).
// TH threads per workgroup: 256 with blocks of 64 / 128 outputs (a few tens of KB of LDS: several workgroups per CU, so one's barriers and exposed
// latencies -- the chunk list, the table reads, the amplitude gather, the conversion passes -- are covered by another's work) whenever the trains of
// such a block fit; 1024 with the longest block that fits for long waveforms / large ratios.
template <int INTERP, int TH>
struct TrainBlock {
    static constexpr uint32_t THREADS = TH, LCAP = 2 * TH, CGRP = TH / GT_CHUNK;
    static constexpr int K = INTERP == 0 ? 2 : interp_taps(INTERP);  // nearest: two trains (the sample at ti, or at ti + 1)
    const GreensParams &P;
    unsigned char *gsm;
    // the block: tap indices ti the edge rule admits (qdas_device.h sample_global): 4 taps: 1 <= ti <= T-3; linear: 0 <= ti <= T-2; nearest: ti >= 0 and ti (+1) < T
    long tlo, thi;                                       // (thi: train 0; trains 1.. of 'nearest': T - 2 -- greens_xtab_kernel)
    int Q;
    uint32_t SB, NSLOT, n, m, tid, lane, wave;
    uint64_t s_lo;
    long cbase;                                          // fine index of slot 0
    long long *H;                                        // [K][NSLOT] {re, im} fixed point
    float bound, Sc;
    bool a_cplx;
    // the wave's queue and the 64 entries it has taken and not retired
    float4 *myq;
    uint32_t qn = 0, pcount = 0;                         // (uniform)
    float4 pit = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 pai = make_float2(0.f, 0.f);

    __device__ TrainBlock(const GreensParams &P_, unsigned char *gsm_, uint32_t blk, uint32_t trace, const unsigned int *bound_bits) : P(P_), gsm(gsm_) {
        const long T = (long)P.T;
        Q = P.q; SB = P.sb;
        tlo = K == 4 ? 1 : 0;
        thi = K == 4 ? T - 3 : (INTERP == 0 ? T - 1 : T - 2);
        NSLOT = (uint32_t)((long)Q * (SB - 1) + thi - tlo + 1);
        n = trace % (uint32_t)P.N; m = trace / (uint32_t)P.N;
        tid = threadIdx.x; lane = tid & 63u;
        wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
        s_lo = (uint64_t)blk * SB;
        cbase = (long)Q * (long)s_lo - thi;
        H = (long long *)gsm;
        myq = (float4 *)(H + (size_t)K * NSLOT * 2) + (size_t)wave * GT_WQ;          // [waves][GT_WQ] {scatterer, r1, r2}: this wave's entries that (may) land in the block
        bound = __uint_as_float(bound_bits[0]);
        a_cplx = __builtin_amdgcn_readfirstlane((int)bound_bits[1]) != 0;
        // scale: 2^46 / largest single contribution -- and never more than 2^62 / (entries of the trace x largest contribution): a slot can at most receive
        // every entry of its trace, so the 64-bit sums cannot wrap however dense the scatterer cloud (beyond 2^16 entries per trace the resolution drops from
        // 2^-46 to 2^-62 x entries of the largest contribution: still 2^-32 at 10^9 entries, far below the fp32 rounding of the result)
        const float ent = (float)P.I * (float)(P.En * P.Em);
        Sc = bound > 0.f ? fminf(fminf(70368744177664.0f, 4.611686018427388e18f / fmaxf(ent, 1.f)) / bound, 1.0e37f) : 0.f;
    }

    // Work-off in two halves: `take` moves 64 queued entries into registers and ISSUES their amplitude gathers; `retire` -- at the next trigger,
    // a few scan passes later -- finishes them.  The gather's latency (the one dependent global read of an entry) runs under the scan.
    __device__ __forceinline__ void take(uint32_t first, uint32_t count) {
        if (lane < count) { pit = myq[first + lane]; pai = ((const float2 *)P.a)[__float_as_uint(pit.x)]; }
        pcount = count;
    }
    __device__ __forceinline__ void retire() {
        if (P.dbg & 2) { pcount = 0; return; }
        if (lane < pcount) {
            const float fs = (float)P.fs, fsr = (float)P.fsr, cinv = (float)P.cinv, R0 = (float)P.R0, toff = (float)(P.t0 - P.s0);
            float r1 = pit.y, r2 = pit.z;
            const float d = (cinv * (r1 + r2) + toff) * fs;                                                   // src/greens.cu:65
            const float ef = (float)Q * d, cf = ceilf(ef);
            const float sl = cf - (float)cbase;                      // slot of the entry (the reference's own fp32 delay arithmetic); a non-finite delay fails the test
            if (sl >= 0.f && sl < (float)NSLOT) {
                const uint32_t slot = (uint32_t)sl;
                const float u = cf - ef;                             // in [0, 1): exact
                if (R0 != 0.f) { r1 = r1 < R0 ? R0 : r1; r2 = r2 < R0 ? R0 : r2; } else { r1 = 1.f; r2 = 1.f; }
                const float g = Sc * __builtin_amdgcn_rcpf(r1 * r2 * fsr);        // (v_rcp_f32: 1 ulp -- the sum is compared at 1e-4; the full division sequence was a tenth of this loop)
                float w[4];
                if constexpr (INTERP == 0) { w[0] = u < 0.5f ? 1.f : 0.f; w[1] = 1.f - w[0]; }
                else interp_weights<INTERP>(u, w);
                unsigned long long *h = (unsigned long long *)(H + (size_t)slot * 2);
                if (a_cplx) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const long long vr = gt_fixed(pai.x * g * w[k]), vi = gt_fixed(pai.y * g * w[k]);
                        atomicAdd(h + (size_t)k * NSLOT * 2, (unsigned long long)vr);
                        atomicAdd(h + (size_t)k * NSLOT * 2 + 1, (unsigned long long)vi);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) atomicAdd(h + (size_t)k * NSLOT * 2, (unsigned long long)gt_fixed(pai.x * g * w[k]));
                }
            }
        }
        pcount = 0;
    }

    // Stage 1.  Scan: one scatterer per lane and load, the (receive, transmit) sub-aperture pairs in the outer loop; two coalesced table reads and a window
    // test on r1 + r2 (a hair wider than the block).  Entries that pass -- about a third of the listed chunks' -- go to the WAVE's queue in LDS (ballot + prefix:
    // no barrier, no atomic); whenever 64 are queued the wave works them off with every lane busy: exact slot, amplitude, weights, fixed point, K adds.
    __device__ void deposit() {
        uint32_t *clist = (uint32_t *)(gsm + P.x_off);               // [LCAP + 16] chunks of 256 scatterers that can reach this block, then the count (behind everything the convolution reuses)
        uint32_t *ccount = clist + LCAP + 16;
        // r1 + r2 of the entries whose slot can lie in [0, NSLOT): ceil(Q d) - cbase in [0, NSLOT) <=> Q d in (cbase - 1, cbase + NSLOT - 1], d = (cinv r + toff) fs
        // (host: path_per_fine = 1 / (Q fs cinv), path_off = (t0 - s0) / cinv -- their roundings are far inside the margin)
        const double rlo_d = ((double)cbase - 1.0) * P.path_per_fine - P.path_off, rhi_d = ((double)cbase + (double)NSLOT - 1.0) * P.path_per_fine - P.path_off;
        // (fp32 roundings of the delay (cinv r + toff) fs: ~3 ulp of r -- and of the offset, expressed as a path length, when |toff| is the larger term)
        const double mar = 4e-6 * (fabs(rlo_d) + fabs(rhi_d) + 2.0 * fabs(P.path_off)) + 1e-30;
        const float rlo = (float)(rlo_d - mar), rhi = (float)(rhi_d + mar);
        const uint64_t I = P.I;
        const uint32_t I32 = (uint32_t)I, nchunk = P.nchunk;         // (I < 2^32: host)
        const int EE = P.En * P.Em;
        for (int sub = 0; sub < EE; ++sub) {
            const int ne = sub % P.En, me = sub / P.En;
            const size_t row1 = (size_t)n + (size_t)ne * P.N, row2 = (size_t)m + (size_t)me * P.M;
            const float *R1 = P.r1tab + row1 * I, *R2 = P.r2tab + row2 * I;
            const float2 *B1 = (const float2 *)P.cb1 + row1 * nchunk, *B2 = (const float2 *)P.cb2 + row2 * nchunk;
            for (uint32_t c0 = 0; c0 < nchunk; c0 += LCAP) {
                // the chunks whose distance sums [min1 + min2, max1 + max2] meet the window (fp32 addition is monotone: the chunk test can only
                // be wider than the entry test below), in any order
                if (tid == 0) *ccount = 0;
                __syncthreads();
                const uint32_t c1 = c0 + LCAP < nchunk ? c0 + LCAP : nchunk;
                for (uint32_t c = c0 + tid; c < c1; c += THREADS) {
                    const float2 b1 = B1[c], b2 = B2[c];
                    if (b1.x + b2.x <= rhi && b1.y + b2.y >= rlo) clist[atomicAdd(ccount, 1u)] = c;
                }
                __syncthreads();
                const uint32_t nact = *ccount;
                if (tid < 16u) clist[nact + tid] = 0xffffffffu;      // (whole passes below)
                __syncthreads();
                // 4 CGRP chunks per pass -- four per lane, eight table loads -- and the NEXT pass's loads are issued before this pass's entries are
                // tested: the table latency is paid once per list, not once per pass
                float r1n[4], r2n[4];
                uint32_t ivn[4];
                auto load_pass = [&](uint32_t e0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t c = clist[e0 + (uint32_t)q * CGRP + wave / (uint32_t)(GT_CHUNK / 64)];
                        const uint32_t i = c * (uint32_t)GT_CHUNK + (tid & (uint32_t)(GT_CHUNK - 1));
                        const bool in = c != 0xffffffffu && i < I32;
                        ivn[q] = i;
                        r1n[q] = in ? R1[i] : INFINITY;              // (an infinite distance: never in a block)
                        r2n[q] = in ? R2[i] : INFINITY;
                    }
                };
                if (nact) load_pass(0u);
                for (uint32_t e0 = 0; e0 < nact; e0 += 4u * CGRP) {
                    float r1v[4], r2v[4];
                    uint32_t iv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { r1v[q] = r1n[q]; r2v[q] = r2n[q]; iv[q] = ivn[q]; }
                    if (e0 + 4u * CGRP < nact) load_pass(e0 + 4u * CGRP);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float r = r1v[q] + r2v[q];
                        const bool hit = r >= rlo && r <= rhi;       // (infinite / NaN distances fail)
                        const uint64_t mask = __ballot(hit);
                        if (mask) {
                            if (hit) myq[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] =
                                         make_float4(__uint_as_float(iv[q]), r1v[q], r2v[q], 0.f);
                            qn += (uint32_t)__builtin_popcountll(mask);
                            __builtin_amdgcn_wave_barrier();
                            if (qn >= 64u) { retire(); qn -= 64u; take(qn, 64u); __builtin_amdgcn_wave_barrier(); }
                        }
                    }
                }
                __syncthreads();                                     // (the list is rebuilt)
            }
        }
        retire();
        take(0u, qn);
        retire();
    }

    // Stage 2.  fixed point -> float2, DE-INTERLEAVED by phase: with t = Q j + p the sum over the taps is
    //     y[s] = sum_k sum_p sum_j A_kp[s - j] x[Q j + p + o_k],   A_kp[s'] = H_k[Q s' - p]:
    // K Q plain FIR filters of ~T / Q taps on unit-stride sequences -- the lanes of a wave read consecutive elements (the interleaved trains had them
    // Q elements apart: a Q-way bank conflict on every read).  Every train goes through registers before the first float is written.
    __device__ float2 *convert(int &jmax_out, uint32_t &alp_out) {
        const float inv = Sc > 0.f ? 1.0f / Sc : 0.f;
        const int jmax = (int)(thi / (long)Q);
        const float rQ = 1.0f / (float)Q;
        const uint32_t AL = SB + (uint32_t)jmax, ALP = AL | 1u;      // (odd: the scatter below walks the phases, ALP elements apart)
        float2 *Af = (float2 *)gsm + 8;                              // [K][Q][ALP], 8 zero elements before it
        float2 v[K][4];                                              // NSLOT <= 4 * THREADS (host)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = tid + q * THREADS;
                v[k][q] = j < NSLOT ? make_float2(gt_float(H[((size_t)k * NSLOT + j) * 2]) * inv, gt_float(H[((size_t)k * NSLOT + j) * 2 + 1]) * inv) : make_float2(0.f, 0.f);
            }
        __syncthreads();
        if (tid < 8u) ((float2 *)gsm)[tid] = make_float2(0.f, 0.f);
        // positions no slot maps to -- the first and last element of a sequence (some phases) and the pad -- are zeroed, the others written: disjoint
        for (uint32_t e = tid; e < (uint32_t)(K * Q) * 3u; e += THREADS) {
            const uint32_t c = e / 3u, w = e - 3u * c, aa = w == 0 ? 0u : (w == 1 ? AL - 1u : ALP - 1u);
            const int ph = (int)(c % (uint32_t)Q), slot = Q * ((int)aa - jmax) - ph + (int)thi;
            if (aa >= AL || slot < 0 || slot >= (int)NSLOT) Af[(size_t)c * ALP + aa] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = tid + q * THREADS;
                if (j < NSLOT) {                                     // slot = Q (a - jmax) - p + thi
                    // (u >= -(Q - 1); floor(n / Q) = floor((n + 1/2) / Q), 1 / (2 Q) away from an integer: exact in fp32 for n < 2^20)
                    const int u = (int)j - (int)thi + Q * jmax, aq = (int)(((float)(u + Q - 1) + 0.5f) * rQ), ph = Q * aq - u;
                    Af[((size_t)k * Q + ph) * ALP + aq] = v[k][q];
                }
            }
        __syncthreads();
        jmax_out = jmax; alp_out = ALP;
        return Af;
    }

    // Stage 3.  The convolution: output s_lo + so, the groups of 8 taps (greens_xtab_kernel) dealt out to PARTS threads per output (a wave: one part, 64
    // outputs); per group three scalar loads (its offset, its taps), eight LDS reads of consecutive elements, sixteen packed multiply-adds.  A padded tap
    // multiplies whatever finite element precedes the sequence by zero (the 8 elements before the first sequence are zeros).
    __device__ void convolve(const float2 *Af) {
        const uint32_t PARTS = THREADS / SB, so = tid % SB;
        const int part = __builtin_amdgcn_readfirstlane((int)(tid / SB));            // (SB >= 64: uniform)
        v2f acc = {0.f, 0.f}, acc2 = {0.f, 0.f};
        const gt_ci grp = (gt_ci)P.segs;
        const gt_cf4 xt = (gt_cf4)P.xtab;
        const int G = grp[0];
        const int g0 = (int)((long)G * part / (long)PARTS), g1 = (int)((long)G * (part + 1) / (long)PARTS);
        const float2 *As = Af + so;
        for (int g = g0; g < g1; ++g) {
            const float2 *Ak = As + grp[1 + g];
            gt_f4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = xt[8 * (long)g + u];
            float2 h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = Ak[-u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc = (v2f){t[u].x, t[u].y} * h[u].x + acc; acc2 = (v2f){t[u].z, t[u].w} * h[u].y + acc2; }
        }
        acc += acc2;
        float2 *red = (float2 *)(gsm + P.pb_off);                    // [PARTS][SB]
        red[(size_t)part * SB + so] = make_float2(acc.x, acc.y);
        __syncthreads();
        if (part == 0) {
            float2 sum = make_float2(acc.x, acc.y);
            for (uint32_t p2 = 1; p2 < PARTS; ++p2) { const float2 r = red[p2 * SB + so]; sum.x += r.x; sum.y += r.y; }
            const uint64_t s = s_lo + so;
            if (!(bound <= 3.0e38f)) sum = make_float2(NAN, NAN);
            if (s < P.S) ((float2 *)P.y)[((size_t)n + (size_t)m * P.N) * P.S + s] = sum;
        }
    }
};

template <int INTERP, int TH>
__global__ void __launch_bounds__(TH) greens_train_kernel(const GreensParams P, const unsigned int *bound_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    // XCD-aware order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2), so workgroup g runs as item
    // (g % 8) * G8 / 8 + g / 8 of a list that walks the blocks of a trace, then the receivers, then the transmits: the workgroups an XCD runs at
    // a time share their distance rows (2 x 4 I bytes per trace) in ITS L2 -- in launch order every XCD saw every receiver's row
    const uint32_t nblk = P.nblk;
    const uint64_t G8 = gridDim.x, item = (uint64_t)(blockIdx.x & 7u) * (G8 >> 3) + (blockIdx.x >> 3);
    if (item >= (uint64_t)nblk * P.N * P.M) return;
    TrainBlock<INTERP, TH> B(P, gsm, (uint32_t)(item % nblk), (uint32_t)(item / nblk), bound_bits);
    for (uint32_t k = B.tid; k < (uint32_t)B.K * B.NSLOT * 2; k += TH) B.H[k] = 0;
    __syncthreads();
    if (P.dbg & 4) return;                               // (QDAS_GREENS_DBG: stage timing, tools/greens_time.py --stages)
    if (B.bound <= 3.0e38f) B.deposit();                 // (a non-finite amplitude: the trains cannot carry it -- the block is written as NaN)
    __syncthreads();
    if (P.dbg & 8) return;
    int jmax;
    uint32_t alp;
    const float2 *Af = B.convert(jmax, alp);
    if (P.dbg & 16) return;
    B.convolve(Af);
}

// 0: launched; 1: not this path (the kernel above runs)
static int launch_greens_train(const GreensParams &P, hipStream_t s) {
    static const bool off = getenv("QDAS_GREENS_NO_TRAINS") != nullptr;
    if (off) return 1;
    // (measured crossover with the direct kernel, whose time grows with the scatterers while this one's is mostly fixed -- zeroing, conversion, convolution --:
    //  ~2600 entries on a 256 x 256 x 2816 record (18.9 / 20.4 / 21.9 ms at 1000 / 2000 / 3000 against 8.2 / 16.2 / 23.9), ~2000 on 64 x 64 x 1955)
    uint64_t min_entries = 2048;
    if (const char *e = getenv("QDAS_GREENS_TRAIN_MIN")) { const long long v = atoll(e); if (v >= 0) min_entries = (uint64_t)v; }
    const double q = floor(P.fsr + 0.5);
    if (q != P.fsr || q < 1 || q > 16 || !(P.cinv > 0) || !(P.fs > 0) || P.I * (uint64_t)(P.En * P.Em) < min_entries || P.T >= (1u << 20)) return 1;
    const int K = P.interp == 0 ? 2 : interp_taps(P.interp);
    const long T = (long)P.T, tlo = K == 4 ? 1 : 0, thi = K == 4 ? T - 3 : (P.interp == 0 ? T - 1 : T - 2);
    if (thi < tlo) return 1;                                         // (a waveform shorter than the interpolator: nothing is ever in support)
    GreensParams p = P;
    p.q = (int)q;
    // (threads, outputs per block): 256 threads with 64 (else 128) outputs if the workgroup stays under 48 KB of LDS (three or more per CU);
    // else 1024 threads with the longest block that fits at all
    size_t lds = 0;
    int th = 0;
    static const struct { int th; uint32_t sb; size_t cap; } shapes[] = {{256, 64, 48 * 1024}, {256, 128, 48 * 1024}, {1024, 512, 150 * 1024}, {1024, 256, 150 * 1024},
                                                                         {1024, 128, 150 * 1024}, {1024, 64, 150 * 1024}};
    int only_th = 0;
    uint32_t only_sb = 0;
    if (const char *e = getenv("QDAS_GREENS_THREADS")) only_th = atoi(e);                 // (experiments)
    if (const char *e = getenv("QDAS_GREENS_SB")) only_sb = (uint32_t)atoi(e);
    for (const auto &sh : shapes) {
        if ((only_th && sh.th != only_th) || (only_sb && sh.sb != only_sb)) continue;
        const uint32_t sb = sh.sb;
        const uint64_t nslot = (uint64_t)q * (sb - 1) + (uint64_t)(thi - tlo) + 1;
        // scan: trains (fixed point) | wave queues ... waveform | chunk list | segments;  convolution: trains (float, de-interleaved) | part sums ... waveform ...
        const uint64_t alp = (sb + (uint64_t)thi / (uint64_t)q) | 1u;
        const size_t scan_end = (size_t)K * nslot * 16 + (size_t)(sh.th / 64) * GT_WQ * 16, pb = ((size_t)K * (size_t)q * alp * 8 + 64 + 15) / 16 * 16;
        const size_t conv_end = pb + (size_t)sh.th * 8;
        const size_t xo = scan_end > conv_end ? scan_end : conv_end;
        lds = xo + (size_t)(2 * sh.th + 20) * 4;
        if (nslot <= 4ull * sh.th && (uint64_t)q * alp <= 2 * nslot && K * (int)q <= 64 && lds <= sh.cap) { p.sb = sb; th = sh.th; p.pb_off = (uint32_t)pb; p.x_off = (uint32_t)xo; break; }
    }
    if (!p.sb) return 1;
    const uint64_t ne_tot = P.N * (uint64_t)P.En, me_tot = P.M * (uint64_t)P.Em, I = P.I;
    if (ne_tot > 65535 || me_tot > 65535 || I + 4096ull >= (1ull << 32) || (ne_tot + me_tot) * I * 4 > (8ull << 30)) return 1;      // (distance tables: at most 8 GiB)
    const bool no_sort = getenv("QDAS_GREENS_NO_SORT") != nullptr;      // (read per call: the tests switch it)
    const bool sorted = !no_sort && I >= 4096;           // (fewer: a handful of chunks, nothing to skip)
    const uint64_t nchunk = (I + GT_CHUNK - 1) / GT_CHUNK;
    // one allocation: bound | bounding box | distance tables | chunk bounds | sorted positions, amplitudes | cell keys | cell counts | segments | taps
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_tab = 256, o_cb = o_tab + up(sizeof(float) * (ne_tot + me_tot) * I), o_ps = o_cb + up(sizeof(float2) * (ne_tot + me_tot) * nchunk),
                 o_a = o_ps + (sorted ? up(12 * I) : 0), o_key = o_a + (sorted ? up(8 * I) : 0), o_hist = o_key + (sorted ? up(4 * I) : 0), o_seg = o_hist + (sorted ? up(4 * (size_t)GT_CELLS) : 0),
                 o_xt = o_seg + up(sizeof(int) * ((size_t)K * (size_t)T / 8 + (size_t)K * (size_t)q + 8)), total = o_xt + up(sizeof(float4) * ((size_t)K * (size_t)T + 8 * (size_t)K * (size_t)q + 8));
    Scratch scratch(s);                                  // (kept arena of this stream, or -- beyond 64 MiB -- a block that lives until this call returns: scratch.hip)
    unsigned char *buf = (unsigned char *)scratch.get(total);
    if (!buf) return 1;
    unsigned int *bound = (unsigned int *)buf;
    float *tabs = (float *)(buf + o_tab);
    float2 *cb = (float2 *)(buf + o_cb);
    auto bail = [&]() { (void)hipGetLastError(); return 1; };
    // bound = 0; any complex amplitude = 0; bounding box {min x y z = all ones, max x y z = 0} (ordered bits)
    if (hipMemsetAsync(buf, 0, 32, s) != hipSuccess || hipMemsetAsync(buf + 8, 0xff, 12, s) != hipSuccess) return bail();
    const float *ps = (const float *)P.Ps;
    const unsigned gI = (unsigned)((I + 255) / 256);
    if (sorted) {
        uint32_t *bb = (uint32_t *)buf + 2, *key = (uint32_t *)(buf + o_key), *hist = (uint32_t *)(buf + o_hist);
        float *pso = (float *)(buf + o_ps);
        float2 *ao = (float2 *)(buf + o_a);
        const int bits = gt_cell_bits(I);
        if (hipMemsetAsync(hist, 0, 4 * ((size_t)1 << (3 * bits)), s) != hipSuccess) return bail();
        greens_bbox_kernel<<<gI, 256, 0, s>>>(ps, I, bb);
        greens_key_kernel<<<gI, 256, 0, s>>>(ps, I, bb, bits, key, hist);
        greens_cellscan_kernel<<<1, 1024, 0, s>>>(hist, 1u << (3 * bits));
        greens_gather_kernel<<<gI, 256, 0, s>>>(ps, (const float2 *)P.a, key, hist, I, pso, ao);
        ps = pso; p.a = ao;
    }
    p.r1tab = tabs; p.r2tab = tabs + ne_tot * I;
    p.cb1 = (const float *)cb; p.cb2 = (const float *)(cb + ne_tot * nchunk);
    p.nchunk = (uint32_t)nchunk;
    p.path_per_fine = 1.0 / (q * P.fs * P.cinv); p.path_off = (P.t0 - P.s0) / P.cinv;
    p.dbg = getenv("QDAS_GREENS_DBG") ? atoi(getenv("QDAS_GREENS_DBG")) : 0;
    greens_dist_kernel<<<dim3((unsigned)nchunk, (unsigned)ne_tot), GT_CHUNK, 0, s>>>(ps, (const float *)P.Pr, tabs, cb, I);
    greens_dist_kernel<<<dim3((unsigned)nchunk, (unsigned)me_tot), GT_CHUNK, 0, s>>>(ps, (const float *)P.Pv, tabs + ne_tot * I, cb + ne_tot * nchunk, I);
    greens_bound_kernel<<<gI, 256, 0, s>>>(P, bound);
    p.segs = buf + o_seg; p.xtab = buf + o_xt;
    {
        const int jmax = (int)(thi / (long)q), alp = (int)((p.sb + (uint32_t)jmax) | 1u);                 // (as the kernel lays the trains out)
        greens_xtab_kernel<<<(unsigned)(((size_t)K * (size_t)T + 8 * (size_t)K * (size_t)q + 255) / 256), 256, 0, s>>>((const float2 *)P.x, K, (int)q, K == 4 ? -1 : 0, (int)tlo, (int)thi,
                                                                                                                     (int)(K == 4 ? T - 3 : T - 2), alp, jmax, (int *)(buf + o_seg), (float4 *)(buf + o_xt));
    }
    p.nblk = (uint32_t)((P.S + p.sb - 1) / p.sb);
    const uint64_t items = (uint64_t)p.nblk * P.N * P.M, g8 = (items + 7) / 8 * 8;
    if (g8 >= (1ull << 31)) return bail();
    const dim3 g((unsigned)g8), b((unsigned)th);
    hipError_t err = hipSuccess;
#define QT(I)                                                                                       \
    do {                                                                                            \
        auto kfn = th == 256 ? greens_train_kernel<I, 256> : greens_train_kernel<I, 1024>;          \
        err = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (err == hipSuccess) kfn<<<g, b, lds, s>>>(p, bound);                                     \
    } while (0)
    switch (P.interp) {
        case 0: QT(0); break;
        case 1: case 4: QT(1); break;
        case 2: QT(2); break;
        case 3: QT(3); break;
        case 5: QT(5); break;
        default: err = hipErrorInvalidValue;
    }
#undef QT
    return (err == hipSuccess && hipGetLastError() == hipSuccess) ? 0 : 1;
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
        case 1: if (launch_greens_train(P, s) == 0) return hipSuccess; return launch_greens_t<st_f32>(P, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
