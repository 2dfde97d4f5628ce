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
constexpr int GT_THREADS = 1024;

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
}

// element-to-scatterer distances, ONCE per launch: the delay is separable, r1 depends on (scatterer, receive element) and r2 on (scatterer, transmit
// element) only -- (N En + M Em) I square roots instead of the N En M Em I x 2 (x blocks per trace) of a scan that recomputes them
__global__ void __launch_bounds__(256) greens_dist_kernel(const float *__restrict__ Ps, const float *__restrict__ Pe, float *__restrict__ R, uint64_t I) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    const size_t k = 3 * (size_t)blockIdx.y;
    const float ax = Ps[3 * i] - Pe[k], ay = Ps[3 * i + 1] - Pe[k + 1], az = Ps[3 * i + 2] - Pe[k + 2];
    R[(size_t)blockIdx.y * I + i] = sqrtf(ax * ax + ay * ay + az * az);          // src/greens.cu:61-62
}

// |t| < 2^53 float -> the nearest 64-bit integer, without the generic conversion sequence: t = hi 2^23 + lo with hi = rint(t 2^-23) (|hi| < 2^30) and
// lo = t - hi 2^23 exactly (one FMA; |lo| <= 2^22), both through v_cvt_i32_f32
__device__ __forceinline__ long long gt_fixed(float t) {
    const float hi = rintf(t * 1.1920928955078125e-7f);
    const float lo = fmaf(hi, -8388608.0f, t);
    return ((long long)(int)hi << 23) + (long long)(int)rintf(lo);
}

// ... and back: (float)v without the generic 64-bit conversion sequence (two 32-bit conversions and an FMA; the sum is rounded once more, 2^-24 relative)
__device__ __forceinline__ float gt_float(long long v) {
    return fmaf((float)(int)(v >> 32), 4294967296.0f, (float)(unsigned int)(v & 0xffffffffll));
}

constexpr int GT_WQ = 128;                               // entries a wave queues before it works 64 of them off

template <int INTERP>
__global__ void __launch_bounds__(GT_THREADS) greens_train_kernel(const GreensParams P, const unsigned int *bound_bits) {
    constexpr int K = INTERP == 0 ? 2 : interp_taps(INTERP);         // nearest: two trains (the sample at ti, or at ti + 1)
    constexpr int O0 = K == 4 ? -1 : 0;                              // tap offset of train 0 (train k: O0 + k)
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const long T = (long)P.T;
    const int Q = P.q;
    const uint32_t SB = P.sb, PARTS = GT_THREADS / SB;
    // tap indices ti the edge rule admits (qdas_device.h sample_global): 4 taps: 1 <= ti <= T-3; linear: 0 <= ti <= T-2; nearest: ti >= 0 and ti (+1) < T
    const long tlo = K == 4 ? 1 : 0;
    const long thi_k0 = K == 4 ? T - 3 : (INTERP == 0 ? T - 1 : T - 2), thi_k1 = K == 4 ? T - 3 : T - 2;      // (trains 2, 3 as train 1)
    const long thi = thi_k0;
    const uint32_t NSLOT = (uint32_t)((long)Q * (SB - 1) + thi - tlo + 1);
    long long *H = (long long *)gsm;                                 // [K][NSLOT] {re, im} fixed point, later float2 in place
    float4 *wq = (float4 *)(H + (size_t)K * NSLOT * 2);              // [waves][GT_WQ] {scatterer, r1, r2}: this wave's entries that (may) land in the block
    float2 *red = (float2 *)(wq + (GT_THREADS / 64) * GT_WQ);        // [PARTS][SB]
    float2 *xl = red + GT_THREADS;                                   // [T]
    // XCD-aware order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2), so workgroup g runs as item
    // (g % 8) * G8 / 8 + g / 8 of a list that walks the blocks of a trace, then the receivers, then the transmits: the workgroups an XCD runs at
    // a time share their distance rows (2 x 4 I bytes per trace) in ITS L2 -- in launch order every XCD saw every receiver's row
    // (100 000 scatterers on a 256 x 256 x 2816 acquisition: 126 -> 104 ms)
    const uint32_t nblk = P.nblk;
    const uint64_t G8 = gridDim.x, item = (uint64_t)(blockIdx.x & 7u) * (G8 >> 3) + (blockIdx.x >> 3);
    if (item >= (uint64_t)nblk * P.N * P.M) return;
    const uint32_t blk = (uint32_t)(item % nblk), trace = (uint32_t)(item / nblk);
    const uint32_t n = trace % (uint32_t)P.N, m = trace / (uint32_t)P.N, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const float2 *a = (const float2 *)P.a, *x = (const float2 *)P.x;
    for (uint32_t k = tid; k < (uint32_t)K * NSLOT * 2; k += GT_THREADS) H[k] = 0;
    for (long k = tid; k < T; k += GT_THREADS) xl[k] = x[k];
    const float bound = __uint_as_float(*bound_bits);
    // scale: 2^46 / largest single contribution -- and never more than 2^62 / (entries of the trace x largest contribution): a slot can at most receive
    // every entry of its trace, so the 64-bit sums cannot wrap however dense the scatterer cloud (beyond 2^16 entries per trace the resolution drops from
    // 2^-46 to 2^-62 x entries of the largest contribution: still 2^-32 at 10^9 entries, far below the fp32 rounding of the result)
    const float ent = (float)P.I * (float)(P.En * P.Em);
    const float Sc = bound > 0.f ? fminf(fminf(70368744177664.0f, 4.611686018427388e18f / fmaxf(ent, 1.f)) / bound, 1.0e37f) : 0.f;
    const float fs = (float)P.fs, fsr = (float)P.fsr, cinv = (float)P.cinv, R0 = (float)P.R0, toff = (float)(P.t0 - P.s0);
    const int EE = P.En * P.Em;
    const uint64_t s_lo = (uint64_t)blk * SB, I = P.I;
    const long cbase = (long)Q * (long)s_lo - thi;                   // fine index of slot 0
    __syncthreads();
    if (bound <= 3.0e38f) {
        // Scan: one scatterer per lane and pass, the (receive, transmit) sub-aperture pairs in the outer loop; two coalesced table reads, the delay,
        // the slot test.  Entries that land in this block -- 1 / (blocks per trace) of them -- go to the WAVE's queue in LDS (ballot + prefix: no
        // barrier, no atomic); whenever 64 are queued the wave works them off with every lane busy: amplitude, weights, fixed point, K complex adds.
        float4 *myq = wq + (size_t)wave * GT_WQ;
        uint32_t qn = 0;                                             // queued entries (uniform)
        // (the scan's test is a window on r1 + r2, a hair wider than the block: the exact slot -- the reference's own fp32 delay arithmetic -- is found here)
        auto work_off = [&](uint32_t first, uint32_t count) {
            if (lane < count) {
                const float4 it = myq[first + lane];
                float r1 = it.y, r2 = it.z;
                const float d = (cinv * (r1 + r2) + toff) * fs;                                               // src/greens.cu:65
                const float ef = (float)Q * d, cf = ceilf(ef);
                const float sl = cf - (float)cbase;                  // slot of the entry; a non-finite delay fails the test
                if (sl >= 0.f && sl < (float)NSLOT) {
                    const uint32_t slot = (uint32_t)sl;
                    const float u = cf - ef;                         // in [0, 1): exact
                    if (R0 != 0.f) { r1 = r1 < R0 ? R0 : r1; r2 = r2 < R0 ? R0 : r2; } else { r1 = 1.f; r2 = 1.f; }
                    const float2 ai = a[__float_as_uint(it.x)];
                    const float g = Sc / (r1 * r2 * fsr);
                    float w[4];
                    if constexpr (INTERP == 0) { w[0] = u < 0.5f ? 1.f : 0.f; w[1] = 1.f - w[0]; }
                    else interp_weights<INTERP>(u, w);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const long long vr = gt_fixed(ai.x * g * w[k]), vi = gt_fixed(ai.y * g * w[k]);
                        unsigned long long *h = (unsigned long long *)(H + ((size_t)k * NSLOT + slot) * 2);
                        if (vr) atomicAdd(h, (unsigned long long)vr);
                        if (vi) atomicAdd(h + 1, (unsigned long long)vi);
                    }
                }
            }
        };
        // r1 + r2 of the entries whose slot can lie in [0, NSLOT): ceil(Q d) - cbase in [0, NSLOT) <=> Q d in (cbase - 1, cbase + NSLOT - 1], d = (cinv r + toff) fs
        const double rlo_d = (((double)cbase - 1.0) / ((double)Q * P.fs) - (P.t0 - P.s0)) / P.cinv, rhi_d = (((double)cbase + (double)NSLOT - 1.0) / ((double)Q * P.fs) - (P.t0 - P.s0)) / P.cinv;
        // (fp32 roundings of the delay (cinv r + toff) fs: ~3 ulp of r -- and of the offset, expressed as a path length, when |toff| is the larger term)
        const double mar = 4e-6 * (fabs(rlo_d) + fabs(rhi_d) + 2.0 * fabs(P.t0 - P.s0) / P.cinv) + 1e-30;
        const float rlo = (float)(rlo_d - mar), rhi = (float)(rhi_d + mar);
        for (int sub = 0; sub < EE; ++sub) {
            const int ne = sub % P.En, me = sub / P.En;
            const float *R1 = P.r1tab + ((size_t)n + (size_t)ne * P.N) * I, *R2 = P.r2tab + ((size_t)m + (size_t)me * P.M) * I;
            // (four scatterers per lane and iteration: the eight table loads are in flight together -- one exposed L2 latency per 4096 entries, not per 1024)
            const uint32_t I32 = (uint32_t)I;                        // (I < 2^32: host)
            for (uint32_t i0 = 0; i0 < I32; i0 += 4 * GT_THREADS) {
                float r1v[4], r2v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t i = i0 + tid + (uint32_t)q * GT_THREADS;
                    r1v[q] = i < I32 ? R1[i] : INFINITY;             // (an infinite distance: never in a block)
                    r2v[q] = i < I32 ? R2[i] : INFINITY;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float r = r1v[q] + r2v[q];
                    const bool hit = r >= rlo && r <= rhi;           // (infinite / NaN distances fail)
                    const uint64_t mask = __ballot(hit);
                    if (mask) {
                        if (hit) myq[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] =
                                     make_float4(__uint_as_float(i0 + tid + (uint32_t)q * GT_THREADS), r1v[q], r2v[q], 0.f);
                        qn += (uint32_t)__builtin_popcountll(mask);
                        __builtin_amdgcn_wave_barrier();
                        if (qn >= 64u) { qn -= 64u; work_off(qn, 64u); __builtin_amdgcn_wave_barrier(); }
                    }
                }
            }
        }
        work_off(0u, qn);
    }
    __syncthreads();
    // fixed point -> float2, in place, train by train (the float image of train k lies inside the fixed-point images of trains <= k)
    const float inv = Sc > 0.f ? 1.0f / Sc : 0.f;
    float2 *Hf = (float2 *)gsm;
    for (int k = 0; k < K; ++k) {
        float2 v[4];                                                 // NSLOT <= 4 * GT_THREADS (host)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid + q * GT_THREADS;
            v[q] = j < NSLOT ? make_float2(gt_float(H[((size_t)k * NSLOT + j) * 2]) * inv, gt_float(H[((size_t)k * NSLOT + j) * 2 + 1]) * inv) : make_float2(0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t j = tid + q * GT_THREADS; if (j < NSLOT) Hf[(size_t)k * NSLOT + j] = v[q]; }
        __syncthreads();
    }
    // the convolution: output s_lo + sl, the (train, tap index) pairs dealt out to PARTS threads per output
    const uint32_t so = tid % SB, part = tid / SB;
    float2 acc = make_float2(0.f, 0.f);
    {
        const long len0 = thi_k0 - tlo + 1, len1 = thi_k1 - tlo + 1;
        const long J = len0 + (K - 1) * len1;                        // pairs per output
        long j0 = J * (long)part / (long)PARTS, j1 = J * (long)(part + 1) / (long)PARTS;
        for (int k = 0; k < K; ++k) {
            const long lenk = k == 0 ? len0 : len1, base = k == 0 ? 0 : len0 + (k - 1) * len1;
            const long a0 = j0 > base ? j0 - base : 0, a1 = (j1 - base) < lenk ? (j1 - base) : lenk;
            const float2 *hk = Hf + (size_t)k * NSLOT + (size_t)Q * so + (size_t)(thi - tlo);    // slot of t = tlo; t + 1 is one slot down
            for (long q = a0; q < a1; ++q) {
                const float2 h = hk[-q], xv = xl[tlo + q + O0 + k];
                acc.x += h.x * xv.x - h.y * xv.y; acc.y += h.x * xv.y + h.y * xv.x;
            }
        }
    }
    red[part * SB + so] = acc;
    __syncthreads();
    if (part == 0) {
        for (uint32_t p2 = 1; p2 < PARTS; ++p2) { const float2 r = red[p2 * SB + so]; acc.x += r.x; acc.y += r.y; }
        const uint64_t s = s_lo + so;
        if (!(bound <= 3.0e38f)) acc = make_float2(NAN, NAN);
        if (s < P.S) ((float2 *)P.y)[((size_t)n + (size_t)m * P.N) * P.S + s] = acc;
    }
}

// 0: launched; 1: not this path (the kernel above runs)
static int launch_greens_train(const GreensParams &P, hipStream_t s) {
    static const bool off = getenv("QDAS_GREENS_NO_TRAINS") != nullptr;
    if (off) return 1;
    uint64_t min_entries = 1024;
    if (const char *e = getenv("QDAS_GREENS_TRAIN_MIN")) { const long long v = atoll(e); if (v >= 0) min_entries = (uint64_t)v; }
    const double q = floor(P.fsr + 0.5);
    if (q != P.fsr || q < 1 || q > 16 || !(P.cinv > 0) || !(P.fs > 0) || P.I * (uint64_t)(P.En * P.Em) < min_entries || P.T >= (1u << 20)) return 1;
    const int K = P.interp == 0 ? 2 : interp_taps(P.interp);
    const long T = (long)P.T, tlo = K == 4 ? 1 : 0, thi = K == 4 ? T - 3 : (P.interp == 0 ? T - 1 : T - 2);
    if (thi < tlo) return 1;                                         // (a waveform shorter than the interpolator: nothing is ever in support)
    GreensParams p = P;
    p.q = (int)q;
    size_t lds = 0;
    for (uint32_t sb : {512u, 256u, 128u, 64u}) {
        const uint64_t nslot = (uint64_t)q * (sb - 1) + (uint64_t)(thi - tlo) + 1;
        lds = (size_t)K * nslot * 16 + (size_t)(GT_THREADS / 64) * GT_WQ * 16 + (size_t)GT_THREADS * 8 + (size_t)T * 8;
        if (nslot <= 4ull * GT_THREADS && lds <= 150 * 1024) { p.sb = sb; break; }
    }
    if (!p.sb) return 1;
    const uint64_t ne_tot = P.N * (uint64_t)P.En, me_tot = P.M * (uint64_t)P.Em;
    if (ne_tot > 65535 || me_tot > 65535 || P.I + 4ull * GT_THREADS >= (1ull << 32) || (ne_tot + me_tot) * P.I * 4 > (8ull << 30)) return 1;      // (distance tables: at most 8 GiB)
    unsigned int *bound = nullptr;
    float *tabs = nullptr;
    if (hipMallocAsync((void **)&bound, 16 + sizeof(float) * (ne_tot + me_tot) * P.I, s) != hipSuccess || !bound) { (void)hipGetLastError(); return 1; }
    if (hipMemsetAsync(bound, 0, sizeof(unsigned int), s) != hipSuccess) { (void)hipGetLastError(); (void)hipFreeAsync(bound, s); return 1; }
    tabs = (float *)((char *)bound + 16);
    p.r1tab = tabs; p.r2tab = tabs + ne_tot * P.I;
    greens_dist_kernel<<<dim3((unsigned)((P.I + 255) / 256), (unsigned)ne_tot), 256, 0, s>>>((const float *)P.Ps, (const float *)P.Pr, tabs, P.I);
    greens_dist_kernel<<<dim3((unsigned)((P.I + 255) / 256), (unsigned)me_tot), 256, 0, s>>>((const float *)P.Ps, (const float *)P.Pv, tabs + ne_tot * P.I, P.I);
    greens_bound_kernel<<<(unsigned)((P.I + 255) / 256), 256, 0, s>>>(p, bound);
    p.nblk = (uint32_t)((P.S + p.sb - 1) / p.sb);
    const uint64_t items = (uint64_t)p.nblk * P.N * P.M, g8 = (items + 7) / 8 * 8;
    if (g8 >= (1ull << 31)) { (void)hipFreeAsync(bound, s); return 1; }
    const dim3 g((unsigned)g8), b(GT_THREADS);
    hipError_t err = hipSuccess;
#define QT(I)                                                                                       \
    do {                                                                                            \
        auto kfn = greens_train_kernel<I>;                                                          \
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
    (void)hipFreeAsync(bound, s);
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
