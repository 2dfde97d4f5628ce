// shiftsum.hip -- transmit synthesis (UltrasoundSystem.focusTx): delay-and-sum over the TRANSMIT ELEMENTS of a
// full-synthetic-aperture record,
//
//   y[t', n, m'] = sum_m  w[m, m'] * x(t' + s[m, m'],  n, m)          t' = 0 .. To-1,  s in samples (any real number; M x Mo, m fastest)
//
// which is what the reference computes with `sample2sep(chd.time, -tau, interp, apd, mdim)` (src/UltrasoundSystem.m:3498 ->
// kern/wsinterpd2.m -> src/interpd.cu:344-396): the sampling positions are the record's own time grid plus ONE offset per (element, synthesised
// transmit).  The general single-delay kernel (wsinterpd.hip) serves that call too -- one lane per output, every term a gather from global memory
// with its own index and weight evaluation (C1: 1.56 ms).  Here the structure is used:
//   * the offset does not depend on t': tap offset floor(s), the K interpolation weights and the range of valid t' are properties of (m, m') --
//     a table of M x Mo entries made by a small kernel (the library's own interp_weights: same numbers as every other kernel; a real weight is
//     folded into them); the main kernel copies an element's entries into LDS together with its window (a chain of dependent scalar loads per
//     entry was the first version's bottleneck);
//   * a workgroup owns 256 x TPT consecutive output samples of one receiver and MOB synthesised transmits; per element m it stages ONE window
//     of the trace x[:, n, m] in LDS (the samples those outputs can reach; prefetched into registers while the previous element is worked on)
//     and every lane reads its K consecutive taps from there: consecutive lanes, consecutive addresses, packed FMAs;
//   * zero weights (the apodization of a walking aperture is mostly zeros) are uniform skips; an element with no weight in the block is not staged.
// Edge rule as everywhere (src/interpd.cu:70-150): a term counts iff all its taps lie in [0, T) and the position is >= 0.
#include "qdas_device.h"
#include "qdas_kernels.h"
#include <type_traits>
#include <stdlib.h>

namespace qdas {

typedef float v2f __attribute__((ext_vector_type(2)));

template <typename R> struct ShiftEntry {    // per (m, m'): 16 + 6 R bytes
    int32_t k0;          // first tap = t' + k0
    int32_t tlo, thi;    // the term is in support for tlo <= t' <= thi
    int32_t on;          // weight != 0 and the range is not empty
    R wr, wi;            // w[m, m']
    R c[4];              // interpolation weights of the K taps
};

template <int INTERP, typename R>
__global__ void __launch_bounds__(256) shift_table_kernel(const R *__restrict__ sh, const void *__restrict__ w, int w_real, uint64_t count, int64_t T, int64_t To,
                                                          ShiftEntry<R> *__restrict__ tab) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    constexpr int K = interp_taps(INTERP);
    constexpr int OFF = (K == 4) ? -1 : 0;
    const R s = sh[i];
    ShiftEntry<R> e;
    e.wr = (R)1; e.wi = (R)0;
    if (w) { if (w_real) e.wr = ((const R *)w)[i]; else { e.wr = ((const R *)w)[2 * i]; e.wi = ((const R *)w)[2 * i + 1]; } }
    e.c[0] = (R)1; e.c[1] = e.c[2] = e.c[3] = (R)0;
    int64_t lo = 0, hi = -1, k0 = 0;
    if (s == s && s > (R)-2.0e9 && s < (R)2.0e9) {     // (a non-finite offset leaves the term out, like a NaN position does)
        if constexpr (K == 1) {                      // nearest: r = floor(t' + s + 1/2), in support iff t' + s >= 0 and r < T   (src/interpd.cu:70-72)
            const R f = qfloor(s + (R)0.5);
            k0 = (int64_t)f;
            lo = (int64_t)ceil((double)-s);           // t' >= -s
            hi = T - 1 - k0;
        } else {
            const R f = qfloor(s);
            interp_weights<INTERP, R>(s - f, e.c);
            k0 = (int64_t)f + OFF;
            lo = -k0;                                 // first tap >= 0 (then t' + s >= 0 too)
            hi = T - K - k0;                          // last tap < T
        }
        if (lo < 0) lo = 0;
        if (hi > To - 1) hi = To - 1;
    }
    const bool on = (e.wr != (R)0 || e.wi != (R)0) && lo <= hi;
    if (e.wi == (R)0) {                              // a real weight rides in the interpolation weights (the main kernel then adds the sample as it is)
        e.c[0] *= e.wr; e.c[1] *= e.wr; e.c[2] *= e.wr; e.c[3] *= e.wr;
        e.wr = (R)1;
    }
    e.k0 = (int32_t)k0; e.tlo = (int32_t)lo; e.thi = (int32_t)hi; e.on = on ? 1 : 0;
    tab[i] = e;
}

constexpr int SS_MOB = 8;        // synthesised transmits per workgroup
constexpr int SS_CAP = 4096;     // samples of one staged window

// per (block of SS_MOB synthesised transmits, element m): smallest / largest tap offset among the weighted entries (kmin > kmax: none) -- what a
// workgroup needs BEFORE it stages a window; one 8-byte scalar load per element instead of a scan over the block's entries
template <typename R>
__global__ void __launch_bounds__(256) shift_block_kernel(const ShiftEntry<R> *__restrict__ tab, uint64_t M, uint64_t Mo, uint32_t mo_blocks, int4 *__restrict__ blk) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * mo_blocks) return;
    const uint64_t m = i % M, mob = i / M;
    int kmin = 0x7fffffff, kmax = -0x7fffffff - 1;
    // .z / .w (round 6): outputs [z, w] lie in the support of EVERY term of the block, all of them weighted by a real number (z > w: no such range) --
    // a wave whose outputs lie inside takes the loop without masks (shift_sum_kernel)
    int lo = 0, hi = 0x7fffffff;
    bool plain = true;
    for (uint64_t mo = mob * SS_MOB; mo < Mo && mo < (mob + 1) * SS_MOB; ++mo) {
        const ShiftEntry<R> &e = tab[m + M * mo];
        if (e.on) { kmin = e.k0 < kmin ? e.k0 : kmin; kmax = e.k0 > kmax ? e.k0 : kmax; }
        if (e.on && e.wi == (R)0) { lo = e.tlo > lo ? e.tlo : lo; hi = e.thi < hi ? e.thi : hi; }
        else plain = false;
    }
    blk[i] = plain ? make_int4(kmin, kmax, lo, hi) : make_int4(kmin, kmax, 0x7fffffff, -2);    // (.w == -2: a zero or complex weight among them -- the general loop)
}

// DT: sample type (float2 / double2 / float / double), R its real type, TPT output samples per lane
// GW: outputs of a 64-lane group.  64, or 65 - K (fp32 complex data, round 6): the lanes of a group then read ONE sample each per term and pass it down the
// wave (v_mov_b32 wave_shl:1, K - 1 times) -- a lane's K taps are its own sample and those of the next K - 1 lanes; the last K - 1 lanes of a group only
// supply taps.  A quarter of the LDS bytes (the kernel is LDS-bound at K reads per output) for K - 1 lane moves per tap.
template <int K, typename DT, typename R, int TPT, int GW = 64>
__global__ void __launch_bounds__(256) shift_sum_kernel(const ShiftParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_lds[];
    DT *const win_base = (DT *)ss_lds;
    DT *win = win_base;                                                          // the window the terms read (class 0: one of two halves, see the element loop)
    constexpr int EW = (int)(sizeof(ShiftEntry<R>) / 4);                         // 32-bit words per table entry
    uint32_t *const ent_base = (uint32_t *)(ss_lds + sizeof(DT) * SS_CAP);       // [2][SS_MOB] this element's entries, copied next to the window
    uint32_t *ent_w = ent_base;
    const ShiftEntry<R> *ent = (const ShiftEntry<R> *)ent_w;
    constexpr bool CPLX = sizeof(DT) == 2 * sizeof(R);
    constexpr int TPB = 4 * GW * TPT;                                           // outputs of a workgroup (4 waves)
    static_assert(GW == 64 || (GW == 65 - K && std::is_same<DT, float2>::value && K > 1), "group width");
    const ShiftEntry<R> *__restrict__ tab = (const ShiftEntry<R> *)P.tab;
    const uint32_t tid = threadIdx.x;
    const int64_t tb = (int64_t)blockIdx.x * TPB;
    const uint64_t n = blockIdx.y;
    const uint32_t mob = blockIdx.z % P.mo_blocks;
    const uint64_t f = blockIdx.z / P.mo_blocks;
    const uint64_t mo0 = (uint64_t)mob * SS_MOB;
    const int nmo = (int)((P.Mo - mo0) < (uint64_t)SS_MOB ? (P.Mo - mo0) : (uint64_t)SS_MOB);
    const int64_t T = (int64_t)P.Tx;                                             // (stored samples: what a load may touch; the support rule is in the table)
    R ar[TPT][SS_MOB], ai[TPT][SS_MOB];
#pragma unroll
    for (int q = 0; q < TPT; ++q)
#pragma unroll
        for (int j = 0; j < SS_MOB; ++j) { ar[q][j] = (R)0; ai[q][j] = (R)0; }

    auto stage = [&](const DT *__restrict__ tr, int64_t wlo, int wlen, const ShiftEntry<R> *row) {   // win[i] = x[wlo + i] (zero outside the record)
        __syncthreads();                                                        // the previous window is consumed
        if (row && (int)tid < nmo * EW)                                          // + the element's table entries (one vector load instead of a scalar-load chain)
            ent_w[tid] = ((const uint32_t *)(row + (uint64_t)(tid / EW) * P.M))[tid % EW];
        for (int i = (int)tid; i < wlen; i += 256) {
            const int64_t g = wlo + i;
            DT v{};
            if (g >= 0 && g < T) v = tr[g];
            win[i] = v;
        }
        __syncthreads();
    };
    // a wave owns 64 x TPT CONSECUTIVE outputs of the block (lane + 64 q within them): whether a term's support covers them all is a per-wave question
    const int wv0 = __builtin_amdgcn_readfirstlane((int)(tid / 64) * (GW * TPT)), lt0 = wv0 + (int)(tid % 64);
    const bool act = GW == 64 || (int)(tid % 64) < GW;                            // (the other lanes of a group only supply taps)
    // all lanes, their TPT outputs, synthesised transmit j.  rel = (first tap of local output 0) - (window start): every lane's taps lie inside the
    // staged window whether or not its output is in support, so the reads need no guard; 32-bit index math throughout
    auto term = [&](const ShiftEntry<R> &e, int j, int rel) {
        const int64_t lo64 = (int64_t)e.tlo - tb, hi64 = (int64_t)e.thi - tb;
        const int lo = lo64 < 0 ? 0 : (int)lo64, hi = hi64 > TPB ? TPB : (int)hi64;            // uniform
        const bool plain = e.wi == (R)0;                                                        // real weight: already in c[]
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            const int lt = lt0 + GW * q;
            const bool ok = act && lt >= lo && lt <= hi;
            const DT *tp = win + (lt + rel);
            if constexpr (std::is_same<DT, float2>::value) {                                    // packed fp32: one v_pk_fma_f32 per tap
                v2f v = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < K; ++k) { const float2 s = tp[k]; v = (v2f){s.x, s.y} * e.c[k] + v; }
                if (!plain) v = (v2f){e.wr * v.x - e.wi * v.y, e.wr * v.y + e.wi * v.x};
                if (ok) { ar[q][j] += v.x; ai[q][j] += v.y; }
            } else {
                R vr = (R)0, vi = (R)0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const DT s = tp[k];
                    if constexpr (CPLX) { vr += e.c[k] * s.x; vi += e.c[k] * s.y; } else vr += e.c[k] * s;
                }
                if constexpr (CPLX) { if (!plain) { const R xr = e.wr * vr - e.wi * vi, xi = e.wr * vi + e.wi * vr; vr = xr; vi = xi; } }
                if (ok) { ar[q][j] += vr; if constexpr (CPLX) ai[q][j] += vi; }
            }
        }
    };

    // the same for a wave INSIDE the support of every term of the block (real weights): no masks, no weight select -- K FMAs per output straight into
    // the accumulator, every tap an immediate offset from one address per term (fp32 complex data; the table entries are read as in `term`)
    // MASK: the wave straddles an end of some term's support -- the same chain under the lanes' support mask (one unsigned compare per output)
    auto lean_term = [&](auto maskc, const ShiftEntry<R> &e, int j, int rel) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(maskc)::value;
        if constexpr (std::is_same<DT, float2>::value) {
            const DT *tp = win + (lt0 + rel);
            float2 sm[TPT][K];
#pragma unroll
            for (int q = 0; q < TPT; ++q) {
                if constexpr (GW == 64) {
#pragma unroll
                    for (int k = 0; k < K; ++k) sm[q][k] = tp[64 * q + k];
                } else {                                                          // one read; tap k = the sample of lane + k (all lanes active here: a lane move
                    sm[q][0] = tp[GW * q];                                        //  reads nothing from a lane that is masked off)
#pragma unroll
                    for (int k = 1; k < K; ++k) {
                        sm[q][k].x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm[q][k - 1].x), 0x130, 0xf, 0xf, true));
                        sm[q][k].y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm[q][k - 1].y), 0x130, 0xf, 0xf, true));
                    }
                }
            }
            if constexpr (!MASK) {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int q = 0; q < TPT; ++q) {
                        const v2f a = (v2f){sm[q][k].x, sm[q][k].y} * e.c[k] + (v2f){ar[q][j], ai[q][j]};
                        ar[q][j] = a.x; ai[q][j] = a.y;
                    }
            } else {
                const int lo = e.tlo - (int)tb;                                   // (the launcher keeps To below 2^30 on this path)
                const uint32_t len = (uint32_t)(e.thi - e.tlo);
#pragma unroll
                for (int q = 0; q < TPT; ++q) {
                    if (act && (uint32_t)(lt0 + GW * q - lo) <= len) {
                        v2f a = {ar[q][j], ai[q][j]};
#pragma unroll
                        for (int k = 0; k < K; ++k) a = (v2f){sm[q][k].x, sm[q][k].y} * e.c[k] + a;
                        ar[q][j] = a.x; ai[q][j] = a.y;
                    }
                }
            }
        }
    };

    // Element loop, software-pipelined: while the lanes work on element m out of LDS, the window (and the table entries) of the next weighted
    // element are already on their way into registers; they are written to LDS behind the barrier that ends m.  Class 0: the window fits the
    // prefetch registers (offsets of the block within ~500 samples of each other -- the usual case); class 1: one window, staged without
    // prefetch; class 2: offsets too far apart for one window -- a window per synthesised transmit.
    const int4 *__restrict__ blk = (const int4 *)P.blk + (uint64_t)mob * P.M;
    constexpr int NPF = TPT + 2;
    DT pre[NPF];
    uint32_t pre_e = 0;
    auto find = [&](uint64_t from, int &kmin, int &kmax, int &all_lo, int &all_hi) -> uint64_t {   // next element with a weight in this block (uniform scalar loads)
        for (uint64_t m = from; m < P.M; ++m) { const int4 kr = blk[m]; if (kr.x <= kr.y) { kmin = kr.x; kmax = kr.y; all_lo = kr.z; all_hi = kr.w; return m; } }
        return P.M;
    };
    auto klass = [&](int kmin, int kmax) -> int {
        const int64_t span = (int64_t)kmax - (int64_t)kmin;
        return span + TPB + K - 1 <= NPF * 256 ? 0 : (span + TPB + K <= SS_CAP ? 1 : 2);
    };
    auto trace = [&](uint64_t m) -> const DT * { return (const DT *)P.x + ((f * P.M + m) * P.N + n) * P.Tx; };
    auto fetch = [&](uint64_t m, int kmin, int wlen) {
        const DT *__restrict__ tr = trace(m);
        const int64_t wlo = tb + kmin;
#pragma unroll
        for (int p = 0; p < NPF; ++p) {
            const int i = (int)tid + 256 * p;
            const int64_t g = wlo + i;
            DT v{};
            if (i < wlen && g >= 0 && g < T) v = tr[g];
            pre[p] = v;
        }
        if ((int)tid < nmo * EW) pre_e = ((const uint32_t *)(tab + m + P.M * (mo0 + tid / EW)))[tid % EW];
    };
    static_assert(NPF * 256 <= SS_CAP / 2, "a class 0 window fits half of the staging area");
    int kmin = 0, kmax = -1, alo = 0, ahi = -1, cur = 0;
    bool big_prev = true;                                                        // (nothing to wait for before the first write; harmless)
    uint64_t m = find(0, kmin, kmax, alo, ahi);
    int kl = m < P.M ? klass(kmin, kmax) : 0;
    if (m < P.M && kl == 0) fetch(m, kmin, (int)((int64_t)kmax - kmin + TPB + K - 1));
    while (m < P.M) {
        const int wl = (int)((int64_t)kmax - (int64_t)kmin + TPB + K - 1);
        if (kl == 0) {
            // two windows (halves of the staging area) take turns: the one written now was last read two elements ago, and every wave has passed a
            // barrier since -- ONE barrier per element (round 6; the window of a class 1 / 2 element spans both halves: a barrier before reusing them)
            if (big_prev) __syncthreads();
            cur ^= 1;
            win = win_base + cur * (SS_CAP / 2);
            ent_w = ent_base + cur * (SS_MOB * EW);
            ent = (const ShiftEntry<R> *)ent_w;
#pragma unroll
            for (int p = 0; p < NPF; ++p) { const int i = (int)tid + 256 * p; if (i < wl) win[i] = pre[p]; }
            if ((int)tid < nmo * EW) ent_w[tid] = pre_e;
            __syncthreads();
            big_prev = false;
        } else {
            win = win_base; ent_w = ent_base; ent = (const ShiftEntry<R> *)ent_w;
            stage(trace(m), tb + kmin, kl == 1 ? wl : 0, tab + m + P.M * mo0);
            big_prev = true;
        }
        int kmin2 = 0, kmax2 = -1, alo2 = 0, ahi2 = -1;
        const uint64_t m2 = find(m + 1, kmin2, kmax2, alo2, ahi2);
        const int kl2 = m2 < P.M ? klass(kmin2, kmax2) : 0;
        if (m2 < P.M && kl2 == 0) fetch(m2, kmin2, (int)((int64_t)kmax2 - kmin2 + TPB + K - 1));       // in flight during the arithmetic below
        // the entries of the mask-free loops come through the SCALAR cache (uniform addresses; the loads of all eight terms are issued together): as broadcast
        // LDS reads they were a third of the kernel's LDS instructions.  -DQDAS_SS_ENT_LDS: from LDS, for A/B runs
#ifdef QDAS_SS_ENT_LDS
#define SS_ENT(j) ent[j]
#else
#define SS_ENT(j) tab[m + P.M * (mo0 + (uint64_t)(j))]
#endif
        const bool all_plain = std::is_same<DT, float2>::value && kl != 2 && nmo == SS_MOB && ahi != -2 && P.To < (1ull << 30);
        if (all_plain && (int64_t)alo <= tb + wv0 && (int64_t)ahi >= tb + wv0 + GW * TPT - 1) {   // (per wave; scalar)
#pragma unroll
            for (int j = 0; j < SS_MOB; ++j) {                                   // (straight-line code: an exit per term costs a copy of every accumulator)
                const ShiftEntry<R> e = SS_ENT(j);
                lean_term(std::false_type{}, e, j, e.k0 - kmin);
            }
        } else if (all_plain) {
#pragma unroll
            for (int j = 0; j < SS_MOB; ++j) {
                const ShiftEntry<R> e = SS_ENT(j);
                lean_term(std::true_type{}, e, j, e.k0 - kmin);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SS_MOB; ++j) {
                if (j >= nmo) break;
                const ShiftEntry<R> e = ent[j];                                  // broadcast LDS reads
                const bool here = e.on && (int64_t)e.thi >= tb && (int64_t)e.tlo < tb + TPB;
                if (!__builtin_amdgcn_readfirstlane((int)here)) continue;        // (uniform)
                if (kl == 2) stage(trace(m), tb + e.k0, TPB + K - 1, nullptr);
                term(e, j, kl == 2 ? 0 : e.k0 - kmin);
            }
        }
        m = m2; kmin = kmin2; kmax = kmax2; kl = kl2; alo = alo2; ahi = ahi2;
    }
#pragma unroll
    for (int j = 0; j < SS_MOB; ++j) {
        if (j >= nmo) break;
        DT *__restrict__ yo = (DT *)P.y + ((f * P.Mo + mo0 + j) * P.N + n) * P.To;
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            const int64_t t = tb + (int64_t)lt0 + GW * q;
            if (act && t < (int64_t)P.To) {
                if constexpr (CPLX) { DT v; v.x = ar[q][j]; v.y = ai[q][j]; yo[t] = v; }
                else yo[t] = ar[q][j];
            }
        }
    }
}

template <typename DT, typename R, int TPT, bool DPP = false>
static hipError_t launch_shift_t(const ShiftParams &P, int interp, const void *sh, const void *w, int w_real, hipStream_t s) {
    const uint64_t count = P.M * P.Mo;
    Scratch scratch(s);                                  // (the stream's arena: scratch.hip)
    ShiftEntry<R> *tab = (ShiftEntry<R> *)scratch.get(sizeof(ShiftEntry<R>) * count);
    if (!tab) return hipErrorOutOfMemory;
    hipError_t e = hipSuccess;
    const unsigned gb = (unsigned)((count + 255) / 256);
    const R *shr = (const R *)sh;
    switch (interp) {
        case 0: shift_table_kernel<0, R><<<gb, 256, 0, s>>>(shr, w, w_real, count, (int64_t)P.T, (int64_t)P.To, tab); break;
        case 1: case 4: shift_table_kernel<1, R><<<gb, 256, 0, s>>>(shr, w, w_real, count, (int64_t)P.T, (int64_t)P.To, tab); break;
        case 2: shift_table_kernel<2, R><<<gb, 256, 0, s>>>(shr, w, w_real, count, (int64_t)P.T, (int64_t)P.To, tab); break;
        case 3: shift_table_kernel<3, R><<<gb, 256, 0, s>>>(shr, w, w_real, count, (int64_t)P.T, (int64_t)P.To, tab); break;
        case 5: shift_table_kernel<5, R><<<gb, 256, 0, s>>>(shr, w, w_real, count, (int64_t)P.T, (int64_t)P.To, tab); break;
        default: return hipErrorInvalidValue;
    }
    ShiftParams p = P;
    p.tab = tab;
    p.mo_blocks = (uint32_t)((P.Mo + SS_MOB - 1) / SS_MOB);
    int4 *blk = (int4 *)scratch.get(sizeof(int4) * P.M * p.mo_blocks);
    if (!blk) return hipErrorOutOfMemory;
    shift_block_kernel<R><<<(unsigned)((P.M * p.mo_blocks + 255) / 256), 256, 0, s>>>(tab, P.M, P.Mo, p.mo_blocks, blk);
    p.blk = blk;
    const int Kt = interp_taps(interp);
    const int gw = DPP && Kt > 1 ? 65 - Kt : 64;
    const uint64_t TPB = 4ull * gw * TPT;
    const dim3 g((unsigned)((P.To + TPB - 1) / TPB), (unsigned)P.N, (unsigned)(p.mo_blocks * P.F));
    const size_t lds = sizeof(DT) * SS_CAP + 2 * sizeof(ShiftEntry<R>) * SS_MOB;
    const int K = interp_taps(interp);
#define QSS(KK, GWW)                                                                                                                    \
    do {                                                                                                                                \
        auto kfn = shift_sum_kernel<KK, DT, R, TPT, GWW>;                                                                                \
        e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        if (e == hipSuccess) kfn<<<g, 256, lds, s>>>(p);                                                                                \
    } while (0)
    if constexpr (DPP) { if (K == 1) QSS(1, 64); else if (K == 2) QSS(2, 63); else QSS(4, 61); }
    else { if (K == 1) QSS(1, 64); else if (K == 2) QSS(2, 64); else QSS(4, 64); }
#undef QSS
    if (e == hipSuccess) e = hipGetLastError();
    return e;
}

// dtype: QDAS_F64 (0) | QDAS_F32 (1); cplx: samples are interleaved complex
hipError_t launch_shift_sum(const ShiftParams &P, int dtype, int cplx, int interp, const void *sh, const void *w, int w_real, hipStream_t s) {
    if (P.To == 0 || P.N == 0 || P.Mo == 0 || P.F == 0) return hipSuccess;
    if (dtype == 1) {
        // outputs per lane: the choice among 4 / 3 / 2 that leaves the fewest idle outputs in the last block of a trace (To = 2200: 3 -> 2304 covered, 4 -> 3072)
        int best = 4;
        uint64_t waste = ~0ull;
        for (int tpt = 4; tpt >= 2; --tpt) {
            const uint64_t tpb = 256ull * tpt, cov = (P.To + tpb - 1) / tpb * tpb - P.To;
            if (cov * 8 < waste * 8 && (waste == ~0ull || (waste - cov) * 16 > P.To)) { waste = cov; best = tpt; }   // a smaller block only if it saves > 6 % of the outputs
        }
        if (const char *e = getenv("QDAS_SS_TPT")) { const int v = atoi(e); if (v >= 2 && v <= 4) best = v; }      // (experiments)
        // taps shared across lanes (groups of 65 - K outputs, shift_sum_kernel GW): measured at 3 outputs per lane, 0.90 of the time per block -- taken when it
        // needs no more blocks per trace than the 64-wide groups do (C1: 2 190 outputs = 3 x 732 either way).  QDAS_SS_NO_DPP: off; QDAS_SS_DPP=1: always
        if (cplx && interp_taps(interp) > 1 && !getenv("QDAS_SS_NO_DPP")) {
            const uint64_t tpd = 12ull * (65 - (uint64_t)interp_taps(interp));
            const bool force = getenv("QDAS_SS_DPP") != nullptr;
            if (force || (best == 3 && (P.To + tpd - 1) / tpd <= (P.To + 767) / 768)) return launch_shift_t<float2, float, 3, true>(P, interp, sh, w, w_real, s);
        }
        if (cplx) return best == 4 ? launch_shift_t<float2, float, 4>(P, interp, sh, w, w_real, s) : best == 3 ? launch_shift_t<float2, float, 3>(P, interp, sh, w, w_real, s)
                                                                                                                 : launch_shift_t<float2, float, 2>(P, interp, sh, w, w_real, s);
        return best == 4 ? launch_shift_t<float, float, 4>(P, interp, sh, w, w_real, s) : best == 3 ? launch_shift_t<float, float, 3>(P, interp, sh, w, w_real, s)
                                                                                                     : launch_shift_t<float, float, 2>(P, interp, sh, w, w_real, s);
    }
    return cplx ? launch_shift_t<double2, double, 2>(P, interp, sh, w, w_real, s) : launch_shift_t<double, double, 4>(P, interp, sh, w, w_real, s);
}

}  // namespace qdas
