// wsinterpd.hip -- the general single-delay flavour: weighted, phase-rotated sampling over an N-D broadcast index space.
//
// Computes what reference kern/wsinterpd.m (and kern/interpd.m: no weights, no sums) computes on the device
// (kernel bodies reference src/interpd.cu:295-342 wsinterpd_temp, :169-192 interpd_temp; launches kern/wsinterpd.m:221-236,
// kern/interpd.m) -- the entry behind ChannelData.sample (src/ChannelData.m:1230-1336) and everything built on it:
//
//     y[j_kept] = sum over the summed dimensions of   w[j] * exp(i*omega*t[j]) * sample(x[:, j], t[j])
//
// where j runs over an index space of up to 8 dimensions; dimension 0 is the sampling dimension (I samples of t against T samples
// of x), and each of t, w, x addresses it through its own element strides (0 = broadcast) -- the reference's matching / outer
// dimension classification (kern/wsinterpd.m:70-93) boils down to exactly these strides.  Differences by design: every output is
// OWNED by one lane and summed in a fixed order (the reference adds with float atomics, src/interpd.cu:339); the summed dimensions
// are the inner loop.  Infinite t are skipped (src/interpd.cu:333); out-of-record samples yield `extrap` (the kernels' no_v), which
// sums treat like MATLAB's sum(..., 'omitnan') when it is NaN (kern/wsinterpd.m:262).
#include "qdas_device.h"
#include "qdas_kernels.h"
#include <type_traits>
#include <stdlib.h>

namespace qdas {

template <int INTERP, typename R, typename ST>
__device__ __forceinline__ bool sample_strided(const ST *__restrict__ tr, long tstride, long T, R s, cplx<R> &out) {
    out = {(R)0, (R)0};
    if (!(s >= (R)0)) return false;                      // tau >= 0; rejects NaN
    if constexpr (INTERP == 0) {                         // nearest (src/interpd.cu:70-72)
        const R r = qfloor(s + (R)0.5);
        if (!(r < (R)T)) return false;
        out = ld(tr, (size_t)((long)r * tstride));
        return true;
    } else {
        const R fl = qfloor(s);
        constexpr int K = interp_taps(INTERP);
        constexpr int OFF = (K == 2) ? 0 : -1;
        if (!(fl + (R)(K - 1 + OFF) < (R)T) || fl + (R)OFF < (R)0) return false;   // all taps in [0,T); rejects +inf
        const long first = (long)fl + OFF;
        R w[4];
        interp_weights<INTERP>(s - fl, w);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const cplx<R> v = ld(tr, (size_t)((first + k) * tstride));
            out.x += w[k] * v.x; out.y += w[k] * v.y;
        }
        return true;
    }
}

// SPL = 4: the four waves of a workgroup share 64 outputs, each sums a quarter of the terms (a contiguous range of the odometer's order) and wave 0 adds
// the four partial sums in range order.  A lane sums its terms one after the other; with few outputs and many terms (ChannelData.sample with sdim at C2
// size: 262 144 outputs x 128 terms = 16 waves per CU, each walking 128 terms) the kernel's time was ONE wave's chain of memory round trips.
template <int INTERP, typename TY, int SPL = 1>
__global__ void __launch_bounds__(256) wsinterpd_kernel(const WsParams P) {
    using R  = typename TY::real;
    using ST = typename TY::store;
    using AR = typename TY::apod_real_t;
    const R *__restrict__ t = (const R *)P.t;
    const ST *__restrict__ x = (const ST *)P.x;
    // kept dimensions, in the order WsParams::kord lists them (fastest first: the host puts the dimension along which x is contiguous there, so
    // that the lanes of a wave read neighbouring samples and write neighbouring outputs -- y has its own strides, WsParams::yst).
    // Lanes run along kord[0] (no per-lane division); the other kept dimensions are decoded from the block id: uniform, scalar arithmetic.
    int64_t tb = 0, xb = 0, wb = 0, yo = 0;
    {
        const int d0 = P.kord[0];
        uint64_t i0 = (uint64_t)blockIdx.x * (256 / SPL) + (threadIdx.x % (256 / SPL));
        if (i0 >= P.n_lane) return;
        if (P.lane2) {                                   // a short fastest dimension: the lanes cover kord[0] x kord[1] (one division per lane)
            const int d1 = P.kord[1];
            const uint32_t s0 = (uint32_t)P.size[d0], i1 = (uint32_t)i0 / s0;      // (n_lane < 2^31: 32-bit division)
            i0 -= (uint64_t)i1 * s0;
            tb = (int64_t)i1 * P.tst[d1]; xb = (int64_t)i1 * P.xst[d1]; wb = (int64_t)i1 * P.wst[d1]; yo = (int64_t)i1 * P.yst[d1];
        }
        tb += (int64_t)i0 * P.tst[d0]; xb += (int64_t)i0 * P.xst[d0]; wb += (int64_t)i0 * P.wst[d0]; yo += (int64_t)i0 * P.yst[d0];
        uint64_t q = (uint64_t)blockIdx.y + (uint64_t)gridDim.y * blockIdx.z;
        if (q >= P.n_rest) return;
        for (int k = P.lane2 ? 2 : 1; k < P.nkd; ++k) {
            const int d = P.kord[k];
            const uint64_t idx = q % P.size[d];
            q /= P.size[d];
            tb += (int64_t)idx * P.tst[d]; xb += (int64_t)idx * P.xst[d]; wb += (int64_t)idx * P.wst[d]; yo += (int64_t)idx * P.yst[d];
        }
    }
    const R omega = (R)P.omega;
    const bool skip_nan = P.any_sum && !(P.extrap == P.extrap);      // sums omit NaN (kern/wsinterpd.m:262)
    cplx<R> acc = {(R)0, (R)0};
    // the summed dimensions: an odometer over the compacted list (WsParams::ssz / sts / sxs / sws).  Counters and running offsets are UNIFORM
    // (every lane sums the same terms of its own output): scalar adds and compares, one carry chain per term -- the reference, and round 2
    // here, decode every term's index with a 64-bit divide and modulo per dimension (src/interpd.cu:316-331)
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t ut = 0, ux = 0, uw = 0;
    auto advance = [&]() {                               // odometer: the next term's uniform offsets
        bool carry = true;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if (d < P.nsd && carry) {
                ut += P.sts[d]; ux += P.sxs[d]; uw += P.sws[d];
                if (++cnt[d] == P.ssz[d]) { cnt[d] = 0; ut -= (int64_t)P.ssz[d] * P.sts[d]; ux -= (int64_t)P.ssz[d] * P.sxs[d]; uw -= (int64_t)P.ssz[d] * P.sws[d]; }
                else carry = false;
            }
        }
    };
    // A lane sums its output's terms one after the other, and every term is a chain of two dependent memory round trips (t, then the taps of x): with one
    // term in flight per lane the sum-over-transmits shape (ChannelData.sample with sdim: 128 terms per output, 16 waves per CU) ran at the memory LATENCY,
    // 0.04-0.08 of the HBM roof (profiles/r04/general_time.txt).  Four terms per pass: their delays are loaded together, then their taps -- from CLAMPED
    // (always in-record) indices, so that no load hides behind a branch; the support test selects afterwards --, then they are added in term order.
    constexpr int U = INTERP == 3 ? 2 : 4;              // (lanczos3: its sinpi weights need the registers -- four terms in flight spilled to scratch)
    constexpr int K = INTERP == 0 ? 1 : interp_taps(INTERP);
    constexpr int OFF = (K <= 2) ? 0 : -1;
    const long T = (long)P.T, xts = (long)P.x_tstride;
    uint64_t r_lo = 0, r_hi = P.n_sum;
    if constexpr (SPL > 1) {                             // this wave's share of the terms; the odometer starts there (divisions once per wave)
        const uint64_t sp = (uint64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / (256 / SPL)));
        r_lo = P.n_sum * sp / SPL; r_hi = P.n_sum * (sp + 1) / SPL;
        uint64_t q = r_lo;
        for (int d = 0; d < 8; ++d) {
            if (d < P.nsd) {
                cnt[d] = (uint32_t)(q % P.ssz[d]); q /= P.ssz[d];
                ut += (int64_t)cnt[d] * P.sts[d]; ux += (int64_t)cnt[d] * P.sxs[d]; uw += (int64_t)cnt[d] * P.sws[d];
            }
        }
    }
    for (uint64_t r = r_lo; r < r_hi; r += U) {
        int64_t xo[U], wo[U];
        R tau[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = r + u < r_hi;
            const int64_t to = tb + ut;
            xo[u] = xb + ux; wo[u] = wb + uw;
            tau[u] = live[u] ? t[to] : (R)0;
            if (live[u]) advance();
        }
        cplx<R> tap[U][K];
        R wk[U][4];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const R sft = INTERP == 0 ? qfloor(tau[u] + (R)0.5) : qfloor(tau[u]);
            // support: tau >= 0 and every tap in [0, T) (sample_strided above; NaN and +-inf fall out of it)
            ok[u] = tau[u] >= (R)0 && sft + (R)(K - 1 + OFF) < (R)T && sft + (R)OFF >= (R)0;
            long first = ok[u] ? (long)sft + OFF : 0;
            if (T < K) first = 0;
            wk[u][0] = (R)1; wk[u][1] = wk[u][2] = wk[u][3] = (R)0;
            if constexpr (K > 1) interp_weights<INTERP>(ok[u] ? tau[u] - sft : (R)0, wk[u]);
#pragma unroll
            for (int k = 0; k < K; ++k) tap[u][k] = (live[u] && T >= K) ? ld(x + xo[u], (size_t)((first + k) * xts)) : cplx<R>{(R)0, (R)0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            if (!(fabs((double)tau[u]) <= 1.0e300) && tau[u] == tau[u]) continue;      // +-inf: excluded (src/interpd.cu:333)
            cplx<R> v = {(R)0, (R)0};
            if (ok[u] && T >= K) {
#pragma unroll
                for (int k = 0; k < K; ++k) { v.x += wk[u][k] * tap[u][k].x; v.y += wk[u][k] * tap[u][k].y; }
            } else {
                if (skip_nan) continue;
                v = {(R)P.extrap, (R)0};
            }
            if (omega != (R)0) {                                                       // src/interpd.cu:334
                R sn, cs;
                if constexpr (sizeof(R) == 8) sincos((double)(omega * tau[u]), (double *)&sn, (double *)&cs);
                else sincosf((float)(omega * tau[u]), (float *)&sn, (float *)&cs);
                v = cmul(v, cplx<R>{cs, sn});
            }
            if (P.w) {
                if (P.w_real) { const R w = (R)ldr((const AR *)P.w, (size_t)wo[u]); v.x *= w; v.y *= w; }
                else v = cmul(v, ld((const ST *)P.w, (size_t)wo[u]));
            }
            acc.x += v.x; acc.y += v.y;
        }
    }
    if constexpr (SPL > 1) {
        __shared__ cplx<R> part[SPL][256 / SPL];
        const uint32_t sp = threadIdx.x / (256 / SPL), ln = threadIdx.x % (256 / SPL);
        part[sp][ln] = acc;
        __syncthreads();
        if (sp != 0) return;
#pragma unroll
        for (int k = 1; k < SPL; ++k) { acc.x += part[k][ln].x; acc.y += part[k][ln].y; }
    }
    st((ST *)P.y, (size_t)yo, acc);
}

// ---- the streaming case: no summed dimension, no weights, no phase rotation (kern/interpd.m; ChannelData.sample without apodization; rectifyt0), every
// extent below 2^31 elements, dimension 0 (the sampled one) NOT among the lane dimensions.  Lean on purpose -- the general kernel above spends ~210 VALU
// instructions per output on 64-bit index arithmetic and the (empty) odometer, as much time as the memory system needs for the data --:
//   * 32-bit lane offsets on top of uniform 64-bit bases;
//   * a lane makes RUN consecutive outputs along dimension 0: for the usual monotone t (rectifyt0, resampling) consecutive outputs share all but
//     one tap -- re-read from L1 instead of L2 / HBM, where 32 output rows in flight at once overran the XCD's L2 (hit rate 25 %);
template <int INTERP, typename TY, int RUN>
__global__ void __launch_bounds__(256) interpd_stream_kernel(const WsParams P) {
    using R  = typename TY::real;
    using ST = typename TY::store;
    const int d0 = P.kord[0];
    uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
    if (i0 >= (uint32_t)P.n_lane) return;
    uint32_t lt = 0, lx = 0, ly = 0;                     // this lane's offsets (elements)
    if (P.lane2) {
        const int d1 = P.kord[1];
        const uint32_t s0 = (uint32_t)P.size[d0], i1 = i0 / s0;
        i0 -= i1 * s0;
        lt = i1 * (uint32_t)P.tst[d1]; lx = i1 * (uint32_t)P.xst[d1]; ly = i1 * (uint32_t)P.yst[d1];
    }
    lt += i0 * (uint32_t)P.tst[d0]; lx += i0 * (uint32_t)P.xst[d0]; ly += i0 * (uint32_t)P.yst[d0];
    // the other kept dimensions from the block id (uniform); dimension 0 in chunks of RUN
    uint64_t q = (uint64_t)blockIdx.y + (uint64_t)gridDim.y * blockIdx.z;
    if (q >= P.n_rest) return;
    int64_t tb = 0, xb = 0, yb = 0;
    uint32_t ic = 0;
    for (int k = P.lane2 ? 2 : 1; k < P.nkd; ++k) {
        const int d = P.kord[k];
        const uint64_t sz = d == 0 ? (P.size[0] + RUN - 1) / RUN : P.size[d];
        const uint64_t idx = q % sz;
        q /= sz;
        if (d == 0) ic = (uint32_t)idx;
        else { tb += (int64_t)idx * P.tst[d]; xb += (int64_t)idx * P.xst[d]; yb += (int64_t)idx * P.yst[d]; }
    }
    const R *__restrict__ t = (const R *)P.t + tb;
    const ST *__restrict__ x = (const ST *)P.x + xb;
    ST *__restrict__ y = (ST *)P.y + yb;
    const uint32_t I = (uint32_t)P.size[0], T = (uint32_t)P.T, xts = (uint32_t)P.x_tstride, tts = (uint32_t)P.tst[0], yts = (uint32_t)P.yst[0];
    constexpr int K = interp_taps(INTERP);
    constexpr int OFF = (K == 1) ? 0 : (K == 2) ? 0 : -1;
#pragma unroll
    for (int r = 0; r < RUN; ++r) {
        const uint32_t i = ic * RUN + r;
        if (i >= I) break;
        const R tau = t[lt + i * tts];
        cplx<R> v = {(R)P.extrap, (R)0};
        // support: tau >= 0 and every tap in [0, T) (the rule of sample_strided above; NaN and +-inf fall out of it)
        const R base = INTERP == 0 ? qfloor(tau + (R)0.5) : qfloor(tau);
        if (tau >= (R)0 && base + (R)(K - 1 + OFF) < (R)T && base + (R)OFF >= (R)0) {
            const uint32_t first = (uint32_t)((int)base + OFF);
            R w[4] = {(R)1, (R)0, (R)0, (R)0};
            if constexpr (K > 1) interp_weights<INTERP>(tau - base, w);
            v = {(R)0, (R)0};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const cplx<R> u = ld(x, (size_t)(lx + (first + k) * xts));
                v.x += w[k] * u.x; v.y += w[k] * u.y;
            }
        }
        st(y, (size_t)(ly + i * yts), v);
    }
}


// ---- one summed dimension, x contiguous along it (a torch-order record summed over its last dimension): lanes along the SUM.  A wave owns RUNW consecutive
// outputs (kept-dimension order of WsParams::kord); for each, lane l takes the terms l, l + 64, ... -- the K taps of 64 neighbouring terms are K gathers
// out of a few memory rows (neighbouring terms sample neighbouring times), 8 lanes per 64-byte sector, the rows shared with the wave's next output -- and
// the wave adds up across its lanes (a fixed tree: reproducible).  The one-output-per-lane kernel above reads such a record with every lane in a row of
// its own; round 5 transposed the record first (0.30 ms at C2 size, against 0.13 ms for the sum itself).
// LW: lanes per output -- 64, or 32 for sums of at most 32 terms (two outputs side by side in a wave: no idle half)
template <int INTERP, typename TY, int RUNW, int LW = 64>
__global__ void __launch_bounds__(256) wsinterpd_lanesum_kernel(const WsParams P) {
    using R  = typename TY::real;
    using ST = typename TY::store;
    using AR = typename TY::apod_real_t;
    const R *__restrict__ t = (const R *)P.t;
    const ST *__restrict__ x = (const ST *)P.x;
    const uint32_t lane = threadIdx.x & (uint32_t)(LW - 1), grp = LW == 64 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : threadIdx.x / (uint32_t)LW;
    const uint64_t o0 = ((uint64_t)blockIdx.x * (256u / LW) + grp) * RUNW;
    if (o0 >= P.n_out) return;                            // (a whole group of LW lanes: the adds across lanes below stay inside it)
    // this group's first output, decoded once (uniform for LW = 64); the following ones by an odometer over the kept dimensions
    uint32_t idx[8];
    int64_t tb = 0, xb = 0, wb = 0, yo = 0;
    {
        uint64_t q = o0;
        for (int k = 0; k < P.nkd; ++k) {
            const int d = P.kord[k];
            idx[k] = (uint32_t)(q % P.size[d]); q /= P.size[d];
            tb += (int64_t)idx[k] * P.tst[d]; xb += (int64_t)idx[k] * P.xst[d]; wb += (int64_t)idx[k] * P.wst[d]; yo += (int64_t)idx[k] * P.yst[d];
        }
    }
    constexpr int K = INTERP == 0 ? 1 : interp_taps(INTERP);
    constexpr int OFF = (K <= 2) ? 0 : -1;
    const long T = (long)P.T, xts = (long)P.x_tstride;
    const uint32_t S = P.ssz[0];
    const int64_t sts = P.sts[0], sws = P.sws[0];
    const R omega = (R)P.omega;
    const bool skip_nan = !(P.extrap == P.extrap);        // sums omit NaN (kern/wsinterpd.m:262)
    // the RUNW outputs' base offsets (uniform odometer over the kept dimensions), then ONE loop over the terms with the outputs innermost: RUNW delays,
    // then RUNW x K taps in flight per lane (one output after the other ran at the latency of its own two dependent round trips)
    int64_t tbr[RUNW], xbr[RUNW], wbr[RUNW], yor[RUNW];
    bool have[RUNW];
#pragma unroll
    for (int r = 0; r < RUNW; ++r) {
        have[r] = o0 + (uint64_t)r < P.n_out;
        tbr[r] = tb; xbr[r] = xb; wbr[r] = wb; yor[r] = yo;
        bool carry = true;
        for (int k = 0; k < P.nkd && carry; ++k) {
            const int d = P.kord[k];
            tb += P.tst[d]; xb += P.xst[d]; wb += P.wst[d]; yo += P.yst[d];
            if (++idx[k] == (uint32_t)P.size[d]) {
                idx[k] = 0;
                tb -= (int64_t)P.size[d] * P.tst[d]; xb -= (int64_t)P.size[d] * P.xst[d]; wb -= (int64_t)P.size[d] * P.wst[d]; yo -= (int64_t)P.size[d] * P.yst[d];
            } else carry = false;
        }
    }
    cplx<R> acc[RUNW];
#pragma unroll
    for (int r = 0; r < RUNW; ++r) acc[r] = {(R)0, (R)0};
    for (uint32_t j = lane; j < ((S + (uint32_t)(LW - 1)) & ~(uint32_t)(LW - 1)); j += (uint32_t)LW) {
        const bool in = j < S;
        R tau[RUNW];
        bool live[RUNW], ok[RUNW];
        cplx<R> tap[RUNW][K];
        R wk[RUNW][4];
#pragma unroll
        for (int r = 0; r < RUNW; ++r) {
            live[r] = in && have[r];
            tau[r] = live[r] ? t[tbr[r] + (int64_t)j * sts] : (R)0;
        }
#pragma unroll
        for (int r = 0; r < RUNW; ++r) {
            const R sft = INTERP == 0 ? qfloor(tau[r] + (R)0.5) : qfloor(tau[r]);
            ok[r] = tau[r] >= (R)0 && sft + (R)(K - 1 + OFF) < (R)T && sft + (R)OFF >= (R)0;       // (the rule of sample_strided above)
            long first = ok[r] ? (long)sft + OFF : 0;
            if (T < K) first = 0;
            wk[r][0] = (R)1; wk[r][1] = wk[r][2] = wk[r][3] = (R)0;
            if constexpr (K > 1) interp_weights<INTERP>(ok[r] ? tau[r] - sft : (R)0, wk[r]);
#pragma unroll
            for (int k = 0; k < K; ++k) tap[r][k] = (live[r] && T >= K) ? ld(x + xbr[r] + j, (size_t)((first + k) * xts)) : cplx<R>{(R)0, (R)0};
        }
#pragma unroll
        for (int r = 0; r < RUNW; ++r) {
            if (!live[r]) continue;
            if (!(fabs((double)tau[r]) <= 1.0e300) && tau[r] == tau[r]) continue;          // +-inf: excluded (src/interpd.cu:333)
            cplx<R> v = {(R)0, (R)0};
            if (ok[r] && T >= K) {
#pragma unroll
                for (int k = 0; k < K; ++k) { v.x += wk[r][k] * tap[r][k].x; v.y += wk[r][k] * tap[r][k].y; }
            } else {
                if (skip_nan) continue;
                v = {(R)P.extrap, (R)0};
            }
            if (omega != (R)0) {                                                           // src/interpd.cu:334
                R sn, cs;
                if constexpr (sizeof(R) == 8) sincos((double)(omega * tau[r]), (double *)&sn, (double *)&cs);
                else sincosf((float)(omega * tau[r]), (float *)&sn, (float *)&cs);
                v = cmul(v, cplx<R>{cs, sn});
            }
            if (P.w) {
                const size_t wo = (size_t)(wbr[r] + (int64_t)j * sws);
                if (P.w_real) { const R w = (R)ldr((const AR *)P.w, wo); v.x *= w; v.y *= w; }
                else v = cmul(v, ld((const ST *)P.w, wo));
            }
            acc[r].x += v.x; acc[r].y += v.y;
        }
    }
#pragma unroll
    for (int r = 0; r < RUNW; ++r) {
#pragma unroll
        for (int sft = LW / 2; sft >= 1; sft >>= 1) { acc[r].x += __shfl_xor(acc[r].x, sft, 64); acc[r].y += __shfl_xor(acc[r].y, sft, 64); }
        if (lane == 0 && have[r]) st((ST *)P.y, (size_t)yor[r], acc[r]);
    }
}

template <typename TY> static hipError_t launch_ws_t(const WsParams &P, hipStream_t s) {
    if (P.stream_ok) {                                    // the lean streaming kernel (conditions checked by the host: qdas_api.hip)
        constexpr int RUN = 8;
        const uint64_t nr = P.n_rest / P.size[0] * ((P.size[0] + RUN - 1) / RUN);       // dimension 0 in chunks of RUN
        const uint64_t gy = nr < 65535 ? (nr ? nr : 1) : 65535, gz = (nr + gy - 1) / gy;
        if (gz <= 65535) {
            WsParams Q = P;
            Q.n_rest = nr;
            const dim3 g((unsigned)((P.n_lane + 255) / 256), (unsigned)gy, (unsigned)(gz ? gz : 1)), b(256);
            switch (P.flag & 7) {
                case 0: interpd_stream_kernel<0, TY, RUN><<<g, b, 0, s>>>(Q); break;
                case 1: case 4: interpd_stream_kernel<1, TY, RUN><<<g, b, 0, s>>>(Q); break;
                case 2: interpd_stream_kernel<2, TY, RUN><<<g, b, 0, s>>>(Q); break;
                case 3: interpd_stream_kernel<3, TY, RUN><<<g, b, 0, s>>>(Q); break;
                case 5: interpd_stream_kernel<5, TY, RUN><<<g, b, 0, s>>>(Q); break;
                default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    if (P.lanesum_ok) {
        const char *rv = getenv("QDAS_WS_RUNW");                              // (experiments: outputs per wave)
        const int runw = rv ? atoi(rv) : 4;
#define QLS(RUNW, LW)                                                                                                      \
        do {                                                                                                               \
            const uint64_t per = (256 / LW) * RUNW, nb = (P.n_out + per - 1) / per;                                        \
            if (nb <= 0x7fffffffull) {                                                                                     \
                const dim3 g((unsigned)nb), b(256);                                                                        \
                switch (P.flag & 7) {                                                                                      \
                    case 0: wsinterpd_lanesum_kernel<0, TY, RUNW, LW><<<g, b, 0, s>>>(P); break;                           \
                    case 1: case 4: wsinterpd_lanesum_kernel<1, TY, RUNW, LW><<<g, b, 0, s>>>(P); break;                   \
                    case 2: wsinterpd_lanesum_kernel<2, TY, RUNW, LW><<<g, b, 0, s>>>(P); break;                           \
                    case 3: wsinterpd_lanesum_kernel<3, TY, RUNW, LW><<<g, b, 0, s>>>(P); break;                           \
                    case 5: wsinterpd_lanesum_kernel<5, TY, RUNW, LW><<<g, b, 0, s>>>(P); break;                           \
                    default: return hipErrorInvalidValue;                                                                  \
                }                                                                                                          \
                return hipGetLastError();                                                                                  \
            }                                                                                                              \
        } while (0)
        if (P.ssz[0] <= 32) QLS(4, 32);
        else if (runw == 2) QLS(2, 64); else if (runw == 8) QLS(8, 64); else QLS(4, 64);
#undef QLS
    }
    // grid: x = blocks along the fastest kept dimension, (y, z) = the other kept dimensions flattened
    const uint64_t n0 = P.n_lane, gy = P.n_rest < 65535 ? (P.n_rest ? P.n_rest : 1) : 65535, gz = (P.n_rest + gy - 1) / gy;
    if (gz > 65535 || (n0 + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
    // many terms per output: four waves share the terms of 64 outputs (fp32 / fp16 data; QDAS_WS_NO_SPLIT keeps one wave per output)
    if (P.n_sum >= 32 && !std::is_same<TY, st_f64>::value && (n0 + 63) / 64 <= 0x7fffffffull && !getenv("QDAS_WS_NO_SPLIT")) {
        const dim3 g4((unsigned)((n0 + 63) / 64), (unsigned)gy, (unsigned)(gz ? gz : 1)), b4(256);
        switch (P.flag & 7) {
            case 0: wsinterpd_kernel<0, TY, 4><<<g4, b4, 0, s>>>(P); break;
            case 1: case 4: wsinterpd_kernel<1, TY, 4><<<g4, b4, 0, s>>>(P); break;
            case 2: wsinterpd_kernel<2, TY, 4><<<g4, b4, 0, s>>>(P); break;
            case 3: wsinterpd_kernel<3, TY, 4><<<g4, b4, 0, s>>>(P); break;
            case 5: wsinterpd_kernel<5, TY, 4><<<g4, b4, 0, s>>>(P); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const dim3 g((unsigned)((n0 + 255) / 256), (unsigned)gy, (unsigned)(gz ? gz : 1)), b(256);
    switch (P.flag & 7) {
        case 0: wsinterpd_kernel<0, TY><<<g, b, 0, s>>>(P); break;
        case 1: case 4: wsinterpd_kernel<1, TY><<<g, b, 0, s>>>(P); break;
        case 2: wsinterpd_kernel<2, TY><<<g, b, 0, s>>>(P); break;
        case 3: wsinterpd_kernel<3, TY><<<g, b, 0, s>>>(P); break;
        case 5: wsinterpd_kernel<5, TY><<<g, b, 0, s>>>(P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_wsinterpd(const WsParams &P, int dtype, hipStream_t s) {
    if (P.n_out == 0) return hipSuccess;
    switch (dtype) {
        case 0: return launch_ws_t<st_f64>(P, s);
        case 1: return launch_ws_t<st_f32>(P, s);
        case 2: return launch_ws_t<st_f16>(P, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
