// tile_pairs.h -- the pair loops of the tiled kernel (internal; included by das_tile_impl.h): for one stage (receiver n, block of
// MB transmits), every lane forms its pixel's MB interpolated samples and adds them to its accumulators.
//
// Transmits (m0+2p, m0+2p+1) ride in the two halves of packed fp32.  Per packed pair:
//     t  = ra[p] + rb                    = tau*fs + OFF - (A+B) - 1/2
//     tm = t + MAGIC;  s = t - (tm - MAGIC)  in [-1/2, 1/2];   LDS byte address of the first tap = bits(tm)*SB + cbase + immediate
//     weights2<INTERP>(s)  (scheduled between the issue of the LDS reads and their wait: that hides the LDS latency)
//     acc += w[k] * tap[k]
// Accumulators: one per (frame, transmit half) with up to two frames per launch, one per frame with four.
#pragma once

// priority of a pair-loop quarter (3, 2, 1, 0); tuning builds may shift the staircase below the stage head's priority 3
#ifndef QDAS_PAIR_PRIO
#define QDAS_PAIR_PRIO(q) (q)
#endif

namespace qdas {

// Plain loop.  TAILV = false: full block (reciprocal mode: block entirely above the diagonal), no bounds tests on m.
// TAILV = true: last, partial transmit block (bounds checks) or -- reciprocal mode -- the block that contains m == n.
// CHECK: the tile touches the ends of the record: edge rule per sample (all taps in [0,T) and tau >= 0; select, not multiply).
// WZ: some table weight of the stage is zero (stage weights from LDS; else: no zero tests at all in the loop).
template <class C> template <bool CHECK, bool TAILV, bool WZ>
__device__ __forceinline__ void Tile<C>::pairs_plain(uint32_t n, uint32_t m0, int bn, float rb, uint32_t cbase, float phB, uint32_t wbase, uint64_t wmask, uint64_t xmask) {
    constexpr int K = C::K, MB = C::MB, WB = C::WB, INTERP = C::INTERP, NHP = C::NHP;
    constexpr bool SYM = C::SYM, FB4 = C::QUAD, FBX = C::FBX, TWO = C::TWO, F32 = C::F32, FMOD = C::FMOD, WTAB = C::WTAB, BF = C::BF;     // (FB4 here: four window sets, two passes)
    // (folded data: any N == M -- the last transmit block may be partial AND holds the diagonal: both rules at once)
    constexpr bool TAIL = (!SYM || C::FOLD) && TAILV, DIAG = SYM && TAILV;
    // (stage weights: does any transmit pair of this stage hold exactly one zero weight?)
    const bool wmixed = C::WST && WZ && ((((wmask ^ (wmask >> 1)) & 0x5555555555555555ull) != 0ull) || (SYM && C::WTAB && (((xmask ^ (xmask >> 1)) & 0x5555555555555555ull) != 0ull)));
    unroll<MB / 2>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        // Fair progress inside a stage: the hardware issues oldest-wave-first, so without help the four waves of a SIMD finish their
        // pair loops one after the other and the last one runs alone -- at a fraction of the issue rate -- until the stage barrier.
        // Lowering the priority as a wave advances (3 .. 0 over the block) lets the waves that are BEHIND go first: they finish
        // together.  Measured (profiles/r02/exp_prio.txt): general kernels -5 ... -7 %, reciprocal mode with 32-transmit stages -6 %.
        if constexpr (!hooks::no_fair_prio && (p * 4) % (MB / 2) == 0) __builtin_amdgcn_s_setprio(QDAS_PAIR_PRIO(3 - (p * 4) / (MB / 2)));
        const uint32_t m = m0 + 2 * p;
        if constexpr (TAIL) { if (m >= M) return; }
        if constexpr (DIAG) { if (m + 1 < n) return; }          // both transmits below the diagonal: their pairs were done as mirrors
        const bool upper = !TAIL || (m + 1 < M);  // the upper half carries a real transmit
        float wr0 = 1.f, wi0 = 0.f, wr1 = 1.f, wi1 = 0.f;
        float xr0 = 1.f, xi0 = 0.f, xr1 = 1.f, xi1 = 0.f;          // reciprocal mode: weights of the MIRROR pairs (receiver m | m+1, transmit n)
        constexpr bool SW = SYM && WTAB;                          // the two traces of an unordered pair carry different weights: separate sums
        constexpr bool SEP = FBX || SW || C::FOLDQ;               // the second window set has its OWN sums (next frame | its own weights | the mirror image of my pixel)
        constexpr bool RECIP = SYM && !SW && !C::FOLD;            // ... or holds the reciprocal trace of the same pair: one sum for both
        v4f wv = {1.f, 0.f, 1.f, 0.f}, xv = {1.f, 0.f, 1.f, 0.f};    // stage weights from LDS: {w[n,m], w[n,m+1]} and, reciprocal mode, {w[m,n], w[m+1,n]}
        if constexpr (C::WST) {
            // zero weights: skipped (src/bf.cu:122,126) -- the stage's non-zero masks are uniform, the tests scalar
            if constexpr (WZ) { if (((wmask >> (2 * p)) & 3ull) == 0ull && (!SW || ((xmask >> (2 * p)) & 3ull) == 0ull)) return; }
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(wv) : "v"(wbase), "n"(2 * p * 8));
            if constexpr (SW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(xv) : "v"(wbase), "n"(MB * 8 + 2 * p * 8));
        } else if constexpr (WTAB) {
            const float2 wa = ((const float2 *)P.wtab)[n + (size_t)N * m];
            const float2 wb_ = upper ? ((const float2 *)P.wtab)[n + (size_t)N * (m + 1)] : make_float2(0.f, 0.f);
            wr0 = wa.x; wi0 = wa.y; wr1 = wb_.x; wi1 = wb_.y;
            if constexpr (!BF) { if (wr0 == 0.f && wi0 == 0.f && wr1 == 0.f && wi1 == 0.f) return; }   // zero weights: skip (src/bf.cu:122,126); 'BF' stores the zeros
        }
        const v2f t = ra[p] + rb;
        const v2f tm = t + MAGIC;
        const v2f s = t - (tm - MAGIC);
        const uint32_t ad0 = (hooks::linear_taps ? (MAGIC_BITS + (uint32_t)lane + (__float_as_uint(tm.x) & 1u)) : __float_as_uint(tm.x)) * (uint32_t)C::SB + cbase;
        const uint32_t ad1 = (hooks::linear_taps ? (MAGIC_BITS + (uint32_t)lane + (__float_as_uint(tm.y) & 1u)) : __float_as_uint(tm.y)) * (uint32_t)C::SB + cbase;
        constexpr bool SPLIT = CHECK || FMOD || WTAB || BF || C::BPIX; // the two halves need separate post-processing
        v2f w[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        // one pass per FRAME PAIR (two passes when four frames share the launch): same tap index and weights
        unroll<NHP>([&](auto hpc) {
            constexpr int hp = decltype(hpc)::value;
            constexpr int GSET = FB4 ? 2 * hp : 0, HSET = FB4 ? 2 * hp + 1 : 1;       // window sets of the pass
            v2f &A0 = *(FB4 ? (hp ? &acc2 : &acc) : &acc), &A1 = *(FB4 ? (hp ? &acc2 : &acc) : ((C::FOLDQ || C::ONEACC) ? &acc : &acc1));      // (FOLDQ: one accumulator per pixel, as MIRQ -- the 32-transmit stages need the registers)
            // (reciprocal + lateral-mirror mode: ONE accumulator per pixel -- my pixel, its mirror image; the register budget of four window sets)
            v2f &B0 = *((C::MIRQ && !C::FOLD) ? (hp ? &acc2 : &acc) : FB4 ? (hp ? &acc3 : &acc1) : &acc2), &B1 = *((C::MIRQ && !C::FOLD) ? (hp ? &acc2 : &acc) : FB4 ? (hp ? &acc3 : &acc1) : ((SYM || C::ONEACC) ? &acc2 : &acc3));
            v2f v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
            v2f u0 = {0.f, 0.f}, u1 = {0.f, 0.f};     // the same two pairs of the second frame (FB2)
            if constexpr (F32) {
                taps_f32 g0, g1, h0, h1;              // direct taps x[:, n, m | m+1]; mirror taps x[:, m | m+1, n]
                if constexpr (hooks::no_tap_reads) { for (int k = 0; k < 4; ++k) { g0.s[k] = h0.s[k] = (v2f){s.x, t.x}; g1.s[k] = h1.s[k] = (v2f){t.y, s.y}; } }
                else {
                    lds_issue<K, (GSET * MB + 2 * p) * WB>(g0, ad0); lds_issue<K, (GSET * MB + 2 * p + 1) * WB>(g1, ad1);
                    if constexpr (TWO) { lds_issue<K, (HSET * MB + 2 * p) * WB>(h0, ad0); lds_issue<K, (HSET * MB + 2 * p + 1) * WB>(h1, ad1); }
                }
                if constexpr (hp == 0) {              // (the next frame pair reuses them)
                    if constexpr (hooks::trivial_weights) { w[0] = s; w[1] = t; w[2] = tm; w[3] = s + t; }
                    else if constexpr (K > 1) weights2<INTERP>(s, w);   // overlaps the LDS latency
                }
                if constexpr (TWO) lds_fence2<K>(g0, g1, h0, h1, w); else lds_fence<K>(g0, g1, w);
                if constexpr (DIAG) {                 // uniform, only in the block that holds the diagonal
                    const v2f z = {0.f, 0.f};
                    if (m < n)      { for (int k = 0; k < K; ++k) g0.s[k] = z; }     // pair (n, m<n): done as the mirror of (m, n)
                    if constexpr (C::FOLD) {                                          // (folded data: the mirror image's trace of the same pair, same rule)
                        if constexpr (TWO) { if (m < n) { for (int k = 0; k < K; ++k) h0.s[k] = z; } }
                    } else {
                        if (m <= n)     { for (int k = 0; k < K; ++k) h0.s[k] = z; }     // m == n: the trace is its own mirror
                        if (m + 1 <= n) { for (int k = 0; k < K; ++k) h1.s[k] = z; }
                    }
                }
                if constexpr (TAIL) {
                    if (!upper) {                       // odd M: no transmit in the upper half (uniform, rare)
#pragma unroll
                        for (int k = 0; k < K; ++k) { g1.s[k] = (v2f){0.f, 0.f}; if constexpr (FBX || C::FOLDQ) h1.s[k] = (v2f){0.f, 0.f}; }
                    }
                }
                if constexpr (K == 1) { v0 = g0.s[0]; v1 = g1.s[0]; if constexpr (SEP) { u0 = h0.s[0]; u1 = h1.s[0]; } }
                else if constexpr (SPLIT) {
#pragma unroll
                    for (int k = 0; k < K; ++k) { v0 = w[k].x * g0.s[k] + v0; v1 = w[k].y * g1.s[k] + v1; }
                    if constexpr (RECIP) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { v0 = w[k].x * h0.s[k] + v0; v1 = w[k].y * h1.s[k] + v1; }
                    }
                    if constexpr (SEP) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { u0 = w[k].x * h0.s[k] + u0; u1 = w[k].y * h1.s[k] + u1; }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) { A0 = w[k].x * g0.s[k] + A0; A1 = w[k].y * g1.s[k] + A1; }
                    if constexpr (TWO) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { B0 = w[k].x * h0.s[k] + B0; B1 = w[k].y * h1.s[k] + B1; }
                    }
                }
                if constexpr (RECIP && K == 1) { v0 += h0.s[0]; v1 += h1.s[0]; }
            } else {
                taps_f16 g0, g1, h0, h1;
                lds_issue<K, (GSET * MB + 2 * p) * WB>(g0, ad0); lds_issue<K, (GSET * MB + 2 * p + 1) * WB>(g1, ad1);
                if constexpr (TWO) { lds_issue<K, (HSET * MB + 2 * p) * WB>(h0, ad0); lds_issue<K, (HSET * MB + 2 * p + 1) * WB>(h1, ad1); }
                if constexpr (hp == 0 && K > 1) weights2<INTERP>(s, w);  // overlaps the LDS latency
                if constexpr (TWO) lds_fence2<K>(g0, g1, h0, h1, w); else lds_fence<K>(g0, g1, w);
                if constexpr (DIAG) {                 // (as for fp32 data above)
                    if (m < n)      { for (int k = 0; k < K; ++k) g0.r[k] = 0u; }
                    if constexpr (C::FOLD) {
                        if constexpr (TWO) { if (m < n) { for (int k = 0; k < K; ++k) h0.r[k] = 0u; } }
                    } else {
                        if (m <= n)     { for (int k = 0; k < K; ++k) h0.r[k] = 0u; }
                        if (m + 1 <= n) { for (int k = 0; k < K; ++k) h1.r[k] = 0u; }
                    }
                }
                if constexpr (TAIL) {
                    if (!upper) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { g1.r[k] = 0u; if constexpr (FBX || C::FOLDQ) h1.r[k] = 0u; }
                    }
                }
                if constexpr (K == 1) {
                    v0 = half2_to_v2f(g0.r[0]); v1 = half2_to_v2f(g1.r[0]);
                    if constexpr (SEP) { u0 = half2_to_v2f(h0.r[0]); u1 = half2_to_v2f(h1.r[0]); }
                    if constexpr (RECIP) { v0 += half2_to_v2f(h0.r[0]); v1 += half2_to_v2f(h1.r[0]); }
                } else if constexpr (SPLIT) {
#pragma unroll
                    for (int k = 0; k < K; ++k) { mix_mac(v0, g0.r[k], w[k].x); mix_mac(v1, g1.r[k], w[k].y); }
                    if constexpr (RECIP) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { mix_mac(v0, h0.r[k], w[k].x); mix_mac(v1, h1.r[k], w[k].y); }
                    }
                    if constexpr (SEP) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { mix_mac(u0, h0.r[k], w[k].x); mix_mac(u1, h1.r[k], w[k].y); }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) { mix_mac(A0, g0.r[k], w[k].x); mix_mac(A1, g1.r[k], w[k].y); }
                    if constexpr (TWO) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { mix_mac(B0, h0.r[k], w[k].x); mix_mac(B1, h1.r[k], w[k].y); }
                    }
                }
            }
            if constexpr (C::WST && hp == 0) {        // (the tap fence above waited for every LDS read of the iteration)
                asm volatile("" : "+v"(wv), "+v"(xv));
                wr0 = wv.x; wi0 = wv.y; wr1 = wv.z; wi1 = wv.w;
                xr0 = xv.x; xi0 = xv.y; xr1 = xv.z; xi1 = xv.w;
            }
            if constexpr (CHECK) {                    // edge rule: all taps in [0,T) and tau >= 0
                const uint32_t mb = upper ? m + 1 : m;
                const int ws0 = Abase[m] + bn, ws1 = Abase[mb] + bn;
                const float lo0 = tapinfo<INTERP>::LO - 0.5f - (float)ws0, hi0 = (float)(T - K + 1 - ws0) - 0.5f;
                const float lo1 = tapinfo<INTERP>::LO - 0.5f - (float)ws1, hi1 = (float)(T - K + 1 - ws1) - 0.5f;
                const bool k0 = (t.x >= lo0) && (t.x < hi0), k1 = (t.y >= lo1) && (t.y < hi1) && upper;
                v0 = k0 ? v0 : (v2f){0.f, 0.f}; v1 = k1 ? v1 : (v2f){0.f, 0.f};
                if constexpr (SEP) { u0 = k0 ? u0 : (v2f){0.f, 0.f}; u1 = k1 ? u1 : (v2f){0.f, 0.f}; }
            }
            if constexpr (C::BPIX) {                  // pixel x block-element weights; a zero weight never samples its trace (src/bf.cu:122,126)
                {
                    const v2f b2 = bw[p];
                    v0 = b2.x != 0.f ? v0 * b2.x : (v2f){0.f, 0.f};
                    v1 = b2.y != 0.f ? v1 * b2.y : (v2f){0.f, 0.f};
                }
            }
            if constexpr (FMOD) {                     // reference src/bf.cu:117: w = exp(2j pi fmod tau), tau*fs = t + 1/2 + (A[m] + B[n]) - OFF
                // phase in cycles = t*f + frac((A[m] + 1/2 - OFF)*f) + frac(B[n]*f), f = fmod/fs: the two constants were tabulated in
                // fp64 by the prologue (Aext / Bext are free by now); v_sin / v_cos take revolutions and reduce the argument themselves
                const uint32_t mb = upper ? m + 1 : m;
                const float f = (float)(P.fmod / fs);
                const v2f ph = t * f + ((v2f){Aext[m], Aext[mb]} + phB);
                const float c0 = __builtin_amdgcn_cosf(ph.x), s0 = __builtin_amdgcn_sinf(ph.x);
                const float c1 = __builtin_amdgcn_cosf(ph.y), s1 = __builtin_amdgcn_sinf(ph.y);
                if constexpr (!BF && !WTAB) {       // rotate and accumulate in one go: two packed FMAs per sample (acc += v*c; acc += (v.y, v.x)*(-s, s))
                    rot_acc(A0, v0, c0, s0); rot_acc(A1, v1, c1, s1);
                    if constexpr (FBX || C::FOLDQ) { rot_acc(B0, u0, c0, s0); rot_acc(B1, u1, c1, s1); }
                } else {
                    v0 = (v2f){v0.x * c0 - v0.y * s0, v0.x * s0 + v0.y * c0};
                    v1 = (v2f){v1.x * c1 - v1.y * s1, v1.x * s1 + v1.y * c1};
                    if constexpr (SEP) {
                        u0 = (v2f){u0.x * c0 - u0.y * s0, u0.x * s0 + u0.y * c0};
                        u1 = (v2f){u1.x * c1 - u1.y * s1, u1.x * s1 + u1.y * c1};
                    }
                }
            }
            if constexpr (BF) {                       // 'BF' (src/bf.cu:134-135): the pair's weighted sample IS the output, plane nm of y
                if constexpr (WTAB) {
                    v0 = (v2f){wr0 * v0.x - wi0 * v0.y, wr0 * v0.y + wi0 * v0.x};
                    v1 = (v2f){wr1 * v1.x - wi1 * v1.y, wr1 * v1.y + wi1 * v1.x};
                }
                if (in_shard()) {
                    float2 *pl0 = (float2 *)P.y + ((size_t)n * P.bf_pn + (size_t)m * P.bf_pm) * P.y_ld;      // uniform
                    asm volatile("" : "+s"(pl0));
                    uint32_t po = pofs;
                    asm volatile("" : "+v"(po));
                    pl0[po] = make_float2(v0.x, v0.y);
                    if (upper) { float2 *pl1 = pl0 + P.bf_pm * P.y_ld; asm volatile("" : "+s"(pl1)); pl1[po] = make_float2(v1.x, v1.y); }
                }
            } else if constexpr (WTAB) {
                if constexpr (C::WST && WZ) {         // a zero weight never samples its trace (non-finite data stay out): uniform, scalar tests --
                  if (wmixed) {                       // and only in stages where a zero shares a transmit pair with a non-zero weight
                    const v2f z = {0.f, 0.f};
                    if (!((wmask >> (2 * p)) & 1ull)) { v0 = z; if constexpr (FBX) u0 = z; }
                    if (!((wmask >> (2 * p + 1)) & 1ull)) { v1 = z; if constexpr (FBX) u1 = z; }
                    if constexpr (SW) { if (!((xmask >> (2 * p)) & 1ull)) { u0 = z; } if (!((xmask >> (2 * p + 1)) & 1ull)) { u1 = z; } }
                  }
                }
                wgt_acc(A0, v0, wr0, wi0); wgt_acc(A1, v1, wr1, wi1);             // complex weight folded into the accumulation
                if constexpr (FBX) { wgt_acc(B0, u0, wr0, wi0); wgt_acc(B1, u1, wr1, wi1); }
                if constexpr (SW) { wgt_acc(B0, u0, xr0, xi0); wgt_acc(B1, u1, xr1, xi1); }      // mirror pairs, their own weights
            } else if constexpr (FMOD) {              // (accumulated by the rotation above)
            } else if constexpr (SPLIT || K == 1) { A0 += v0; A0 += v1; if constexpr (FBX || C::FOLDQ) { B0 += u0; B0 += u1; } }
        });
    });
}

// fp64 data: one transmit at a time, everything in double -- the reference's double kernel computes delays, weights and sums in
// fp64 (src/bf.cu:144-151) and the parity bar of this precision is 1e-10 of the image maximum.  Same scheme as above: t = ra + rb,
// k = rint(t) read off the low word of (t + 1.5*2^52), s = t - k, K x ds_read_b128, weights between issue and wait, 2K v_fma_f64.
template <class C> template <bool CHECK, bool TAILV>
__device__ __forceinline__ void Tile<C>::pairs_f64(uint32_t n, uint32_t m0, int bn, double rb, uint32_t cbase) {
    constexpr int K = C::K, MB = C::MB, WB = C::WB, INTERP = C::INTERP;
    double lead[4];
    weights1_lead<INTERP, !C::FMOD>(lead);
    const double fq = C::FMOD ? P.fmod / fs : 0.0;    // cycles per sample (uniform)
    unroll<MB>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (!hooks::no_fair_prio && (p * 4) % MB == 0) __builtin_amdgcn_s_setprio(QDAS_PAIR_PRIO(3 - (p * 4) / MB));     // fair progress, see pairs_plain
        const uint32_t m = m0 + p;
        if constexpr (TAILV) { if (m >= M) return; }
        if constexpr (C::FMOD) __builtin_amdgcn_sched_barrier(0);   // one transmit at a time: the phasor polynomial's registers never meet the next transmit's taps
        double wr = 1.0, wi = 0.0;                                   // (the other three waves of the SIMD cover the LDS latency: ~60 fp64 operations per pair)
        if constexpr (C::WTAB) {
            const double2 wv = ((const double2 *)P.wtab)[n + (size_t)N * m];
            wr = wv.x; wi = wv.y;
            if (wr == 0.0 && wi == 0.0) return;          // zero weight: skip (src/bf.cu:122,126)
        }
        const double t = rad[p] + rb;
        const double tm = t + MAGIC64;
        const double s = t - (tm - MAGIC64);
        const uint32_t ad = (uint32_t)__double2loint(tm) * 16u + cbase;
        taps_f64 g;
        lds_issue<K, p * WB>(g, ad);
        double w[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (K > 1) weights1<INTERP>(s, w, lead);  // overlaps the LDS latency
        lds_fence<K>(g, w);
        double &ar = dacc[2 * (p & 1)], &ai = dacc[2 * (p & 1) + 1];
        if constexpr (K > 1 && !CHECK && !C::WTAB && !C::FMOD) {
#pragma unroll
            for (int k = 0; k < K; ++k) { ar = __builtin_fma(w[k], g.s[k].x, ar); ai = __builtin_fma(w[k], g.s[k].y, ai); }
        } else {
            double vr = g.s[0].x, vi = g.s[0].y;
            if constexpr (K > 1) {
                vr *= w[0]; vi *= w[0];
#pragma unroll
                for (int k = 1; k < K; ++k) { vr = __builtin_fma(w[k], g.s[k].x, vr); vi = __builtin_fma(w[k], g.s[k].y, vi); }
            }
            if constexpr (CHECK) {                    // edge rule: all taps in [0,T) and tau >= 0
                const int ws = Abase[m] + bn;
                const double lo = (double)tapinfo<INTERP>::LO - 0.5 - (double)ws, hi = (double)(T - K + 1 - ws) - 0.5;
                const bool keep = (t >= lo) && (t < hi);
                vr = keep ? vr : 0.0; vi = keep ? vi : 0.0;
            }
            if constexpr (C::FMOD) {                  // src/bf.cu:117 in double: exp(2j pi fmod tau), tau*fs = t + 1/2 + (A[m] + B[n]) - OFF
                __builtin_amdgcn_sched_barrier(0);    // after the taps are consumed: the polynomial's temporaries never overlap them
                const double ph = (t + ((double)(Abase[m] + bn) + (0.5 - (double)tapinfo<INTERP>::OFF))) * fq;
                double c, sn;
                sincos2pi_f64(ph - __builtin_rint(ph), c, sn);
                const double xr = vr * c - vi * sn, xi = vr * sn + vi * c;
                vr = xr; vi = xi;
            }
            if constexpr (C::WTAB) { ar += wr * vr - wi * vi; ai += wr * vi + wi * vr; }
            else { ar += vr; ai += vi; }
        }
    });
}

// Software-pipelined loop (two window sets, 4 taps, full interior block, no per-sample post-processing): the first-set taps of
// iteration p+1 are requested before the MACs of iteration p, so the LDS pipe always has work queued and the counted wait (newest
// 8 reads stay in flight) rarely stalls.  A unit = (transmit pair p, frame pair hp); hp only with four frames per launch.
template <class C> __device__ __forceinline__ void Tile<C>::pairs_pipelined(float rb, uint32_t cbase) {
    constexpr int K = C::K, MB = C::MB, WB = C::WB, INTERP = C::INTERP, NHP = C::NHP;
    constexpr bool FB4 = C::QUAD, F32 = C::F32;      // (four window sets, two passes: four frames, or reciprocal + lateral-mirror mode)
    constexpr int NP = MB / 2, NU = NP * NHP;
    using taps_t = std::conditional_t<F32, taps_f32, taps_f16>;
    taps_t gd0[2], gd1[2];                 // first-set taps of the two halves, double-buffered over units
    v2f sv[2];
    uint32_t a0v[2], a1v[2];
    v2f w[4];
    auto index = [&](auto uc) {            // index math (first unit of a pair) + first-set reads of unit u
        constexpr int u = decltype(uc)::value, p = u / NHP, hp = u % NHP, GSET = FB4 ? 2 * hp : 0;
        if constexpr (hp == 0) {
            const v2f t = ra[p] + rb;
            const v2f tm = t + MAGIC;
            sv[p & 1] = t - (tm - MAGIC);
            a0v[p & 1] = __float_as_uint(tm.x) * (uint32_t)C::SB + cbase;
            a1v[p & 1] = __float_as_uint(tm.y) * (uint32_t)C::SB + cbase;
        }
        if constexpr (F32 && hooks::no_tap_reads) { for (int k = 0; k < 4; ++k) { gd0[u & 1].s[k] = (v2f){sv[p & 1].x, rb}; gd1[u & 1].s[k] = (v2f){rb, sv[p & 1].y}; } }
        else { lds_issue<K, (GSET * MB + 2 * p) * WB>(gd0[u & 1], a0v[p & 1]); lds_issue<K, (GSET * MB + 2 * p + 1) * WB>(gd1[u & 1], a1v[p & 1]); }
    };
    index(std::integral_constant<int, 0>{});
    unroll<NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value, p = u / NHP, hp = u % NHP, HSET = FB4 ? 2 * hp + 1 : 1;
        if constexpr (!hooks::no_fair_prio && (u * 4) % NU == 0) __builtin_amdgcn_s_setprio(QDAS_PAIR_PRIO(3 - (u * 4) / NU));     // fair progress, see pairs_plain
        taps_t h0, h1;
        if constexpr (F32 && hooks::no_tap_reads) { for (int k = 0; k < 4; ++k) { h0.s[k] = sv[p & 1]; h1.s[k] = (v2f){sv[p & 1].y, sv[p & 1].x}; } }
        else { lds_issue<K, (HSET * MB + 2 * p) * WB>(h0, a0v[p & 1]); lds_issue<K, (HSET * MB + 2 * p + 1) * WB>(h1, a1v[p & 1]); }
        if constexpr (u + 1 < NU) index(std::integral_constant<int, u + 1>{});
        if constexpr (hp == 0) {
            if constexpr (hooks::trivial_weights) { w[0] = sv[p & 1]; w[1] = sv[p & 1] + 1.f; w[2] = sv[p & 1] * 2.f; w[3] = 1.f - sv[p & 1]; }
            else weights2<INTERP>(sv[p & 1], w);
        }
        if constexpr (u + 1 < NU) lds_fence2_keep<8>(gd0[u & 1], gd1[u & 1], h0, h1, w);
        else                      lds_fence2_keep<0>(gd0[u & 1], gd1[u & 1], h0, h1, w);
        v2f &A0 = *(FB4 ? (hp ? &acc2 : &acc) : &acc), &A1 = *(FB4 ? (hp ? &acc2 : &acc) : ((C::FOLDQ || C::ONEACC) ? &acc : &acc1));      // (FOLDQ: one accumulator per pixel, as MIRQ -- the 32-transmit stages need the registers)
        // (reciprocal mode: both mirror halves share ONE accumulator -- three in all; with 32-transmit stages the fourth costs the two
        //  registers that would otherwise spill, and measures the same: profiles/r02/exp_prio.txt)
        v2f &B0 = *((C::MIRQ && !C::FOLD) ? (hp ? &acc2 : &acc) : FB4 ? (hp ? &acc3 : &acc1) : &acc2), &B1 = *((C::MIRQ && !C::FOLD) ? (hp ? &acc2 : &acc) : FB4 ? (hp ? &acc3 : &acc1) : ((C::SYM || C::ONEACC) ? &acc2 : &acc3));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            tap_mac(A0, gd0[u & 1], k, w[k].x); tap_mac(A1, gd1[u & 1], k, w[k].y);
            tap_mac(B0, h0, k, w[k].x);         tap_mac(B1, h1, k, w[k].y);
        }
    });
}

}  // namespace qdas
