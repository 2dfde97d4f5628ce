#pragma once
// das_tile_impl.h -- the fused, LDS-staged delay-and-sum kernel for gfx950 (MI355X).
//
// Replaces the reference launch of `DASf` / `DASh` (reference src/bf.cu:153-171, body src/bf.cu:49-142) for the bulk of the
// work: 'DAS' (sum over both apertures) with fp32 / fp16 data, 'SYN' / 'MUL' (keep one aperture) with fp32 data; scalar sound
// speed or a per-pixel map; pixel-independent apodization (folded by the host into one N x M table) plus one pixel x receiver
// array or generated rule.  Everything else is served by das_generic.hip.  (Instantiated per launch configuration in
// das_tile_{f32,sym,f16,f32x2,f16x2}.hip; dispatch in das_tile.hip.)
//
// Design (MI355X-first, not a re-tiling of the reference's one-thread-per-pixel loop):
//
//  * A workgroup (16 waves) owns a TILE of 1024 pixels, 2^t (fast image axis I1 = depth) x 1024/2^t columns; a wave covers
//    2^w x 64/2^w of them.  The plan probes the tile footprint (largest that fits the LDS window) and picks the wave footprint
//    from an LDS bank model (8 x 8 pixels on a lambda/4 grid: the 32 lanes of an access group read <= 32 consecutive samples).
//    Two consecutive transmits (m, m+1) of the same pixel ride in the two halves of packed-fp32 (v_pk_*_f32) instructions.
//  * Time of flight is separable: tau*fs + off = a(i,m) + b(i,n).  The prologue computes tile-wide integer window bases
//    A[m] <= a, B[n] <= b and extents (fp32 estimates with explicit error margins); afterwards each lane only carries the
//    small fp32 residuals of the fp64 delays, ra = a - A[m] - 1/2, rb = b - B[n] (exact to ~1e-5 sample; the reference's fp32
//    tau carries ~1e-4 sample at tau*fs ~ 2000).  Per pair:
//        t = ra[m] + rb;  k = rint(t) (magic-number add);  s = t - k  in [-1/2, 1/2];
//        first tap = window[k], weights = even/odd polynomials in s.
//    "Block" elements (the MB transmits of a stage) and the "stage" element (its receiver) each have a delay kind {distance,
//    signed distance, plane wave}; for 'MUL' the host swaps the two apertures' roles.
//  * For every STAGE (receiver n, block of MB transmits) the workgroup stages MB fast-time WINDOWS (W samples starting at
//    A[m]+B[n]) of the channel data into LDS with LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write;
//    out-of-buffer lanes deliver 0), coalesced along fast time and double-buffered against the compute of the previous stage;
//    one buffer descriptor per transmit block, B[n] prefetched a stage ahead, every LDS read of a stage issued before its DMA
//    (hipcc orders a later LDS read behind the DMA's vmcnt).  Taps are gathered with ds_read_b64 / ds_read_b32 from inline asm
//    with immediate (window, tap) offsets (hipcc would merge them into ds_read2_b64: 2x slower for this gather).
//  * A stage's SECOND window set holds, depending on the mode: the mirror traces x[:,m,n] (reciprocal mode, SYM: Pv == Pr, so
//    tau(n,m) == tau(m,n) and index + weights serve both traces of an unordered pair), or the same traces of the NEXT FRAME
//    (FB2: index + weights serve two frames).
//  * All resident workgroups walk the traces in the same order (columns-fastest tile order, XCD-aware remap), so the channel
//    data is served by L2 to all but the first of the tiles of a depth band.
//  * Tiles whose windows all lie inside [0, T) run a branch-free loop; tiles that touch the ends of the record run the checked
//    loop (edge rule of SURVEY.md section 8 a5).  A tile whose delay spread does not fit W appends itself to a fallback list
//    and is processed by the generic kernel afterwards -- results never depend on the geometry being "image like".
//  * Lanczos weights: even/odd-split polynomials (lanczos_poly.h), no transcendentals; fp16 taps are consumed by
//    v_fma_mix_f32 (fp16 x fp32 + fp32) and accumulated in fp32 (the reference accumulates in half2, src/bf.cu:170).
//  * A pixel x receiver weight (an I1 x I2 x I3 x N array, or a rule evaluated from the geometry: qdas.h QDAS_RXAPOD_*) does
//    not depend on the transmit: it multiplies the stage's partial sum once per (pixel, receiver), one stage ahead, and a wave
//    whose 64 weights are all zero skips the stage's gathers altogether.
//  * 'SYN' / 'MUL': a stage belongs to one plane of the output; its sum is added with non-returning fp32 atomics.
//  * Few tiles (pixel slab of a multi-GPU job, small image): ksplit workgroups per tile, each a slice of the aperture,
//    partial images reduced in a fixed order.
#include "qdas_device.h"
#include "qdas_kernels.h"
#include "lanczos_poly.h"
#include "das_tile_cfg.h"
#include <type_traits>
#include <utility>
#include <cstdlib>

#ifndef QDAS_ABL
#define QDAS_ABL 0   // ablation bits for profiling builds only (tools/ablate.sh); 0 in the product
#endif

#ifndef QDAS_PROF
#define QDAS_PROF 0  // 1: in-kernel phase timers (s_memtime) of waves 0 and 15 of every workgroup -> tools/phase_timers.py; 0 in the product
#endif
#if QDAS_PROF
__device__ unsigned long long qdas_prof_buf[2 * 8 * 8192];
extern "C" int qdas_debug_read_prof(unsigned long long *dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(qdas_prof_buf), n * sizeof(unsigned long long));
}
#define QDAS_TICK() ({ asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long t_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_; })
#endif

namespace qdas {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float MAGIC = 12582912.0f;          // 1.5 * 2^23: (t + MAGIC) has rint(t) in its low mantissa bits
constexpr uint32_t MAGIC_BITS = 0x4B400000u;

template <int INTERP> struct tapinfo {
    static constexpr int K = interp_taps(INTERP);
    // offset folded into a(i,m) so that floor(a + b) is the FIRST tap:
    //   nearest: round(tau) = floor(tau + 1/2); linear: floor(tau); 4-tap: floor(tau) - 1
    static constexpr double OFF = (INTERP == 0) ? 0.5 : (K == 2 ? 0.0 : -1.0);
    // lowest admissible value of (tau*fs + OFF): tau >= 0 AND first tap >= 0
    static constexpr float LO = (INTERP == 0) ? 0.5f : 0.0f;
};

template <int D> __device__ __forceinline__ v2f horner2(const float (&c)[D + 1], v2f q) {
    v2f r = {c[D], c[D]};
#pragma unroll
    for (int k = D - 1; k >= 0; --k) r = r * q + (v2f){c[k], c[k]};
    return r;
}

// Tap weights for s = u - 1/2 (two columns packed).  w[k] multiplies tap (first + k).
template <int INTERP> __device__ __forceinline__ void weights2(v2f s, v2f w[4]) {
    if constexpr (INTERP == 1 || INTERP == 4) {            // lerp (reference src/interpd.cu:84)
        w[0] = 0.5f - s; w[1] = 0.5f + s;
    } else if constexpr (INTERP == 2) {                    // Catmull-Rom, exact even/odd split about u = 1/2
        const v2f q = s * s;
        const v2f ei = 0.5625f - 0.25f * q, oi = -1.375f + 1.5f * q;
        const v2f eo = -0.0625f + 0.25f * q, oo = 0.125f - 0.5f * q;
        w[1] = ei + s * oi; w[2] = ei - s * oi; w[0] = eo + s * oo; w[3] = eo - s * oo;
    } else if constexpr (INTERP == 3) {                    // Lanczos (a = 2), lanczos_poly.h
        constexpr float EI[] = QDAS_LANCZOS_EI, OI[] = QDAS_LANCZOS_OI, EO[] = QDAS_LANCZOS_EO, OO[] = QDAS_LANCZOS_OO;
        const v2f q = s * s;
        const v2f ei = horner2<sizeof(EI) / 4 - 1>(EI, q), oi = horner2<sizeof(OI) / 4 - 1>(OI, q);
        const v2f eo = horner2<sizeof(EO) / 4 - 1>(EO, q), oo = horner2<sizeof(OO) / 4 - 1>(OO, q);
        w[1] = ei + s * oi; w[2] = ei - s * oi; w[0] = eo + s * oo; w[3] = eo - s * oo;
    } else if constexpr (INTERP == 5) {                    // the Horner lines the device code executes (src/interpd.cu:103-106)
        const v2f u = s + 0.5f;
        w[0] = 0.5f * (u * (-1.0f + u * (2.0f * u - 1.0f)));
        w[1] = 0.5f * (2.0f + u * (u * (-5.0f * u + 3.0f)));
        w[2] = 0.5f * (u * (1.0f + u * (4.0f * u - 3.0f)));
        w[3] = 0.5f * (u * (u * (1.0f - u)));
    }
}

// DPP wave reductions (VALU speed; the result is valid in lane 63 only): quad swaps, half-row / row mirrors, then the row
// broadcasts of GFX9 (lane 15 -> next row, lane 31 -> rows 2-3).  __shfl_xor compiles to ds_bpermute_b32: six dependent
// LDS-pipe round trips per reduction.
template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_min63(float v) {
    v = fminf(v, dppf<0xB1>(v)); v = fminf(v, dppf<0x4E>(v)); v = fminf(v, dppf<0x141>(v)); v = fminf(v, dppf<0x140>(v));
    v = fminf(v, dppf<0x142, 0xa>(v)); v = fminf(v, dppf<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ float wave_max63(float v) {
    v = fmaxf(v, dppf<0xB1>(v)); v = fmaxf(v, dppf<0x4E>(v)); v = fmaxf(v, dppf<0x141>(v)); v = fmaxf(v, dppf<0x140>(v));
    v = fmaxf(v, dppf<0x142, 0xa>(v)); v = fmaxf(v, dppf<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- LDS tap gathers.  fp32 data: 4 x ds_read_b64 from inline asm; the results are only usable
//      after lds_fence(), which ties the registers through the s_waitcnt so the compiler cannot
//      hoist a consumer above it (cdna_hip_programming.md section 5.4 rule 18 / 5.7).
struct taps_f32 { v2f s[4]; };
template <int K, int OFF> __device__ __forceinline__ void lds_issue(taps_f32 &t, uint32_t addr) {
    if constexpr (K == 4)
        asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\tds_read_b64 %3, %4 offset:%8"
                     : "=&v"(t.s[0]), "=&v"(t.s[1]), "=&v"(t.s[2]), "=&v"(t.s[3]) : "v"(addr), "n"(OFF), "n"(OFF + 8), "n"(OFF + 16), "n"(OFF + 24));
    else if constexpr (K == 2)
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(t.s[0]), "=&v"(t.s[1]) : "v"(addr), "n"(OFF), "n"(OFF + 8));
    else
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(t.s[0]) : "v"(addr), "n"(OFF));
}
// The weights are tied through the wait as well, so that their evaluation is scheduled BEFORE it
// (between the issue of the loads and the wait: that is what hides the LDS latency).
__device__ __forceinline__ void lds_fence(taps_f32 &a, taps_f32 &b, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(b.s[2]), "+v"(b.s[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
__device__ __forceinline__ void lds_fence2(taps_f32 &a, taps_f32 &b, taps_f32 &c, taps_f32 &d, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(b.s[2]), "+v"(b.s[3]),
                   "+v"(c.s[0]), "+v"(c.s[1]), "+v"(c.s[2]), "+v"(c.s[3]), "+v"(d.s[0]), "+v"(d.s[1]), "+v"(d.s[2]), "+v"(d.s[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
// counted variant for the software-pipelined loop: the NEWEST `KEEP` LDS reads (the next iteration's direct taps) stay in flight
template <int KEEP> __device__ __forceinline__ void lds_fence2_keep(taps_f32 &a, taps_f32 &b, taps_f32 &c, taps_f32 &d, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%20)"
                 : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(b.s[2]), "+v"(b.s[3]),
                   "+v"(c.s[0]), "+v"(c.s[1]), "+v"(c.s[2]), "+v"(c.s[3]), "+v"(d.s[0]), "+v"(d.s[1]), "+v"(d.s[2]), "+v"(d.s[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])
                 : "n"(KEEP));
}
template <int... Is, typename F> __device__ __forceinline__ void unroll_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void unroll(F &&f) { unroll_impl(std::make_integer_sequence<int, N>{}, f); }

// fp16 data: 4-byte samples {re, im}.  K x ds_read_b32 with immediate offsets from inline asm (the same issue / weights / fence
// pattern as fp32); the MAC is v_fma_mix_f32 -- fp16 tap x fp32 weight + fp32 accumulator in ONE instruction per component, so the
// taps are never converted (hipcc's own choice is 2 v_cvt_f32_f16 + 1 v_pk_fma_f32 per tap: 1.8x the issue cycles).
struct taps_f16 { uint32_t r[4]; };
template <int K, int OFF> __device__ __forceinline__ void lds_issue(taps_f16 &t, uint32_t addr) {
    if constexpr (K == 4)
        asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8"
                     : "=&v"(t.r[0]), "=&v"(t.r[1]), "=&v"(t.r[2]), "=&v"(t.r[3]) : "v"(addr), "n"(OFF), "n"(OFF + 4), "n"(OFF + 8), "n"(OFF + 12));
    else if constexpr (K == 2)
        asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4" : "=&v"(t.r[0]), "=&v"(t.r[1]) : "v"(addr), "n"(OFF), "n"(OFF + 4));
    else
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(t.r[0]) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_fence(taps_f16 &a, taps_f16 &b, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
__device__ __forceinline__ void lds_fence2(taps_f16 &a, taps_f16 &b, taps_f16 &c, taps_f16 &d, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]),
                   "+v"(c.r[0]), "+v"(c.r[1]), "+v"(c.r[2]), "+v"(c.r[3]), "+v"(d.r[0]), "+v"(d.r[1]), "+v"(d.r[2]), "+v"(d.r[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
template <int KEEP> __device__ __forceinline__ void lds_fence2_keep(taps_f16 &a, taps_f16 &b, taps_f16 &c, taps_f16 &d, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%20)"
                 : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]),
                   "+v"(c.r[0]), "+v"(c.r[1]), "+v"(c.r[2]), "+v"(c.r[3]), "+v"(d.r[0]), "+v"(d.r[1]), "+v"(d.r[2]), "+v"(d.r[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])
                 : "n"(KEEP));
}
__device__ __forceinline__ void mix_mac(v2f &acc, uint32_t tap, float w) {       // acc += w * (float2)tap
    float ar = acc.x, ai = acc.y;
    asm("v_fma_mix_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(ar), "+v"(ai) : "v"(tap), "v"(w));
    acc = (v2f){ar, ai};
}
// acc += w * tap k, either data type (software-pipelined loop)
__device__ __forceinline__ void tap_mac(v2f &acc, const taps_f32 &t, int k, float w) { acc = w * t.s[k] + acc; }
__device__ __forceinline__ void tap_mac(v2f &acc, const taps_f16 &t, int k, float w) { mix_mac(acc, t.r[k], w); }
__device__ __forceinline__ v2f half2_to_v2f(uint32_t v) {
    return (v2f){__half2float(__ushort_as_half((unsigned short)(v & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(v >> 16)))};
}

__device__ __forceinline__ float2   zero_of(const float2 *)   { return make_float2(0.f, 0.f); }
__device__ __forceinline__ uint32_t zero_of(const uint32_t *) { return 0u; }

// ---- cold fp64 code, kept OUT of line on purpose.  Inlined into the stage loop (16 unrolled copies of the transmit-block refresh,
//      the fp64 acos / cos of the generated receive apodization) it inflated the live ranges around the pair loop until the
//      register allocator parked the lane's transmit residuals ra[] in scratch and re-loaded them every stage (round 1: 170-230
//      spilled VGPRs, 340-416 B of scratch per lane in every general instantiation).  As calls they cost a few scalar
//      instructions once per transmit block / stage and the kernels have no scratch at all (tools/kernel_regs.py).
typedef __attribute__((address_space(3))) const float lds_cfloat;
// a(i,m) - A[m] - 1/2 of one (pixel, block element): geometry tables in LDS, fp64 (reference src/bf.cu:104-108,114)
static __device__ __noinline__ float block_residual(float px, float py, float pz, double cf, double fs, int kindB, lds_cfloat *Pv, lds_cfloat *Nv,
                                                    uint32_t m, int Abase_m, double off) {
    const double rx = (double)px - (double)Pv[4 * m], ry = (double)py - (double)Pv[4 * m + 1], rz = (double)pz - (double)Pv[4 * m + 2];
    const double dot = kindB ? rx * (double)Nv[3 * m] + ry * (double)Nv[3 * m + 1] + rz * (double)Nv[3 * m + 2] : 0.0;
    double dv = dot;
    if (kindB != 2) {
        const double d2 = rx * rx + ry * ry + rz * rz;
        const float s0 = __builtin_sqrtf((float)d2);                  // fp32 seed + one Newton step (as dsqrt in the kernel)
        const double sd = (double)s0;
        const double r = __builtin_fma(-sd, sd, d2);
        const double len = __builtin_fma(r, (double)(0.5f * __builtin_amdgcn_rcpf(fmaxf(s0, 1.0e-30f))), sd);
        dv = kindB == 0 ? len : copysign(len, dot);
    }
    return (float)((dv * cf - (double)Pv[4 * m + 3] * fs + off) - ((double)Abase_m + 0.5));
}
// generated pixel x receiver weight (qdas.h QDAS_RXAPOD_*): element position from the LDS record, normal by scalar loads
static __device__ __noinline__ float rx_apod_generated(int kind, double p0, double p1, float px, float py, float pz, float ex, float ey, float ez,
                                                       const float *rxn, uint32_t n) {
    const float nx = rxn ? rxn[3 * n] : 0.f, ny = rxn ? rxn[3 * n + 1] : 0.f, nz = rxn ? rxn[3 * n + 2] : 1.f;
    return (float)rx_apod_weight(kind, p0, p1, (double)px - (double)ex, (double)py - (double)ey, (double)pz - (double)ez,
                                 (double)nx, (double)ny, (double)nz, (double)px, (double)pz, (double)ex);
}

// CFG: WAVES waves (= image columns) per workgroup, MB transmits per stage, W samples per window,
//      NBUF window buffers (NBUF-1 stages of LDS-DMA in flight), PSZ bytes per lane and DMA piece (12|16),
//      BPC workgroups per CU the register budget is sized for.
//      PROBE: plan-time variant that stops after the window-fit test (a kernel of its own name, so that profiles of
//      das_tile_kernel<..., false> hold full frames only).
//      FB2: two FRAMES per launch (x, x + x_fstride -> y, y + y_fstride): the second window set of a stage holds the same
//      traces of the next frame, so tap index and weights -- which depend on the geometry only -- are computed once for
//      both frames (the reference launches one kernel per frame, kern/das_spec.m:371).  Structurally the reciprocal mode's
//      "mirror" set with another source and a separate sum.
//      FB4: four frames per launch: four window sets of MB = 8 transmits; the pair loop makes two passes (frames 0-1, 2-3).
template <int INTERP, typename ST, bool FMOD, bool WTAB, bool SYM, bool FB2, bool FB4, int WAVES, int MB, int W, int NBUF, int PSZ, int BPC, bool PROBE, bool BIG = false, bool LUT = false>
__global__ void __launch_bounds__(WAVES * 64, WAVES * BPC / 4)
das_tile_kernel(const TileParams P) {
    constexpr bool FBX = FB2 || FB4;          // more than one frame per launch
    constexpr bool TWO = SYM || FBX;          // (at least) two window sets per stage: direct + (mirror | next frame)
    constexpr int NHP = FB4 ? 2 : 1;          // passes of the pair loop: one per frame pair
    constexpr int NFR = FB4 ? 4 : (FB2 ? 2 : 1);   // frames per launch
    constexpr int NW = FB4 ? 4 * MB : (TWO ? 2 * MB : MB);     // windows per LDS buffer
    static_assert(!SYM || !WTAB, "reciprocal mode: no weight table");
    static_assert(!(SYM && FBX) && !(FB2 && FB4), "reciprocal mode runs one frame per launch");
    static_assert(!BIG || (!SYM && !FBX), "the re-basing general kernel runs one frame per launch");
    static_assert(!LUT || (!SYM && !FBX && !BIG), "table-driven delays: general mode, one frame per launch");
    constexpr int K = tapinfo<INTERP>::K;
    constexpr int THREADS = WAVES * 64;
    constexpr int TX = WAVES;                 // waves per workgroup; a wave holds 1, 2 or 4 image columns (tz_log2)
    constexpr int WPW = FB4 ? 1 : MB / WAVES; // windows staged per wave and window set
    constexpr int SB = (int)sizeof(ST);       // bytes per complex sample
    constexpr bool F32 = (SB == 8);
    static_assert(FB4 ? (2 * MB == WAVES) : (MB % WAVES == 0 && MB % 2 == 0), "staging split");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if QDAS_PROF
    unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long pstart_ = QDAS_TICK();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: keeps everything derived from it in SGPRs
    const uint32_t M = (uint32_t)P.M, N = (uint32_t)P.N;
    const int T = (int)P.T;
    int   *Abase = (int *)smem;                       // [M]
    float *Aext  = (float *)(Abase + M);              // [M]
    float *Bext  = Aext + M;                          // [N]
    float4 *nrec = (float4 *)(smem + (((2 * M + N) * 4 + 15) & ~15u));   // [N] per receiver {window base B (int bits), x, y, z}: ONE broadcast read per stage
    float *PvL   = (float *)(nrec + N);               // [4M] (virtual) sources + t0
    float *NvL   = PvL + 4 * M;                       // [3M] transmit normals
    const uint32_t hdr = ((((2 * M + N) * 4 + 15) & ~15u) + 16 * N + 7 * M * 4 + 15) & ~15u;
    ST *win = (ST *)(smem + hdr);                     // [NBUF][NW][W]
    float *part = (float *)(smem + hdr);              // prologue scratch, aliases the windows
    const uint32_t win_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem) + hdr;

    // ---- which tile (XCD-aware: consecutive tile ids -> same XCD; dispatch is round-robin mod 8)
    const uint32_t nb = gridDim.x;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t q = nb / 8, r = nb % 8, xcd = bid % 8, k = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective remap
    }
    // columns fastest: the tiles an XCD runs concurrently sit at the SAME depth band, so they stage (nearly)
    // the same window of every trace and the XCD's L2 serves all but the first of them
    // split = slowest index: the workgroups an XCD runs concurrently work on the SAME slice of the aperture of neighbouring tiles
    const uint32_t ntile = P.tiles_x * P.tiles_z;
    const uint32_t split = bid / ntile, S = P.ksplit;
    bid -= split * ntile;
    const uint32_t tz = bid / P.tiles_x, txi = P.tile_x0 + bid % P.tiles_x;
    const uint32_t tile_id = tz + P.tiles_z * txi;

    // ---- my pixel: lane -> depth, wave -> column.  Out-of-image lanes are clamped onto a real
    //      pixel (keeps them inside the tile's delay window) and masked at the store.
    const uint64_t ncols = P.I2 * P.I3, i_end = P.i_begin + P.i_count;
    // Tile and wave footprints (uniform, chosen by the plan from the scan's delay gradient; qdas_api.hip choose_tile_shape):
    // the tile is (1 << tzl) pixels of I1 x (1024 >> tzl) columns -- as deep as the LDS window allows; inside it a wave covers
    // (1 << wzl) x (64 >> wzl) pixels -- as shallow as needed for the 32 lanes of an LDS access group to read <= 32
    // consecutive samples (conflict-free), e.g. 8 x 8 when the delay advances 2 samples per pixel of depth.
    const int tzl = P.tz_log2, wzl = P.wz_log2;
    const uint32_t wave_z = (uint32_t)wave & ((1u << (tzl - wzl)) - 1u), wave_c = (uint32_t)wave >> (tzl - wzl);
    const uint64_t i1 = ((uint64_t)tz << tzl) + (wave_z << wzl) + (uint32_t)(lane & ((1 << wzl) - 1));
    const uint64_t col = (uint64_t)txi * ((uint32_t)(TX * 64) >> tzl) + (wave_c << (6 - wzl)) + (uint32_t)(lane >> wzl);
    float px, py, pz;                                 // widened to fp64 where they are used
    const double fs = P.fs;
    double cf = P.cinv_fs;                            // samples per metre: scalar sound speed, or this pixel's entry of a sound-speed map
    // LUT: the delays come from host-supplied tables (tau_tx: I x M, tau_rx: I x N, in samples; the split-delay flavour
    // bfDASLUT / sample2sep / wsinterpd2 of the reference) instead of the geometry; everything downstream is the same kernel.
    uint64_t ipl = 0;                                 // my (clamped) pixel's row of the tables
    if constexpr (LUT) {
        px = py = pz = 0.f;
        const uint64_t i = (i1 < P.I1 ? i1 : P.I1 - 1) + P.I1 * (col < ncols ? col : ncols - 1);
        ipl = i < i_end ? i : i_end - 1;
    } else {
        const uint64_t i = (i1 < P.I1 ? i1 : P.I1 - 1) + P.I1 * (col < ncols ? col : ncols - 1);
        px = P.Pi[3 * i]; py = P.Pi[3 * i + 1]; pz = P.Pi[3 * i + 2];
        if (P.cinv_pix) cf = (double)P.cinv_pix[i] * fs;
    }
    const uint64_t Ilut = i_end;                      // (table-driven plans cover [0, I))
    // Delay model of the BLOCK elements (the MB "transmits" of a stage) and of the STAGE elements (its "receiver"):
    // 0 = distance, 1 = signed distance (focused wave: copysign by the normal, src/bf.cu:106-108), 2 = plane wave (dot product).
    // 'DAS' / 'SYN': block = transmits (kind from VS / DV), stage = receivers (distance).  'MUL' (keep the transmit dimension) runs
    // the same kernel with the roles swapped by the host: block = receivers, stage = transmits -- tables, strides and kinds swap.
    const int kindB = P.kindB, kindS = P.kindS;

    // sqrt in fp64 from an fp32 seed + one Newton step (rel. error ~1e-14; v_sqrt_f32 is 1 ulp)
    auto dsqrt = [](double d2) -> double {
        const float s0 = __builtin_sqrtf((float)d2);
        const double sd = (double)s0;
        const double r = __builtin_fma(-sd, sd, d2);
        // (v_rcp_f32 is plenty for the correction term; s0 == 0 gives r == 0 and inf*0 -> guard with a max)
        return __builtin_fma(r, (double)(0.5f * __builtin_amdgcn_rcpf(fmaxf(s0, 1.0e-30f))), sd);
    };
    // The geometry tables are read from global memory in the prologue and from their LDS copies in the
    // main loop: a vector-memory load there would sit behind the stage's LDS-DMA in the in-order vmcnt
    // queue and expose the DMA latency every stage (measured: 15 of 64 ms).
    const float *gPv = P.Pv, *gNv = P.Nv;
    auto a_of = [&](uint32_t m) -> double {              // tau_tx*fs - t0*fs + OFF, reference src/bf.cu:104-108,114
        if constexpr (LUT) return (double)P.lut_tx[ipl + Ilut * m] + tapinfo<INTERP>::OFF;
        const double rx = (double)px - (double)gPv[4 * m], ry = (double)py - (double)gPv[4 * m + 1], rz = (double)pz - (double)gPv[4 * m + 2];
        const double dot = kindB ? rx * (double)gNv[3 * m] + ry * (double)gNv[3 * m + 1] + rz * (double)gNv[3 * m + 2] : 0.0;
        double dv = dot;
        if (kindB != 2) { const double len = dsqrt(rx * rx + ry * ry + rz * rz); dv = kindB == 0 ? len : copysign(len, dot); }
        return dv * cf - (double)gPv[4 * m + 3] * fs + tapinfo<INTERP>::OFF;
    };
    auto b_at = [&](float ex, float ey, float ez) -> double {      // tau_rx*fs for a receiver at (ex,ey,ez), reference src/bf.cu:110
        const double rx = (double)px - (double)ex, ry = (double)py - (double)ey, rz = (double)pz - (double)ez;
        return dsqrt(rx * rx + ry * ry + rz * rz) * cf;
    };
    // delay of STAGE element n at (ex,ey,ez): a receiver (kind 0), or -- roles swapped -- a transmit with {t0, normal} in P.St (scalar loads)
    auto s_at = [&](uint32_t n, float ex, float ey, float ez) -> double {
        if constexpr (LUT) return (double)P.lut_rx[ipl + Ilut * n];
        if (!P.St) return b_at(ex, ey, ez);
        const double rx = (double)px - (double)ex, ry = (double)py - (double)ey, rz = (double)pz - (double)ez;
        const double dot = kindS ? rx * (double)P.St[4 * n + 1] + ry * (double)P.St[4 * n + 2] + rz * (double)P.St[4 * n + 3] : 0.0;
        double dv = dot;
        if (kindS != 2) { const double len = dsqrt(rx * rx + ry * ry + rz * rz); dv = kindS == 0 ? len : copysign(len, dot); }
        return dv * cf - (double)P.St[4 * n] * fs;
    };

    // ---- prologue: tile-wide window bases / extents per transmit and per receiver
    const uint32_t MX = M > N ? M : N;
    float a_lo = INFINITY, a_hi = -INFINITY, a_ext = 0.f;            // per-thread partials of tile-wide stats
    // reciprocal mode: a(i,m) = b(i,m) + C with C = OFF - t0*fs, so A[m] := B[m] + floor(C) (filled below)
    const double symC = tapinfo<INTERP>::OFF - (double)P.Pv[3] * fs;
    const int symCi = (int)floor(symC);
    // The window bases / extents only need the delays to a small fraction of a sample: fp32 estimates with an explicit error
    // margin (DLT, below) -- a quarter of the fp64 cost.  Focused transmits keep fp64: their delay flips sign with
    // (Pi - Pv).Nv (copysign, src/bf.cu:107) and the two precisions must agree on the sign of a dot product that may be ~0.
    const bool pro32 = !LUT && kindB != 1 && kindS != 1;
    const float cf32 = (float)cf, fs32 = (float)fs;
    auto a_est = [&](uint32_t m) -> float {
        if (!pro32) return (float)a_of(m);
        const float rx = px - gPv[4 * m], ry = py - gPv[4 * m + 1], rz = pz - gPv[4 * m + 2];
        const float d = kindB != 2 ? __builtin_sqrtf(rx * rx + ry * ry + rz * rz) : rx * gNv[3 * m] + ry * gNv[3 * m + 1] + rz * gNv[3 * m + 2];
        return d * cf32 - gPv[4 * m + 3] * fs32 + (float)tapinfo<INTERP>::OFF;
    };
    auto b_est = [&](uint32_t n) -> float {
        if constexpr (LUT) return P.lut_rx[ipl + Ilut * n];
        if (kindS == 1) return (float)s_at(n, P.Pr[3 * n], P.Pr[3 * n + 1], P.Pr[3 * n + 2]);
        const float rx = px - P.Pr[3 * n], ry = py - P.Pr[3 * n + 1], rz = pz - P.Pr[3 * n + 2];
        if (!P.St) return __builtin_sqrtf(rx * rx + ry * ry + rz * rz) * cf32;
        if (kindS == 0) return __builtin_sqrtf(rx * rx + ry * ry + rz * rz) * cf32 - P.St[4 * n] * fs32;
        return (rx * P.St[4 * n + 1] + ry * P.St[4 * n + 2] + rz * P.St[4 * n + 3]) * cf32 - P.St[4 * n] * fs32;
    };
    // |fp32 estimate - fp64 delay| <= ~4e-7 * (|distance*cf| + |t0*fs|), and |distance*cf| <= |a| + |t0*fs| + 1: 1e-6 is generous
    auto margin = [](float mn, float mx, float t0fs) -> float { return 1.0e-6f * (fmaxf(fabsf(mn), fabsf(mx)) + 2.0f * fabsf(t0fs) + 2.0f); };
    // (four elements per pass: independent reduction chains overlap)
    auto minmax4 = [&](float (&v)[4], uint32_t e0, uint32_t cnt) {
        float lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = hi[q] = (v[q] == v[q]) ? v[q] : INFINITY; }   // a NaN delay poisons the tile's extent
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = wave_min63(lo[q]); hi[q] = wave_max63(hi[q]); }
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (e0 + q < cnt) { part[wave * MX + e0 + q] = lo[q]; part[(WAVES + wave) * MX + e0 + q] = hi[q]; }
        }
    };
    if constexpr (!SYM) {
        for (uint32_t m = 0; m < M; m += 4) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a_est(m + q < M ? m + q : M - 1);
            minmax4(v, m, M);
        }
        __syncthreads();
        for (uint32_t m = tid; m < M; m += THREADS) {
            float mn = part[m], mx = part[WAVES * MX + m];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, part[w * MX + m]); mx = fmaxf(mx, part[(WAVES + w) * MX + m]); }
            const float dlt = margin(mn, mx, LUT ? 0.f : P.Pv[4 * m + 3] * fs32);
            const float fl = floorf(mn - dlt) - 1.0f;        // margin: the estimate may lie above the true minimum
            const bool fin = fabsf(fl) < 1.0e9f;
            const float e = fin ? ((mx + dlt) - fl) + 0.01f : INFINITY;
            Abase[m] = fin ? (int)fl : 0;
            Aext[m] = e;
            a_lo = fminf(a_lo, fl); a_hi = fmaxf(a_hi, fl + e); a_ext = fmaxf(a_ext, e);
        }
        __syncthreads();
    }
    for (uint32_t n = 0; n < N; n += 4) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = b_est(n + q < N ? n + q : N - 1);
        minmax4(v, n, N);
    }
    __syncthreads();
    float b_lo = INFINITY, b_hi = -INFINITY, b_ext = 0.f;
    for (uint32_t n = tid; n < N; n += THREADS) {
        float mn = part[n], mx = part[WAVES * MX + n];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, part[w * MX + n]); mx = fmaxf(mx, part[(WAVES + w) * MX + n]); }
        const float dlt = margin(mn, mx, LUT ? 0.f : SYM ? P.Pv[3] * fs32 : (P.St ? P.St[4 * n] * fs32 : 0.f));
        const float fl = floorf(mn - dlt) - 1.0f;
        const bool fin = fabsf(fl) < 1.0e9f;
        const float e = fin ? ((mx + dlt) - fl) + 0.01f : INFINITY;
        if constexpr (LUT) nrec[n] = make_float4(__int_as_float(fin ? (int)fl : 0), 0.f, 0.f, 0.f);
        else nrec[n] = make_float4(__int_as_float(fin ? (int)fl : 0), P.Pr[3 * n], P.Pr[3 * n + 1], P.Pr[3 * n + 2]);
        Bext[n] = e;
        b_lo = fminf(b_lo, fl); b_hi = fmaxf(b_hi, fl + e); b_ext = fmaxf(b_ext, e);
        if constexpr (SYM) {                             // a - A = (b - B) + frac(C) in [1, Bext + 1)
            Abase[n] = (fin ? (int)fl : 0) + symCi;
            Aext[n] = e + 1.0f;
            a_lo = fminf(a_lo, fl + (float)symCi); a_hi = fmaxf(a_hi, fl + (float)symCi + e + 1.0f); a_ext = fmaxf(a_ext, e + 1.0f);
        }
    }
    __syncthreads();                                   // part[] is free again
    a_lo = wave_min(a_lo); b_lo = wave_min(b_lo);
    a_hi = wave_max(a_hi); b_hi = wave_max(b_hi); a_ext = wave_max(a_ext); b_ext = wave_max(b_ext);
    if (lane == 0) { float *q = part + wave * 8; q[0] = a_lo; q[1] = b_lo; q[2] = a_hi; q[3] = b_hi; q[4] = a_ext; q[5] = b_ext; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const float *q = part + w * 8;
        a_lo = fminf(a_lo, q[0]); b_lo = fminf(b_lo, q[1]); a_hi = fmaxf(a_hi, q[2]); b_hi = fmaxf(b_hi, q[3]);
        a_ext = fmaxf(a_ext, q[4]); b_ext = fmaxf(b_ext, q[5]);
    }
    // every lane's last tap (+1 for the rint/floor ambiguity at exact integers) must be inside the staged window
    if (!(a_ext + b_ext + (float)(K + 1) <= (float)W)) {
        if (tid == 0 && split == 0) {
            const uint32_t slot = atomicAdd(&P.fallback_list[0], 1u);
            if (slot < P.fallback_cap) P.fallback_list[1 + slot] = tile_id;
        }
        return;                                        // uniform exit: generic kernel takes this tile
    }
    if constexpr (PROBE) return;                       // plan-time shape selection only wants the fit verdict
    // every window of every stage strictly inside the record?  (uniform) -> branch-free loop
    const bool tile_interior = (a_lo + b_lo >= 1.0f) && (a_hi + b_hi + (float)(K + 1) < (float)T);
    if constexpr (!LUT) {
        for (uint32_t k = tid; k < 4 * M; k += THREADS) PvL[k] = P.Pv[k];
        for (uint32_t k = tid; k < 3 * M; k += THREADS) NvL[k] = P.Nv[k];
        gPv = PvL; gNv = NvL;
    }
    __syncthreads();

#if QDAS_PROF
    pt_[0] = QDAS_TICK() - pstart_;                    // prologue
#endif
    // ---- main loop over stages (mb = transmit block, n = receiver; n is the inner index)
    // This workgroup's share of the aperture (ksplit workgroups per tile when the image has too few tiles to fill the GPU;
    // their partial sums are added in a fixed order by tile_reduce_kernel):
    //   reciprocal mode: every S-th transmit block, dealt out boustrophedon (0..S-1, S-1..0, ...) because block kb pairs with
    //   16(kb+1) receivers -- the triangular work is balanced;   otherwise: a contiguous range of receivers, all transmit blocks.
    auto blk = [&](uint32_t r) -> uint32_t {             // first transmit of my r-th block (>= M: exhausted)
        if constexpr (SYM) return (r * S + ((r & 1u) ? S - 1u - split : split)) * MB;
        else return r * MB;
    };
    const uint32_t n_lo = SYM ? 0u : (uint32_t)((uint64_t)N * split / S);
    const uint32_t n_hi = (uint32_t)((uint64_t)N * (split + 1) / S);
    // receivers paired with transmit block m0: my range, or -- reciprocal mode -- all n <= the block's last transmit
    auto nlim = [&](uint32_t m0) -> uint32_t { return SYM ? (m0 + MB < N ? m0 + MB : N) : n_hi; };
    uint32_t nstage = 0;
    for (uint32_t r = 0; blk(r) < M; ++r) nstage += nlim(blk(r)) - n_lo;
    v2f acc = {0.f, 0.f};                              // (re, im) of this lane's pixel
    v2f acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f}, acc3 = {0.f, 0.f};   // independent partial sums: no back-to-back dependent packed FMAs
    // accumulator -> frame: one frame: all four; two frames: (acc, acc1 | acc2, acc3); four frames: one each
    auto frame_sums = [&](v2f (&Sf)[4]) {
        if constexpr (FB4) { Sf[0] = acc; Sf[1] = acc1; Sf[2] = acc2; Sf[3] = acc3; }
        else if constexpr (FB2) { Sf[0] = acc + acc1; Sf[1] = acc2 + acc3; }
        else Sf[0] = (acc + acc1) + (acc2 + acc3);
    };
    v2f ra[MB / 2];                                    // tx residuals a - A[m] - 1/2 of transmits (2p, 2p+1), packed
    // ---- LDS-DMA staging.  One buffer descriptor per stage, based at trace (n, m0): window j starts
    //      (j*strM + A[m0+j] + B[n]) samples after it.  A lane moves 16 bytes; a wave-instruction 1 KiB.
    //      Offsets before the base wrap to >= num_records and, like offsets past the end of x, deliver 0.
    //      Samples outside [0, T) of a trace but inside x read the neighbouring trace: they are only ever
    //      touched by lanes that the checked loop masks out (select, not multiply).
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int WB = W * SB;                         // bytes per window
    // (only 16-byte pieces give a contiguous LDS image: a 12-byte piece still advances 16 bytes per lane --
    //  measured with tools/scratch/dma12.hip)
    constexpr int PB = 1024;                           // bytes per full DMA piece (one wave-instruction x 16 B)
    constexpr int PCS = (WB + PB - 1) / PB;            // pieces per window; the last one may use fewer lanes
    constexpr int NDMA = WPW * PCS * (TWO ? 2 : 1);    // DMA instructions per wave and stage (four frames: two of the four sets per wave)
    // window sets this wave stages: (0, 1) in general; with four frames the lower / upper half of the waves take frames (0, 2) / (1, 3)
    const int fa = FB4 ? __builtin_amdgcn_readfirstlane(wave / MB) : 0, fb = FB4 ? fa + 2 : 1;
    auto wjr = [&](int r) -> int { return FB4 ? __builtin_amdgcn_readfirstlane(wave % MB) : __builtin_amdgcn_readfirstlane(wave + WAVES * r); };
    static_assert(WB % 16 == 0 && PSZ == 16, "window must be a whole number of 16-byte lanes");
    const uint64_t xbytes = (uint64_t)P.N * P.M * P.T * SB;
    // Per-wave DMA state: this wave stages windows j_r = wave + WAVES*r.  One buffer descriptor per TRANSMIT BLOCK, based at
    // trace (rx n_lo, tx m0) (mirror: (rx m0, tx 0)); everything that does not depend on the receiver -- A[m], the window's
    // trace offset -- is folded into one scalar per window when the block starts (dma_block).  A stage then costs two scalar
    // adds per window: offset = soff (running receiver offset, 32-bit by the plan-time check) + wb[r] + B[n]*SB.
    int wb[WPW], wb2[WPW];                             // A[m_r]*SB + j_r*strM*SB   (mirror: + j_r*strN*SB)
    uint32_t soff = 0, soff2 = 0;
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)P.x, 0, 0, 0x00020000), rsM = rsD;
    // descriptor based `o` bytes into the frame (+ `extra`: the frame itself); records beyond the frame read as zeros
    auto make_rs = [&](uint64_t o, uint64_t extra) -> __amdgpu_buffer_rsrc_t {
        const uint64_t rem = xbytes > o ? xbytes - o : 0;   // (m0 >= M when the split is exhausted: nothing more is issued)
        return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)P.x + (rem ? o + extra : 0)), 0, rem > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)rem, 0x00020000);
    };
    // Reciprocal mode walks the whole frame inside one transmit block (the mirror "transmits", or the receivers of transposed
    // data): its running offsets are kept below 2^30 by re-basing the descriptor when they get there (uniform, rare).  The
    // general kernels keep one descriptor per block (plan-time check: the walk stays below 2^31 bytes); transposed fp32 frames
    // beyond that run the BIG instantiation (launch configuration 9), which re-bases as well.
    constexpr uint32_t REBASE = 1u << 30;
    uint64_t offD = 0, offM = 0;
    auto dma_block = [&](uint32_t m0) {
        const uint64_t o = ((uint64_t)m0 * P.strM + (uint64_t)n_lo * P.strN) * SB;
        if constexpr (SYM || BIG) offD = o;
        rsD = make_rs(o, (uint64_t)fa * P.x_fstride);
        soff = 0;
#pragma unroll
        for (int r = 0; r < WPW; ++r) {
            const int j = wjr(r);
            const uint32_t m = m0 + j;
            const int am = __builtin_amdgcn_readfirstlane(Abase[m < M ? m : M - 1]);
            wb[r] = am * SB + (int)((long)j * (long)P.strM * SB);
            if constexpr (SYM) wb2[r] = am * SB + (int)((long)j * (long)P.strN * SB);
        }
        if constexpr (FBX) rsM = make_rs(o, (uint64_t)fb * P.x_fstride);   // the same traces of the next frame (four frames: of frame fb)
        if constexpr (SYM) {                           // mirror traces x[:, rx = m0 + j, tx = n] (reciprocal mode starts every block at n = 0)
            offM = (uint64_t)m0 * P.strN * SB;
            rsM = make_rs(offM, 0);
            soff2 = 0;
        }
    };
    auto stage_dma = [&](int bn, int buf) {           // stage (receiver with window base bn = B[n], current DMA transmit block)
        const int bs = bn * SB;
#pragma unroll
        for (int r = 0; r < WPW; ++r) {
            const int j = wjr(r);
            const int so = (int)soff + wb[r] + bs;
#pragma unroll
            for (int q = 0; q < ((QDAS_ABL & 64) ? 1 : PCS); ++q) {
                lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + fa * MB + j) * WB + q * PB));
                if (lane * 16 < WB - q * PB)             // trailing partial piece: upper lanes masked off
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, dst, 16, lane * 16, so + q * PB, 0, 0);
            }
        }
        if constexpr (FBX) {                           // next frame: same offsets, other descriptor, another window set
#pragma unroll
            for (int r = 0; r < WPW; ++r) {
                const int j = wjr(r);
                const int so = (int)soff + wb[r] + bs;
#pragma unroll
                for (int q = 0; q < PCS; ++q) {
                    lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + fb * MB + j) * WB + q * PB));
                    if (lane * 16 < WB - q * PB)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, dst, 16, lane * 16, so + q * PB, 0, 0);
                }
            }
        }
        soff += (uint32_t)P.strN * SB;                 // next receiver, same transmit block
        if constexpr (SYM || BIG) {
            if (soff >= REBASE) { offD += soff; soff = 0; rsD = make_rs(offD, 0); }
        }
        if constexpr (SYM) {                           // same window start A[m] + B[n] in the mirror trace
#pragma unroll
            for (int r = 0; r < WPW; ++r) {
                const int j = __builtin_amdgcn_readfirstlane(wave + WAVES * r);
                const int so = (int)soff2 + wb2[r] + bs;
#pragma unroll
                for (int q = 0; q < PCS; ++q) {
                    lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + MB + j) * WB + q * PB));
                    if (lane * 16 < WB - q * PB)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, dst, 16, lane * 16, so + q * PB, 0, 0);
                }
            }
            soff2 += (uint32_t)P.strM * SB;            // next "transmit" n of the mirror traces
            if (soff2 >= REBASE) { offM += soff2; soff2 = 0; rsM = make_rs(offM, 0); }
        }
    };

    // ---- per-(pixel, receiver) apodization (optional; uniform runtime switch, nothing of it in the pair loop)
    const bool wpix = !SYM && (P.apix != nullptr || P.gen_kind != 0);   // weights from an I x N array, or generated from the geometry
    const uint64_t Itot = P.I1 * P.I2 * P.I3;
    const uint64_t ipc = (i1 < P.I1 ? i1 : P.I1 - 1) + P.I1 * (col < ncols ? col : ncols - 1);   // my (clamped) pixel
    auto wload = [&](uint32_t n) -> v2f {
        if (!SYM && P.gen_kind) {                         // qdas.h QDAS_RXAPOD_*: element from the LDS record, normal by scalar loads (never in reciprocal mode)
            const float4 e = nrec[n];
            return (v2f){rx_apod_generated(P.gen_kind, P.gen_p0, P.gen_p1, px, py, pz, e.y, e.z, e.w, P.rxn, n), 0.f};
        }
        const uint64_t k = ipc + Itot * n;
        if (P.apix_real) {
            if constexpr (F32) return (v2f){((const float *)P.apix)[k], 0.f};
            else return (v2f){__half2float(__ushort_as_half(((const unsigned short *)P.apix)[k])), 0.f};
        } else {
            if constexpr (F32) { const float2 v = ((const float2 *)P.apix)[k]; return (v2f){v.x, v.y}; }
            else { const uint32_t v = ((const uint32_t *)P.apix)[k];
                   return (v2f){__half2float(__ushort_as_half((unsigned short)(v & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(v >> 16)))}; }
        }
    };
    v2f tot[NFR];                                      // weighted totals per frame when wpix (acc.. then hold one stage's partial sums)
#pragma unroll
    for (int f = 0; f < NFR; ++f) tot[f] = (v2f){0.f, 0.f};
    const bool syn = !SYM && F32 && P.syn;             // keep the receive dimension: one output plane per receiver
    const bool in_shard = (i1 < P.I1) && (col < ncols) && (i1 + P.I1 * col >= P.i_begin) && (i1 + P.I1 * col < i_end);

    auto run = [&](auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        v2f wcur = {1.f, 0.f}, wnext = {1.f, 0.f};
        if (wpix) wcur = wload(n_lo);
        float tbc = 0.f, tbn = 0.f;                        // LUT: this / the next stage's receive delay of my pixel
        if constexpr (LUT) tbc = P.lut_rx[ipl + Ilut * n_lo];
        uint32_t pr = 0, pn = n_lo, pm0 = blk(0);          // stage the DMA front is at (NBUF-1 stages ahead)
        dma_block(pm0);
        // B[n] of the stage at the DMA front travels in a VGPR, loaded one stage before it is needed: every LDS read of a stage
        // is issued BEFORE the stage's LDS-DMA in program order (the compiler orders a later LDS read behind the DMA's vmcnt).
        float vbn = nrec[pn < N ? pn : N - 1].x;
        auto dma_next = [&](int buf) {
            const int bn = __builtin_amdgcn_readfirstlane(__float_as_int(vbn));
            const uint32_t qn = (pn + 1 == nlim(pm0)) ? n_lo : pn + 1;     // receiver of the stage after this one
            vbn = nrec[qn < N ? qn : N - 1].x;
            stage_dma(bn, buf);
            if (++pn == nlim(pm0)) { pn = n_lo; pm0 = blk(++pr); dma_block(pm0); }
        };
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b)
            if ((uint32_t)b < nstage) dma_next(b);
        // counted wait: everything but the newest (NBUF-2) stages has landed; then publish to the workgroup
        if (nstage >= (uint32_t)(NBUF - 1)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 2) * NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        int buf = 0;
        uint32_t cr = 0, n = n_lo, m0 = blk(0);
        for (uint32_t st = 0; st < nstage; ++st) {
#if QDAS_PROF
            const unsigned long long ts0_ = QDAS_TICK();
#endif
            const bool more = st + (NBUF - 1) < nstage;
            // next stage's pixel weight: requested BEFORE this stage's DMA, so the end-of-stage wait covers it
            if (wpix && st + 1 < nstage) wnext = wload(n + 1 == nlim(m0) ? n_lo : n + 1);
            if constexpr (LUT) { if (st + 1 < nstage) tbn = P.lut_rx[ipl + Ilut * (n + 1 == nlim(m0) ? n_lo : n + 1)]; }
            const bool skip = wpix && (__ballot(wcur.x != 0.f || wcur.y != 0.f) == 0ull);   // whole wave weightless: no gathers
            const float4 rec = nrec[n];                // {B[n], receiver position}: one broadcast LDS read, issued ahead of the DMA
            // The next stage's staging is issued at the start of this stage by the younger half of the waves and AFTER the pair
            // loop by the older half: the hardware favours older waves, they finish their pair loop early and would only wait at
            // the barrier -- their (scalar-heavy) issue phase then overlaps the younger waves' arithmetic instead of everybody's.
            const bool dma_late = SYM && !(QDAS_ABL & 1024) && wave < WAVES / 2;
            if (!(QDAS_ABL & 1) && more && !dma_late) dma_next((buf + NBUF - 1) % NBUF);   // lands during the next NBUF-1 stages

#if QDAS_PROF
            const unsigned long long ts1_ = QDAS_TICK();
#endif
            if (n == n_lo) {                           // new transmit block: refresh the tx residuals
#pragma unroll
                for (int p = 0; p < MB / 2; ++p) {
                    const uint32_t ma = m0 + 2 * p < M ? m0 + 2 * p : M - 1, mb = m0 + 2 * p + 1 < M ? m0 + 2 * p + 1 : M - 1;
                    if constexpr (SYM) {               // a - A - 1/2 = (b - B) + frac(C) - 1/2, from the receiver records
                        const float4 ea = nrec[ma], eb = nrec[mb];
                        const double fc = symC - (double)symCi - 0.5;
                        ra[p] = (v2f){(float)(b_at(ea.y, ea.z, ea.w) - (double)__float_as_int(ea.x) + fc),
                                      (float)(b_at(eb.y, eb.z, eb.w) - (double)__float_as_int(eb.x) + fc)};
                    } else if constexpr (LUT)
                        ra[p] = (v2f){(float)(a_of(ma) - ((double)Abase[ma] + 0.5)), (float)(a_of(mb) - ((double)Abase[mb] + 0.5))};
                    else
                        ra[p] = (v2f){block_residual(px, py, pz, cf, fs, kindB, (lds_cfloat *)PvL, (lds_cfloat *)NvL, ma, Abase[ma], tapinfo<INTERP>::OFF),
                                      block_residual(px, py, pz, cf, fs, kindB, (lds_cfloat *)PvL, (lds_cfloat *)NvL, mb, Abase[mb], tapinfo<INTERP>::OFF)};
                    __builtin_amdgcn_sched_barrier(0);      // one pair at a time: keeps this cold block from inflating the register budget
                }
            }
#if QDAS_PROF
            const unsigned long long ts2_ = QDAS_TICK();
#endif
            if (!skip) {
            const int bn = __float_as_int(rec.x);
            const float rb = LUT ? tbc - (float)bn : (QDAS_ABL & 2) ? (float)(lane * 2 + 3) + 0.37f * (float)(n & 7) : (float)(s_at(n, rec.y, rec.z, rec.w) - (double)bn);
            // LDS byte address of a tap = bits(t + MAGIC)*SB + cbase + (window, tap) immediate
            const uint32_t cbase = win_off + (uint32_t)buf * (NW * WB) - (MAGIC_BITS * (uint32_t)SB);

            auto pairs = [&](auto pc, auto tailc) {       // transmits (m0+2p, m0+2p+1) ride in the two halves
                constexpr int p = decltype(pc)::value;
                constexpr bool TAIL = !SYM && decltype(tailc)::value;  // last, partial transmit block: bounds checks
                constexpr bool DIAG = SYM && decltype(tailc)::value;   // reciprocal mode, block that contains m == n
                const uint32_t m = m0 + 2 * p;
                if constexpr (TAIL) { if (m >= M) return; }
                if constexpr (DIAG) { if (m + 1 < n) return; }          // both transmits below the diagonal: their pairs were done as mirrors
                const bool upper = !TAIL || (m + 1 < M);  // the upper half carries a real transmit
                float wr0 = 1.f, wi0 = 0.f, wr1 = 1.f, wi1 = 0.f;
                if constexpr (WTAB) {
                    const float2 wa = ((const float2 *)P.wtab)[n + (size_t)N * m];
                    const float2 wb = upper ? ((const float2 *)P.wtab)[n + (size_t)N * (m + 1)] : make_float2(0.f, 0.f);
                    wr0 = wa.x; wi0 = wa.y; wr1 = wb.x; wi1 = wb.y;
                    if (wr0 == 0.f && wi0 == 0.f && wr1 == 0.f && wi1 == 0.f) return;   // zero weights: skip (src/bf.cu:122,126)
                }
                const v2f t = ra[p] + rb;                 // = tau*fs + OFF - (A+B) - 1/2
                const v2f tm = t + MAGIC;
                const v2f s = t - (tm - MAGIC);           // in [-1/2, 1/2]
                const uint32_t ad0 = ((QDAS_ABL & 128) ? (MAGIC_BITS + (uint32_t)lane + (__float_as_uint(tm.x) & 1u)) : __float_as_uint(tm.x)) * (uint32_t)SB + cbase;
                const uint32_t ad1 = ((QDAS_ABL & 128) ? (MAGIC_BITS + (uint32_t)lane + (__float_as_uint(tm.y) & 1u)) : __float_as_uint(tm.y)) * (uint32_t)SB + cbase;
                constexpr bool SPLIT = CHECK || FMOD || WTAB;       // the two halves need separate post-processing
                v2f w[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
                // one pass per FRAME PAIR (two passes when four frames share the launch): same tap index and weights
                unroll<NHP>([&](auto hpc) {
                constexpr int hp = decltype(hpc)::value;
                constexpr int GSET = FB4 ? 2 * hp : 0, HSET = FB4 ? 2 * hp + 1 : 1;       // window sets of the pass
                // accumulators: one per (frame, transmit half) with up to two frames, one per frame with four
                v2f &A0 = *(FB4 ? (hp ? &acc2 : &acc) : &acc), &A1 = *(FB4 ? (hp ? &acc2 : &acc) : &acc1);
                v2f &B0 = *(FB4 ? (hp ? &acc3 : &acc1) : &acc2), &B1 = *(FB4 ? (hp ? &acc3 : &acc1) : &acc3);
                v2f v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
                v2f u0 = {0.f, 0.f}, u1 = {0.f, 0.f};     // the same two pairs of the second frame (FB2)
                if constexpr (F32) {
                    taps_f32 g0, g1, h0, h1;              // direct taps x[:, n, m | m+1]; mirror taps x[:, m | m+1, n]
                    if constexpr ((QDAS_ABL & 4) != 0) { for (int k = 0; k < 4; ++k) { g0.s[k] = h0.s[k] = (v2f){s.x, t.x}; g1.s[k] = h1.s[k] = (v2f){t.y, s.y}; } }
                    else {
                        lds_issue<K, (GSET * MB + 2 * p) * WB>(g0, ad0); lds_issue<K, (GSET * MB + 2 * p + 1) * WB>(g1, ad1);
                        if constexpr (TWO) { lds_issue<K, (HSET * MB + 2 * p) * WB>(h0, ad0); lds_issue<K, (HSET * MB + 2 * p + 1) * WB>(h1, ad1); }
                        if constexpr (K < 4) { g0.s[2] = g0.s[3] = g1.s[2] = g1.s[3] = h0.s[2] = h0.s[3] = h1.s[2] = h1.s[3] = (v2f){0.f, 0.f}; }
                        if constexpr (K < 2) { g0.s[1] = g1.s[1] = h0.s[1] = h1.s[1] = (v2f){0.f, 0.f}; }
                    }
                    if constexpr (hp == 0) {              // (the next frame pair reuses them)
                        if constexpr ((QDAS_ABL & 8) != 0) { w[0] = s; w[1] = t; w[2] = tm; w[3] = s + t; }
                        else if constexpr (K > 1) weights2<INTERP>(s, w);   // overlaps the LDS latency
                    }
                    if constexpr (TWO) lds_fence2(g0, g1, h0, h1, w); else lds_fence(g0, g1, w);
                    if constexpr (DIAG) {                 // uniform, only in the block that holds the diagonal
                        const v2f z = {0.f, 0.f};
                        if (m < n)      { for (int k = 0; k < 4; ++k) g0.s[k] = z; }     // pair (n, m<n): done as the mirror of (m, n)
                        if (m <= n)     { for (int k = 0; k < 4; ++k) h0.s[k] = z; }     // m == n: the trace is its own mirror
                        if (m + 1 <= n) { for (int k = 0; k < 4; ++k) h1.s[k] = z; }
                    }
                    if constexpr (TAIL) {
                        if (!upper) {                       // odd M: no transmit in the upper half (uniform, rare)
#pragma unroll
                            for (int k = 0; k < 4; ++k) { g1.s[k] = (v2f){0.f, 0.f}; if constexpr (FBX) h1.s[k] = (v2f){0.f, 0.f}; }
                        }
                    }
                    if constexpr (K == 1) { v0 = g0.s[0]; v1 = g1.s[0]; if constexpr (FBX) { u0 = h0.s[0]; u1 = h1.s[0]; } }
                    else if constexpr (SPLIT) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { v0 = w[k].x * g0.s[k] + v0; v1 = w[k].y * g1.s[k] + v1; }
                        if constexpr (SYM) {
#pragma unroll
                            for (int k = 0; k < K; ++k) { v0 = w[k].x * h0.s[k] + v0; v1 = w[k].y * h1.s[k] + v1; }
                        }
                        if constexpr (FBX) {
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
                    if constexpr (SYM && K == 1) { v0 += h0.s[0]; v1 += h1.s[0]; }
                } else {
                    taps_f16 g0, g1, h0, h1;
                    lds_issue<K, (GSET * MB + 2 * p) * WB>(g0, ad0); lds_issue<K, (GSET * MB + 2 * p + 1) * WB>(g1, ad1);
                    if constexpr (TWO) { lds_issue<K, (HSET * MB + 2 * p) * WB>(h0, ad0); lds_issue<K, (HSET * MB + 2 * p + 1) * WB>(h1, ad1); }
                    if constexpr (K < 4) { g0.r[2] = g0.r[3] = g1.r[2] = g1.r[3] = h0.r[2] = h0.r[3] = h1.r[2] = h1.r[3] = 0u; }
                    if constexpr (K < 2) { g0.r[1] = g1.r[1] = h0.r[1] = h1.r[1] = 0u; }
                    if constexpr (hp == 0 && K > 1) weights2<INTERP>(s, w);  // overlaps the LDS latency
                    if constexpr (TWO) lds_fence2(g0, g1, h0, h1, w); else lds_fence(g0, g1, w);
                    if constexpr (DIAG) {                 // (as for fp32 data above)
                        if (m < n)      { for (int k = 0; k < 4; ++k) g0.r[k] = 0u; }
                        if (m <= n)     { for (int k = 0; k < 4; ++k) h0.r[k] = 0u; }
                        if (m + 1 <= n) { for (int k = 0; k < 4; ++k) h1.r[k] = 0u; }
                    }
                    if constexpr (TAIL) {
                        if (!upper) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { g1.r[k] = 0u; if constexpr (FBX) h1.r[k] = 0u; }
                        }
                    }
                    if constexpr (K == 1) {
                        v0 = half2_to_v2f(g0.r[0]); v1 = half2_to_v2f(g1.r[0]);
                        if constexpr (FBX) { u0 = half2_to_v2f(h0.r[0]); u1 = half2_to_v2f(h1.r[0]); }
                        if constexpr (SYM) { v0 += half2_to_v2f(h0.r[0]); v1 += half2_to_v2f(h1.r[0]); }
                    } else if constexpr (SPLIT) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { mix_mac(v0, g0.r[k], w[k].x); mix_mac(v1, g1.r[k], w[k].y); }
                        if constexpr (SYM) {
#pragma unroll
                            for (int k = 0; k < K; ++k) { mix_mac(v0, h0.r[k], w[k].x); mix_mac(v1, h1.r[k], w[k].y); }
                        }
                        if constexpr (FBX) {
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
                if constexpr (CHECK) {                    // edge rule: all taps in [0,T) and tau >= 0
                    const uint32_t mb = upper ? m + 1 : m;
                    const int ws0 = Abase[m] + bn, ws1 = Abase[mb] + bn;
                    const float lo0 = tapinfo<INTERP>::LO - 0.5f - (float)ws0, hi0 = (float)(T - K + 1 - ws0) - 0.5f;
                    const float lo1 = tapinfo<INTERP>::LO - 0.5f - (float)ws1, hi1 = (float)(T - K + 1 - ws1) - 0.5f;
                    const bool k0 = (t.x >= lo0) && (t.x < hi0), k1 = (t.y >= lo1) && (t.y < hi1) && upper;
                    v0 = k0 ? v0 : (v2f){0.f, 0.f}; v1 = k1 ? v1 : (v2f){0.f, 0.f};
                    if constexpr (FBX) { u0 = k0 ? u0 : (v2f){0.f, 0.f}; u1 = k1 ? u1 : (v2f){0.f, 0.f}; }
                }
                if constexpr (FMOD) {                     // reference src/bf.cu:117: w = exp(2j pi fmod tau)
                    const uint32_t mb = upper ? m + 1 : m;
                    const double f = P.fmod / fs;         // tau*fs = t + 1/2 + ws - OFF
                    const double p0 = ((double)(Abase[m] + bn) + 0.5 - tapinfo<INTERP>::OFF) * f;
                    const double p1 = ((double)(Abase[mb] + bn) + 0.5 - tapinfo<INTERP>::OFF) * f;
                    const v2f ph = t * (float)f + (v2f){(float)(p0 - floor(p0)), (float)(p1 - floor(p1))};   // cycles
                    const float c0 = __builtin_amdgcn_cosf(ph.x), s0 = __builtin_amdgcn_sinf(ph.x);
                    const float c1 = __builtin_amdgcn_cosf(ph.y), s1 = __builtin_amdgcn_sinf(ph.y);
                    v0 = (v2f){v0.x * c0 - v0.y * s0, v0.x * s0 + v0.y * c0};
                    v1 = (v2f){v1.x * c1 - v1.y * s1, v1.x * s1 + v1.y * c1};
                    if constexpr (FBX) {
                        u0 = (v2f){u0.x * c0 - u0.y * s0, u0.x * s0 + u0.y * c0};
                        u1 = (v2f){u1.x * c1 - u1.y * s1, u1.x * s1 + u1.y * c1};
                    }
                }
                if constexpr (WTAB) {
                    A0 += (v2f){wr0 * v0.x - wi0 * v0.y, wr0 * v0.y + wi0 * v0.x};
                    A0 += (v2f){wr1 * v1.x - wi1 * v1.y, wr1 * v1.y + wi1 * v1.x};
                    if constexpr (FBX) {
                        B0 += (v2f){wr0 * u0.x - wi0 * u0.y, wr0 * u0.y + wi0 * u0.x};
                        B0 += (v2f){wr1 * u1.x - wi1 * u1.y, wr1 * u1.y + wi1 * u1.x};
                    }
                } else if constexpr (SPLIT || K == 1) { A0 += v0; A0 += v1; if constexpr (FBX) { B0 += u0; B0 += u1; } }
                            });
            };
            // full block (reciprocal mode: block entirely above the diagonal): check-free; else the tail / diagonal variant
            if (SYM ? (n < m0) : (m0 + MB <= M)) {
                if constexpr (TWO && (F32 || SYM) && K == 4 && !(CHECK || FMOD || WTAB) && !(QDAS_ABL & 256)) {
                    // Software-pipelined: the direct taps of iteration p+1 are requested before the MACs of iteration p, so the
                    // LDS pipe always has work queued and the counted wait (newest 8 reads stay in flight) rarely stalls.
                    // A unit = (transmit pair p, frame pair hp); hp only with four frames per launch: same index and weights.
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
                            a0v[p & 1] = __float_as_uint(tm.x) * (uint32_t)SB + cbase;
                            a1v[p & 1] = __float_as_uint(tm.y) * (uint32_t)SB + cbase;
                        }
                        if constexpr (F32 && (QDAS_ABL & 4) != 0) { for (int k = 0; k < 4; ++k) { gd0[u & 1].s[k] = (v2f){sv[p & 1].x, rb}; gd1[u & 1].s[k] = (v2f){rb, sv[p & 1].y}; } }
                        else { lds_issue<K, (GSET * MB + 2 * p) * WB>(gd0[u & 1], a0v[p & 1]); lds_issue<K, (GSET * MB + 2 * p + 1) * WB>(gd1[u & 1], a1v[p & 1]); }
                    };
                    index(std::integral_constant<int, 0>{});
                    unroll<NU>([&](auto uc) {
                        constexpr int u = decltype(uc)::value, p = u / NHP, hp = u % NHP, HSET = FB4 ? 2 * hp + 1 : 1;
                        taps_t h0, h1;
                        if constexpr (F32 && (QDAS_ABL & 4) != 0) { for (int k = 0; k < 4; ++k) { h0.s[k] = sv[p & 1]; h1.s[k] = (v2f){sv[p & 1].y, sv[p & 1].x}; } }
                        else { lds_issue<K, (HSET * MB + 2 * p) * WB>(h0, a0v[p & 1]); lds_issue<K, (HSET * MB + 2 * p + 1) * WB>(h1, a1v[p & 1]); }
                        if constexpr (u + 1 < NU) index(std::integral_constant<int, u + 1>{});
                        if constexpr (hp == 0) {
                            if constexpr ((QDAS_ABL & 8) != 0) { w[0] = sv[p & 1]; w[1] = sv[p & 1] + 1.f; w[2] = sv[p & 1] * 2.f; w[3] = 1.f - sv[p & 1]; }
                            else weights2<INTERP>(sv[p & 1], w);
                        }
                        if constexpr (u + 1 < NU) lds_fence2_keep<8>(gd0[u & 1], gd1[u & 1], h0, h1, w);
                        else                      lds_fence2_keep<0>(gd0[u & 1], gd1[u & 1], h0, h1, w);
                        v2f &A0 = *(FB4 ? (hp ? &acc2 : &acc) : &acc), &A1 = *(FB4 ? (hp ? &acc2 : &acc) : &acc1);
                        v2f &B0 = *(FB4 ? (hp ? &acc3 : &acc1) : &acc2), &B1 = *(FB4 ? (hp ? &acc3 : &acc1) : &acc3);
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            tap_mac(A0, gd0[u & 1], k, w[k].x); tap_mac(A1, gd1[u & 1], k, w[k].y);
                            tap_mac(B0, h0, k, w[k].x);         tap_mac(B1, h1, k, w[k].y);
                        }
                    });
                } else {
                    unroll<MB / 2>([&](auto pc) { pairs(pc, std::false_type{}); });
                }
            } else {
                unroll<MB / 2>([&](auto pc) { pairs(pc, std::true_type{}); });
            }

            }   // !skip
            if (!(QDAS_ABL & 1) && more && dma_late) dma_next((buf + NBUF - 1) % NBUF);
#if QDAS_PROF
            const unsigned long long ts3_ = QDAS_TICK();
#endif

            // stage st+1 must have landed (all but the NBUF-2 newest DMA groups), all my LDS reads are done
            if (!(QDAS_ABL & 16)) {
                if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NBUF - 2) * NDMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
#if QDAS_PROF
            { const unsigned long long ts4_ = QDAS_TICK();
              pt_[1] += ts1_ - ts0_; pt_[2] += ts2_ - ts1_; pt_[3] += ts3_ - ts2_; pt_[4] += ts4_ - ts3_; pt_[6] += 1; }
#endif
            buf = (buf + 1 == NBUF) ? 0 : buf + 1;
            if (syn) {                                 // 'SYN' (src/bf.cu:131-133): the stage's sum over its transmits joins plane n of y.
                // Non-returning fp32 atomics: the planes are zero-filled by the host, the transmit blocks of one (pixel, n) are
                // visited in order by this lane only (and receiver ranges of a split aperture are disjoint) -> deterministic.
                v2f Sf[4];                             // the stage's sum per frame
                frame_sums(Sf);
#pragma unroll
                for (int f = 0; f < NFR; ++f) {
                    if (wpix) Sf[f] = (v2f){wcur.x * Sf[f].x - wcur.y * Sf[f].y, wcur.x * Sf[f].y + wcur.y * Sf[f].x};
                    if (in_shard) {
                        float *q = (float *)((float2 *)P.y + (size_t)f * P.y_fstride + (size_t)(i1 + P.I1 * col - P.i_begin) + (size_t)n * P.y_ld);
                        unsafeAtomicAdd(q, Sf[f].x); unsafeAtomicAdd(q + 1, Sf[f].y);
                    }
                }
                if (wpix) wcur = wnext;
                acc = acc1 = acc2 = acc3 = (v2f){0.f, 0.f};
            } else if (wpix) {                         // weight the stage's partial sum (the weight does not depend on m)
                v2f Sf[4];
                frame_sums(Sf);
#pragma unroll
                for (int f = 0; f < NFR; ++f) tot[f] += (v2f){wcur.x * Sf[f].x - wcur.y * Sf[f].y, wcur.x * Sf[f].y + wcur.y * Sf[f].x};
                acc = acc1 = acc2 = acc3 = (v2f){0.f, 0.f};
                wcur = wnext;
            }
            if constexpr (LUT) tbc = tbn;
            if (++n == nlim(m0)) { n = n_lo; m0 = blk(++cr); }
        }
    };
    if (tile_interior) run(std::false_type{}); else run(std::true_type{});

#if QDAS_PROF
    pt_[5] = QDAS_TICK() - pstart_;                    // whole workgroup
    if (lane == 0 && (wave == 0 || wave == WAVES - 1) && blockIdx.x < 8192) {
        unsigned long long *o = qdas_prof_buf + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
        for (int k = 0; k < 8; ++k) o[k] = pt_[k];
    }
#endif
    if (syn) return;                                   // every stage already added its share to its plane
    v2f res[4];
    frame_sums(res);
    // ---- epilogue: y[i] = pix  (reference src/bf.cu:140); one image per frame of the launch
    {
        const uint64_t ig = i1 + P.I1 * col;
        if ((i1 < P.I1) && (col < ncols) && (ig >= P.i_begin) && (ig < i_end)) {
#pragma unroll
            for (int f = 0; f < NFR; ++f) {
                const v2f r = wpix ? tot[f] : res[f];
                // partial images of a split aperture are laid out [split][frame][pixel]
                if (S > 1) P.part[((size_t)split * NFR + f) * P.i_count + (size_t)(ig - P.i_begin)] = make_float2(r.x, r.y);
                else st((ST *)P.y + (size_t)f * P.y_fstride, (size_t)(ig - P.i_begin), cplx<float>{r.x, r.y});
            }
        }
    }
}

template <int INTERP, typename ST, int CI>
static hipError_t launch_tile_i(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    constexpr Cfg G = CFGS[CI];
    constexpr bool SYM = (CI == 1 || CI == 7 || CI == 8), FB2 = (CI == 3 || CI == 4), FB4 = (CI == 5 || CI == 6), BIG = (CI == 9), LUT = (CI == 10 || CI == 11);
    const bool fm = P.fmod != 0.0, wt = P.wtab != nullptr;
    const dim3 g(ntiles * (P.probe ? 1u : P.ksplit)), b(G.waves * 64);
#define QDAS_LAUNCH(FM, WT) QDAS_LAUNCH_P(FM, WT, false)
#define QDAS_LAUNCH_P(FM, WT, PR)                                                                        \
    do {                                                                                                 \
        auto kfn = das_tile_kernel<INTERP, ST, FM, WT, SYM, FB2, FB4, G.waves, G.mb, G.w, G.nbuf, G.psz, G.bpc, PR, BIG && !PR, LUT>; \
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                   \
        kfn<<<g, b, lds, s>>>(P);                                                                        \
    } while (0)
    if (P.probe) {
        if constexpr (FB2 || FB4) return hipErrorInvalidValue;    // the window fit does not depend on the frame count: probes use one frame
        else { QDAS_LAUNCH_P(false, false, true); return hipGetLastError(); }
    }
    if constexpr (SYM) {
        if (wt) return hipErrorInvalidValue;
        if (fm) QDAS_LAUNCH(true, false); else QDAS_LAUNCH(false, false);
    } else {
        if (fm && wt) QDAS_LAUNCH(true, true);
        else if (fm)  QDAS_LAUNCH(true, false);
        else if (wt)  QDAS_LAUNCH(false, true);
        else          QDAS_LAUNCH(false, false);
    }
#undef QDAS_LAUNCH
#undef QDAS_LAUNCH_P
    return hipGetLastError();
}


}  // namespace qdas
