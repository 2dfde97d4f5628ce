// tile_taps.h -- LDS tap gathers of the tiled kernel (internal): inline-asm ds_read issue / fence pairs for fp32 and fp16 data and
// the fp16 x fp32 + fp32 MAC (v_fma_mix_f32).
#pragma once
#include "tile_util.h"

namespace qdas {

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
// (K: taps per sample -- only the registers that were actually loaded are tied, so that 1- and 2-tap interpolators do not pin
//  registers for taps they never read)
template <int K> __device__ __forceinline__ void lds_fence(taps_f32 &a, taps_f32 &b, v2f (&w)[4]) {
    if constexpr (K == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(b.s[2]), "+v"(b.s[3]),
                       "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (K == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(w[0]), "+v"(w[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.s[0]), "+v"(b.s[0]));
}
template <int K> __device__ __forceinline__ void lds_fence2(taps_f32 &a, taps_f32 &b, taps_f32 &c, taps_f32 &d, v2f (&w)[4]) {
    if constexpr (K == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(b.s[2]), "+v"(b.s[3]),
                       "+v"(c.s[0]), "+v"(c.s[1]), "+v"(c.s[2]), "+v"(c.s[3]), "+v"(d.s[0]), "+v"(d.s[1]), "+v"(d.s[2]), "+v"(d.s[3]),
                       "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (K == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(c.s[0]), "+v"(c.s[1]), "+v"(d.s[0]), "+v"(d.s[1]), "+v"(w[0]), "+v"(w[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.s[0]), "+v"(b.s[0]), "+v"(c.s[0]), "+v"(d.s[0]));
}
// counted variant for the software-pipelined loop: the NEWEST `KEEP` LDS reads (the next iteration's direct taps) stay in flight
template <int KEEP> __device__ __forceinline__ void lds_fence2_keep(taps_f32 &a, taps_f32 &b, taps_f32 &c, taps_f32 &d, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%20)"
                 : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(b.s[0]), "+v"(b.s[1]), "+v"(b.s[2]), "+v"(b.s[3]),
                   "+v"(c.s[0]), "+v"(c.s[1]), "+v"(c.s[2]), "+v"(c.s[3]), "+v"(d.s[0]), "+v"(d.s[1]), "+v"(d.s[2]), "+v"(d.s[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])
                 : "n"(KEEP));
}
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
template <int K> __device__ __forceinline__ void lds_fence(taps_f16 &a, taps_f16 &b, v2f (&w)[4]) {
    if constexpr (K == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]),
                       "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (K == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(w[0]), "+v"(w[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.r[0]), "+v"(b.r[0]));
}
template <int K> __device__ __forceinline__ void lds_fence2(taps_f16 &a, taps_f16 &b, taps_f16 &c, taps_f16 &d, v2f (&w)[4]) {
    if constexpr (K == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]),
                       "+v"(c.r[0]), "+v"(c.r[1]), "+v"(c.r[2]), "+v"(c.r[3]), "+v"(d.r[0]), "+v"(d.r[1]), "+v"(d.r[2]), "+v"(d.r[3]),
                       "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (K == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(c.r[0]), "+v"(c.r[1]), "+v"(d.r[0]), "+v"(d.r[1]), "+v"(w[0]), "+v"(w[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.r[0]), "+v"(b.r[0]), "+v"(c.r[0]), "+v"(d.r[0]));
}
template <int KEEP> __device__ __forceinline__ void lds_fence2_keep(taps_f16 &a, taps_f16 &b, taps_f16 &c, taps_f16 &d, v2f (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%20)"
                 : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]),
                   "+v"(c.r[0]), "+v"(c.r[1]), "+v"(c.r[2]), "+v"(c.r[3]), "+v"(d.r[0]), "+v"(d.r[1]), "+v"(d.r[2]), "+v"(d.r[3]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])
                 : "n"(KEEP));
}
// fp64 data: 16-byte samples {re, im}: K x ds_read_b128 (windows are 16-byte aligned), one sample at a time
typedef double v2d __attribute__((ext_vector_type(2)));
struct taps_f64 { v2d s[4]; };
template <int K, int OFF> __device__ __forceinline__ void lds_issue(taps_f64 &t, uint32_t addr) {
    if constexpr (K == 4)
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=&v"(t.s[0]), "=&v"(t.s[1]), "=&v"(t.s[2]), "=&v"(t.s[3]) : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48));
    else if constexpr (K == 2)
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(t.s[0]), "=&v"(t.s[1]) : "v"(addr), "n"(OFF), "n"(OFF + 16));
    else
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(t.s[0]) : "v"(addr), "n"(OFF));
}
template <int K> __device__ __forceinline__ void lds_fence(taps_f64 &a, double (&w)[4]) {
    if constexpr (K == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(a.s[2]), "+v"(a.s[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (K == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.s[0]), "+v"(a.s[1]), "+v"(w[0]), "+v"(w[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.s[0]));
}
__device__ __forceinline__ void mix_mac(v2f &acc, uint32_t tap, float w) {       // acc += w * (float2)tap
    float ar = acc.x, ai = acc.y;
    asm("v_fma_mix_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(ar), "+v"(ai) : "v"(tap), "v"(w));
    acc = (v2f){ar, ai};
}
// acc += v * (c + i s): complex rotation folded into the accumulation ('modulation': two packed FMAs per sample)
__device__ __forceinline__ void rot_acc(v2f &acc, v2f v, float c, float s) {
    acc = v * c + acc;
    acc = (v2f){v.y, v.x} * (v2f){-s, s} + acc;
}
// acc += v * (wr + i wi), a weight-table entry; plan-specialised builds of a plan whose table is real (apodization windows) drop the second FMA
__device__ __forceinline__ void wgt_acc(v2f &acc, v2f v, float wr, float wi) {
    if (QSPEC(WREAL, 0)) acc = v * wr + acc;
    else rot_acc(acc, v, wr, wi);
}
// acc += w * tap k, either data type (software-pipelined loop)
__device__ __forceinline__ void tap_mac(v2f &acc, const taps_f32 &t, int k, float w) { acc = w * t.s[k] + acc; }
__device__ __forceinline__ void tap_mac(v2f &acc, const taps_f16 &t, int k, float w) { mix_mac(acc, t.r[k], w); }
}  // namespace qdas
