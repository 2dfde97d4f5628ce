// tile_util.h -- small device-side building blocks of the tiled kernel (internal): packed-fp32 tap weights, DPP wave reductions,
// compile-time unrolling, and the cold fp64 helpers that are deliberately kept out of line.
#pragma once
#include "qdas_device.h"
#include "lanczos_poly.h"
#include "lanczos_poly64.h"
#include <type_traits>
#include <utility>

namespace qdas {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr float MAGIC = 12582912.0f;          // 1.5 * 2^23: (t + MAGIC) has rint(t) in its low mantissa bits
constexpr uint32_t MAGIC_BITS = 0x4B400000u;
constexpr double MAGIC64 = 6755399441055744.0;   // 1.5 * 2^52: the LOW WORD of (t + MAGIC64) is rint(t) as a two's-complement int32

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

// fp64 data: the same weights for ONE sample, in double (cubic exact; Lanczos: degree-6 polynomials in q, |error| < 2e-12: lanczos_poly64.h)
// Four polynomials of the same degree in lock-step: four independent FMA chains (a dependent v_fma_f64 cannot issue back to back).
// Every coefficient passes through an opaque scalar register right where it is used: left alone, the compiler keeps all the double
// constants live across the 16 unrolled samples of a stage -- > 50 SGPRs, which spill into VGPR lanes and from there into scratch;
// one s_mov_b64 per use rides on the scalar unit, beside the other waves' vector work.  The LEADING coefficients arrive in vector
// registers (`lead`, made once per stage by the caller): an FMA reads one scalar operand only.
template <int D> __device__ __forceinline__ void horner4(const double (&a)[D + 1], const double (&b)[D + 1], const double (&c)[D + 1], const double (&d)[D + 1],
                                                         const double (&lead)[4], double q, double &ra, double &rb, double &rc, double &rd) {
    auto k_ = [](double v) { asm volatile("" : "+s"(v)); return v; };
    ra = __builtin_fma(lead[0], q, k_(a[D - 1])); rb = __builtin_fma(lead[1], q, k_(b[D - 1]));
    rc = __builtin_fma(lead[2], q, k_(c[D - 1])); rd = __builtin_fma(lead[3], q, k_(d[D - 1]));
#pragma unroll
    for (int k = D - 2; k >= 0; --k) {
        ra = __builtin_fma(ra, q, k_(a[k])); rb = __builtin_fma(rb, q, k_(b[k]));
        rc = __builtin_fma(rc, q, k_(c[k])); rd = __builtin_fma(rd, q, k_(d[k]));
    }
}
// the four leading coefficients of the fp64 Lanczos polynomials, in vector registers (nothing for the other interpolators)
template <int INTERP, bool PIN = true> __device__ __forceinline__ void weights1_lead(double (&lead)[4]) {
    if constexpr (INTERP == 3) {
        constexpr double EI[] = QDAS_LANCZOS64_EI, OI[] = QDAS_LANCZOS64_OI, EO[] = QDAS_LANCZOS64_EO, OO[] = QDAS_LANCZOS64_OO;
        constexpr int D = sizeof(EI) / 8 - 1;
        lead[0] = EI[D]; lead[1] = OI[D]; lead[2] = EO[D]; lead[3] = OO[D];
        if constexpr (PIN) asm volatile("" : "+v"(lead[0]), "+v"(lead[1]), "+v"(lead[2]), "+v"(lead[3]));   // (kept in registers unless the variant has none to spare)
    } else { lead[0] = lead[1] = lead[2] = lead[3] = 0.0; }
}
template <int INTERP> __device__ __forceinline__ void weights1(double s, double (&w)[4], const double (&lead)[4]) {
    if constexpr (INTERP == 1 || INTERP == 4) {
        w[0] = 0.5 - s; w[1] = 0.5 + s;
    } else if constexpr (INTERP == 2) {
        const double q = s * s;
        const double ei = 0.5625 - 0.25 * q, oi = -1.375 + 1.5 * q;
        const double eo = -0.0625 + 0.25 * q, oo = 0.125 - 0.5 * q;
        w[1] = ei + s * oi; w[2] = ei - s * oi; w[0] = eo + s * oo; w[3] = eo - s * oo;
    } else if constexpr (INTERP == 3) {
        constexpr double EI[] = QDAS_LANCZOS64_EI, OI[] = QDAS_LANCZOS64_OI, EO[] = QDAS_LANCZOS64_EO, OO[] = QDAS_LANCZOS64_OO;
        const double q = s * s;
        static_assert(sizeof(EI) == sizeof(OI) && sizeof(EI) == sizeof(EO) && sizeof(EI) == sizeof(OO), "one degree");
        double ei, oi, eo, oo;
        horner4<sizeof(EI) / 8 - 1>(EI, OI, EO, OO, lead, q, ei, oi, eo, oo);
        w[1] = ei + s * oi; w[2] = ei - s * oi; w[0] = eo + s * oo; w[3] = eo - s * oo;
    } else if constexpr (INTERP == 5) {
        const double u = s + 0.5;
        w[0] = 0.5 * (u * (-1.0 + u * (2.0 * u - 1.0)));
        w[1] = 0.5 * (2.0 + u * (u * (-5.0 * u + 3.0)));
        w[2] = 0.5 * (u * (1.0 + u * (4.0 * u - 3.0)));
        w[3] = 0.5 * (u * (u * (1.0 - u)));
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
// Four minima and four maxima at once, one instruction per value and step (the result is valid in lane 63 only).  Written as
// v_min/v_max_f32_dpp in ONE asm block: through the builtin each step compiles to a copy, a v_mov_dpp, a canonicalising v_max and
// the v_min -- four instructions per value and step, 192 per call of the prologue's inner loop instead of 48.  The eight chains are
// interleaved, so a DPP read follows the write of its register by seven instructions (the 2-wait-state hazard needs no s_nop);
// the leading s_nop covers whatever wrote the inputs.  The inputs must not be NaN (the caller maps NaN to +-inf).
__device__ __forceinline__ void wave_minmax63x4(float (&lo)[4], float (&hi)[4]) {
#define QDAS_DPP_STEP(ctrl)                                                                                                       \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_min_f32_dpp %1, %1, %1 " ctrl "\n\tv_min_f32_dpp %2, %2, %2 " ctrl "\n\tv_min_f32_dpp %3, %3, %3 " ctrl "\n\t" \
    "v_max_f32_dpp %4, %4, %4 " ctrl "\n\tv_max_f32_dpp %5, %5, %5 " ctrl "\n\tv_max_f32_dpp %6, %6, %6 " ctrl "\n\tv_max_f32_dpp %7, %7, %7 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 QDAS_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 QDAS_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 QDAS_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 QDAS_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 QDAS_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 QDAS_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
#undef QDAS_DPP_STEP
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

template <int... Is, typename F> __device__ __forceinline__ void unroll_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void unroll(F &&f) { unroll_impl(std::make_integer_sequence<int, N>{}, f); }

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
        const float s0 = __builtin_amdgcn_sqrtf((float)d2);                  // fp32 seed + one Newton step (as dsqrt in the kernel)
        const double sd = (double)s0;
        const double r = __builtin_fma(-sd, sd, d2);
        const double len = __builtin_fma(r, (double)(0.5f * __builtin_amdgcn_rcpf(fmaxf(s0, 1.0e-30f))), sd);
        dv = kindB == 0 ? len : copysign(len, dot);
    }
    return (float)((dv * cf - (double)Pv[4 * m + 3] * fs + off) - ((double)Abase_m + 0.5));
}
// fp64 data: the same from fp64 geometry tables, returned in double (IEEE sqrt: the residual is the whole delay precision here)
typedef __attribute__((address_space(3))) const double lds_cdouble;
static __device__ __noinline__ double block_residual64(double px, double py, double pz, double cf, double fs, int kindB, lds_cdouble *Pv, lds_cdouble *Nv,
                                                       uint32_t m, int Abase_m, double off) {
    const double rx = px - Pv[4 * m], ry = py - Pv[4 * m + 1], rz = pz - Pv[4 * m + 2];
    const double dot = kindB ? rx * Nv[3 * m] + ry * Nv[3 * m + 1] + rz * Nv[3 * m + 2] : 0.0;
    double dv = dot;
    if (kindB != 2) { const double len = sqrt(rx * rx + ry * ry + rz * rz); dv = kindB == 0 ? len : copysign(len, dot); }
    return (dv * cf - Pv[4 * m + 3] * fs + off) - ((double)Abase_m + 0.5);
}
// cos and sin of 2 pi r for |r| <= 1/2 in fp64 (remodulation of fp64 data: the reference's cospi / sinpi, src/bf.cu:117): quarter-turn
// reduction, then the Taylor polynomials on |y| <= pi/4 (truncation below 1e-16); branch-free, about 27 fp64 operations
static __device__ __forceinline__ void sincos2pi_f64(double r, double &c, double &s) {
    const double q = __builtin_rint(4.0 * r);                             // -2 .. 2
    const double y = (r - 0.25 * q) * 6.283185307179586476925;
    const double y2 = y * y;
    double sp = -1.0 / 1307674368000.0;
    sp = __builtin_fma(sp, y2, 1.0 / 6227020800.0);
    sp = __builtin_fma(sp, y2, -1.0 / 39916800.0);
    sp = __builtin_fma(sp, y2, 1.0 / 362880.0);
    sp = __builtin_fma(sp, y2, -1.0 / 5040.0);
    sp = __builtin_fma(sp, y2, 1.0 / 120.0);
    sp = __builtin_fma(sp, y2, -1.0 / 6.0);
    const double sy = __builtin_fma(sp * y2, y, y);
    double cp = 1.0 / 20922789888000.0;
    cp = __builtin_fma(cp, y2, -1.0 / 87178291200.0);
    cp = __builtin_fma(cp, y2, 1.0 / 479001600.0);
    cp = __builtin_fma(cp, y2, -1.0 / 3628800.0);
    cp = __builtin_fma(cp, y2, 1.0 / 40320.0);
    cp = __builtin_fma(cp, y2, -1.0 / 720.0);
    cp = __builtin_fma(cp, y2, 1.0 / 24.0);
    cp = __builtin_fma(cp, y2, -0.5);
    const double cy = __builtin_fma(cp, y2, 1.0);
    const int qi = (int)q & 3;                                           // quarter turns: 0: (c, s); 1: (-s, c); 2: (-c, -s); 3: (s, -c)
    const double cc = (qi & 1) ? sy : cy, ss = (qi & 1) ? cy : sy;
    c = (qi == 1 || qi == 2) ? -cc : cc;
    s = (qi >= 2) ? -ss : ss;
}
// generated pixel x receiver weight (qdas.h QDAS_RXAPOD_*): element position from the LDS record, normal by scalar loads
static __device__ __noinline__ float rx_apod_generated(int kind, double p0, double p1, float px, float py, float pz, float ex, float ey, float ez,
                                                       const float *rxn, uint32_t n) {
    const float nx = rxn ? rxn[3 * n] : 0.f, ny = rxn ? rxn[3 * n + 1] : 0.f, nz = rxn ? rxn[3 * n + 2] : 1.f;
    return (float)rx_apod_weight(kind, p0, p1, (double)px - (double)ex, (double)py - (double)ey, (double)pz - (double)ez,
                                 (double)nx, (double)ny, (double)nz, (double)px, (double)pz, (double)ex);
}

}  // namespace qdas
