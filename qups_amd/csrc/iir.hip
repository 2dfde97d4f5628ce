// iir.hip -- recursive (IIR) filtering of the traces in front of the DAS path: qdas_iir.
//
// Replaces what ChannelData.filter does with an IIR digitalFilter (reference src/ChannelData.m:857-888: filter(D, x) along the time dimension, then
// t0 -= filtord(D) / fs): MATLAB runs a digitalFilter's second-order sections one after the other, each as the direct-form II transposed recursion
//     y[t] = b0 x[t] + s1;   s1 = b1 x[t] - a1 y[t] + s2;   s2 = b2 x[t] - a2 y[t]                    (coefficients normalised by a0)
// The recursion is sequential in time and independent between traces: a lane owns a trace, a wave 64 traces.  Traces are time-contiguous (T x K), so the
// wave moves TILES of 64 traces x 32 samples through LDS -- global loads / stores run along time (coalesced: 256 bytes of one trace per half wave), the
// recursion reads its trace's row of the tile (padded rows: conflict-free) -- and all sections are applied to the tile before it is written back: one read
// and one write of the record whatever the filter order.  HBM-bound: 2 x sizeof(record).  The filter state is kept in double for fp32 data (the reference
// computes in the data's precision; the extra digits cost nothing in a memory-bound kernel and keep high-order filters from drifting).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/qdas.h"

namespace qdas {

constexpr int IIR_TS = 32;            // samples of a trace per tile
constexpr int IIR_MAXSEC = 16;

struct IirParams {
    const void *x; void *y;
    uint64_t T, K;
    int nsec;
    double sos[IIR_MAXSEC][5];       // b0 b1 b2 a1 a2 (a0 divided out), gain folded into the first section
};

template <typename R, bool CPLX>
__global__ void __launch_bounds__(64) iir_kernel(const IirParams P) {
    constexpr int W = CPLX ? 2 : 1;                       // reals per sample
    __shared__ R tile[64][IIR_TS * W + 1];                // (+1: rows start in different banks)
    const uint32_t lane = threadIdx.x;
    const uint64_t k0 = (uint64_t)blockIdx.x * 64;
    const uint64_t kk = k0 + lane;
    const R *x = (const R *)P.x;
    R *y = (R *)P.y;
    double s1r[IIR_MAXSEC], s2r[IIR_MAXSEC], s1i[IIR_MAXSEC], s2i[IIR_MAXSEC];
#pragma unroll
    for (int q = 0; q < IIR_MAXSEC; ++q) { s1r[q] = s2r[q] = s1i[q] = s2i[q] = 0.0; }
    constexpr int RW = IIR_TS * W;                        // reals per trace and tile
    constexpr int TPI = 64 / RW > 0 ? 64 / RW : 1;        // traces one wave-wide load covers (complex: 1, real: 2)
    for (uint64_t t0 = 0; t0 < P.T; t0 += IIR_TS) {
        const uint32_t nt = (uint32_t)(P.T - t0 < IIR_TS ? P.T - t0 : IIR_TS);
        // ---- load: lanes along time
        for (uint32_t r0 = 0; r0 < 64; r0 += TPI) {
            const uint32_t r = r0 + lane / RW, c = lane % RW;
            if (lane < (uint32_t)(TPI * RW) && k0 + r < P.K && c < nt * W) tile[r][c] = x[((k0 + r) * P.T + t0) * W + c];
        }
        __builtin_amdgcn_s_barrier();
        // ---- the recursion: every lane its own trace, section after section over the tile's samples
        if (kk < P.K) {
#pragma unroll
            for (int q = 0; q < IIR_MAXSEC; ++q) {
                if (q < P.nsec) {                          // (uniform)
                    const double b0 = P.sos[q][0], b1 = P.sos[q][1], b2 = P.sos[q][2], a1 = P.sos[q][3], a2 = P.sos[q][4];
                    for (uint32_t j = 0; j < nt; ++j) {
                        const double xr = (double)tile[lane][j * W];
                        const double yr = b0 * xr + s1r[q];
                        s1r[q] = b1 * xr - a1 * yr + s2r[q];
                        s2r[q] = b2 * xr - a2 * yr;
                        tile[lane][j * W] = (R)yr;
                        if constexpr (CPLX) {
                            const double xi = (double)tile[lane][j * W + 1];
                            const double yi = b0 * xi + s1i[q];
                            s1i[q] = b1 * xi - a1 * yi + s2i[q];
                            s2i[q] = b2 * xi - a2 * yi;
                            tile[lane][j * W + 1] = (R)yi;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        // ---- store: lanes along time
        for (uint32_t r0 = 0; r0 < 64; r0 += TPI) {
            const uint32_t r = r0 + lane / RW, c = lane % RW;
            if (lane < (uint32_t)(TPI * RW) && k0 + r < P.K && c < nt * W) y[((k0 + r) * P.T + t0) * W + c] = tile[r][c];
        }
        __builtin_amdgcn_s_barrier();
    }
}

hipError_t launch_iir(const IirParams &P, int dtype, int cplx, hipStream_t s) {
    const unsigned nb = (unsigned)((P.K + 63) / 64);
    if (dtype == QDAS_F64) { if (cplx) iir_kernel<double, true><<<nb, 64, 0, s>>>(P); else iir_kernel<double, false><<<nb, 64, 0, s>>>(P); }
    else                   { if (cplx) iir_kernel<float, true><<<nb, 64, 0, s>>>(P);  else iir_kernel<float, false><<<nb, 64, 0, s>>>(P); }
    return hipGetLastError();
}

}  // namespace qdas

void qdas_internal_set_error(const char *msg);          // qdas_api.hip: the library's thread-local last-error string

extern "C" int qdas_iir(const qdas_iir_desc *d, const void *x, void *y, void *stream) {
    if (!d) { qdas_internal_set_error("null argument"); return QDAS_EINVAL; }
    if (d->dtype != QDAS_F64 && d->dtype != QDAS_F32) { qdas_internal_set_error("iir: datatype must be double or single"); return QDAS_EINVAL; }
    if (d->nsec < 1 || d->nsec > qdas::IIR_MAXSEC || !d->sos) { qdas_internal_set_error("iir: 1 to 16 second-order sections [b0 b1 b2 a0 a1 a2]"); return QDAS_EINVAL; }
    if (d->K >= (1ull << 37)) { qdas_internal_set_error("iir: too many traces for one launch"); return QDAS_EUNSUPPORTED; }
    if (d->T == 0 || d->K == 0) return QDAS_OK;
    if (!x || !y) { qdas_internal_set_error("null data pointer"); return QDAS_EINVAL; }
    qdas::IirParams p{};
    p.x = x; p.y = y; p.T = d->T; p.K = d->K; p.nsec = d->nsec;
    for (int q = 0; q < d->nsec; ++q) {
        const double *c = d->sos + 6 * q;
        if (c[3] == 0.0 || !(c[3] == c[3])) { qdas_internal_set_error("iir: a0 of a section is zero"); return QDAS_EINVAL; }
        const double g = (q == 0 ? (d->gain != 0.0 ? d->gain : 1.0) : 1.0) / c[3];
        p.sos[q][0] = c[0] * g; p.sos[q][1] = c[1] * g; p.sos[q][2] = c[2] * g; p.sos[q][3] = c[4] / c[3]; p.sos[q][4] = c[5] / c[3];
    }
    int prev = -1;
    if (d->device >= 0) { if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(d->device) != hipSuccess) { qdas_internal_set_error("hipSetDevice failed"); return QDAS_EHIP; } }
    const hipError_t e = qdas::launch_iir(p, d->dtype, d->cplx ? 1 : 0, (hipStream_t)stream);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) { qdas_internal_set_error(hipGetErrorString(e)); return QDAS_EHIP; }
    return QDAS_OK;
}
