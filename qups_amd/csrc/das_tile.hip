// das_tile.hip -- host side of the tiled kernel: launch configurations, LDS budget, dispatch to the per-configuration
// translation units (das_tile_{f32,f32big,lut,luth,bf,sym,symw,symh,f16,f32x2,f16x2,f32x4,f16x4,f64}.hip, kernel in das_tile_impl.h) and the fixed-order
// reduce of a split aperture.  -DQDAS_UNITY compiles everything as ONE translation unit (profiling / ablation builds that
// pass -DQDAS_ABL / -DQDAS_PROF: tools/ablate.sh).
#ifdef QDAS_UNITY
#include "das_tile_f32.hip"
#include "das_tile_f16.hip"
#include "das_tile_f32x2.hip"
#include "das_tile_f16x2.hip"
#include "das_tile_f32x4.hip"
#include "das_tile_f16x4.hip"
#include "das_tile_symh.hip"
#include "das_tile_f32big.hip"
#include "das_tile_lut.hip"
#include "das_tile_luth.hip"
#include "das_tile_bf.hip"
#include "das_tile_f64.hip"
#include "das_tile_f32w.hip"
#include "das_tile_symqh.hip"
#include "das_tile_fold.hip"
#else
#include "qdas_device.h"
#include "das_tile_cfg.h"
#endif
#include "qdas_kernels.h"
#include "jit.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>

namespace qdas {

hipError_t launch_tile_f32(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f16(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f32x2(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f16x2(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f32x4(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f16x4(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_symh(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f32big(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_lut(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_luth(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_bf(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f64(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_f32w(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_symqh(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);
hipError_t launch_tile_fold(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s);

// y[i] = sum over the ksplit partial images, in split order (deterministic)
// (fp64 data: complex128 partial images)
__global__ void __launch_bounds__(256) tile_reduce_kernel_f64(const double2 *part, double2 *y, uint64_t count, uint32_t ksplit) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double2 a = part[i];
    for (uint32_t j = 1; j < ksplit; ++j) { const double2 b = part[(size_t)j * count + i]; a.x += b.x; a.y += b.y; }
    y[i] = a;
}
template <typename ST>
__global__ void __launch_bounds__(256) tile_reduce_kernel(const float2 *part, ST *y, uint64_t count, uint32_t ksplit, uint64_t stride) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float2 a = part[i];
    for (uint32_t j = 1; j < ksplit; ++j) { const float2 b = part[(size_t)j * stride + i]; a.x += b.x; a.y += b.y; }
    st(y, (size_t)i, cplx<float>{a.x, a.y});
}



// Which prebuilt instantiations does a workload launch?  $QDAS_KERNEL_CENSUS=<file> appends one line "ci interp sample_bytes fm wt probe" per distinct
// instantiation and process (tools/kernel_census.py sums the files of a test run: the prebuilt matrix is what the suite and the benches use, the rest
// of the template's variants are built on demand -- das_tile_cfg.h tile_prebuilt).
void tile_census(int ci, int interp, int sample_bytes, bool fm, bool wt, bool probe) {
    static const char *path = getenv("QDAS_KERNEL_CENSUS");
    if (!path || !*path) return;
    static std::mutex mu;
    static std::set<uint32_t> seen;
    const uint32_t key = (uint32_t)ci | ((uint32_t)interp << 8) | ((uint32_t)sample_bytes << 12) | (fm ? 1u << 20 : 0u) | (wt ? 1u << 21 : 0u) | (probe ? 1u << 22 : 0u);
    std::lock_guard<std::mutex> lk(mu);
    if (!seen.insert(key).second) return;
    if (FILE *f = fopen(path, "a")) { fprintf(f, "%d %d %d %d %d %d\n", ci, interp, sample_bytes, (int)fm, (int)wt, (int)probe); fclose(f); }
}

// Resolving a plan's kernels without launching them (qdas_plan_create): instantiations that libqdas.so does not carry are compiled here, so that
// no execute ever waits for a compiler (das_tile_impl.h QDAS_LAUNCH_P, jit.hip lazy_tile_launch).
static thread_local bool g_prepare_only = false;
bool tile_prepare_only() { return g_prepare_only; }
hipError_t prepare_tile(const TileParams &P, int dtype, unsigned ntiles, std::string *built) {
    lazy_tile_reset();
    g_prepare_only = true;
    const hipError_t e = launch_tile(P, dtype, ntiles ? ntiles : 1u, nullptr);
    g_prepare_only = false;
    if (built) *built = lazy_tile_last();
    return e;
}

// the launcher's view of a parameter block (plan_modes.h launch_legal: the admission rules, shared with the GPU-less mode tests)
modes::LaunchShape launch_shape(const TileParams &P, int dtype, bool jit) {
    modes::LaunchShape L;
    L.dtype = dtype; L.sym = P.sym ? 1 : 0; L.fold = P.fold; L.mir = P.mir; L.narrow_raw = P.narrow; L.big = P.big; L.bf = P.bf; L.syn = P.syn; L.stage_shift = P.stage_shift;
    L.probe = P.probe; L.nfr = P.nfr; L.lut = P.lut_tx != nullptr; L.has_wtab = P.wtab != nullptr; L.has_apix = P.apix != nullptr; L.has_bpix = P.bpix != nullptr;
    L.has_part = P.part != nullptr; L.jit = jit; L.gen_kind = P.gen_kind; L.fmod = P.fmod; L.act_bytes = P.act_bytes; L.ksplit = P.ksplit; L.N = P.N; L.M = P.M;
    return L;
}

hipError_t launch_tile(const TileParams &P, int dtype, unsigned ntiles, hipStream_t s, hipFunction_t jit, size_t jit_lds) {
    if (ntiles == 0) return hipSuccess;
    modes::LaunchChoice ch;
    if (modes::launch_legal(launch_shape(P, dtype, jit != nullptr), &ch)) return hipErrorInvalidValue;
    const int sym = P.sym ? 1 : 0;
    if (dtype == 0) {                                    // fp64 data: one frame, one workgroup per tile, plain 'DAS' sum, prebuilt kernels
        const size_t lds64 = ch.lds;
        hipError_t e64 = jit ? jit_launch(jit, P, ntiles * P.ksplit, (unsigned)CFGS[13].waves * 64u, lds64, s) : launch_tile_f64(P, ntiles, lds64, s);
        if (e64 != hipSuccess || P.probe || P.ksplit <= 1 || g_prepare_only) return e64;
        tile_reduce_kernel_f64<<<(unsigned)((P.i_count + 255) / 256), 256, 0, s>>>((const double2 *)P.part, (double2 *)P.y, P.i_count, P.ksplit);
        return hipGetLastError();
    }
    const int narrow = ch.narrow, fold = ch.fold, mirq = ch.mirq, nfr = ch.nfr, nf = ch.nf;
    const bool probe_f32sym = ch.probe_f32sym;
    const size_t lds = ch.lds;
    hipError_t e = jit ? jit_launch(jit, P, ntiles * P.ksplit, (unsigned)CFGS[cfg_index(dtype, sym, 1, narrow, mirq, fold)].waves * 64u, jit_lds ? jit_lds : lds, s) : (fold || probe_f32sym) ? launch_tile_fold(P, ntiles, lds, s) : P.lut_tx ? (dtype == 2 ? launch_tile_luth(P, ntiles, lds, s) : launch_tile_lut(P, ntiles, lds, s)) : (mirq && dtype == 2) ? launch_tile_symqh(P, ntiles, lds, s) : (sym && dtype == 2) ? launch_tile_symh(P, ntiles, lds, s) : sym ? hipErrorInvalidValue
                 : nf == 4 ? (dtype == 2 ? launch_tile_f16x4(P, ntiles, lds, s) : launch_tile_f32x4(P, ntiles, lds, s))
                 : nf == 2 ? (dtype == 2 ? launch_tile_f16x2(P, ntiles, lds, s) : launch_tile_f32x2(P, ntiles, lds, s))
                           : (dtype == 2 ? launch_tile_f16(P, ntiles, lds, s) : narrow == 2 ? launch_tile_f32w(P, ntiles, lds, s) : (P.bf && !P.probe) ? launch_tile_bf(P, ntiles, lds, s) : (P.big && !P.probe) ? launch_tile_f32big(P, ntiles, lds, s) : launch_tile_f32(P, ntiles, lds, s));
    if (e != hipSuccess || P.probe || P.ksplit <= 1 || P.syn || P.bf || g_prepare_only) return e;   // ('SYN' planes are accumulated in place, 'BF' planes stored by their owners)
    const uint64_t oc = P.mir == 2 ? 2 * P.i_count : P.i_count;     // pixels the plan writes (a mirror slab: slab A and its image)
    const unsigned rb = (unsigned)((oc + 255) / 256);
    for (int f = 0; f < nfr; ++f) {                      // partial images: [split][frame][pixel]
        const float2 *src = P.part + (size_t)f * oc;
        const uint64_t stride = (uint64_t)nfr * oc;
        if (dtype == 2) tile_reduce_kernel<uint32_t><<<rb, 256, 0, s>>>(src, (uint32_t *)P.y + (size_t)f * P.y_fstride, oc, P.ksplit, stride);
        else            tile_reduce_kernel<float2><<<rb, 256, 0, s>>>(src, (float2 *)P.y + (size_t)f * P.y_fstride, oc, P.ksplit, stride);
    }
    return hipGetLastError();
}


}  // namespace qdas
