// qdas_kernels.h -- host<->kernel parameter blocks and launchers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>
#include <string>
#include "tile_params.h"
#include "plan_modes.h"

#ifndef QDAS_MAX_APOD
#define QDAS_MAX_APOD 6
#endif

namespace qdas {

// Parameter block of the generic kernel == the reference kernel's argument list
// (reference src/bf.cu:50-54) plus the size constants (src/sizes.cu) and the shard window.
struct GenericParams {
    const void *Pi, *Pr, *Pv, *Nv, *apod, *cinv, *x;
    void *y;
    uint64_t T, N, M, I1, I2, I3;
    uint64_t i_begin, i_count, y_ld;
    uint64_t cst[6];                    // cstride  (reference kern/das_spec.m:259)
    uint64_t ast[6 * QDAS_MAX_APOD];    // astride  (reference kern/das_spec.m:260)
    double fs, fmod;
    int32_t S, flag, VS, DV, apod_real;
    // generated receive apodization (qdas.h QDAS_RXAPOD_*): kind, parameters, element normals (3 x N, same type as Pr)
    int32_t gen_kind;
    double gen_p0, gen_p1;
    const void *rxn;
    // fallback-tile mode (tile_list != nullptr): process only the listed (1 << tile_zl) x tile_cols tiles;
    // tile_list[0] = count, tile_list[1..] = tile ids (written by the tiled kernel)
    const uint32_t *tile_list;
    uint32_t blocks_per_tile, tile_cols, tile_zl;
    uint64_t tiles_z;
};

hipError_t launch_generic(const GenericParams &P, int dtype, unsigned grid, hipStream_t s);
hipError_t launch_delays(const GenericParams &P, int dtype, void *tau, double cinv, hipStream_t s);

// ---- tiled kernel (das_tile_impl.h); parameter block in tile_params.h; TileConfig / tile_config / tile_lds_bytes / tile_lds_limit: plan_modes.h (pure host code)
// jit: plan-specialised kernel (jit.hip) to launch instead of the prebuilt instantiation; one frame per launch only
hipError_t launch_tile(const TileParams &P, int dtype, unsigned ntiles, hipStream_t s, hipFunction_t jit = nullptr, size_t jit_lds = 0);
// resolve (build on demand, if libqdas.so does not carry it) the instantiation launch_tile(P, ...) would run, without launching; *built = its cache key or ""
hipError_t prepare_tile(const TileParams &P, int dtype, unsigned ntiles, std::string *built);

// ---- reciprocity fold (fold.hip): xs[:,n,m] = w[n,m] x[:,n,m] + w[m,n] x[:,m,n] for n < m, xs[:,n,n] = w[n,n] x[:,n,n]; complex64, traces of T samples at
//      (n*strN + m*strM) samples; wtab: N x N float2 [n + N*m] or null (ones); only the upper triangle n <= m of xs is written
hipError_t launch_fold(const void *x, void *xs, const void *wtab, uint64_t T, uint64_t N, uint64_t strN, uint64_t strM, hipStream_t s, int in_f16 = 0);   // in_f16: x holds fp16 samples (xs is complex64 always)
hipError_t launch_y32_to_y16(const void *y32, void *y16, uint64_t n, hipStream_t s);       // complex64 image -> complex32 image (fp16 plans on the folded fp32 kernels)

// ---- split-delay kernel (das_lut.hip)
struct LutParams {
    const void *tau_rx, *tau_tx, *w, *x;
    void *y;
    uint64_t T, N, M, I;
    uint64_t wst[3];
    double omega;
    int32_t flag, w_real;
};
hipError_t launch_lut(const LutParams &P, int dtype, hipStream_t s);

// ---- general single-delay flavour over an N-D broadcast index space (wsinterpd.hip)
struct WsParams {
    const void *t, *w, *x;
    void *y;
    uint64_t T, x_tstride;           // samples per trace of x, element stride between consecutive samples
    int32_t nd;                      // dimensions of the index space (<= 8); dimension 0 is the sampling dimension
    uint64_t size[8];
    int64_t tst[8], xst[8], wst[8];  // element strides of t / x (trace base) / w per dimension; 0 = broadcast
    uint8_t sum[8];                  // 1: the dimension is summed
    uint64_t n_out, n_sum;           // product of the kept / summed sizes
    // the kept dimensions in decode order (kord[0]: the one the lanes run along), the product of the sizes of kord[1..], and y's element strides
    int32_t nkd, kord[8];
    uint64_t n_rest, n_lane;         // (n_lane: outputs the lanes of grid.x cover -- size[kord[0]], times size[kord[1]] with lane2)
    int32_t lane2;
    int32_t stream_ok;               // the lean streaming kernel applies (no sums / weights / rotation, 32-bit extents, dimension 0 outside the lane dimensions)
    int64_t yst[8];
    // the summed dimensions alone, compacted (fastest first): the kernel walks them like an odometer -- uniform scalar adds per term
    // instead of a 64-bit divide + modulo per dimension and term
    int32_t nsd;
    uint32_t ssz[8];
    int64_t sts[8], sxs[8], sws[8];
    double omega, extrap;
    int32_t flag, w_real, any_sum;
    int32_t lanesum_ok;              // one summed dimension, and x is contiguous along it: the lanes of a wave run along the SUM (wsinterpd_lanesum_kernel)
};
hipError_t launch_wsinterpd(const WsParams &P, int dtype, hipStream_t s);

// ---- transmit synthesis (shiftsum.hip): y[t', n, m'] = sum_m w[m, m'] x(t' + s[m, m'], n, m)
struct ShiftParams {
    const void *x; void *y; const void *tab, *blk;
    uint64_t T, To, N, M, Mo, F;
    uint32_t mo_blocks;
    uint64_t Tx;                     // samples per trace actually stored (T - tpad): the rest of [0, T) reads as zero
};
hipError_t launch_shift_sum(const ShiftParams &P, int dtype, int cplx, int interp, const void *sh, const void *w, int w_real, hipStream_t s);

// ---- point-scatterer simulator (greens.hip)
struct GreensParams {
    const void *Ps, *a, *Pr, *Pv, *x;
    void *y;
    uint64_t S, T, N, M, I;
    int32_t En, Em, interp, x_in_lds;
    double s0, t0, fs, fsr, cinv, R0;
    int32_t q;                       // impulse-train kernel (greens.hip): the integer waveform-to-data sampling ratio, output samples per workgroup
    uint32_t sb, nblk;
    const float *r1tab, *r2tab;      // ... and the scatterer-to-element distances, [N En][I] and [M Em][I]
    const float *cb1, *cb2;          // ... their {min, max} per chunk of 256 scatterers, [N En][nchunk][2] and [M Em][nchunk][2]
    uint32_t nchunk;
    int32_t dbg;
    double path_per_fine, path_off;  // path length r1 + r2 per fine (waveform-rate) sample, and of the time offset t0 - s0
    const void *segs, *xtab;         // the convolution's groups of 8 taps: {count, element offset per group}, and the taps (greens_xtab_kernel)
    uint32_t pb_off, x_off;          // LDS byte offsets: slice sums of the convolution, the waveform
};
hipError_t launch_greens(const GreensParams &P, int dtype, hipStream_t s);

// ---- temporaries of one call on one stream (scratch.hip): from a kept, hipMalloc'ed arena per (device, stream) -- not from the stream-ordered pool
class Scratch {
public:
    explicit Scratch(hipStream_t s);
    ~Scratch();
    void *get(size_t bytes);         // 256-byte aligned; nullptr: no memory
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
private:
    hipStream_t stream_;
    void *arena_ = nullptr;
    size_t off_ = 0;
    std::vector<void *> big_;
};
void scratch_trim();

// ---- batched 1-D convolution (conv.hip)
struct ConvParams {
    const void *x, *y;
    void *z;
    uint64_t C, M, N, L, S;          // x: C x M x S, y: C x N x S, z: C x L x S
    int64_t off;                     // index of output 0 in the full convolution
    uint64_t xcs, xts, xss;          // element strides of x: column, time, slice (0 = broadcast)
    uint64_t ycs, yts, yss;
};
hipError_t launch_conv(const ConvParams &P, int dtype, int cplx, int y_real, hipStream_t s);
// FFT convolution of K complex64 traces of M samples with one filter of `ntaps` (real fp32 | complex64) taps: outputs [off, off + L) of the linear convolution
// (pre.hip).  0: launched, 1: not this path, 2: HIP error
int fftconv_launch(const void *x, const void *taps, int taps_real, void *z, uint64_t M, uint64_t ntaps, uint64_t K, uint64_t off, uint64_t L, hipStream_t s);

// ---- layout.hip: out[c][b][a] = in[a][b][c]
hipError_t launch_permute3(const void *in, void *out, uint64_t A, uint64_t B, uint64_t C, int elem_bytes, hipStream_t s);

}  // namespace qdas
