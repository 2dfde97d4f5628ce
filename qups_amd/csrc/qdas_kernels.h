// qdas_kernels.h -- host<->kernel parameter blocks and launchers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QDAS_MAX_APOD 6

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

// ---- tiled kernel (das_tile_impl.h)
struct TileParams {
    // fp32 geometry (QDAS_F32 / QDAS_F16 only).  Pv (4 x M: position, t0) and Nv (3 x M) describe the BLOCK elements of a stage,
    // Pr (3 x N) and -- when kindS != 0 -- St (4 x N: t0, normal) the STAGE elements; kindB / kindS: 0 distance, 1 signed distance,
    // 2 plane wave.  'DAS' / 'SYN': block = transmits, stage = receivers; 'MUL': roles (and N, M, strN, strM, wtab) swapped by the host.
    const float *Pi, *Pr, *Pv, *Nv, *St;
    int32_t kindB, kindS;
    const void *x;
    void *y;
    const void *wtab;                   // optional N x M table of folded apodization weights (float2), may be null
    const void *apix;                   // optional I x N apodization (data precision; real if apix_real), may be null
    int32_t apix_real;
    int32_t gen_kind;                   // generated pixel x receiver apodization (QDAS_RXAPOD_*), exclusive with apix
    double gen_p0, gen_p1;
    const float *rxn;                   // 3 x N element normals (device)
    uint64_t T, N, M, I1, I2, I3;
    uint64_t i_begin, i_count;
    uint64_t strN, strM;                // trace strides of x in samples: (T, T*N) or (T*M, T) when transposed
    double cinv_fs;                     // cinv * fs (scalar sound speed; first pixel's value when cinv_pix is set)
    const float *cinv_pix;              // optional per-pixel 1/c (I1 x I2 x I3, contiguous): sound-speed map; the delay stays separable
    double fs, fmod;
    int32_t flag, VS, DV;
    int32_t narrow;                     // reciprocal mode: the 128-sample-window configuration (chosen by the plan when every tile fits)
    int32_t big;                        // general mode, fp32: re-base the DMA descriptors along the receiver walk (transposed frames > 2 GiB)
    int32_t sym;                        // reciprocal mode: Pv == Pr, one t0 (checked by the host) -> tau(n,m) == tau(m,n)
    int32_t tz_log2;                    // tile footprint: (1 << tz_log2) pixels of I1 x (waves * 64 >> tz_log2) columns; 3..6
    int32_t wz_log2;                    // wave footprint inside the tile: (1 << wz_log2) pixels of I1 x (64 >> wz_log2) columns; <= tz_log2
    int32_t probe;                      // 1: stop after the window-fit test (plan-time shape selection; only fallback_list is written)
    uint32_t tiles_z, tiles_x, tile_x0; // tile grid over (I1 >> tz_log2) x (columns / tile columns); first column tile of the shard
    int32_t syn;                        // 1: 'SYN' -- keep the receive dimension: y is I x N planes (leading dimension y_ld), zero-filled by the host
    uint64_t y_ld;
    int32_t nfr;                        // frames per launch: 1, or 2 / 4 (frame f at x + f*x_fstride -> y + f*y_fstride); 1 with sym
    uint64_t x_fstride, y_fstride;      // frame strides: BYTES of x, ELEMENTS of y
    uint32_t ksplit;                    // workgroups per tile (>= 1): each sums a slice of the aperture into part[], then reduced into y
    float2 *part;                       // [ksplit][nfr][i_count] partial images (ksplit > 1 only)
    uint32_t *fallback_list;            // [0] = count, [1..] = tile ids that did not fit the LDS window
    uint32_t fallback_cap;
    // table-driven delays (launch configuration 10, qdas_das_lut): tau_tx (I x M) and tau_rx (I x N) in samples, fp32
    const float *lut_tx, *lut_rx;
};

struct TileConfig { int waves; int mb; int window; size_t lds_bytes; int threads; };
TileConfig tile_config(int dtype, int sym, int narrow = 0);
size_t tile_lds_bytes(int dtype, int sym, uint64_t N, uint64_t M, int narrow = 0);   // dynamic LDS of one workgroup
size_t tile_lds_limit(int sym);                               // LDS budget of one workgroup in that configuration
hipError_t launch_tile(const TileParams &P, int dtype, unsigned ntiles, hipStream_t s);

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

// ---- point-scatterer simulator (greens.hip)
struct GreensParams {
    const void *Ps, *a, *Pr, *Pv, *x;
    void *y;
    uint64_t S, T, N, M, I;
    int32_t En, Em, interp, x_in_lds;
    double s0, t0, fs, fsr, cinv, R0;
};
hipError_t launch_greens(const GreensParams &P, int dtype, hipStream_t s);

// ---- batched 1-D convolution (conv.hip)
struct ConvParams {
    const void *x, *y;
    void *z;
    uint64_t C, M, N, L, S;          // x: C x M x S, y: C x N x S, z: C x L x S
    int64_t off;                     // index of output 0 in the full convolution
    uint64_t xcs, xts, xss;          // element strides of x: column, time, slice (0 = broadcast)
    uint64_t ycs, yts, yss;
};
hipError_t launch_conv(const ConvParams &P, int dtype, int cplx, hipStream_t s);

// ---- layout.hip: out[c][b][a] = in[a][b][c]
hipError_t launch_permute3(const void *in, void *out, uint64_t A, uint64_t B, uint64_t C, int elem_bytes, hipStream_t s);

}  // namespace qdas
