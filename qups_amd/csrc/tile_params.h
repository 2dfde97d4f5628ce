// tile_params.h -- parameter block of the tiled kernel (internal).  Plain data only: this header is also compiled by hiprtc
// (plan-specialised kernels, jit.hip), which has no host runtime types.
#pragma once
#include "tile_rtc.h"

namespace qdas {

struct TileParams {
    // fp32 geometry (QDAS_F32 / QDAS_F16 only).  Pv (4 x M: position, t0) and Nv (3 x M) describe the BLOCK elements of a stage,
    // Pr (3 x N) and -- when kindS != 0 -- St (4 x N: t0, normal) the STAGE elements; kindB / kindS: 0 distance, 1 signed distance,
    // 2 plane wave.  'DAS' / 'SYN': block = transmits, stage = receivers; 'MUL': roles (and N, M, strN, strM, wtab) swapped by the host.
    const float *Pi, *Pr, *Pv, *Nv, *St;
    int32_t kindB, kindS;
    // kindS == 3: ONE-SIDED signed distance.  A focused transmit's delay flips sign at the plane through its focus ((Pi - Pv).Nv = 0,
    // src/bf.cu:106-108): discontinuous, so a tile that the plane crosses has no window that fits.  The host then lists every transmit twice
    // -- stage element 2m: the pixels BEFORE the plane (delay -|r|), 2m + 1: those behind it (+|r|) -- with gen_kind 5 (1 on the element's
    // side, else 0) as its pixel weight: each side is smooth, lanes of the other side stay out of the window bases, and the tile's stage
    // list drops the side it does not have.  stage_shift = 1: stage element e reads the traces of transmit e >> 1.
    int32_t stage_shift;
    const void *x;
    void *y;
    const void *wtab;                   // optional N x M table of folded apodization weights (float2), may be null
    const void *apix;                   // optional I x N apodization (data precision; real if apix_real), may be null
    int32_t apix_real;
    int32_t apix_pixel_only;            // the array is I1 x I2 x I3 only (a spatial weight / region-of-interest mask): the same entry for every stage element
    int32_t gen_kind;                   // generated pixel x receiver apodization (QDAS_RXAPOD_*), exclusive with apix; 5: side of the focal plane (kindS == 3);
                                        // 6: that side rule times apix[pixel + I * (n >> 1)] (pixel x transmit weights of a two-sided focused plan)
    double gen_p0, gen_p1;
    const float *rxn;                   // 3 x N element normals (device)
    uint32_t act_bytes;                 // LDS bytes of the tile's stage list: 8 * (N + 1) when a pixel x receiver weight (apix / gen_kind) is set, else 0
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
    int32_t bf;                         // 1: 'BF' -- keep both aperture dimensions: plane (n*bf_pn + m*bf_pm) of y per pair (launch configuration 12)
    uint64_t bf_pn, bf_pm;              // plane strides of the stage / block element: (1, N) or -- transposed data -- (M, 1): the DATA's aperture order (src/bf.cu:100,135)
    int32_t nfr;                        // frames per launch: 1, or 2 / 4 (frame f at x + f*x_fstride -> y + f*y_fstride); 1 with sym
    uint64_t x_fstride, y_fstride;      // frame strides: BYTES of x, ELEMENTS of y
    uint32_t ksplit;                    // workgroups per tile (>= 1): each sums a slice of the aperture into part[], then reduced into y
    float2 *part;                       // [ksplit][nfr][i_count] partial images (ksplit > 1 only)
    uint32_t *fallback_list;            // [0] = count, [1..] = tile ids that did not fit the LDS window
    uint32_t fallback_cap;
    // table-driven delays (launch configuration 10, qdas_das_lut): tau_tx (I x M) and tau_rx (I x N) in samples, fp32
    const float *lut_tx, *lut_rx;
    // a SECOND pixel-dependent weight, per (pixel, BLOCK element): real fp32, [pixel + I * block element]; applied per pair (launch configuration 14
    // without table / remodulation only: TileCfg::BPIX) -- the receive-side mask of a focused plan whose stage weight is the transmit-side rule
    const float *bpix;
    // LATERAL-MIRROR mode (two-window-set instantiations, TileCfg::FBX with nfr == 1): array, sequence and scan are mirror-symmetric about
    // the plane x = 0 -- Pr[N-1-n] = mirror(Pr[n]), {Pv, Nv, t0}[M-1-m] = mirror({Pv, Nv, t0}[m]), pixel column I2-1-c = mirror(column c),
    // bit for bit (checked by the host) -- so tau(pixel', N-1-n, M-1-m) == tau(pixel, n, m): the tile grid covers the columns c < (I2+1)/2
    // only, the second window set of a stage holds the traces x[:, N-1-n, M-1-m], and tap index + interpolation weights -- half of the
    // pair loop's instructions -- are computed once for a pixel and its mirror image.  Every product of both sums is still formed.
    // mir == 2 (QDAS_PLAN_MIRROR_SLAB): the plan's pixels [i_begin, i_begin + i_count) are whole columns of the first half; the mirror images go
    // to y[i_count + (pixel' - (I - i_begin - i_count))] -- slab B behind slab A, natural pixel order.
    int32_t mir;
    // RECIPROCITY-FOLDED data (das_tile_impl.h TileCfg::FOLD; reciprocal fp32 plans): x points at the plan's folded copy of the frame
    // (fold.hip: xs[:,n,m] = w[n,m] x[:,n,m] + w[m,n] x[:,m,n], n <= m) and the stage loop walks the upper triangle only; wtab is null
    int32_t fold;
    // PROLOGUE TABLES (round 5).  The window bases A[m], B[n], their extents and the tile-wide window statistics depend on the GEOMETRY only -- not on the
    // frame --, yet every workgroup of every execute recomputed them: (M + N) wave-wide min / max reductions per wave, a quarter of BASELINE C5's kernel time
    // (8 workgroups per tile, each with the whole prologue), a tenth of C2's.  The plan computes them ONCE (a probe launch with pro_out set, after the tile
    // shape is final) and the stage kernels load them: per tile 2 (M + N) + 8 floats -- {base (int bits), extent} per block / stage element, then
    // {a_lo, b_lo, a_hi, b_hi, a_ext, b_ext}.  Tile slot = tz + tiles_z * (column tile - tile_x0).  Null: computed in the kernel, as before.
    const float *pro_tab;
    float *pro_out;
    // ... and, for plans with a pixel x stage-element weight, the tile's ACTIVITY MASK behind the statistics: ceil(N / 32) words, bit n = stage element n carries
    // weight for some pixel of the tile (or, lateral-mirror mode, for the mirror image of one) -- what plan_stages() found by loading every candidate's weights in
    // every workgroup of every execute.  pro_mask = 1: the table holds it (stride 2 (M + N) + 8 + ceil(N / 32) floats).
    int32_t pro_mask;
    // TWO-DIMENSIONAL SPLIT of the aperture (plans with a stage list): the ksplit workgroups of a tile are ksplit_m groups over the TRANSMIT BLOCKS (group g takes
    // blocks g, g + ksplit_m, ...) times ksplit / ksplit_m interleaved classes of stage elements.  A workgroup that owns few transmit blocks refreshes the block
    // residuals (one fp64 delay per pixel and transmit) rarely -- BASELINE C5 with 8 receiver classes x all 6 blocks: a refresh every 3.4 stages, 9-15 % of the
    // kernel; one block per workgroup: once.  0 / 1: the one-dimensional split.
    uint32_t ksplit_m;
    // plan-time probes only: the window length [samples] the fit test assumes instead of the probe kernel's own (0: its own) -- lets the plan ask "would these
    // tiles fit the 128-sample windows of a plan-specialised build?" with the prebuilt 192-sample probe kernels
    int32_t probe_w;
};

}  // namespace qdas
