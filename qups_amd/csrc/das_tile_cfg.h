// das_tile_cfg.h -- launch configurations of the tiled kernel, shared by its translation units (internal).
#pragma once
// elements per pass of the prologue's tile-wide reductions (LDS scratch = 2 * waves * min(max(N, M), chunk) floats, aliasing the windows)
#define QDAS_PROLOGUE_CHUNK 512u
namespace qdas {
// ------------------------------------------------------------------------------------------
// Launch configurations.  cfg 0: 16-wave workgroup = 64 x 16 pixel tile, 32 transmits per stage, 2 window buffers, one
// workgroup per CU (general case);  cfg 1: the same tile with 16 transmits per stage and direct + mirror windows
// (reciprocal mode).
struct Cfg { int waves, mb, w, nbuf, psz, bpc; };
// cfg 3 / 4: two frames per launch (fp32 / fp16 data): 16 transmits x 2 frames per stage -- the reciprocal mode's window layout;
// cfg 5 / 6: four frames per launch: 8 transmits x 4 frames per stage
// cfg 7: reciprocal mode with 128-sample windows (a third less staging traffic; used when every tile of some footprint fits them)
// cfg 8: reciprocal mode for fp16 data (256-sample windows = one 1024-byte DMA piece, like cfg 7)
// cfg 9: cfg 0 with descriptor re-basing (transposed fp32 frames beyond 2 GiB)
// cfg 10: cfg 0 with table-driven delays (the split-delay flavour: qdas_das_lut)
// cfg 11: cfg 2 (fp16 data) with table-driven delays
// cfg 12: cfg 0 keeping BOTH aperture dimensions ('BF': one output plane per (receiver, transmit) pair, nothing is summed)
// cfg 13: fp64 data (16-byte samples): 16 transmits per stage, 192-sample windows -- the LDS image of cfg 0
// cfg 14: fp32 data with 384-sample windows and 16 transmits per stage (the LDS image of cfg 0): the second attempt of a plan whose
//         tiles do not fit 192 samples -- pixel grids coarser than about lambda/2 (volumes, previews), steep delay gradients
// cfg 15 / 16: reciprocal + lateral-mirror mode (fp32 / fp16 data): FOUR window sets of 16 one-KiB windows per buffer -- the LDS image of the
//         32-transmit reciprocal stages hiprtc builds run
// cfg 17 / 18 / 19: reciprocity-FOLDED fp32 data (TileCfg::FOLD, fold.hip) -- with the lateral-mirror mode: TWO window sets of 32 one-KiB windows
//         (cfg 17; the LDS image of cfg 15) or of 16 192-sample windows (cfg 18: tiles that do not fit 128 samples); without it: ONE set of 32
//         192-sample windows (cfg 19: the LDS image of cfg 0)
// cfg 20 / 21: two FRAMES of folded data per launch -- mirror mode: four window sets of 16 one-KiB windows (the LDS image of cfg 15); without: two sets of 16 x 192
static constexpr Cfg CFGS[22] = {{16, 32, 192, 2, 16, 1}, {16, 16, 192, 2, 16, 1}, {16, 32, 384, 2, 16, 1},
                                {16, 16, 192, 2, 16, 1}, {16, 16, 384, 2, 16, 1}, {16, 8, 192, 2, 16, 1}, {16, 8, 384, 2, 16, 1},
                                {16, 16, 128, 2, 16, 1}, {16, 16, 256, 2, 16, 1}, {16, 32, 192, 2, 16, 1}, {16, 32, 192, 2, 16, 1}, {16, 32, 384, 2, 16, 1}, {16, 32, 192, 2, 16, 1},
                                {16, 16, 192, 2, 16, 1}, {16, 16, 384, 2, 16, 1}, {16, 16, 128, 2, 16, 1}, {16, 16, 256, 2, 16, 1},
                                {16, 32, 128, 2, 16, 1}, {16, 16, 192, 2, 16, 1}, {16, 32, 192, 2, 16, 1}, {16, 16, 128, 2, 16, 1}, {16, 16, 192, 2, 16, 1}};
// fb: frames per launch (1 | 2 | 4)
// narrow: window variant -- 1: reciprocal mode with 128-sample windows (cfg 7); 2: general mode, fp32 data, 384-sample windows (cfg 14)
// mirq: reciprocal + lateral-mirror mode (cfg 15 / 16)
// fold: reciprocity-folded fp32 data (cfg 17 / 18 with mirq: narrow / 192-sample windows; cfg 19 without)
static inline int cfg_index(int dtype, int sym, int fb = 1, int narrow = 0, int mirq = 0, int fold = 0) { return (fold && sym && dtype == 1) ? (fb == 2 ? (mirq ? 20 : 21) : mirq ? (narrow ? 17 : 18) : 19) : (mirq && sym && dtype != 0) ? (dtype == 2 ? 16 : 15) : dtype == 0 ? 13 : (!sym && narrow == 2 && dtype == 1 && fb == 1) ? 14 : sym ? (dtype == 2 ? 8 : (narrow ? 7 : 1)) : (fb == 4 ? (dtype == 2 ? 6 : 5) : (fb == 2 ? (dtype == 2 ? 4 : 3) : (dtype == 2 ? 2 : 0))); }
// ------------------------------------------------------------------------------------------
// Which points of the template's matrix libqdas.so carries.  Everything else is built on demand by hiprtc from the same template arguments (jit.hip
// lazy_tile_launch; ~2 s once per variant and machine, cached on disk) -- the reference builds ALL its kernels per system that way
// (src/UltrasoundSystem.m:5527-5625).  The prebuilt set is what the BASELINE configurations, the benches and the test suite launch
// (QDAS_KERNEL_CENSUS, tools/kernel_census.py: the census of a GPU run of all of them), so that none of those ever waits for a compiler.
// Probe instantiations (the plan-time window-fit test: the prologue only, ~8 KB each) are all prebuilt.
// Row = launch configuration, column = interpolator flag (0 nearest, 1 linear, 2 cubic, 3 lanczos3, 5 cubic_dev), value = mask over the variants
// 1: plain, 2: remodulation, 4: weight table, 8: both.  The set: what `bench.py` launches for C1 ... C5 and its switches (general / reciprocal / mirror,
// frame streams, fp16, fp64, windows; prebuilt and hiprtc-specialised), what __graft_entry__.smoke() launches, plus the plain variant of every
// interpolator on the general fp32 / fp16 configurations and of `cubic` (the reference's default, src/UltrasoundSystem.m:3289) on the folded ones.
// `tools/warm_cache.py` (python -m qups_amd.warm) builds any other set ahead of time, in parallel; tests/conftest.py does so for the GPU suite.
static constexpr unsigned char TILE_PREBUILT[22][6] = {
    {0x1, 0x1, 0x5, 0x7, 0x0, 0x0},   // cfg 0   fp32, general
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 1   (unfolded fp32 reciprocal: hiprtc-specialised builds only)
    {0x1, 0x1, 0x1, 0x1, 0x0, 0x0},   // cfg 2   fp16, general
    {0x0, 0x1, 0x5, 0x7, 0x0, 0x0},   // cfg 3   fp32, two frames per launch / lateral-mirror mode
    {0x0, 0x0, 0x1, 0x1, 0x0, 0x0},   // cfg 4   fp16, two frames per launch / lateral-mirror mode
    {0x0, 0x1, 0x1, 0x1, 0x0, 0x0},   // cfg 5   fp32, four frames per launch
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 6   fp16, four frames per launch
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 7   (as cfg 1)
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 8   fp16 reciprocal, unfolded (QDAS_PLAN_NO_FOLD)
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 9   fp32, transposed frames beyond 2 GiB
    {0x0, 0x1, 0x1, 0x1, 0x0, 0x0},   // cfg 10  fp32, table-driven delays (bfDASLUT)
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 11  fp16, table-driven delays
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 12  'BF'
    {0x0, 0x0, 0x3, 0x0, 0x0, 0x0},   // cfg 13  fp64
    {0x0, 0x1, 0x0, 0x0, 0x0, 0x0},   // cfg 14  fp32, 384-sample windows
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 15  (as cfg 1)
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 16  fp16 reciprocal + mirror, unfolded
    {0x0, 0x0, 0x1, 0x3, 0x0, 0x0},   // cfg 17  folded, mirror, 128-sample windows (the headline: C3)
    {0x0, 0x0, 0x0, 0x0, 0x0, 0x0},   // cfg 18  folded, mirror, 192-sample windows
    {0x0, 0x0, 0x1, 0x3, 0x0, 0x0},   // cfg 19  folded
    {0x0, 0x0, 0x1, 0x1, 0x0, 0x0},   // cfg 20  folded, mirror, two frames per launch
    {0x0, 0x0, 0x1, 0x1, 0x0, 0x0},   // cfg 21  folded, two frames per launch
};
constexpr bool tile_prebuilt(int ci, int interp, bool fm, bool wt, bool probe) {
    if (probe) return true;
    return ((TILE_PREBUILT[ci][interp] >> ((fm ? 1 : 0) + (wt ? 2 : 0))) & 1) != 0;
}
}  // namespace qdas
