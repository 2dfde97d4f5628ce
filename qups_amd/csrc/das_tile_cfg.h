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
}  // namespace qdas
