// das_tile_cfg.h -- launch configurations of the tiled kernel, shared by its translation units (internal).
#pragma once
namespace qdas {
// ------------------------------------------------------------------------------------------
// Launch configurations.  cfg 0: 16-wave workgroup = 64 x 16 pixel tile, 32 transmits per stage, 2 window buffers, one
// workgroup per CU (general case);  cfg 1: the same tile with 16 transmits per stage and direct + mirror windows
// (reciprocal mode).
struct Cfg { int waves, mb, w, nbuf, psz, bpc; };
static constexpr Cfg CFGS[3] = {{16, 32, 192, 2, 16, 1}, {16, 16, 192, 2, 16, 1}, {16, 32, 384, 2, 16, 1}};
static inline int cfg_index(int dtype, int sym) { return sym ? 1 : (dtype == 2 ? 2 : 0); }
}  // namespace qdas
