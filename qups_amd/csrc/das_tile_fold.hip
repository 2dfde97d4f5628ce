// das_tile_fold.hip -- instantiations of the tiled kernel for launch configurations 17 / 18 / 19 (reciprocity-folded fp32 data, TileCfg::FOLD:
// with the lateral-mirror mode and 128- or 192-sample windows, without it); one translation unit per family so that they compile in parallel (make -j).
#include "das_tile_impl.h"

namespace qdas {

template <int CI> static hipError_t launch_fold_ci(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    switch (P.flag & 7) {
        case 0: return launch_tile_i<0, float2, CI>(P, ntiles, lds, s);
        case 1: case 4: return launch_tile_i<1, float2, CI>(P, ntiles, lds, s);
        case 2: return launch_tile_i<2, float2, CI>(P, ntiles, lds, s);
        case 3: return launch_tile_i<3, float2, CI>(P, ntiles, lds, s);
        case 5: return launch_tile_i<5, float2, CI>(P, ntiles, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_tile_fold(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    if (P.probe) return P.narrow ? launch_fold_ci<17>(P, ntiles, lds, s) : launch_fold_ci<19>(P, ntiles, lds, s);     // (window-fit probes: 128- / 192-sample windows)
    if (P.nfr == 2 && !P.probe) return P.mir ? (P.narrow ? launch_fold_ci<20>(P, ntiles, lds, s) : hipErrorInvalidValue) : launch_fold_ci<21>(P, ntiles, lds, s);
    if (P.nfr > 2) return hipErrorInvalidValue;
    if (!P.mir) return launch_fold_ci<19>(P, ntiles, lds, s);
    return P.narrow ? launch_fold_ci<17>(P, ntiles, lds, s) : launch_fold_ci<18>(P, ntiles, lds, s);
}

}  // namespace qdas
