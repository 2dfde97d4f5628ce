// das_tile_symq.hip -- instantiations of the tiled kernel for launch configuration 15 (reciprocal + lateral-mirror mode, fp32 data: four window
// sets per stage); one translation unit per configuration so that they compile in parallel (make -j).
#include "das_tile_impl.h"

namespace qdas {

hipError_t launch_tile_symq(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    switch (P.flag & 7) {
        case 0: return launch_tile_i<0, float2, 15>(P, ntiles, lds, s);
        case 1: case 4: return launch_tile_i<1, float2, 15>(P, ntiles, lds, s);
        case 2: return launch_tile_i<2, float2, 15>(P, ntiles, lds, s);
        case 3: return launch_tile_i<3, float2, 15>(P, ntiles, lds, s);
        case 5: return launch_tile_i<5, float2, 15>(P, ntiles, lds, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
