// das_tile_luth.hip -- instantiations of the tiled kernel for launch configuration 11 (fp16 data, delays from host-supplied tables);
// one translation unit per configuration (make -j).
#include "das_tile_impl.h"

namespace qdas {

hipError_t launch_tile_luth(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    switch (P.flag & 7) {
        case 0: return launch_tile_i<0, uint32_t, 11>(P, ntiles, lds, s);
        case 1: case 4: return launch_tile_i<1, uint32_t, 11>(P, ntiles, lds, s);
        case 2: return launch_tile_i<2, uint32_t, 11>(P, ntiles, lds, s);
        case 3: return launch_tile_i<3, uint32_t, 11>(P, ntiles, lds, s);
        case 5: return launch_tile_i<5, uint32_t, 11>(P, ntiles, lds, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
