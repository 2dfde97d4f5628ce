// das_tile_f64.hip -- instantiations of the tiled kernel for launch configuration 13 (fp64 data: geometry, delays, weights
// and sums in double; das_tile_impl.h "F64"); one translation unit per configuration so that they compile in parallel (make -j).
#include "das_tile_impl.h"

namespace qdas {

hipError_t launch_tile_f64(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    switch (P.flag & 7) {
        case 0: return launch_tile_i<0, double2, 13>(P, ntiles, lds, s);
        case 1: case 4: return launch_tile_i<1, double2, 13>(P, ntiles, lds, s);
        case 2: return launch_tile_i<2, double2, 13>(P, ntiles, lds, s);
        case 3: return launch_tile_i<3, double2, 13>(P, ntiles, lds, s);
        case 5: return launch_tile_i<5, double2, 13>(P, ntiles, lds, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace qdas
