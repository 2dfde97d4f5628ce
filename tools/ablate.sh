#!/bin/bash
# Build ablation / profiling variants of libqdas.so into tools/abl/ (scratch, git-ignored).  The tiled kernel is compiled as ONE
# translation unit (-DQDAS_UNITY) with the given QDAS_ABL bit mask (see das_tile_impl.h); extra flags via EXTRA, e.g.
#   tools/ablate.sh 1 4 8            -> tools/abl/libqdas_abl{1,4,8}.so
#   EXTRA=-DQDAS_PROF=1 tools/ablate.sh 0 && mv tools/abl/libqdas_abl0.so tools/abl/libqdas_prof.so    (tools/phase_timers.py)
# then  QDAS_LIB=tools/abl/libqdas_ablN.so python bench.py --no-cpu
set -e
cd "$(dirname "$0")/../qups_amd/csrc"
mkdir -p ../../tools/abl
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DQDAS_UNITY -DQDAS_ABL=$a $EXTRA -c das_tile.hip -o ../../tools/abl/das_tile_$a.o &
done
wait
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/libqdas_abl$a.so qdas_api.o das_generic.o das_lut.o wsinterpd.o greens.o pre.o conv.o layout.o jit.o sharded.o ../../tools/abl/das_tile_$a.o -L/opt/rocm/lib -lhipfft -ldl -Wl,-rpath,/opt/rocm/lib
done
