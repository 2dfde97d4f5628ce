#!/bin/bash
# Build ablation variants of libqdas.so (QDAS_ABL bit mask, see das_tile.hip) into tools/abl/ (scratch, git-ignored)
# usage: tools/ablate.sh 1 2 4 8 16 ...   then   QDAS_LIB=tools/abl/libqdas_ablN.so python bench.py --no-cpu
set -e
cd "$(dirname "$0")/../qups_amd/csrc"
mkdir -p ../../tools/abl
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DQDAS_ABL=$a -c das_tile.hip -o ../../tools/abl/das_tile_$a.o &
done
wait
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/libqdas_abl$a.so qdas_api.o das_generic.o das_lut.o greens.o ../../tools/abl/das_tile_$a.o
done
