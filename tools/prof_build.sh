#!/bin/bash
# Profiling build for tools/phase_timers.py: ONE translation unit of the tiled kernel compiled with -DQDAS_PROF=1 (in-kernel phase
# timers, tile_hooks.h), linked with the product's other objects into tools/abl/libqdas_prof.so (scratch, git-ignored).
#   tools/prof_build.sh das_tile_f16      (then: python tools/phase_timers.py c5)
#   tools/prof_build.sh das_tile_sym      (C3, prebuilt reciprocal kernel)
# Run `make -C qups_amd/csrc` first; plans must use the prebuilt kernels (no QDAS_PLAN_JIT).
set -e
TU=${1:-das_tile_f32}
cd "$(dirname "$0")/../qups_amd/csrc"
mkdir -p ../../tools/abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DQDAS_PROF=1 $EXTRA -c $TU.hip -o ../../tools/abl/${TU}_prof.o
OBJS=$(ls *.o | grep -v "^$TU.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/libqdas_prof.so $OBJS ../../tools/abl/${TU}_prof.o -L/opt/rocm/lib -lhipfft -ldl -Wl,-rpath,/opt/rocm/lib
echo "tools/abl/libqdas_prof.so: $TU instrumented"
