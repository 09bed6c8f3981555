#!/bin/bash
# Builds the tree's native sources -- or another checkout's -- into felics_amd/_variants/<name>/libfelics.so, for same-box A/B
# runs (profiles/tools/ab.py).  Runs in the build container (hipcc cross-compiles); the variants travel to the GPU box with the
# snapshot and stay out of git (felics_amd/_variants/ is ignored).
#   profiles/tools/variant.sh <name> [-DFOO ...]            the working tree, extra compiler flags
#   SRC=/tmp/base profiles/tools/variant.sh base            another checkout (e.g. `git worktree add /tmp/base HEAD`)
set -eo pipefail
name=$1
shift
R=$(cd "$(dirname "$0")/../.." && pwd)
SRC=${SRC:-$R}
O=$R/felics_amd/_variants/$name
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter $*"
cd "$SRC/felics_amd/csrc"
for f in felics_kernels felics_wide felics_gpudecode; do hipcc $F -c $f.hip -o "$O/$f.o" & done
hipcc $F -x hip -c felics_api.cpp -o "$O/felics_api.o" &
hipcc -O3 -std=c++17 -fPIC -c felics_decode.cpp -o "$O/felics_decode.o" &
wait
hipcc --offload-arch=gfx950 -shared -o "$O/libfelics.so" "$O"/felics_kernels.o "$O"/felics_wide.o "$O"/felics_gpudecode.o "$O"/felics_api.o "$O"/felics_decode.o -Wl,-rpath,/opt/rocm/lib
rm -f "$O"/*.o
ls -la "$O/libfelics.so"
