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
KERNELS="felics_kernels felics_wide felics_gpudecode"
[ -f felics_chain.hip ] && KERNELS="$KERNELS felics_chain"   # (round 5 on; an older checkout has no such file)
OBJS=""
for f in $KERNELS; do hipcc $F -c $f.hip -o "$O/$f.o" & OBJS="$OBJS $O/$f.o"; done
hipcc $F -x hip -c felics_api.cpp -o "$O/felics_api.o" &
hipcc -O3 -std=c++17 -fPIC -c felics_decode.cpp -o "$O/felics_decode.o" &
wait
hipcc --offload-arch=gfx950 -shared -o "$O/libfelics.so" $OBJS "$O"/felics_api.o "$O"/felics_decode.o -Wl,-rpath,/opt/rocm/lib
rm -f "$O"/*.o
ls -la "$O/libfelics.so"
