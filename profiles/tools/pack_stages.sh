#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/pack_stages.sh <tag>
# Instruction counts of k_pack_g's stages: four diagnostic builds of the library that leave the kernel after stage 1 .. 4
#   make -C felics_amd/csrc lib OUT=../../scratch/pstop$n CXXFLAGS="-O3 -std=c++17 -fPIC -DFELICS_PACK_STOP=$n"    (n = 1 .. 4)
# and the product build, each under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD (counters only), two
# blocking 64-frame steps without output checks (the early-exit builds produce no streams).
set -eo pipefail
tag=$1
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p "$O"
export TMPDIR=/tmp FELICS_SERIAL=1 FELICS_SLICES=1
cd /tmp
for n in 1 2 3 4 full; do
  lib=$R/scratch/pstop$n/libfelics.so; [ $n = full ] && lib=$R/felics_amd/_build/libfelics.so
  rm -rf "$O/ps_$n"
  FELICS_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d "$O/ps_$n" -- python3 "$R/scratch/r3/run_only.py" > "$O/ps_$n.log" 2>&1 || { tail -5 "$O/ps_$n.log"; exit 1; }
  echo "stop after stage $n: $(python3 "$R/profiles/tools/pmc_agg.py" "$O/ps_$n" 2 | grep k_pack_g)"
  rm -rf "$O/ps_$n"
done
