#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/pmc_serial.sh <tag> [pass ...]
# Per-kernel hardware counters with every kernel running ALONE (FELICS_SERIAL=1: one stream; FELICS_SLICES=1: one launch
# per stage), two blocking steps of the headline batch per pass.  Counters only (--pmc), one pass per counter group.
# -> gpurun_out/<tag>/pmc_<pass>.txt (per kernel: counter sums per step, in millions)
set -eo pipefail
tag=$1
shift
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p "$O"
export TMPDIR=/tmp FELICS_SERIAL=1 FELICS_SLICES=1
declare -A G
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
G[sq2]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G[sq3]="SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT"
G[ta]="TA_TA_BUSY TA_TOTAL_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES"
G[tcp2]="TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_ATOMIC_WITH_RET_REQ TCP_TCP_TA_DATA_STALL_CYCLES"
G[tcc1]="TCC_REQ TCC_HIT TCC_MISS TCC_ATOMIC"
G[tcc2]="TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_TAG_STALL"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
passes=${@:-sq1 sq2 ta tcp tcc1 tcc2}
cd /tmp
for p in $passes; do
  echo "[pmc] $p: ${G[$p]}"
  rm -rf "$O/pmc_$p"
  timeout -k 10 300 rocprofv3 --pmc ${G[$p]} --output-format csv -d "$O/pmc_$p" -- python3 "$R/bench.py" --steps 1 --warmup 0 --synchronous --no-blocking-extra --no-side-configs --no-decode-leg --cpu-seconds 0 --check-frames 1 $PMC_BENCH_ARGS > "$O/pmc_$p.log" 2>&1 || { tail -5 "$O/pmc_$p.log"; exit 1; }
  python3 "$R/profiles/tools/pmc_agg.py" "$O/pmc_$p" 2 > "$O/pmc_$p.txt"
  rm -rf "$O/pmc_$p"
  cat "$O/pmc_$p.txt"
done
