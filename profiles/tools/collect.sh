#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/collect.sh <tag>
# 1. the default bench line (cpu_baseline, side configs, decode leg)      -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 0 --no-blocking-extra` (timed steps through the
#    submission queue, as in 1.)                                           -> gpurun_out/<tag>/kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY ..., --pmc SQ_INSTS_LDS
#    SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT ... (four separate passes, counters only: no trace domains) of ONE checked + ONE timed
#    blocking step
#                                                                          -> gpurun_out/<tag>/traffic.json
# Copy what should be judged from gpurun_out/<tag>/ into profiles/<round>/ (and traffic.json to profiles/) afterwards.
set -eo pipefail
tag=${1:-run}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p "$O"
export TMPDIR=/tmp
echo "[collect] bench line"
timeout -k 10 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" || { tail -5 "$O/bench.err"; exit 1; }
cd /tmp
# (The counter passes run blocking steps, the headline goes through the submission queue: the same kernels on the same data.)
PMC_ARGS="--steps 1 --warmup 0 --synchronous --no-blocking-extra --no-side-configs --no-decode-leg --cpu-seconds 0"
echo "[collect] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --steps 10 --warmup 0 --cpu-seconds 0 --no-blocking-extra --no-side-configs --no-decode-leg > "$O/kt.log" 2>&1 || { tail -5 "$O/kt.log"; exit 1; }
echo "[collect] FETCH_SIZE"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 "$R/bench.py" $PMC_ARGS > "$O/fetch.log" 2>&1 || { tail -5 "$O/fetch.log"; exit 1; }
echo "[collect] WRITE_SIZE"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 "$R/bench.py" $PMC_ARGS > "$O/write.log" 2>&1 || { tail -5 "$O/write.log"; exit 1; }
echo "[collect] SQ counters"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -- python3 "$R/bench.py" $PMC_ARGS > "$O/sq.log" 2>&1 || { tail -5 "$O/sq.log"; exit 1; }
echo "[collect] SQ counters, second pass (LDS, memory instructions)"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d "$O/sq2" -- python3 "$R/bench.py" $PMC_ARGS > "$O/sq2.log" 2>&1 || { tail -5 "$O/sq2.log"; exit 1; }
cd "$R"
python3 profiles/tools/summarize.py "$O"
rm -rf "$O/kt" "$O/fetch" "$O/write" "$O/sq" "$O/sq2"
ls -la "$O"
