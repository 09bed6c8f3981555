#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/collect.sh r01_final
# 1. bench.py line (with cpu_baseline)            -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 0 --no-blocking-extra` (13 steps, 12 of them
#    through the submission queue like the timed ones) -> gpurun_out/<tag>/kernel_stats.csv (felics:: kernels only)
# 3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace domains) of
#    `bench.py --steps 1 --warmup 0 --synchronous` -> gpurun_out/<tag>/traffic.json (profiles/tools/summarize.py)
# Copy what should be judged from gpurun_out/<tag>/ into profiles/<round>/ afterwards.
set -eo pipefail
tag=${1:-run}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p "$O"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --steps 10 --warmup 0 --cpu-seconds 0 --no-blocking-extra > "$O/kt.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 "$R/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --synchronous > "$O/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 "$R/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --synchronous > "$O/write.log" 2>&1
cd "$R"
python3 profiles/tools/summarize.py "$O"
rm -rf "$O/kt" "$O/fetch" "$O/write"
ls -la "$O"
