#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/timeline.sh <tag> <bench.py arguments...>
# start / end of every felics:: kernel of the last steps of `python3 bench.py <arguments>` -> gpurun_out/<tag>/timeline.txt
set -eo pipefail
tag=$1
shift
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 "$R/bench.py" "$@" > "$O/kt.log" 2>&1 || { tail -20 "$O/kt.log"; exit 1; }
cd "$R"
python3 profiles/tools/timeline.py "$O/kt" 3 > "$O/timeline.txt"
rm -rf "$O/kt"
head -120 "$O/timeline.txt"
