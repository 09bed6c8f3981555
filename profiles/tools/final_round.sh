#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/final_round.sh <tag>
# Everything profiles/<round>/ holds about the build in the tree, in one call (then copy from gpurun_out/<tag>/):
#   collect.sh          -> bench.json, kernel_stats.csv (queued), traffic.json (four counter passes)
#   kstats.sh x 2       -> every kernel alone (FELICS_SERIAL=1 FELICS_SLICES=1), S1 and S2 frames
#   content_ab.py       -> S1 / noise / flat / natural-like, queued and blocking
#   timeline.sh         -> start / end of every kernel of queued steps
#   source hash         -> the build all of the above belong to
set -eo pipefail
tag=${1:-final}
O=gpurun_out/$tag; mkdir -p "$O"
python3 -c "from felics_amd import build; print(build.source_hash())" > "$O/source_sha256.txt"
echo "[final] collect"; bash profiles/tools/collect.sh "$tag" > "$O/collect.log" 2>&1 || { tail -5 "$O/collect.log"; exit 1; }
echo "[final] kernels alone, S1"; FELICS_SERIAL=1 FELICS_SLICES=1 bash profiles/tools/kstats.sh "$tag/s1_alone" --steps 10 --warmup 0 --synchronous --cpu-seconds 0 --no-blocking-extra --no-side-configs --no-decode-leg > "$O/s1_alone.log" 2>&1 || { tail -5 "$O/s1_alone.log"; exit 1; }
echo "[final] kernels alone, S2"; FELICS_SERIAL=1 FELICS_SLICES=1 bash profiles/tools/kstats.sh "$tag/s2_alone" --kind S2 --steps 10 --warmup 0 --synchronous --cpu-seconds 0 --no-blocking-extra --no-side-configs --no-decode-leg > "$O/s2_alone.log" 2>&1 || { tail -5 "$O/s2_alone.log"; exit 1; }
echo "[final] kernels in the queue, S2"; bash profiles/tools/kstats.sh "$tag/s2_queue" --kind S2 --steps 10 --warmup 0 --cpu-seconds 0 --no-blocking-extra --no-side-configs --no-decode-leg > "$O/s2_queue.log" 2>&1 || { tail -5 "$O/s2_queue.log"; exit 1; }
echo "[final] content"; timeout -k 10 300 python3 profiles/tools/content_ab.py > "$O/content_sensitivity.txt" 2> "$O/content.err" || { tail -5 "$O/content.err"; exit 1; }
echo "[final] timeline"; timeout -k 10 200 bash profiles/tools/timeline.sh "$tag/tl" --steps 8 --warmup 2 --no-side-configs --no-decode-leg --cpu-seconds 0 --no-blocking-extra > /dev/null 2>&1 || true
for f in s1_alone s2_alone s2_queue; do sed -i "1i # source_sha256 $(cat $O/source_sha256.txt)" "$O/$f/kernel_stats.csv"; done
cat "$O/content_sensitivity.txt"; cat "$O/s1_alone/kernel_stats.csv"
