#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/sweep_queue.sh <tag> ["lanes list"] ["slices list"] [bench.py arguments...]
# The headline step against submissions in flight (FELICS_LANES) x slices per queued submission (FELICS_SLICES_QUEUED).
set -eo pipefail
tag=$1; lanes=${2:-"2 3 4"}; slices=${3:-"2 3 4 6"}; shift; shift || true; shift || true
O=gpurun_out/$tag; mkdir -p "$O"
for l in $lanes; do for s in $slices; do
  FELICS_LANES=$l FELICS_SLICES_QUEUED=$s python3 bench.py --steps 20 --warmup 3 --no-side-configs --no-decode-leg --no-blocking-extra --cpu-seconds 0 "$@" > "$O/l${l}_s${s}.json" 2> "$O/err.txt" || { tail -3 "$O/err.txt"; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('$O/l${l}_s${s}.json')); print('lanes $l slices $s: %.3f ms/step' % d['ms_per_step'], {k: round(v,2) for k,v in d['pipeline']['stage_ms_sum_of_launches'].items() if v>0})"
done; done | tee "$O/sweep.txt"
