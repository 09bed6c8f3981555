#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/sweep_slices_one_frame.sh
# One 4K gray / RGB frame per blocking call by slices per call (FELICS_SLICES).
for s in 2 3 4 6 8 12; do
for c in 2 4; do
  FELICS_SLICES=$s timeout -k 10 100 python3 bench.py --steps 30 --warmup 3 --config $c --synchronous --no-decode-leg --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('FELICS_SLICES=$s config $c one frame blocking: %.3f ms' % d['ms_per_step'])"
done; done
