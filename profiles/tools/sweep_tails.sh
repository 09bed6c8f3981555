# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/sweep_tails.sh
# The headline step with the lanes sharing one tail stream (default) against a tail stream per lane (FELICS_OWN_TAILS=1: tiles by ticket),
# by slices per queued submission.
set -eo pipefail
O=gpurun_out/r5/tails; mkdir -p $O
for t in 0 1; do for s in 2 3 4; do
  if [ $t = 0 ]; then unset FELICS_OWN_TAILS; else export FELICS_OWN_TAILS=$t; fi
  FELICS_SLICES_QUEUED=$s timeout -k 10 100 python3 bench.py --steps 20 --warmup 3 --no-side-configs --no-decode-leg --no-blocking-extra --cpu-seconds 0 > $O/t${t}_s$s.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/t${t}_s$s.json')); print('own_tails $t slices $s: %.3f ms/step' % d['ms_per_step'], d.get('fallbacks'), {k: round(v,2) for k,v in d['pipeline']['stage_ms_sum_of_launches'].items() if v>0})"
done; done | tee $O/sweep.txt
