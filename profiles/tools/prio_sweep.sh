for p in 101 111 110 100 011 001; do
  FELICS_EXP_PRIO=$p python3 bench.py --steps 20 --warmup 3 --no-side-configs --no-decode-leg --no-blocking-extra --cpu-seconds 0 > gpurun_out/r5/prio_$p.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r5/prio_$p.json')); print('prio(spine,front,tail) $p: %.3f ms/step' % d['ms_per_step'], {k: round(v,2) for k,v in d['pipeline']['stage_ms_sum_of_launches'].items() if v>0})"
done
