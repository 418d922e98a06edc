mkdir -p gpurun_out/s56
O=gpurun_out/s56
for rep in 1 2 3 4; do
  timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('20 steps', round(j['ms_per_step'],4), 'steady', round(j['steady_state']['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> $O/b.txt
done
