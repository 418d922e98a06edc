mkdir -p gpurun_out/s46
O=gpurun_out/s46
export F2N_DEBUG_BUILD=1
for rep in 1 2; do
for nb in 128 64; do
  echo "== fresh nb $nb" >> $O/ab.txt
  F2N_BIN_NB=$nb timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fresh', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> $O/ab.txt
done
done
for nb in 128 64; do
  echo "== llff nb $nb" >> $O/ab.txt
  F2N_BIN_NB=$nb timeout 300 python bench.py --preset llff --steps 200 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('llff', round(j['ms_per_step'],4), j['value'])" >> $O/ab.txt
done
