mkdir -p gpurun_out/s48
O=gpurun_out/s48
for rep in 1 2 3; do
for k in "" "--knob fused_tail=1"; do
  echo "== fresh $k" >> $O/ab.txt
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 $k 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fresh', round(j['ms_per_step'],4), round(j['steady_state']['ms_per_step'],4) if j.get('steady_state') else None)" >> $O/ab.txt
done
done
for k in "" "--knob fused_tail=1"; do
  echo "== llff $k" >> $O/ab.txt
  timeout 300 python bench.py --preset llff --steps 200 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 $k 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('llff', round(j['ms_per_step'],4), j['value'])" >> $O/ab.txt
  echo "== big20 $k" >> $O/ab.txt
  timeout 300 python bench.py --preset wanjinyou_big --log2 20 --steps 200 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 $k 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('big20', round(j['ms_per_step'],4), j['value'])" >> $O/ab.txt
done
