mkdir -p gpurun_out/s11
for k in 1 0 1 0; do
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 --knob fused_tail=$k 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused_tail=$k', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s11/fresh_ab.txt
done
for k in 1 0; do
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 --preset llff --knob fused_tail=$k 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('llff fused_tail=$k', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s11/fresh_ab.txt
done
