mkdir -p gpurun_out/s12
for k in 2 0 1; do
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 --knob fused_tail=$k 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fresh fox, fused_tail=$k', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s12/fresh_ab.txt
done
timeout 300 python tools/converged_steps.py --steps 200 --fused-tail-sweep >> gpurun_out/s12/fresh_ab.txt 2>&1
timeout 200 python tools/converged_steps.py --native --steps 300 >> gpurun_out/s12/fresh_ab.txt 2>&1
