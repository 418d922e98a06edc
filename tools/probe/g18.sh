mkdir -p gpurun_out/s18
run() { timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s18/fresh.log; }
run
run --knob fused_tail=1 --knob spec_begin_before_scatter=1
run --knob fused_tail=0 --knob spec_begin_before_scatter=1
run
run --knob fused_tail=1 --knob spec_begin_before_scatter=1
run --preset llff
run --preset llff --knob fused_tail=1 --knob spec_begin_before_scatter=1
