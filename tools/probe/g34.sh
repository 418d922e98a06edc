mkdir -p gpurun_out/s34
run() { timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s34/big.log; }
run --preset wanjinyou_big --log2 22
run --preset wanjinyou_big --log2 22 --knob fused_tail=1
run --preset wanjinyou_big --log2 20
run --preset wanjinyou_big --log2 20 --knob fused_tail=1
run --preset wanjinyou_big --log2 21
run --preset wanjinyou_big --log2 21 --knob fused_tail=1
