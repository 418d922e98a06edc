mkdir -p gpurun_out/s38
O=gpurun_out/s38
export F2N_DEBUG_BUILD=1
timeout 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" >> $O/cnt.txt
run() { timeout 300 python bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-converged --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', round(j['ms_per_step'],4), j['scatter_counters'])" >> $O/cnt.txt; }
run
run --preset llff
run --preset nerf-360
run --preset wanjinyou_big --log2 20
run --preset wanjinyou_big --log2 22
