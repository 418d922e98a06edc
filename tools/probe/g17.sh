mkdir -p gpurun_out/s17
export F2N_DEBUG_BUILD=1
for p in 0 1 0 1; do
echo "F2N_SIDE_PRIO=$p" >> gpurun_out/s17/prio.log
F2N_SIDE_PRIO=$p timeout 300 python tools/converged_steps.py --restore --steps 200 --block-waves-sweep 4 2>&1 | grep -v "^/opt" >> gpurun_out/s17/prio.log
done
for p in 0 1; do
echo "fresh F2N_SIDE_PRIO=$p" >> gpurun_out/s17/prio.log
F2N_SIDE_PRIO=$p timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s17/prio.log
done
