mkdir -p gpurun_out/s36
timeout 500 bash tools/kernel_size_sweep.sh r06i --factors 1,3 > gpurun_out/s36/sweep.log 2>&1
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fresh', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" > gpurun_out/s36/fresh.txt
timeout 300 python tools/converged_steps.py --native --steps 300 2>&1 | grep "native loop" >> gpurun_out/s36/fresh.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hash or gather" 2>&1 | tail -2 >> gpurun_out/s36/fresh.txt
