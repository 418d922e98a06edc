mkdir -p gpurun_out/s35
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "fused_step" 2>&1 | tail -3 > gpurun_out/s35/pytest.txt
run() { timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', round(j['ms_per_step'],4), j['value'])" >> gpurun_out/s35/big.log; }
run --preset wanjinyou_big --log2 22
run --preset wanjinyou_big --log2 20
run
