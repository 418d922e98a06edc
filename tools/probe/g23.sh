mkdir -p gpurun_out/s23
run() { timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> gpurun_out/s23/fresh.log; }
run
run --knob fused_tail=0
run
run --knob fused_tail=0
run --preset llff
run --preset llff --knob fused_tail=0
run --preset nerf-360
run --preset nerf-360 --knob fused_tail=0
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_determinism.py -x -q -k "step_tail or fused_step or determinism or streaming_step or deferred_fin" 2>&1 | tail -3 >> gpurun_out/s23/fresh.log
timeout 300 python tools/converged_steps.py --native --steps 300 2>&1 | grep -v amdgpu.ids >> gpurun_out/s23/fresh.log
