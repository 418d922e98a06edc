set -x
mkdir -p gpurun_out/s4
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q -k "step_tail or bucketed or deferred_finiteness or streaming_step or hooks_on_one or adam or config1" 2>&1 | tail -8 > gpurun_out/s4/pytest.txt
timeout 300 python tools/converged_steps.py --steps 200 --fused-tail-sweep > gpurun_out/s4/tail.log 2>&1
timeout 200 python tools/converged_steps.py --native --steps 300 > gpurun_out/s4/native.log 2>&1
