set -x
mkdir -p gpurun_out/s6
python tools/probe/event_cost.py > gpurun_out/s6/event_cost.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_determinism.py tests/test_gpu_scale.py -x -q -k "step_tail or bucketed or deferred_finiteness or streaming_step or hooks_on_one or fused_step or determinism or speculative_training or prefetched" 2>&1 | tail -8 > gpurun_out/s6/pytest.txt
timeout 300 python tools/converged_steps.py --steps 200 --fused-tail-sweep > gpurun_out/s6/tail.log 2>&1
timeout 200 python tools/converged_steps.py --native --steps 300 > gpurun_out/s6/native.log 2>&1
timeout 300 bash tools/converged_timeline.sh r06d --native > gpurun_out/s6/tl.log 2>&1
