set -x
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/s1/pytest.txt
timeout 500 bash tools/kernel_size_sweep.sh r06a > gpurun_out/s1/sweep.log 2>&1
timeout 300 bash tools/converged_timeline.sh r06a --native > gpurun_out/s1/tl.log 2>&1
timeout 200 python tools/kernel_size_sweep.py --factors 1 --speculation 2 > gpurun_out/s1/events_spec.log 2>&1
