set -x
mkdir -p gpurun_out/s2
timeout 500 bash tools/kernel_size_sweep.sh r06b > gpurun_out/s2/sweep.log 2>&1
timeout 300 bash tools/converged_timeline.sh r06b --native > gpurun_out/s2/tl.log 2>&1
timeout 200 python tools/converged_steps.py --native --steps 200 > gpurun_out/s2/native.log 2>&1
timeout 200 python tools/converged_steps.py --steps 100 >> gpurun_out/s2/native.log 2>&1
