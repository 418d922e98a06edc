set -x
mkdir -p gpurun_out/s5
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "step_tail" --tb=short 2>&1 | grep -v "^  \|^$" | tail -40 > gpurun_out/s5/pytest.txt
timeout 300 bash tools/converged_timeline.sh r06c --native > gpurun_out/s5/tl.log 2>&1
