mkdir -p gpurun_out/s9
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s9/pytest.txt
timeout 200 python tools/converged_steps.py --native --steps 300 > gpurun_out/s9/native.log 2>&1
