mkdir -p gpurun_out/s21
timeout 300 python tools/converged_steps.py --native --steps 300 2>&1 | grep -v amdgpu.ids > gpurun_out/s21/native.txt
timeout 300 python tools/converged_steps.py --native --steps 300 2>&1 | grep -v amdgpu.ids >> gpurun_out/s21/native.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_converged.py -x -q 2>&1 | tail -3 >> gpurun_out/s21/native.txt
