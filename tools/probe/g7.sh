mkdir -p gpurun_out/s7
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "fused_step" --tb=long 2>&1 | tail -60 > gpurun_out/s7/pytest.txt
