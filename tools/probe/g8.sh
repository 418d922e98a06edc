mkdir -p gpurun_out/s8
python tools/probe/event_cost.py > gpurun_out/s8/event_cost.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "fused_step or deferred_finiteness" --tb=short 2>&1 | tail -15 > gpurun_out/s8/pytest.txt
