mkdir -p gpurun_out/s19
python tools/probe/rowwalk_tail.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s19/tail.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "occupancy or votes or early_stop" 2>&1 | tail -3 >> gpurun_out/s19/tail.txt
