mkdir -p gpurun_out/s20
python tools/probe/rowwalk_tail.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s20/tail.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "occupancy or votes or early_stop or hash_backward or composite or step_tail" 2>&1 | tail -3 >> gpurun_out/s20/tail.txt
timeout 500 bash tools/kernel_size_sweep.sh r06h --factors 1,3 > gpurun_out/s20/sweep.log 2>&1
timeout 300 python tools/converged_steps.py --native --steps 300 2>&1 | grep -v amdgpu.ids >> gpurun_out/s20/tail.txt
