set -x
mkdir -p gpurun_out/s3
timeout 300 python -m pytest tests/test_gpu_scale.py -x -q -k "persistent_march or speculative_training" 2>&1 | tail -5 > gpurun_out/s3/pytest.txt
timeout 300 python tools/converged_steps.py --steps 100 --block-waves-sweep 1,2,4,8,16 --kernel-timing > gpurun_out/s3/waves.log 2>&1
