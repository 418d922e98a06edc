mkdir -p gpurun_out/s13
timeout 500 bash tools/kernel_size_sweep.sh r06f > gpurun_out/s13/sweep.log 2>&1
timeout 500 bash tools/kernel_size_sweep.sh r06g --fused-tail 0 > gpurun_out/s13/sweep0.log 2>&1
