mkdir -p gpurun_out/s10
timeout 200 python tools/converged_steps.py --native --steps 300 > gpurun_out/s10/native.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --psnr-runs 0 --psnr-ref-runs 0 > gpurun_out/s10/bench.json 2> gpurun_out/s10/bench.err
timeout 300 bash tools/converged_timeline.sh r06e --native > gpurun_out/s10/tl.log 2>&1
