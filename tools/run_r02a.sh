mkdir -p gpurun_out
python -m pytest tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r02a_scale_tests.log
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r02a_other_tests.log
python bench.py --steps 200 --warmup 20 --breakdown > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
tail -5 gpurun_out/r02a_scale_tests.log; tail -5 gpurun_out/r02a_other_tests.log; head -c 3000 gpurun_out/r02a_bench.json
