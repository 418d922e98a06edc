mkdir -p gpurun_out
python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "rig_presets or big_table" 2>&1 | tail -25 > gpurun_out/r02l_tests.log; tail -14 gpurun_out/r02l_tests.log
for p in llff nerf-360; do python bench.py --preset $p --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02l_bench_$p.json 2> gpurun_out/r02l_bench_$p.err; tail -c 300 gpurun_out/r02l_bench_$p.err; head -c 900 gpurun_out/r02l_bench_$p.json; echo; done
python bench.py --preset wanjinyou_big --steps 100 --warmup 10 --no-cpu-baseline --no-converged > gpurun_out/r02l_bench_big20.json 2> gpurun_out/r02l_bench_big20.err; head -c 700 gpurun_out/r02l_bench_big20.json; echo
python bench.py --preset wanjinyou_big --log2 22 --steps 100 --warmup 10 --no-cpu-baseline --no-converged > gpurun_out/r02l_bench_big22.json 2> gpurun_out/r02l_bench_big22.err; head -c 700 gpurun_out/r02l_bench_big22.json; echo
