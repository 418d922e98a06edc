mkdir -p gpurun_out/s57
O=gpurun_out/s57
( time timeout -k 5 900 python bench.py --steps 20 --warmup 5 ) > $O/r06_bench_as_the_driver_runs_it.json 2> $O/bench.err
