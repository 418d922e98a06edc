mkdir -p gpurun_out/s22
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/s22/pytest.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/s22/bench_driver.json 2> gpurun_out/s22/bench_driver.err
timeout 300 bash tools/converged_timeline.sh r06 --native > gpurun_out/s22/tl.log 2>&1
PASSES="stats" PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06 > gpurun_out/s22/prof.log 2>&1
