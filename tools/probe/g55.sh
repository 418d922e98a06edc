mkdir -p gpurun_out/s55
O=gpurun_out/s55
timeout -k 5 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > $O/pytest.txt
timeout -k 5 300 python -c "
import __graft_entry__ as g
g.smoke()" > $O/smoke.txt 2>&1
( time timeout -k 5 900 python bench.py --steps 20 --warmup 5 ) > $O/r06_bench_as_the_driver_runs_it.json 2> $O/bench.err
PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06 > $O/prof.log 2>&1
timeout -k 5 400 bash tools/converged_timeline.sh r06 --native > $O/tl.log 2>&1
timeout -k 5 500 bash tools/kernel_size_sweep.sh r06 > $O/sweep.log 2>&1
timeout -k 5 300 python tools/scatter_bench.py --reps 50 2>&1 | grep scatter_bench > $O/scatter_bench.txt
