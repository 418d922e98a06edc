mkdir -p gpurun_out/s43
O=gpurun_out/s43
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > $O/pytest.txt
python -c "
import __graft_entry__ as g
g.smoke()" > $O/smoke.txt 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/r06_bench_as_the_driver_runs_it.json 2> $O/bench.err
PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06 > $O/prof.log 2>&1
timeout 400 bash tools/converged_timeline.sh r06 --native > $O/tl.log 2>&1
timeout 500 bash tools/kernel_size_sweep.sh r06 > $O/sweep.log 2>&1
