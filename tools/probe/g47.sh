mkdir -p gpurun_out/s47
O=gpurun_out/s47
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > $O/pytest.txt
python -c "
import __graft_entry__ as g
g.smoke()" > $O/smoke.txt 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/r06_bench_as_the_driver_runs_it.json 2> $O/bench.err
PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06 > $O/prof.log 2>&1
timeout 400 bash tools/converged_timeline.sh r06 --native > $O/tl.log 2>&1
timeout 500 bash tools/kernel_size_sweep.sh r06 > $O/sweep.log 2>&1
BENCH_EXTRA="--preset wanjinyou_big --log2 22" PASSES="stats fetch write" PASS_TIMEOUT=300 bash profiles/run_profiles.sh r06_big22 > $O/big22.log 2>&1
BENCH_EXTRA="--preset wanjinyou_big --log2 20" PASSES="stats fetch write" PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06_big20 > $O/big20.log 2>&1
timeout 300 python tools/scatter_bench.py --reps 50 2>&1 | grep scatter_bench > $O/scatter_bench.txt
