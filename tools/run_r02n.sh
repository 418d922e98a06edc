mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "data_parallel" 2>&1 | tail -25 > gpurun_out/r02n_tests.log; tail -8 gpurun_out/r02n_tests.log
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged > gpurun_out/r02n_bench_plain.json 2>/dev/null
F2N_BENCH_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-converged > gpurun_out/r02n_bench_dp1.json 2> gpurun_out/r02n_bench_dp1.err
F2N_BENCH_FORCE_DP=1 F2N_DP_OVERLAP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-converged > gpurun_out/r02n_bench_dp1_overlap.json 2> gpurun_out/r02n_bench_dp1o.err
python - <<'PY'
import json
for f in ("plain","dp1","dp1_overlap"):
    try:
        d=json.loads(open('gpurun_out/r02n_bench_%s.json'%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
