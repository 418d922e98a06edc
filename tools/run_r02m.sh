mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py -m gpu -x -q -k "data_parallel or shade_fused" 2>&1 | tail -25 > gpurun_out/r02m_tests.log; tail -12 gpurun_out/r02m_tests.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-converged > gpurun_out/r02m_bench_plain.json 2>/dev/null
F2N_BENCH_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-converged > gpurun_out/r02m_bench_dp1.json 2> gpurun_out/r02m_bench_dp1.err
python - <<'PY'
import json
for f in ("plain","dp1"):
    try:
        d=json.loads(open('gpurun_out/r02m_bench_%s.json'%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
tail -5 gpurun_out/r02m_bench_dp1.err
