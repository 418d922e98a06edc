mkdir -p gpurun_out/s26
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py -x -q -k "data_parallel or bucketed or step_tail or fused_step" 2>&1 | tail -4 > gpurun_out/s26/pytest.txt
export F2N_BENCH_FORCE_DP=1
for rep in 1 2; do
for k in 2 0; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --knob fused_tail=$k 2>gpurun_out/s26/err.txt | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); d=j['data_parallel']; print('forced one-rank RCCL world, fused_tail=$k:', round(j['ms_per_step'],4), 'exchange', d['dp_exchange_ms'], 'wait', d['dp_wait_ms'], 'early', d['small_buffers_exchanged_beside_the_scatter_rank0'], j['replicas']['identical'])" >> gpurun_out/s26/dp.txt
done
done
