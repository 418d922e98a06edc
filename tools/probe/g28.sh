mkdir -p gpurun_out/s28
export F2N_BENCH_FORCE_DP=1
for q in 4 8 6; do
GPU_MAX_HW_QUEUES=$q timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 2>gpurun_out/s28/err.txt | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); d=j['data_parallel']; print('forced one-rank RCCL world, GPU_MAX_HW_QUEUES=$q:', round(j['ms_per_step'],4), 'exchange', d['dp_exchange_ms'], 'wait', d['dp_wait_ms'], j['replicas']['identical'])" >> gpurun_out/s28/dp.txt
done
unset F2N_BENCH_FORCE_DP
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 300 python tools/converged_steps.py --native --steps 300 2>&1 | grep "native loop" | sed "s/^/single GPU, GPU_MAX_HW_QUEUES=$q: /" >> gpurun_out/s28/dp.txt
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('single GPU fresh, GPU_MAX_HW_QUEUES=$q:', round(j['ms_per_step'],4))" >> gpurun_out/s28/dp.txt
done
