mkdir -p gpurun_out/s30
export F2N_BENCH_FORCE_DP=1
for cp in gloo nccl gloo nccl; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --control-plane $cp 2>gpurun_out/s30/err_$cp.txt | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); d=j['data_parallel']; print('forced one-rank RCCL world, control plane $cp:', round(j['ms_per_step'],4), 'exchange', d['dp_exchange_ms'], 'wait', d['dp_wait_ms'], j['replicas']['identical'])" >> gpurun_out/s30/dp.txt
done
