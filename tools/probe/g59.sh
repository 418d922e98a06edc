mkdir -p gpurun_out/s59
export F2N_BENCH_FORCE_DP=1
timeout -k 5 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 2>gpurun_out/s59/err.txt | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); d=j['data_parallel']; print('forced one-rank RCCL world:', round(j['ms_per_step'],4), 'exchange', d['dp_exchange_ms'], 'wait', d['dp_wait_ms'], j['replicas']['identical'], j['scatter_counters'])" > gpurun_out/s59/dp.txt
tail -5 gpurun_out/s59/err.txt >> gpurun_out/s59/dp.txt
