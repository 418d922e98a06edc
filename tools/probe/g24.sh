mkdir -p gpurun_out/s24
export F2N_BENCH_FORCE_DP=1
for b in 4 1; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --dp-buckets $b 2>gpurun_out/s24/err$b.txt | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('forced one-rank RCCL world, table buckets $b:', round(j['ms_per_step'],4), j['data_parallel'], j['replicas']['identical'])" >> gpurun_out/s24/dp.txt
done
