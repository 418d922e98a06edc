mkdir -p gpurun_out/s29
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "data_parallel or bucketed" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > gpurun_out/s29/pytest.txt
export F2N_BENCH_FORCE_DP=1
for rep in 1 2; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 2>gpurun_out/s29/err.txt | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); d=j['data_parallel']; print('forced one-rank RCCL world, tail chain on the communicator stream:', round(j['ms_per_step'],4), 'exchange', d['dp_exchange_ms'], 'wait', d['dp_wait_ms'], j['replicas']['identical'])" >> gpurun_out/s29/dp.txt
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/s29/trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 $R/bench.py --gpus 1 --steps 12 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --no-steady --marker-pause > $R/gpurun_out/s29/run.log 2> $R/gpurun_out/s29/run.err
cd $R
for DB in $(find gpurun_out/s29/trace -name "*.db"); do python profiles/timeline_rocpd.py $DB 2 > gpurun_out/s29/tl.txt 2>&1; done
find gpurun_out/s29 -name "*.db" -delete
