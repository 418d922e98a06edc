mkdir -p gpurun_out/s31
export F2N_BENCH_FORCE_DP=1
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/s31/trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 $R/bench.py --gpus 1 --steps 12 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --no-steady --marker-pause > $R/gpurun_out/s31/run.log 2> $R/gpurun_out/s31/run.err
cd $R
for DB in $(find gpurun_out/s31/trace -name "*.db"); do python profiles/timeline_rocpd.py $DB 3 > gpurun_out/s31/tl.txt 2>&1; done
find gpurun_out/s31 -name "*.db" -delete
