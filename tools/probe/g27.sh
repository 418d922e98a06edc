mkdir -p gpurun_out/s27
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py -x -q -k "data_parallel or bucketed or step_tail or fused_step" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > gpurun_out/s27/pytest.txt
export F2N_BENCH_FORCE_DP=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/s27/trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 $R/bench.py --gpus 1 --steps 12 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --no-steady --marker-pause > $R/gpurun_out/s27/run.log 2> $R/gpurun_out/s27/run.err
cd $R
for DB in $(find gpurun_out/s27/trace -name "*.db"); do python profiles/timeline_rocpd.py $DB 2 > gpurun_out/s27/tl.txt 2>&1; done
find gpurun_out/s27 -name "*.db" -delete
