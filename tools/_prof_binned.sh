cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F2N_BINNED_GATHER_P0=${P0:-2} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_binned -- python $R/bench.py --preset wanjinyou_big --log2 22 --steps 20 --warmup 5 --no-cpu-baseline --no-converged --other-configs 0 --no-steady > /dev/null 2> $R/gpurun_out/prof_binned.err
cd $R
DB=$(find gpurun_out/prof_binned -name "*.db" | head -1)
python profiles/summarize_rocpd.py stats $DB gpurun_out/binned_kernel_stats.csv
find gpurun_out/prof_binned -name "*.db" -delete
head -12 gpurun_out/binned_kernel_stats.csv
