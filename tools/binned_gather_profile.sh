#!/bin/bash
# rocprofv3 kernel trace of the fresh wanjinyou_big step at 2^22 entries per level with the slice-binned gather from level pair $P0
# (default 1): per-kernel durations of gather_request / gather_serve / gather_blend and the coarse pairs' partitioned gather
# -> gpurun_out/binned_kernel_stats.csv (profiles/r04_binned_gather.txt).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
F2N_DEBUG_BUILD=1 F2N_BINNED_GATHER_P0=${P0:-1} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_binned -- python $R/bench.py --preset wanjinyou_big --log2 22 --steps 20 --warmup 5 --no-cpu-baseline --no-converged --other-configs 0 --no-steady > /dev/null 2> $R/gpurun_out/prof_binned.err
cd $R
DB=$(find gpurun_out/prof_binned -name "*.db" | head -1)
python profiles/summarize_rocpd.py stats $DB gpurun_out/binned_kernel_stats.csv
find gpurun_out/prof_binned -name "*.db" -delete
head -12 gpurun_out/binned_kernel_stats.csv
