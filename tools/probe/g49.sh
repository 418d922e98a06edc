mkdir -p gpurun_out/s49
O=$GRAFT_REPO_ROOT/gpurun_out/s49
export F2N_DEBUG_BUILD=1
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2; do
for only in "47%" "fresh"; do
  tag=$(echo $only | tr -d '%')
  F2N_BIN_DISSECT=$d timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_${d}_$tag -- python $GRAFT_REPO_ROOT/tools/scatter_bench.py --reps 40 --amps 2e-4 --only "$only" > $O/run_${d}_$tag.log 2>&1
  F=$(find $O/t_${d}_$tag -name "*kernel_stats.csv" | head -1)
  echo "== dissect $d set $only" >> $O/dissect.txt
  grep -i "hash_bin" $F | cut -c1-200 >> $O/dissect.txt
  grep scatter_bench $O/run_${d}_$tag.log | cut -c1-200 >> $O/dissect.txt
  find $O/t_${d}_$tag -name "*.db" -delete
done
done
