#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers (run on the GPU box through gpurun):
#   1. --kernel-trace --stats : per-kernel durations          -> profiles/<tag>_kernel_stats.csv
#   2. --pmc FETCH_SIZE       : HBM-side read traffic          -> profiles/<tag>_pmc_fetch_size.csv   (own pass: 3 of 4 TCC slots)
#   2b. --pmc WRITE_SIZE      : HBM-side write traffic         -> profiles/<tag>_pmc_write_size.csv  (own pass)
#       both -> profiles/<tag>_traffic.json (bytes per launch; FETCH_SIZE doubled, the gfx950 correction of the guide)
#   3. --pmc SQ_*             : issue / wait breakdown         -> profiles/<tag>_pmc_sq.csv
#   4. --pmc TCC_HIT_sum TCC_MISS_sum : L2 hit / miss requests -> profiles/<tag>_pmc_tcc.csv   (PASSES="... tcc")
# Counter passes never share a run with tracing (gpurun refuses that combination).  Every pass runs under `timeout` (round 5: rocprofv3 was seen
# to hang in its finalisation behind a finished workload; the results database is complete by then and is summarised all the same).
# Environment: BENCH_EXTRA = extra bench.py arguments (e.g. "--preset wanjinyou_big --log2 22"), PASSES = which passes to run.
set -u
TAG=${1:-r05}
BENCH_EXTRA=${BENCH_EXTRA:-}
PASSES=${PASSES:-"stats fetch write sq"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-converged $BENCH_EXTRA"
for k in $PASSES; do
  case $k in
    stats) timeout ${PASS_TIMEOUT:-200} rocprofv3 --kernel-trace --stats -d $OUT/stats -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err ;;
    fetch) timeout ${PASS_TIMEOUT:-200} rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- $BENCH > /dev/null 2> $OUT/fetch.err ;;
    write) timeout ${PASS_TIMEOUT:-200} rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- $BENCH > /dev/null 2> $OUT/write.err ;;
    sq) timeout ${PASS_TIMEOUT:-200} rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $OUT/sq -- $BENCH > /dev/null 2> $OUT/sq.err ;;
    tcc) timeout ${PASS_TIMEOUT:-200} rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -- $BENCH > /dev/null 2> $OUT/tcc.err ;;
  esac
done
cd $ROOT
for k in $PASSES; do
  DB=$(find $OUT/$k -name "*.db" | head -1)
  echo "$k: $DB"
  case $k in
    stats) python profiles/summarize_rocpd.py stats $DB $OUT/${TAG}_kernel_stats.csv ;;
    fetch) python profiles/summarize_rocpd.py pmc $DB $OUT/${TAG}_pmc_fetch_size.csv ;;
    write) python profiles/summarize_rocpd.py pmc $DB $OUT/${TAG}_pmc_write_size.csv ;;
    sq) python profiles/summarize_rocpd.py pmc $DB $OUT/${TAG}_pmc_sq.csv ;;
    tcc) python profiles/summarize_rocpd.py pmc $DB $OUT/${TAG}_pmc_tcc.csv ;;
  esac
done
if [ -f $OUT/${TAG}_pmc_fetch_size.csv ] && [ -f $OUT/${TAG}_pmc_write_size.csv ]; then
  python profiles/summarize_rocpd.py traffic $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv $OUT/${TAG}_traffic.json
fi
find $OUT -name "*.db" -delete
ls -la $OUT
