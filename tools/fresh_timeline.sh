#!/bin/bash
# rocprofv3 kernel trace of the fresh-scene bench (bench.py --no-converged, marker pause before the timed steps) -> per-step
# timeline of the queues (profiles/timeline_rocpd.py).  Usage: fresh_timeline.sh <tag> [bench.py args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/bench.py --steps 12 --warmup 30 --no-cpu-baseline --no-converged --other-configs 0 --no-steady --marker-pause "$@" > $OUT/run.log 2> $OUT/run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python profiles/timeline_rocpd.py $DB 2 > $OUT/${TAG}_fresh_timeline.txt 2>&1
find $OUT -name "*.db" -delete
