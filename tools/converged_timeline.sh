#!/bin/bash
# rocprofv3 kernel trace of tools/converged_steps.py (20 000 training iterations, then timed steps behind a 0.3 s pause) ->
# per-step timeline of both queues (profiles/timeline_rocpd.py).  Usage: converged_timeline.sh <tag> [converged_steps.py args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/tools/converged_steps.py --steps 12 "$@" > $OUT/run.log 2> $OUT/run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python profiles/timeline_rocpd.py $DB 2 > $OUT/${TAG}_converged_timeline.txt 2>&1
find $OUT -name "*.db" -delete
tail -2 $OUT/run.log
