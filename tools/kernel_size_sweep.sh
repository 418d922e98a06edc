#!/bin/bash
# rocprofv3 kernel trace of tools/kernel_size_sweep.py -> kernel x batch-size table (profiles/sweep_rocpd.py).
# Usage: kernel_size_sweep.sh <tag> [kernel_size_sweep.py args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sweep_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/tools/kernel_size_sweep.py --no-events "$@" > $OUT/run.log 2> $OUT/run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
LABELS=$(grep PHASE $OUT/run.log | sed 's/.*meaningful\/step \([0-9]*\).*/\1/' | tr '\n' ' ')
(grep PHASE $OUT/run.log; python profiles/sweep_rocpd.py $DB $LABELS) > $OUT/${TAG}_kernel_size_sweep.txt 2>&1
find $OUT -name "*.db" -delete
tail -30 $OUT/${TAG}_kernel_size_sweep.txt
