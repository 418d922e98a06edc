ROOT=$(pwd); mkdir -p gpurun_out; OUT=$ROOT/gpurun_out/prof_r02h; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-converged --marker-pause > $OUT/run.log 2> $OUT/run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python profiles/timeline_rocpd.py $DB 1 > gpurun_out/r02h_fresh_timeline.txt 2>&1
find $OUT -name "*.db" -delete
head -c 600 $OUT/run.log; echo; head -100 gpurun_out/r02h_fresh_timeline.txt
