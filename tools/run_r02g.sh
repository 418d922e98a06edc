ROOT=$(pwd); mkdir -p gpurun_out; OUT=$ROOT/gpurun_out/prof_r02g; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/tools/converged_steps.py --iters 20000 --steps 12 > $OUT/run.log 2> $OUT/run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1); ls -la $DB
python profiles/summarize_rocpd.py schema $DB $OUT/schema.txt
python profiles/timeline_rocpd.py $DB 1 > gpurun_out/r02g_converged_timeline.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/run.log; head -120 gpurun_out/r02g_converged_timeline.txt
