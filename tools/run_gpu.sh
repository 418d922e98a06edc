#!/bin/bash
# scratch driver for one gpurun call; not part of the product
ROOT=$(pwd)
bash profiles/run_profiles.sh r02 > gpurun_out/r02_profiles.log 2>&1
OUT=$ROOT/gpurun_out/prof_r02
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-converged --marker-pause > $OUT/trace_run.log 2> $OUT/trace_run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python profiles/timeline_rocpd.py $DB 1 > $OUT/r02_fresh_timeline.txt 2>&1
find $OUT -name "*.db" -delete
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace2 -- python $ROOT/tools/converged_steps.py --iters 20000 --steps 12 > $OUT/trace2_run.log 2> $OUT/trace2_run.err
cd $ROOT
DB=$(find $OUT/trace2 -name "*.db" | head -1)
python profiles/timeline_rocpd.py $DB 1 > $OUT/r02_converged_timeline.txt 2>&1
find $OUT -name "*.db" -delete
for P in llff nerf-360; do
  python bench.py --no-cpu-baseline --no-converged --preset $P --steps 300 > $OUT/r02_bench_$P.json 2> $OUT/bench_$P.err
done
python bench.py --no-cpu-baseline --no-converged --preset wanjinyou_big --steps 300 > $OUT/r02_bench_wanjinyou_big_log2_20.json 2> $OUT/bench_big20.err
python bench.py --no-cpu-baseline --no-converged --preset wanjinyou_big --log2 22 --steps 300 > $OUT/r02_bench_wanjinyou_big_log2_22.json 2> $OUT/bench_big22.err
ls -la $OUT
