mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02i_tests.log; tail -6 gpurun_out/r02i_tests.log
python bench.py --steps 200 --warmup 20 > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02i_bench.json'))
print("fresh ms/step",d["ms_per_step"],"value",d["value"], d["roofline"]["timed_calls_ms_per_step"])
c=d["converged"]; print("converged",{k:c[k] for k in ("train_wall_s","psnr_test_mean","ms_per_step","value","rays_per_batch","rho_marched_over_meaningful") if k in c}, c.get("error"))
PY
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r02i; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -- python $ROOT/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-converged --marker-pause > $OUT/run.log 2> $OUT/run.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python profiles/timeline_rocpd.py $DB 1 > gpurun_out/r02i_fresh_timeline.txt 2>&1
find $OUT -name "*.db" -delete
head -60 gpurun_out/r02i_fresh_timeline.txt
