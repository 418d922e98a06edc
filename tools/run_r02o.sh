mkdir -p gpurun_out
bash profiles/run_profiles.sh r02 > gpurun_out/r02o_profiles.log 2>&1
tail -15 gpurun_out/r02o_profiles.log
python bench.py > gpurun_out/r02o_bench_default.json 2> gpurun_out/r02o_bench_default.err
head -c 1500 gpurun_out/r02o_bench_default.json
