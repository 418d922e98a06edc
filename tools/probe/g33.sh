mkdir -p gpurun_out/s33
BENCH_EXTRA="--preset llff" PASSES="fetch write" PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06_llff > gpurun_out/s33/llff.log 2>&1
BENCH_EXTRA="--preset nerf-360" PASSES="fetch write" PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06_nerf-360 > gpurun_out/s33/n360.log 2>&1
BENCH_EXTRA="--preset wanjinyou_big --log2 20" PASSES="stats fetch write" PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06_big20 > gpurun_out/s33/big20.log 2>&1
BENCH_EXTRA="--preset wanjinyou_big --log2 22" PASSES="fetch write" PASS_TIMEOUT=300 bash profiles/run_profiles.sh r06_big22 > gpurun_out/s33/big22.log 2>&1
