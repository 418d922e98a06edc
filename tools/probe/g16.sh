mkdir -p gpurun_out/s16
PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06 > gpurun_out/s16/prof.log 2>&1
timeout 300 bash tools/converged_timeline.sh r06 --native > gpurun_out/s16/tl.log 2>&1
