mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02f_tests.log; tail -8 gpurun_out/r02f_tests.log
python bench.py --steps 200 --warmup 20 --breakdown > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f_bench.json'))
print("fresh ms/step",d["ms_per_step"],"value",d["value"], d["roofline"]["timed_calls_ms_per_step"])
c=d["converged"]; print("converged",{k:c[k] for k in ("train_wall_s","psnr_test_mean","ms_per_step","value","rays_per_batch") if k in c}, c.get("error"))
print(c.get("timed_calls_ms_per_step"))
PY
grep -v amdgpu gpurun_out/r02f_bench.err | head -30
