mkdir -p gpurun_out
python tools/gather_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02e_gather_ab.txt; cat gpurun_out/r02e_gather_ab.txt
python -m pytest tests/test_gpu_scale.py tests/test_gpu_e2e.py -m gpu -x -q -k "dataset_rays or trajectory or config2" 2>&1 | tail -30 > gpurun_out/r02e_tests.log; tail -12 gpurun_out/r02e_tests.log
python bench.py --steps 200 --warmup 20 --breakdown > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02e_bench.json'))
print("fresh ms/step",d["ms_per_step"],"value",d["value"], d["roofline"]["timed_calls_ms_per_step"])
c=d["converged"]; print("converged",{k:c[k] for k in ("train_wall_s","psnr_test_mean","ms_per_step","value","rays_per_batch") if k in c}, c.get("error"))
print(c.get("timed_calls_ms_per_step"))
PY
