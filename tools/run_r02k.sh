mkdir -p gpurun_out
python -m pytest tests/test_gpu_scale.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02k_tests.log; tail -12 gpurun_out/r02k_tests.log
python bench.py --steps 200 --warmup 20 > gpurun_out/r02k_bench.json 2> gpurun_out/r02k_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02k_bench.json'))
print("fresh ms/step",d["ms_per_step"],"value",d["value"])
c=d["converged"]; print("converged",{k:c[k] for k in ("train_wall_s","psnr_test_mean","ms_per_step","value","rays_per_batch","octree_nodes") if k in c}, c.get("error"))
PY
