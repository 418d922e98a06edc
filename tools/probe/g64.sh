mkdir -p gpurun_out/s64
O=gpurun_out/s64
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refnum.py -x -q -k "hash_backward or step_tail or owner" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/pytest.txt
cp tools/probe/libf2n_hip_wc2.so f2-nerf_amd/libf2n_hip.so
for v in ix2 wc2 ix2 wc2; do
  cp tools/probe/libf2n_hip_$v.so f2-nerf_amd/libf2n_hip.so
  echo "== $v" >> $O/ab.txt
  timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-200 >> $O/ab.txt
  timeout -k 5 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
  timeout -k 5 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fresh', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> $O/ab.txt
done
cp tools/probe/libf2n_hip_wc2.so f2-nerf_amd/libf2n_hip.so
