mkdir -p gpurun_out/s53
O=gpurun_out/s53
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refnum.py -x -q -k "hash_backward or step_tail or owner" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/pytest.txt
for lay in 0 1; do
  echo "== debug variant, record layout $lay (0: [level][slice][chunk], 1: [level][chunk][slice]); the pair, then the producers alone" >> $O/ab.txt
  F2N_DEBUG_BUILD=1 F2N_BIN_LAYOUT=$lay timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-140 >> $O/ab.txt
  F2N_DEBUG_BUILD=1 F2N_BIN_LAYOUT=$lay F2N_BIN_DISSECT=4 timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-140 >> $O/ab.txt
done
for rep in 1 2; do
for v in ovt pm; do
  cp tools/probe/libf2n_hip_$v.so f2-nerf_amd/libf2n_hip.so
  echo "== $v" >> $O/ab.txt
  timeout -k 5 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
  timeout -k 5 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fresh', round(j['ms_per_step'],4), j['roofline']['timed_calls_ms_per_step'])" >> $O/ab.txt
done
done
cp tools/probe/libf2n_hip_pm.so f2-nerf_amd/libf2n_hip.so
