mkdir -p gpurun_out/s45
O=gpurun_out/s45
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refnum.py -x -q -k "hash_backward or step_tail or owner" 2>&1 | tail -15 > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -5 >> $O/pytest.txt
for rep in 1 2 3; do
for v in cur ovt; do
  cp tools/probe/libf2n_hip_$v.so f2-nerf_amd/libf2n_hip.so
  echo "== $v" >> $O/ab.txt
  timeout 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
  [ $rep != 3 ] && timeout 300 python bench.py --preset wanjinyou_big --log2 22 --steps 200 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('big22', round(j['ms_per_step'],4), j['scatter_counters'])" >> $O/ab.txt
done
done
cp tools/probe/libf2n_hip_ovt.so f2-nerf_amd/libf2n_hip.so
