mkdir -p gpurun_out/s40
O=gpurun_out/s40
timeout 1200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_converged.py -x -q 2>&1 | tail -5 > $O/pytest.txt
for rep in 1 2 3; do
  echo "== draws on main" >> $O/ab.txt
  timeout 300 python tools/converged_steps.py --native --steps 400 --draws-on-main 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
  echo "== draws on the tail stream" >> $O/ab.txt
  timeout 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
done
