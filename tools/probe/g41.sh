mkdir -p gpurun_out/s41
O=gpurun_out/s41
for rep in 1 2 3; do
  for f in "" "--spec-start-event" "--draws-on-main"; do
  echo "== $f" >> $O/ab.txt
  timeout 300 python tools/converged_steps.py --native --steps 400 $f 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
  done
done
timeout 1200 python -m pytest tests/test_gpu_determinism.py -x -q 2>&1 | tail -5 > $O/pytest.txt
