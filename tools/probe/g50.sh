mkdir -p gpurun_out/s50
O=gpurun_out/s50
export F2N_DEBUG_BUILD=1
for rep in 1 2; do
for d in 0 1; do
  echo "== dissect $d" >> $O/dissect.txt
  F2N_BIN_DISSECT=$d timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-140 >> $O/dissect.txt
done
done
