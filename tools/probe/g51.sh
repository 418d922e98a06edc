mkdir -p gpurun_out/s51
O=gpurun_out/s51
export F2N_DEBUG_BUILD=1
for d in 4 5 6 4 5; do
  echo "== dissect $d (4: producers alone; 5: ... without their record stores; 6: ... without slot atomics and stores)" >> $O/dissect.txt
  F2N_BIN_DISSECT=$d timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-140 >> $O/dissect.txt
done
