mkdir -p gpurun_out/s54
O=gpurun_out/s54
export F2N_DEBUG_BUILD=1
for d in 4 12 20 4 12 20; do
  echo "== producers alone, dissect $d (4: 8-byte records; 12: 4-byte stores 4 bytes apart; 20: 6-byte records as three 2-byte stores)" >> $O/dissect.txt
  F2N_BIN_DISSECT=$d timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-140 >> $O/dissect.txt
done
