mkdir -p gpurun_out
for cfg in "0 0 256" "1 0 256" "0 1 256" "0 2 256" "0 4 256" "0 0 128" "0 0 512" "0 0 1024"; do
  set -- $cfg
  echo "== COMBINE=$1 DBG=$2 BPP=$3"
  F2N_GATHER_COMBINE=$1 F2N_GATHER_DBG=$2 F2N_GATHER_BPP=$3 python tools/gather_ab.py 2>&1 | grep "gather"
done > gpurun_out/r02d_gather_exp.txt
cat gpurun_out/r02d_gather_exp.txt
