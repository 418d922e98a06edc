mkdir -p gpurun_out/s39
O=gpurun_out/s39
for rep in 1 2; do
for v in new cur; do
  cp tools/probe/libf2n_hip_$v.so f2-nerf_amd/libf2n_hip.so
  echo "== $v" >> $O/ab.txt
  timeout 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
done
done
cp tools/probe/libf2n_hip_cur.so f2-nerf_amd/libf2n_hip.so
export F2N_DEBUG_BUILD=1
for rep in 1 2; do
for nb in 32 64 128; do
  echo "== nb $nb" >> $O/ab.txt
  F2N_BIN_NB=$nb timeout 300 python tools/converged_steps.py --native --steps 400 2>&1 | grep "native loop" | cut -c1-60 >> $O/ab.txt
done
done
