#!/bin/bash
# Where does hash_bin_accumulate_kernel's time go?  Its duration (rocprofv3 kernel trace of 12 converged python-driven steps) with
# parts of it switched off by F2N_ACC_DBG (1 no LDS adds, 2 no records at all, 4 no flush; results are wrong then).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in 0 1 2 6; do
  F2N_ACC_DBG=$d bash tools/converged_timeline.sh accdbg$d --depth 2 --iters 3000 > /dev/null 2>&1
  echo "F2N_ACC_DBG=$d: $(grep -E 'e_kernel12F2nBinQueues|DF16_llPKt12F2nBinQueues' gpurun_out/tl_accdbg$d/accdbg${d}_converged_timeline.txt | tail -2 | tr '\n' ' ')"
done
