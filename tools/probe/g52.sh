mkdir -p gpurun_out/s52
O=gpurun_out/s52
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refnum.py -x -q -k "hash_backward or step_tail or owner" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 > $O/pytest.txt
timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-200 > $O/sb_wc.txt
F2N_DEBUG_BUILD=1 F2N_SCATTER_WC=0 timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-200 > $O/sb_nowc.txt
F2N_DEBUG_BUILD=1 F2N_BIN_DISSECT=4 timeout -k 5 200 python tools/scatter_bench.py --reps 60 --amps 2e-4 2>&1 | grep scatter_bench | cut -c1-140 > $O/sb_wc_producers_alone.txt
