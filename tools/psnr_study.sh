#!/bin/bash
# PSNR@20k, product numerics against the reference-numerics build, N paired seeds (2022 .. 2022+N-1), one training at a time
# (round-4 verdict, next 3).  Each worker is `bench.py --psnr-worker N`; tools/psnr_pool.py turns the two result files (+ the earlier
# rounds' studies) into profiles/r05_psnr_study.json.  ~17 s per product training, ~28 s per reference-numerics training.
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-10}
mkdir -p gpurun_out/psnr
F2N_REFERENCE_NUMERICS=0 timeout 900 python bench.py --psnr-worker $N --psnr-no-rerun --psnr-partial gpurun_out/psnr/product_partial.json --train-iters 20000 --factor 2 --preset wanjinyou > gpurun_out/psnr/product.json 2> gpurun_out/psnr/product.err
F2N_REFERENCE_NUMERICS=1 timeout 900 python bench.py --psnr-worker $N --psnr-no-rerun --psnr-partial gpurun_out/psnr/refnum_partial.json --train-iters 20000 --factor 2 --preset wanjinyou > gpurun_out/psnr/refnum.json 2> gpurun_out/psnr/refnum.err
tail -c 600 gpurun_out/psnr/product.json; echo; tail -c 600 gpurun_out/psnr/refnum.json; echo
