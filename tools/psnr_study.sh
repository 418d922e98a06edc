#!/bin/bash
# PSNR@20k, product numerics against the reference-numerics build, N paired seeds (SEED0 .. SEED0+N-1), one training at a time
# (round-4 verdict, next 3).  Each worker is `bench.py --psnr-worker N`; tools/psnr_session.py turns the two result files into one more
# entry of profiles/psnr_estimates.json (what bench.py pools).  ~15 s per product training, ~28 s per reference-numerics training.
# Usage: psnr_study.sh [N=10] [SEED0=2022] [TAG=psnr]
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-10}; SEED0=${2:-2022}; TAG=${3:-psnr}
mkdir -p gpurun_out/$TAG
F2N_REFERENCE_NUMERICS=0 timeout $((N * 25 + 120)) python bench.py --psnr-worker $N --psnr-seed0 $SEED0 --psnr-no-rerun --psnr-partial gpurun_out/$TAG/product_partial.json --train-iters 20000 --factor 2 --preset wanjinyou > gpurun_out/$TAG/product.json 2> gpurun_out/$TAG/product.err
F2N_REFERENCE_NUMERICS=1 timeout $((N * 45 + 120)) python bench.py --psnr-worker $N --psnr-seed0 $SEED0 --psnr-no-rerun --psnr-partial gpurun_out/$TAG/refnum_partial.json --train-iters 20000 --factor 2 --preset wanjinyou > gpurun_out/$TAG/refnum.json 2> gpurun_out/$TAG/refnum.err
tail -c 600 gpurun_out/$TAG/product.json; echo; tail -c 600 gpurun_out/$TAG/refnum.json; echo
