cd ${GRAFT_REPO_ROOT:-/root/repo}
# a reference-numerics worker as background load (what bench.py's psnr_numerics_ab runs beside the product worker)
F2N_REFERENCE_NUMERICS=1 timeout 400 python bench.py --psnr-worker 3 --train-iters 20000 --factor 2 --preset wanjinyou > gpurun_out/det_load_refnum.json 2>/dev/null &
LOAD=$!
timeout 400 python tools/determinism_probe.py --iters 20000 --stride 500 --runs 3 > gpurun_out/det_under_load.txt 2>&1
kill $LOAD 2>/dev/null
tail -n 14 gpurun_out/det_under_load.txt
