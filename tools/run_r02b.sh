mkdir -p gpurun_out
F2N_GATHER_COMBINE=0 python tools/gather_ab.py > gpurun_out/r02b_gather_ab.txt 2>&1
F2N_GATHER_COMBINE=1 python tools/gather_ab.py >> gpurun_out/r02b_gather_ab.txt 2>&1
cat gpurun_out/r02b_gather_ab.txt
python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "dataset_rays" 2>&1 | tail -40 > gpurun_out/r02b_e2e.log; tail -40 gpurun_out/r02b_e2e.log
