mkdir -p gpurun_out
F2N_GATHER_COMBINE=0 python tools/gather_ab.py > gpurun_out/r02c_gather_ab.txt 2>&1
F2N_GATHER_COMBINE=1 python tools/gather_ab.py >> gpurun_out/r02c_gather_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r02c_gather_ab.txt
python -m pytest tests/test_gpu_scale.py tests/test_gpu_e2e.py -m gpu -x -q -k "dataset_rays or trajectory" 2>&1 | tail -40 > gpurun_out/r02c_tests.log; tail -30 gpurun_out/r02c_tests.log
