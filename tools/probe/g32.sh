mkdir -p gpurun_out/s32
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > gpurun_out/s32/pytest.txt
python -c "
import __graft_entry__ as g
g.smoke()" > gpurun_out/s32/smoke.txt 2>&1
