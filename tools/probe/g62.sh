mkdir -p gpurun_out/s62
O=gpurun_out/s62
timeout -k 5 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > $O/pytest.txt
timeout -k 5 300 python -c "
import __graft_entry__ as g
g.smoke()" > $O/smoke.txt 2>&1
PASSES="stats" PASS_TIMEOUT=240 bash profiles/run_profiles.sh r06 > $O/prof.log 2>&1
