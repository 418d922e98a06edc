#!/bin/bash
# First GPU call of the round that picks up branch wip/next (launch floor + LDS octree walk + optimistic pack; none of it has
# been run beyond six targeted tests of the launch-floor part).  One call, ~2 GPU-minutes:
#   1. the parity tests that cover what changed (bit-exact against the oracle / against the unchanged kernels);
#   2. fresh-scene step time for each knob setting, two rounds interleaved (bench.py --no-converged, 300 steps).
# Usage on the GPU box:  bash tools/next_round_ab.sh > gpurun_out/next_round_ab.log 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_e2e.py -q -x \
  -k "sampler or segment_scan or nonfinite or train_loss or speculative or streaming_step or deferred_finiteness or aux_states" 2>&1 | tail -5
run() {  # run <label> <bench args...>
  local label=$1; shift
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 "$@" 2>/dev/null |
    python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('%-34s fresh ms %.4f  value %.4e' % ('$label', d['ms_per_step'], d['value']))"
}
for round in 1 2; do
  run "lds 0 / optimistic 0"  --lds-octree 0 --optimistic-pack 0
  run "lds 1 / optimistic 0"  --lds-octree 1 --optimistic-pack 0
  run "lds 0 / optimistic 1"  --lds-octree 0 --optimistic-pack 1
  run "lds 1 / optimistic 1"  --lds-octree 1 --optimistic-pack 1
done
