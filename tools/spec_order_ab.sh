#!/bin/bash
# A/B of Renderer::spec_order_ (bench.py --speculation-order): fresh-scene step time for each setting, two rounds, interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for o in ${ORDERS:-0 1}; do
    python bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 --speculation-order $o 2>/dev/null |
      python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('order $o  fresh ms %.4f  value %.4e' % (d['ms_per_step'], d['value']))"
  done
done
