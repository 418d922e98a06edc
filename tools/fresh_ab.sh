#!/bin/bash
# Fresh-scene step time (bench.py --no-converged, 300 steps) for a list of knob settings, two rounds interleaved.
# Usage: fresh_ab.sh "<label>|<bench args>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for spec in "$@"; do
    label=${spec%%|*}; args=${spec#*|}
    python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-converged --other-configs 0 $args 2>/dev/null |
      python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('%-44s fresh ms %.4f  value %.4e' % ('$label', d['ms_per_step'], d['value']))"
  done
done
