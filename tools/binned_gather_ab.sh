#!/bin/bash
# A/B of the big-table gather (BASELINE config 5): XCD-partitioned gather of all eight level pairs (F2N_BINNED_GATHER_P0=8)
# against the slice-binned pipeline for the level pairs >= P0 (f2n_hash_gather_planes_binned).  HIP-event time of the call and
# the step time of the fresh scene at 2^22 and 2^21 entries per level.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lg in ${LGS:-22 21}; do
  for p0 in ${P0S:-8 0 1 2 3}; do
    echo "== wanjinyou_big log2 $lg, F2N_BINNED_GATHER_P0=$p0"
    F2N_DEBUG_BUILD=1 F2N_BINNED_GATHER_P0=$p0 python bench.py --preset wanjinyou_big --log2 $lg --steps 60 --warmup 10 --no-cpu-baseline --no-converged \
      --other-configs 0 --no-steady --breakdown 2>$$.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'samples/s %.3g' % d['value'])"
    grep -E "hash_gather|field_mlp_prepass|sum of" $$.err; rm -f $$.err
  done
done
