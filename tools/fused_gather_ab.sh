#!/bin/bash
# A/B of the pre-pass's hash gather -> density MLP: XCD-partitioned gather into f16 planes + plane-fed MLP (the product) against
# ONE kernel that gathers all 16 levels of a 16-sample tile per wave and feeds the matrix cores from registers (F2N_FUSED_GATHER=1).
# Converged fox scene (python-driven steps, HIP-event timing of the calls) and the fresh scene at 2^22 entries per level.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in 0 1; do
  echo "== converged scene, F2N_FUSED_GATHER=$f"
  F2N_DEBUG_BUILD=1 F2N_FUSED_GATHER=$f python tools/converged_steps.py --steps 100 --depth 2 --kernel-timing 2>/dev/null | tail -2
done
for f in 0 1; do
  echo "== wanjinyou_big log2 22 (fresh), F2N_FUSED_GATHER=$f"
  F2N_DEBUG_BUILD=1 F2N_FUSED_GATHER=$f python bench.py --preset wanjinyou_big --log2 22 --steps 60 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --no-steady --breakdown 2>&1 >/dev/null | grep -E "hash_gather|field_mlp_prepass|field_prepass|sum of" 
done
for f in 0 1; do
  echo "== wanjinyou log2 19 (fresh headline), F2N_FUSED_GATHER=$f"
  F2N_DEBUG_BUILD=1 F2N_FUSED_GATHER=$f python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-converged --other-configs 0 --no-steady --breakdown 2>&1 >/dev/null | grep -E "hash_gather|field_mlp_prepass|field_prepass|sum of"
done
