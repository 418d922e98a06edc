#!/bin/bash
# Kernel-level A/B of the three MLP kernels (field_shade_fwd, shade_bwd, field_bwd) between two builds of libf2n_hip.so kept under
# f2-nerf_amd/_ab/ (base; MFMA results in VGPR form: F2N_EXTRA_HIPCC="shade.hip:-mllvm -amdgpu-mfma-vgpr-form=1;field.hip:..."):
# rocprofv3 kernel trace of the fresh 8192-ray step with the sampler on the main stream (--speculation off: nothing runs beside them).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-base vgpr base vgpr}; do
  cp $R/f2-nerf_amd/_ab/libf2n_hip_$v.so $R/f2-nerf_amd/libf2n_hip.so
  rm -rf $R/gpurun_out/prof_mlp_ab
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_mlp_ab -- python $R/bench.py --steps 40 --warmup 10 --speculation off --no-cpu-baseline --no-converged --other-configs 0 --no-steady > /dev/null 2> $R/gpurun_out/prof_mlp_ab.err
  DB=$(find $R/gpurun_out/prof_mlp_ab -name "*.db" | head -1)
  python $R/profiles/summarize_rocpd.py stats $DB $R/gpurun_out/mlp_ab_$v.csv > /dev/null
  echo "== $v"; grep -E "^\"?(field_shade_fwd|shade_bwd|field_bwd|field_fwd_kernel)" $R/gpurun_out/mlp_ab_$v.csv | cut -d, -f1-4,7-8
done
cp $R/f2-nerf_amd/_ab/libf2n_hip_base.so $R/f2-nerf_amd/libf2n_hip.so
rm -rf $R/gpurun_out/prof_mlp_ab
