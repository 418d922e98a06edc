mkdir -p gpurun_out/s15
export F2N_DEBUG_BUILD=1
timeout 400 python tools/converged_steps.py --restore --steps 200 --env-sweep F2N_SHADE_BWD_BLOCKS=512,256,384 > gpurun_out/s15/shade_blocks.log 2>&1
timeout 400 python tools/converged_steps.py --restore --steps 200 --env-sweep F2N_FIELD_BWD_BLOCKS=768,512,256 > gpurun_out/s15/field_blocks.log 2>&1
unset F2N_DEBUG_BUILD
timeout 400 python tools/converged_steps.py --restore --steps 200 --fused-tail-sweep > gpurun_out/s15/tail.log 2>&1
timeout 400 python tools/converged_steps.py --restore --steps 200 --block-waves-sweep 1,2,4,8 > gpurun_out/s15/waves.log 2>&1
