mkdir -p gpurun_out/s14
export F2N_DEBUG_BUILD=1
timeout 400 python tools/converged_steps.py --steps 200 --env-sweep F2N_SHADE_BWD_BLOCKS=512,256,384 > gpurun_out/s14/shade_blocks.log 2>&1
timeout 400 python tools/converged_steps.py --steps 200 --env-sweep F2N_FIELD_BWD_BLOCKS=768,512,256 > gpurun_out/s14/field_blocks.log 2>&1
