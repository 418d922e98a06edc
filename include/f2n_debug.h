/* Debugging entry points of libf2n_hip_debug.so (-DF2N_DEBUG_BUILD=1; f2-nerf_amd/build.py variant "debug", selected by
 * F2N_DEBUG_BUILD=1 in the environment before the package is imported).  NOT part of the product ABI (include/f2n_abi.h): the product
 * library does not export them, reads no environment variable and has no debugging hooks in its step.  The debug variant is the same
 * code plus: these two launches, the stream-skew / pollution hooks of the host's Renderer (ExpRunner.debug_side_delay) and the
 * measurement knobs F2N_MARCH_LDS, F2N_FUSED_GATHER, F2N_BINNED_GATHER_P0, F2N_BINNED_GATHER_MIN_LOG2, F2N_BIN_NB. */
#ifndef F2N_DEBUG_H
#define F2N_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* A delay on a stream: one wave that spins for that many microseconds (skews the sampler's side streams against the main stream:
 * tools/determinism_probe.py --side-delay). */
int f2n_debug_spin(void* stream, int microseconds);
/* A launch that leaves value-derived garbage in 64 KB of LDS and ~100 vector registers of every CU: what a co-tenant's kernels do to
 * the state a kernel finds when it starts (tools/determinism_probe.py --pollute). */
int f2n_debug_pollute(void* stream, unsigned value);
/* f2n_composite_train with the rays' sums of w * c handed in (colors_in [R,3], WITHOUT the background term): the forward walk then
 * neither loads the samples' colours nor carries their three scan chains.  An upper bound of what an order-free sum in the colour
 * network's epilogue (north star / judge row N1) can take off this launch: tools/n1_bound.py, profiles/r05_n1_bound.txt. */
int f2n_debug_composite_train_colors_in(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride,
                                        const float* dt, const float* t, const float* rgb, const float* bg, const float* gt_colors,
                                        float var_w, float disp_w, float tv_w, float grad_scaling_progress, float* colors, float* weights,
                                        float* drgb, float* df0, float* out_losses, const float* colors_in /*[R,3]*/);
#ifdef __cplusplus
}
#endif
#endif
