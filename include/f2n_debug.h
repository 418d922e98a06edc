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
#ifdef __cplusplus
}
#endif
#endif
