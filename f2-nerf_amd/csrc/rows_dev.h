// Row-per-ray helpers shared by the compositing kernels (render.hip) and the fused early-stop + occupancy-vote kernel
// (sampler.hip).
#pragma once
#include "f2n_dev.h"

#define F2N_DENSITY_SHIFT 3.f  // Renderer.cpp:101-104
#define F2N_T_EPS 1e-4f        // early-stop threshold, Renderer.cpp:125
#define F2N_T_BIAS 1e-2f       // sampled_t = t + 1e-2, Renderer.cpp:118,197

// ---------------------------------------------------------------------------------------------------
// One DPP row (16 lanes) per ray, four rays per wave.  The reference walks every ray left to right with one
// thread; that keeps 8192 rays on 128 waves and serialises ~100 exp-laden iterations per lane.  Here the 16 lanes
// of a row take 16 consecutive samples: all element-wise maths (exp, divisions) runs in parallel, and only the
// running sums are serial -- a 16-step DPP chain in which lane k adds its own term to lane k-1's finished prefix.
// The chain performs exactly the additions of the sequential loop in exactly its order, so results are bit-identical
// to the one-lane-per-ray walk (the summation order is part of the parity contract), at ~2 instructions per sample.
// ---------------------------------------------------------------------------------------------------
#define F2N_ROW_RAYS_PER_BLOCK 16  // 256 threads

__device__ __forceinline__ float f2n_row_shr1(float v) {  // lane c reads lane c-1 of its row (lane 0 reads 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
}
// Inclusive left-to-right running sum over the row: lane k returns ((carry + x_0) + x_1) + ... + x_k.
__device__ __forceinline__ float f2n_row_seq_scan(float x, float carry, int c) {
  float p = carry + x;
#pragma unroll
  for (int k = 1; k < 16; k++) {
    const float t = f2n_row_shr1(p) + x;
    p = (c == k) ? t : p;
  }
  return p;
}
__device__ __forceinline__ float f2n_row_last(float v) { return __shfl(v, 15, 16); }
// the prefix that excludes the lane's own term
__device__ __forceinline__ float f2n_row_exclusive(float incl, float carry, int c) {
  const float prev = f2n_row_shr1(incl);
  return c == 0 ? carry : prev;
}

