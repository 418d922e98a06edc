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
//
// One step of the chain is ONE instruction (round 6): `v_add_f32_dpp p, p, x row_shr:1` with bound_ctrl off -- every lane k >= 1
// takes p[k-1] + x[k], lane 0 (no source lane) is left alone.  p starts as carry + x in every lane, which is lane 0's final value;
// step k makes lane k final, and a lane that is final stays so (it is recomputed from the same final neighbour and the same x), so
// 15 unconditional steps leave the sequential sums in every lane -- the additions of the one-thread walk in its order, as before.
// The builtin spelling (shift with 0 shifted in, add, select lane k) compiled to add + v_cndmask + s_nop per step, and a ray's
// walk is bound by how many instructions its one wave has to issue (converged batch: the longest ray is 25 chunks of 16 samples
// with 8 chains forward).  The compiler neither sees inside the asm nor pads it: VALU write -> DPP read of the same register needs
// two wait states (s_nop 1, or two other instructions: f2n_row_chainN interleave the steps of N independent chains), a VALU write of EXEC five.
#if defined(__HIP_DEVICE_COMPILE__)
#define F2N_DPP_STEP_(P, X) "v_add_f32_dpp " P ", " P ", " X " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define F2N_REP15_(S) S S S S S S S S S S S S S S S
__device__ __forceinline__ float f2n_row_seq_scan(float x, float carry, int c) {
  float p = carry + x;
  asm volatile("s_nop 4\n\t" F2N_REP15_(F2N_DPP_STEP_("%0", "%1") "s_nop 1\n\t") : "+v"(p) : "v"(x));
  return p;
}
// Chains that keep their state from one 16-sample chunk of a ray to the next: p holds the previous chunk's inclusive sums (or the
// initial carry in every lane), and the new chunk starts from `p[15] + x[0]` in lane 0 -- `row_ror:1` hands lane 15's value to lane 0
// inside the add, so no broadcast (ds_bpermute + wait) sits between two chunks.  Totals are read once, behind the walk (f2n_row_last).
#define F2N_DPP_HEAD_(P, X) "v_add_f32_dpp " P ", " P ", " X " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void f2n_row_chain1(float x, float& p) {
  asm volatile("s_nop 4\n\t" F2N_DPP_HEAD_("%0", "%1") "s_nop 1\n\t" F2N_REP15_(F2N_DPP_STEP_("%0", "%1") "s_nop 1\n\t") : "+v"(p) : "v"(x));
}
// ... and the prefix that excludes the lane's own term (lane 0: the previous chunk's total)
__device__ __forceinline__ void f2n_row_chain1x(float x, float& p, float& excl) {
  asm volatile("s_nop 4\n\t"
               "v_mov_b32_dpp %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t" F2N_DPP_HEAD_("%0", "%2") "s_nop 1\n\t" F2N_REP15_(
                   F2N_DPP_STEP_("%0", "%2") "s_nop 1\n\t") "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                                            "s_nop 1\n\t"
               : "+v"(p), "=&v"(excl)
               : "v"(x));
}
__device__ __forceinline__ void f2n_row_chain2(float x0, float x1, float& p0, float& p1) {
  asm volatile("s_nop 4\n\t" F2N_DPP_HEAD_("%0", "%2") F2N_DPP_HEAD_("%1", "%3") "s_nop 0\n\t" F2N_REP15_(
                   F2N_DPP_STEP_("%0", "%2") F2N_DPP_STEP_("%1", "%3") "s_nop 0\n\t") "s_nop 0\n\t"
               : "+v"(p0), "+v"(p1)
               : "v"(x0), "v"(x1));
}
__device__ __forceinline__ void f2n_row_chain3(float x0, float x1, float x2, float& p0, float& p1, float& p2) {
  asm volatile("s_nop 4\n\t" F2N_DPP_HEAD_("%0", "%3") F2N_DPP_HEAD_("%1", "%4") F2N_DPP_HEAD_("%2", "%5") F2N_REP15_(
                   F2N_DPP_STEP_("%0", "%3") F2N_DPP_STEP_("%1", "%4") F2N_DPP_STEP_("%2", "%5")) "s_nop 1\n\t"
               : "+v"(p0), "+v"(p1), "+v"(p2)
               : "v"(x0), "v"(x1), "v"(x2));
}
__device__ __forceinline__ void f2n_row_chain4(float x0, float x1, float x2, float x3, float& p0, float& p1, float& p2, float& p3) {
  asm volatile("s_nop 4\n\t" F2N_DPP_HEAD_("%0", "%4") F2N_DPP_HEAD_("%1", "%5") F2N_DPP_HEAD_("%2", "%6") F2N_DPP_HEAD_("%3", "%7") F2N_REP15_(
                   F2N_DPP_STEP_("%0", "%4") F2N_DPP_STEP_("%1", "%5") F2N_DPP_STEP_("%2", "%6") F2N_DPP_STEP_("%3", "%7")) "s_nop 1\n\t"
               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}
#else
// The same chain in portable C++ (this branch is what a host compiler sees; nothing of the product runs it -- the CPU wavefront
// emulation of tests/ does).
__device__ __forceinline__ float f2n_row_seq_scan(float x, float carry, int c) {
  float p = carry + x;
#pragma unroll
  for (int k = 1; k < 16; k++) {
    const float t = f2n_row_shr1(p) + x;
    p = (c == 0) ? p : t;
  }
  return p;
}
__device__ __forceinline__ float f2n_row_last_(float v) { return __shfl(v, 15, 16); }
__device__ __forceinline__ void f2n_row_chain1(float x, float& p) { p = f2n_row_seq_scan(x, f2n_row_last_(p), (int) (threadIdx.x & 15)); }
__device__ __forceinline__ void f2n_row_chain1x(float x, float& p, float& excl) {
  const int c = threadIdx.x & 15;
  const float carry = f2n_row_last_(p);
  p = f2n_row_seq_scan(x, carry, c);
  const float prev = f2n_row_shr1(p);
  excl = c == 0 ? carry : prev;
}
__device__ __forceinline__ void f2n_row_chain2(float x0, float x1, float& p0, float& p1) {
  f2n_row_chain1(x0, p0);
  f2n_row_chain1(x1, p1);
}
__device__ __forceinline__ void f2n_row_chain3(float x0, float x1, float x2, float& p0, float& p1, float& p2) {
  f2n_row_chain1(x0, p0);
  f2n_row_chain1(x1, p1);
  f2n_row_chain1(x2, p2);
}
__device__ __forceinline__ void f2n_row_chain4(float x0, float x1, float x2, float x3, float& p0, float& p1, float& p2, float& p3) {
  f2n_row_chain1(x0, p0);
  f2n_row_chain1(x1, p1);
  f2n_row_chain1(x2, p2);
  f2n_row_chain1(x3, p3);
}
#endif
__device__ __forceinline__ float f2n_row_last(float v) { return __shfl(v, 15, 16); }
// the prefix that excludes the lane's own term
__device__ __forceinline__ float f2n_row_exclusive(float incl, float carry, int c) {
  const float prev = f2n_row_shr1(incl);
  return c == 0 ? carry : prev;
}

